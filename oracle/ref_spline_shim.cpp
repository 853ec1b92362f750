// C entry points around the reference's own tk::spline (src/tools/spline.cpp, std-only), compiled from where it lies under
// /root/reference by oracle/Makefile into oracle/_ref/libref_spline.so.  TEST INFRASTRUCTURE: it validates the spline
// restatement in oracle/corridor_oracle.py bit for bit; nothing in the product links or loads it.
#include <vector>

#include "tools/spline.h"

using PathOptimizationNS::tk::spline;

extern "C" {
void* ref_spline_new(int n, const double* x, const double* y) {
    spline* s = new spline();
    s->set_points(std::vector<double>(x, x + n), std::vector<double>(y, y + n));
    return s;
}
double ref_spline_eval(void* s, double x) { return (*static_cast<spline*>(s))(x); }
double ref_spline_deriv(void* s, int order, double x) { return static_cast<spline*>(s)->deriv(order, x); }
void ref_spline_free(void* s) { delete static_cast<spline*>(s); }
}
