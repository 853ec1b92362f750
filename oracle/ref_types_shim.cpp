// C entry points around the std-only pieces of the reference's data types and helpers, compiled from where they lie under
// /root/reference by oracle/Makefile into oracle/_ref/libref_types.so:
//   include/tools/tools.hpp:24-35                     constrainAngle (header-only template)
//   include/data_struct/data_struct.hpp:14-32,74-93   State, SlState, VehicleStateBound (layouts by sizeof / offsetof)
//   src/data_struct/vehicle_state_frenet.cpp:29-47    VehicleState getters / getInitError / setInitError
// TEST INFRASTRUCTURE: it pins constrain_angle() of the oracle and of the HIP kernels bit for bit and the layouts of
// include/pqp_types.hpp field by field; nothing in the product links or loads it.
#include <cstddef>
#include <vector>

#include "data_struct/data_struct.hpp"
#include "data_struct/vehicle_state_frenet.hpp"
#include "tools/tools.hpp"

using namespace PathOptimizationNS;

extern "C" {

double ref_constrain_angle(double a) { return constrainAngle(a); }

// sizes and field offsets in bytes, in a fixed order (tests/cpp/types_layout.cpp prints the same list for pqp_types.hpp)
int ref_type_layout(int* out, int cap) {
    const int v[] = {
        (int)sizeof(State), (int)offsetof(State, x), (int)offsetof(State, y), (int)offsetof(State, heading), (int)offsetof(State, k),
        (int)offsetof(State, d_k), (int)offsetof(State, s), (int)offsetof(State, v), (int)offsetof(State, a),
        (int)sizeof(SlState), (int)offsetof(SlState, l), (int)offsetof(SlState, d_heading),
        (int)sizeof(VehicleStateBound::SingleBound), (int)offsetof(VehicleStateBound::SingleBound, ub),
        (int)offsetof(VehicleStateBound::SingleBound, lb), (int)offsetof(VehicleStateBound::SingleBound, x),
        (int)offsetof(VehicleStateBound::SingleBound, y), (int)offsetof(VehicleStateBound::SingleBound, heading),
        (int)sizeof(VehicleStateBound), (int)offsetof(VehicleStateBound, front), (int)offsetof(VehicleStateBound, rear),
        (int)offsetof(VehicleStateBound, center)};
    const int n = (int)(sizeof(v) / sizeof(v[0]));
    for (int i = 0; i < n && i < cap; ++i) out[i] = v[i];
    return n;
}

// State's 7-argument constructor (data_struct.hpp:16-17): (x, y, heading, k, s, v, a) -> the 8 fields in declaration order
void ref_state_ctor(const double* a7, double* out8) {
    State s(a7[0], a7[1], a7[2], a7[3], a7[4], a7[5], a7[6]);
    out8[0] = s.x; out8[1] = s.y; out8[2] = s.heading; out8[3] = s.k; out8[4] = s.d_k; out8[5] = s.s; out8[6] = s.v; out8[7] = s.a;
}

// SingleBound::set takes {ub, lb} in that order (data_struct.hpp:82-88)
void ref_single_bound_set(const double* two, const double* center_xyh, double* out5) {
    VehicleStateBound::SingleBound b;
    State c(center_xyh[0], center_xyh[1], center_xyh[2]);
    b.set(std::vector<double>{two[0], two[1]}, c);
    out5[0] = b.ub; out5[1] = b.lb; out5[2] = b.x; out5[3] = b.y; out5[4] = b.heading;
}

// VehicleState(start, end, offset, heading_error) -> getInitError(), start.k, target.heading; then setInitError (vehicle_state_frenet.cpp:14-51)
void ref_vehicle_state(const double* start_xyhk, const double* target_xyhk, double offset, double heading_error, double set_offset,
                       double set_heading_error, double* out6) {
    State s(start_xyhk[0], start_xyhk[1], start_xyhk[2], start_xyhk[3]);
    State t(target_xyhk[0], target_xyhk[1], target_xyhk[2], target_xyhk[3]);
    VehicleState vs(s, t, offset, heading_error);
    const std::vector<double> e0 = vs.getInitError();
    out6[0] = e0[0]; out6[1] = e0[1];
    out6[2] = vs.getStartState().k; out6[3] = vs.getTargetState().heading;
    vs.setInitError(set_offset, set_heading_error);
    const std::vector<double> e1 = vs.getInitError();
    out6[4] = e1[0]; out6[5] = e1[1];
}
}
