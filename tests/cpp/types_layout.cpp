// Prints sizes and field offsets of include/pqp_types.hpp in the order oracle/ref_types_shim.cpp::ref_type_layout reports them for the
// reference's include/data_struct/data_struct.hpp, then the results of the same small scenarios (State's constructor, VehicleState's
// getters).  tests/test_ref_types.py compares the two line by line.
#include <cstddef>
#include <cstdio>

#include "pqp_types.hpp"

using namespace PathOptimizationNS;

int main() {
    const int v[] = {
        (int)sizeof(State), (int)offsetof(State, x), (int)offsetof(State, y), (int)offsetof(State, heading), (int)offsetof(State, k),
        (int)offsetof(State, d_k), (int)offsetof(State, s), (int)offsetof(State, v), (int)offsetof(State, a),
        (int)sizeof(SlState), (int)offsetof(SlState, l), (int)offsetof(SlState, d_heading),
        (int)sizeof(VehicleStateBound::SingleBound), (int)offsetof(VehicleStateBound::SingleBound, ub),
        (int)offsetof(VehicleStateBound::SingleBound, lb), (int)offsetof(VehicleStateBound::SingleBound, x),
        (int)offsetof(VehicleStateBound::SingleBound, y), (int)offsetof(VehicleStateBound::SingleBound, heading),
        (int)sizeof(VehicleStateBound), (int)offsetof(VehicleStateBound, front), (int)offsetof(VehicleStateBound, rear),
        (int)offsetof(VehicleStateBound, center)};
    std::printf("layout");
    for (int x : v) std::printf(" %d", x);
    std::printf("\n");
    State s(1.5, -2.5, 0.25, 0.125, 7.0, 3.0, -1.0);
    std::printf("state %.17g %.17g %.17g %.17g %.17g %.17g %.17g %.17g\n", s.x, s.y, s.heading, s.k, s.d_k, s.s, s.v, s.a);
    State st(0.5, 0.75, -0.5, 0.0625), tg(20.0, 3.0, 0.375, -0.03125);
    VehicleState vs(st, tg, 0.4, -0.1);
    const std::vector<double> e0 = vs.getInitError();
    vs.setInitError(-0.2, 0.05);
    const std::vector<double> e1 = vs.getInitError();
    std::printf("vehicle %.17g %.17g %.17g %.17g %.17g %.17g\n", e0[0], e0[1], vs.getStartState().k, vs.getTargetState().heading, e1[0], e1[1]);
    return 0;
}
