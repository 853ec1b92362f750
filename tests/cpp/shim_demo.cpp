// Drives the BaseSolver shim exactly like PathOptimizer::optimizePath (reference src/path_optimizer.cpp:124-161):
// input_path from the reference states (l = d_heading = 0, k = k_ref), solve(), updateProblemFormulationAndSolve(out, out).
// Reads a scenario (n, then n rows "s k heading x y flb fub rlb rub clb cub", then 6 scalars) from stdin, prints the
// path "x y heading l d_heading k d_k" per line.
#include <cstdio>
#include <vector>

#include "../../include/pqp_base_solver.hpp"

using namespace PathOptimizationNS;

int main(int argc, char** argv) {
    int n = 0;
    if (std::scanf("%d", &n) != 1 || n < 2) return 2;
    std::vector<State> states(n);
    std::vector<VehicleStateBound> bounds(n);
    for (int i = 0; i < n; ++i) {
        State& st = states[i];
        VehicleStateBound& b = bounds[i];
        if (std::scanf("%lf %lf %lf %lf %lf %lf %lf %lf %lf %lf %lf", &st.s, &st.k, &st.heading, &st.x, &st.y, &b.front.lb, &b.front.ub,
                       &b.rear.lb, &b.rear.ub, &b.center.lb, &b.center.ub) != 11) return 2;
    }
    double sc[6];
    for (double& v : sc) if (std::scanf("%lf", &v) != 1) return 2;
    ReferencePath ref;
    ref.setReferenceStates(states);
    ref.setBounds(bounds);
    State start, target;
    start.k = sc[2];
    target.heading = sc[3];
    VehicleState vs(start, target, sc[0], sc[1]);
    std::vector<SlState> input_path;                      // path_optimizer.cpp:128-137
    for (const auto& rs : ref.getReferenceStates()) {
        SlState in;
        in.x = rs.x; in.y = rs.y; in.heading = rs.heading; in.s = rs.s; in.k = rs.k;
        input_path.push_back(in);
    }
    BaseSolver solver(ref, vs, input_path);
    solver.setMaxSteeringAngle(sc[5]);
    for (int a = 1; a < argc; ++a) {
        pqp_params p = solver.params();
        double len = 0.0;
        if (std::sscanf(argv[a], "rough=%lf", &len) == 1) {   // FLAGS_rough_constraints_far_away with FLAGS_precise_planning_length = len
            p.rough_constraints_far_away = 1; p.precise_planning_length = len;
        } else {                                           // "polish": the engine's production setting
            p.eps_abs = p.eps_rel = 1e-4; p.polish = 1; p.polish_every = 25; p.adaptive_rho_interval = 25; p.polish_warm_set = 1; p.polish_refine_iter = 3;
        }
        solver.setParams(p);
    }
    std::vector<SlState> final_path;
    if (!solver.solve(&final_path)) { std::fprintf(stderr, "Pre solving failed! (status %d)\n", solver.lastStatus()); return 1; }
    const int it0 = solver.lastIterations();
    if (!solver.updateProblemFormulationAndSolve(final_path, &final_path)) { std::fprintf(stderr, "Solving failed!\n"); return 1; }
    std::fprintf(stderr, "vars %zu cons %zu precise %zu iters %d + %d\n", solver.vars(), solver.cons(), solver.precisePlanningSize(), it0, solver.lastIterations());
    for (const auto& p : final_path) std::printf("%.17g %.17g %.17g %.17g %.17g %.17g %.17g\n", p.x, p.y, p.heading, p.l, p.d_heading, p.k, p.d_k);
    return 0;
}
