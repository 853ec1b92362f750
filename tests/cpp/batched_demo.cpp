// Drives PathOptimizationNS::BatchedPathSolver (include/pqp_batched_solver.hpp): reads "B" then B scenarios (n, then n rows
// "s k heading x y flb fub rlb rub clb cub", then 6 scalars) from stdin - the scenarios may differ in n -, solves them in one batch over
// argv[1] shards (default 1; more than the GPUs present is refused by the engine) and prints per scenario "ok n" and the path rows.
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../include/pqp_batched_solver.hpp"

using namespace PathOptimizationNS;

int main(int argc, char** argv) {
    const int devices = argc > 1 ? std::atoi(argv[1]) : 1;
    int B = 0;
    if (std::scanf("%d", &B) != 1 || B < 1) return 2;
    std::vector<ReferencePath> refs(B);
    std::vector<VehicleState> vss(B);
    std::vector<double> steer(B);
    for (int q = 0; q < B; ++q) {
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
        refs[q].setReferenceStates(states);
        refs[q].setBounds(bounds);
        State start, target;
        start.k = sc[2];
        target.heading = sc[3];
        vss[q] = VehicleState(start, target, sc[0], sc[1]);
        steer[q] = sc[5];
    }
    BatchedPathSolver solver(devices);
    for (int q = 0; q < B; ++q) solver.add(refs[q], vss[q], steer[q]);
    std::vector<std::vector<SlState>> paths;
    std::vector<bool> ok;
    if (!solver.optimizePaths(&paths, &ok)) { std::fprintf(stderr, "Solving failed!\n"); return 1; }
    for (int q = 0; q < B; ++q) {
        std::printf("%d %zu\n", ok[q] ? 1 : 0, paths[q].size());
        for (const auto& p : paths[q]) std::printf("%.17g %.17g %.17g %.17g %.17g %.17g %.17g\n", p.x, p.y, p.heading, p.l, p.d_heading, p.k, p.d_k);
    }
    // the single-scenario form, as PathOptimizer::solve would call it
    MapFreePathOptimizer one;
    std::vector<SlState> final_path;
    if (!one.solve(refs[0], vss[0], &final_path) || final_path.size() != paths[0].size()) { std::fprintf(stderr, "MapFreePathOptimizer failed\n"); return 1; }
    return 0;
}
