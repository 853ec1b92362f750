// The reference's call pattern, timed: one BaseSolver per planning cycle at batch 1 (src/path_optimizer.cpp:138-153: construct ->
// solve() -> updateProblemFormulationAndSolve(out, out) -> destruct), K cycles on the same scenario.  Reads the scenario format of
// shim_demo.cpp from stdin; argv: cycles [cache 0/1] [polish 0/1].  Prints one JSON object: microseconds of the first cycle (it pays
// pqp_create) and of the later ones (median / min), split into construct+solve, update, destruct.
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../include/pqp_base_solver.hpp"

using namespace PathOptimizationNS;
using Clock = std::chrono::steady_clock;

static double us(Clock::time_point a, Clock::time_point b) { return std::chrono::duration<double, std::micro>(b - a).count(); }
static double median(std::vector<double> v) { std::sort(v.begin(), v.end()); return v.empty() ? 0.0 : v[v.size() / 2]; }

int main(int argc, char** argv) {
    const int cycles = argc > 1 ? std::atoi(argv[1]) : 50;
    const bool cache = argc > 2 ? std::atoi(argv[2]) != 0 : true;
    const bool polish = argc > 3 ? std::atoi(argv[3]) != 0 : false;
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
    std::vector<SlState> input_path;
    for (const auto& rs : ref.getReferenceStates()) {
        SlState in;
        in.x = rs.x; in.y = rs.y; in.heading = rs.heading; in.s = rs.s; in.k = rs.k;
        input_path.push_back(in);
    }
    std::vector<double> total, t_solve, t_update, t_destruct;
    double first = 0.0;
    int iters0 = 0, iters1 = 0;
    for (int c = 0; c < cycles; ++c) {
        const auto t0 = Clock::now();
        Clock::time_point t1, t2;
        {
            BaseSolver solver(ref, vs, input_path);
            solver.setMaxSteeringAngle(sc[5]);
            solver.setHandleCaching(cache);
            if (polish) { pqp_params p; pqp_production_params(&p); solver.setParams(p); }
            std::vector<SlState> path;
            if (!solver.solve(&path)) { std::fprintf(stderr, "Pre solving failed!\n"); return 1; }
            iters0 = solver.lastIterations();
            t1 = Clock::now();
            if (!solver.updateProblemFormulationAndSolve(path, &path)) { std::fprintf(stderr, "Solving failed!\n"); return 1; }
            iters1 = solver.lastIterations();
            t2 = Clock::now();
        }
        const auto t3 = Clock::now();
        if (c == 0) first = us(t0, t3);
        else { total.push_back(us(t0, t3)); t_solve.push_back(us(t0, t1)); t_update.push_back(us(t1, t2)); t_destruct.push_back(us(t2, t3)); }
    }
    std::printf("{\"n\": %d, \"cycles\": %d, \"handle_cache\": %s, \"setting\": \"%s\", \"first_cycle_us\": %.1f, \"cycle_us_median\": %.1f, \"cycle_us_min\": %.1f, "
                "\"construct_and_solve_us\": %.1f, \"update_and_solve_us\": %.1f, \"destruct_us\": %.1f, \"admm_iters\": [%d, %d]}\n",
                n, cycles, cache ? "true" : "false", polish ? "production" : "reference (eps 2e-3, no polish)", first, median(total),
                total.empty() ? 0.0 : *std::min_element(total.begin(), total.end()), median(t_solve), median(t_update), median(t_destruct), iters0, iters1);
    return 0;
}
