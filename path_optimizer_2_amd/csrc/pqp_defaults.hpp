// Default pqp_params: the reference's gflags defaults (src/config/planning_flags.cpp), the constants
// hard-coded in src/solver/base_solver.cpp, and OSQP's documented defaults.
#pragma once
#include "../../include/pqp.h"

namespace pqp {
inline void default_params(pqp_params* p) {
    p->front_length = 3.9;                          // planning_flags.cpp:20
    p->rear_length = -1.0;                          // planning_flags.cpp:18
    p->wheel_base = 2.5;                            // planning_flags.cpp:16
    p->expected_safety_margin = 0.6;                // planning_flags.cpp:95
    p->precise_planning_length = 30.0;              // planning_flags.cpp:114
    p->constraint_end_heading = 1;                  // planning_flags.cpp:98
    p->rough_constraints_far_away = 0;              // planning_flags.cpp:112
    p->weight_l = 0.0;                              // base_solver.cpp:123
    p->weight_kappa = 20.0;                         // base_solver.cpp:124
    p->weight_dkappa = 100.0;                       // base_solver.cpp:125
    p->weight_slack = 10.0;                         // base_solver.cpp:126
    p->end_l_bound = 1.0;                           // base_solver.cpp:250-251
    p->end_psi_tol = 0.087;                         // base_solver.cpp:257-258
    p->end_psi_max = 70.0 * 3.14159265358979323846 / 180.0;   // base_solver.cpp:256
    p->min_clearance = 0.1;                         // base_solver.cpp:292
    p->eps_abs = 2e-3;                              // base_solver.cpp:61
    p->eps_rel = 2e-3;                              // base_solver.cpp:62
    p->rho = 0.1;
    p->sigma = 1e-6;
    p->alpha = 1.6;
    p->max_iter = 4000;
    p->scaling = 10;
    p->adaptive_rho = 1;
    p->adaptive_rho_interval = 100;
    p->adaptive_rho_tolerance = 5.0;
    p->check_termination = 25;
    p->polish = 0;
    p->polish_refine_iter = 4;
    p->polish_every = 0;
    p->polish_warm_set = 0;
    p->polish_max_rounds = 40;
    p->polish_reseed = 0;
    p->polish_diverge = 0;
    p->polish_reseed_factor = 1.0;
    p->eps_prim_inf = 1e-4;
    p->polish_patience = 0;
    p->prim_inf_after = 0;                          // OSQP: the certificate at every check
    p->polish_lazy = 0;
    p->polish_final_refine = 0;
    p->polish_delta = 1e-6;
    p->polish_tol = 1e-7;
    p->tension2_deviation_weight = 0.005;           // planning_flags.cpp:57
    p->tension2_curvature_weight = 1.0;             // planning_flags.cpp:59
    p->tension2_curvature_rate_weight = 10.0;       // planning_flags.cpp:61
    p->cartesian_curvature_weight = 1.0;            // planning_flags.cpp:51
    p->cartesian_curvature_rate_weight = 50.0;      // planning_flags.cpp:53
    p->cartesian_deviation_weight = 0.0;            // planning_flags.cpp:55
}

// The engine's production setting on top of the defaults: ADMM to 1e-4, KKT-verified polish (a returned path is the exact
// optimum of its QP), 4 Ruiz passes, residual check / rho adaptation / polish attempt every 5 (paths of up to 90 waypoints) or 8 iterations, 2 refinement solves per active-set
// round (the first 5 rounds of an attempt move their rows after the first of them), at most max(24, n/5 - 8) rounds per attempt, pass 2 starts from pass 1's active set and equilibration, an attempt that
// gives up re-seeds ADMM with its best point, a QP whose polish cannot be verified ends like OSQP's (ADMM point, unpolished), the
// infeasibility certificate evaluated outside the ADMM loop from iteration 100 on (prim_inf_after = 0 gives OSQP's every-check test back).  Tuned on MI355X (DESIGN.md sections 2, 5);
// bench.py, smoke() and the parity tests run this setting.
inline void production_params(pqp_params* p) {
    default_params(p);
    p->eps_abs = 1e-4;
    p->eps_rel = 1e-4;
    p->scaling = -4;                                // 4 Ruiz passes instead of OSQP's 10: the polish returns the exact optimum whatever the
                                                    // metric, the ADMM iterations before it only have to predict the active set (+0.5 % solves,
                                                    // -6 passes of 2.5 us: +4.5 % paths/s, profiles/r02h_policy_sweep.txt).  Negative (round 4):
                                                    // the passes evaluated on one interior waypoint's blocks and taken by every waypoint - on
                                                    // the tested scenario families the same D, E, c as the full passes, without their
                                                    // exchanges and reductions; the polish makes the result independent of it (+1.2 % paths/s, profiles/r04l_nominal_scaling_ab.txt)
    p->adaptive_rho_interval = -1;                  // the three intervals by path length (path_interval() below): 5 iterations up to 90 waypoints, 8 beyond
    p->check_termination = -1;
    p->polish = 1;
    p->polish_refine_iter = 2;
    p->polish_every = -1;                           // (round 1: 15.  A factorisation now costs 2 solves, not 3.3: earlier, cheaper attempts win: +6 %.
                                                    //  round 5: 8 at every length -> 5 up to 90 waypoints: short paths' first attempt needs no more
                                                    //  iterations than that - headline +2.2 %, N = 60 +12 %, one launch at a time +7 %; at 96 / 110 / 128 / 200
                                                    //  waypoints 5 costs 0.5 / 1.1 / 2.4 / 4.1 %: profiles/r05x_polish_every_seeds_configs.txt, r05aa_intervals.txt)
    p->polish_warm_set = 2;
    p->polish_max_rounds = 0;                       // auto: max(24, n/5 - 8)
    p->polish_reseed = 1;
    p->adaptive_rho_tolerance = 2.0;                // re-balance rho sooner: the few slow QPs of a batch need 175 instead of 350 iterations
    p->max_iter = 1000;                             // per pass (OSQP's 4000 in the defaults): every feasible QP of the sweeps ends within 500; an
                                                    // infeasible one, which this setting cannot certify, then holds its batch up for 4 ms, not 15
    p->polish_patience = 5;                         // a QP whose polish cannot be verified (e.g. infeasible by 1e-5) ends like OSQP's,
                                                    // after attempts at 8, 24, 56, 120, 248 iterations (5, 15, 35, 75, 155 on short paths)
    p->polish_lazy = 5;                             // the first 5 rounds of an attempt move rows after one solve: -11 % solve+factor cost on the
                                                    // emulator sweeps with the tails unchanged (8 and more: the tails grow), +4..10 % paths/s
                                                    // on MI355X (profiles/r02i_lazy_refinement.txt)
    p->prim_inf_after = 100;                        // the lean kernel (no certificate work inside the ADMM loop: 12 % faster iterations); from
                                                    // iteration 100 on the certificate is evaluated between checks on y_now - y_previous_check,
                                                    // so an infeasible QP ends PRIMAL_INFEASIBLE after ~130 iterations instead of holding its
                                                    // batch up until max_iter (every feasible QP of the sweeps is long done by then)
}

// pqp_params::adaptive_rho_interval / check_termination / polish_every < 0: by path length (production_params).
inline int path_interval(int n) { return n <= 90 ? 5 : 8; }
// What a solve of paths of (up to) n waypoints runs with: the intervals resolved, and long paths' extra refinement of accepted points
// (pqp.h: polish_final_refine).  Shared by the launcher (pqp_kernels.hip) and the host emulation (tests/emu/lane_emu.cpp).
inline void resolve_path_params(pqp_params* p, int n) {
    const int k = path_interval(n);
    if (p->adaptive_rho_interval < 0) p->adaptive_rho_interval = k;
    if (p->check_termination < 0) p->check_termination = k;
    if (p->polish_every < 0) p->polish_every = k;
    const int r = n > 256 ? 3 : (n > 128 ? 1 : 0);
    if (p->polish_final_refine < r) p->polish_final_refine = r;
}
// the smoother QPs (generic banded core): the same fields < 0 mean 8
inline void resolve_banded_params(pqp_params* p) {
    if (p->adaptive_rho_interval < 0) p->adaptive_rho_interval = 8;
    if (p->check_termination < 0) p->check_termination = 8;
    if (p->polish_every < 0) p->polish_every = 8;
}
}  // namespace pqp
