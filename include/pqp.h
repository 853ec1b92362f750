/*
 * pqp.h — C ABI of the MI355X-native batched path-QP engine (libpqp_hip.so).
 *
 * Drop-in boundary for the ONE hot path of LiJiangnanBit/path_optimizer_2: assembling and solving
 * the Frenet path QP behind PathOptimizer::optimizePath / BaseSolver.  Every entry point cites the
 * reference interface it replaces (file:line relative to the reference tree).
 *
 *   reference                                                     this ABI
 *   ------------------------------------------------------------  ---------------------------------
 *   BaseSolver::BaseSolver            src/solver/base_solver.cpp:15-39   pqp_path_sizes / pqp_path_pattern
 *   BaseSolver::setCost               src/solver/base_solver.cpp:119-148 pqp_path_assemble (P diagonal)
 *   BaseSolver::setConstraints        src/solver/base_solver.cpp:150-261 pqp_path_assemble (A values, l, u)
 *   BaseSolver::getSoftBounds         src/solver/base_solver.cpp:290-295 (inside the assemble kernel)
 *   BaseSolver::solve                 src/solver/base_solver.cpp:56-95   pqp_path_solve (warm = 0, passes = 0)
 *   BaseSolver::updateProblemFormulationAndSolve      :97-117            pqp_path_solve (warm = 1, lin = input)
 *   BaseSolver::getOptimizedPath      src/solver/base_solver.cpp:263-288 (unpack, fused into the solve kernel)
 *   PathOptimizer::optimizePath       src/path_optimizer.cpp:124-161     pqp_path_solve (passes = 1)
 *   OsqpEigen::Solver initSolver/solve (third party, called at base_solver.cpp:87-88,110)
 *                                                                 the ADMM loop inside the solve kernel
 *   TensionSmoother2::osqpSmooth      src/reference_path_smoother/tension_smoother_2.cpp:20-158   pqp_smooth_tension2
 *   TensionSmoother::osqpSmooth       src/reference_path_smoother/tension_smoother.cpp:49-177     pqp_smooth_tension
 *   ReferencePathSmoother::postSmooth src/reference_path_smoother/reference_path_smoother.cpp:526-636 (QP part) pqp_post_smooth
 *   ReferencePath::updateBounds -> ReferencePathImpl::updateBoundsImproved
 *                                     src/data_struct/reference_path.cpp:61, reference_path_impl.cpp:177-312    pqp_corridor_bounds
 *   ReferencePathImpl::buildReferenceFromSpline  reference_path_impl.cpp:314-338, PathOptimizer::processInitState path_optimizer.cpp:73-85
 *                                                                                                         pqp_reference_states
 *   ReferencePathSmoother::postSmooth (tail)    reference_path_smoother.cpp:559-573                         pqp_offsets_to_points
 *   PathOptimizer::setReferencePathLength       path_optimizer.cpp:87-104                                   pqp_reference_length
 *   ReferencePathSmoother::bSpline              reference_path_smoother.cpp:490-521                         pqp_bspline_resample
 *   ReferencePathSmoother::segmentRawReference  reference_path_smoother.cpp:48-85                           pqp_segment_raw_reference
 *   tk::spline::set_points            src/tools/spline.cpp:161-249                                         pqp_spline_fit
 *   ReferencePathSmoother::graphSearchDp  src/reference_path_smoother/reference_path_smoother.cpp:142-295   pqp_dp_corridor
 *
 * Conventions
 *   - plain C, no C++/torch types; all reals are IEEE fp64, all indices int32.
 *   - every function returns 0 on success or a negative pqp_error; nothing throws across the ABI.
 *   - `*_device` entry points take DEVICE pointers and enqueue on the handle's HIP stream without
 *     synchronising; pqp_sync() waits.  The non-suffixed entry points take HOST pointers, copy in,
 *     run, copy out and synchronise (that is what the C++ BaseSolver shim uses with batch = 1).
 *   - the handle owns all device memory and its stream; callers own every buffer they pass; no
 *     caller pointer is retained after a call returns (device entry points: after pqp_sync()).
 *   - a handle is single-owner and not re-entrant; use one handle per GPU / host thread.
 *   - there is NO CPU fallback: if no HIP device is usable, pqp_create fails with PQP_ERR_NO_DEVICE.
 *
 * Array layouts (C-contiguous, fp64) — AoS per waypoint, the way State/SlState vectors are walked
 * (reference include/data_struct/data_struct.hpp:14-32,74-93):
 *   ref    [batch][n][5]   s, k, heading, x, y          ReferencePath::getReferenceStates()
 *   lin    [batch][n][3]   l, d_heading, k              input_path_ (the linearisation point); NULL =>
 *                                                       (0, 0, k_ref) as path_optimizer.cpp:128-137
 *   bounds [batch][n][6]   front lb, ub, rear lb, ub, center lb, ub      ReferencePath::getBounds()
 *   scal   [batch][6]      init_error[0], init_error[1], start_state.k, target_state.heading,
 *                          blocked (0/1: ReferencePath::isBlocked()!=nullptr), max_steering_angle
 *   out    [batch][n][7]   x, y, heading, l, d_heading, k, d_k           the SlState fields
 *                                                       getOptimizedPath fills (s, v, a stay 0 there)
 */
#ifndef PQP_H_
#define PQP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PQP_REF_STRIDE 5
#define PQP_LIN_STRIDE 3
#define PQP_BOUNDS_STRIDE 6
#define PQP_SCAL_STRIDE 6
#define PQP_OUT_STRIDE 7
#define PQP_INFO_STRIDE 8

typedef enum pqp_error {
    PQP_OK = 0,
    PQP_ERR_INVALID = -1,      /* bad argument (null pointer, n or batch out of range)            */
    PQP_ERR_NO_DEVICE = -2,    /* no usable HIP device: the product path has no CPU fallback      */
    PQP_ERR_HIP = -3,          /* a HIP runtime call failed; see pqp_last_error()                 */
    PQP_ERR_CAPACITY = -4      /* batch / n exceed what the handle was created for                */
} pqp_error;

/* Per-QP termination status written to status[batch] (osqp-eigen collapses this to bool). */
typedef enum pqp_status {
    PQP_STATUS_UNSOLVED = 0,
    PQP_STATUS_SOLVED = 1,         /* both residual tests passed (and, with polish on, the KKT-verified polish
                                      was accepted or ADMM reached 1e-10)             -> solve() == true  */
    PQP_STATUS_MAX_ITER = 2,       /* max_iter reached                                  -> solve() == false */
    PQP_STATUS_NUMERICAL = 3,      /* NaN/Inf in the iterates, or a scenario that is not a number: NaN / Inf among its reference states,
                                      bounds, linearisation point or start state, or an arclength s that does not increase (the reference
                                      divides by ds, base_solver.cpp:174,180).  Checked inside both path kernels (host- and device-pointer
                                      entry points alike): such a QP ends here before its first iteration, its output record is all
                                      zeros, the other QPs of the batch are not affected.  An absent bound is +-1e30 (OSQP_INFTY),
                                      not an IEEE infinity                              -> solve() == false */
    PQP_STATUS_PRIMAL_INFEASIBLE = 4   /* OSQP's primal infeasibility certificate holds  -> solve() == false */
} pqp_status;

/* The scalars the path reads.  Defaults (pqp_default_params) are the reference's gflags defaults
 * (src/config/planning_flags.cpp) and the constants hard-coded in base_solver.cpp. */
typedef struct pqp_params {
    /* car / planning flags */
    double front_length;              /* planning_flags.cpp:20   3.9   */
    double rear_length;               /* planning_flags.cpp:18  -1.0   */
    double wheel_base;                /* planning_flags.cpp:16   2.5   */
    double expected_safety_margin;    /* planning_flags.cpp:95   0.6   */
    double precise_planning_length;   /* planning_flags.cpp:114  30.0  */
    int32_t constraint_end_heading;   /* planning_flags.cpp:98   1     */
    int32_t rough_constraints_far_away; /* planning_flags.cpp:112 0    */
    /* constants of base_solver.cpp */
    double weight_l;                  /* :123  0    */
    double weight_kappa;              /* :124  20   */
    double weight_dkappa;             /* :125  100  */
    double weight_slack;              /* :126  10   */
    double end_l_bound;               /* :250-251  1.0   */
    double end_psi_tol;               /* :257-258  0.087 */
    double end_psi_max;               /* :256  70 deg    */
    double min_clearance;             /* :292  0.1       */
    /* solver settings (OSQP names; base_solver.cpp:59-62 sets eps to 2e-3, the rest are OSQP defaults) */
    double eps_abs;
    double eps_rel;
    double rho;                       /* 0.1   */
    double sigma;                     /* 1e-6  */
    double alpha;                     /* 1.6   */
    int32_t max_iter;                 /* 4000  */
    int32_t scaling;                  /* 10 Ruiz passes (OSQP's default).  k < 0 (path QP only; production: -4): |k| passes of the same equilibration
                                         evaluated on ONE interior waypoint's blocks and taken by every waypoint (no exchange, no reduction: ~1 us
                                         instead of ~2 us per pass).  A valid positive diagonal scaling - any is - that equals what |k| full passes
                                         give on the tested scenario families (uniform spacing, default flags, a pass linearised around the
                                         reference line); it ignores what only single waypoints carry (first / last waypoint's rows, rough rows,
                                         varied spacing).  Meant for handles with polish != 0, whose result does not depend on the metric: with
                                         polish == 0 the eps-accurate iterates then differ from OSQP's.  |k| <= 64 (pqp_set_params). */
    int32_t adaptive_rho;             /* 1     */
    int32_t adaptive_rho_interval;    /* 100 (OSQP's "auto" value without wall-clock profiling).  This field, check_termination and polish_every < 0
                                         (pqp_production_params: -1): by path length - 5 iterations for paths of up to 90 waypoints, 8 beyond
                                         (measured: profiles/r05x_polish_every_seeds_configs.txt; the smoother QPs: 8) */
    double adaptive_rho_tolerance;    /* 5     */
    int32_t check_termination;        /* 25    (< 0: see adaptive_rho_interval) */
    /* solution polishing (OSQP paper section 4.2; OSQP default and the reference: off).  When on, the active
     * set the ADMM iterate predicts is solved as an equality-constrained QP (regularised KKT + iterative
     * refinement) and ACCEPTED ONLY IF the polished point passes a KKT check (primal feasibility of inactive
     * rows, dual signs of active rows, stationarity) - i.e. it is then the exact optimum of the QP.  A
     * rejected polish resumes ADMM with a 10x tighter internal tolerance and tries again later. */
    int32_t polish;                   /* 0 (reference) ; bench and parity tests use 1.  Smoother QPs with polish != 0 return exact optima
                                         with iters = 0 where the QP's structure allows: TensionSmoother2's (equality rows only: a linear-
                                         quadratic control problem) by one Riccati sweep per scenario; with polish == 1 postSmooth's (a box
                                         QP in the offsets) and TensionSmoother's (a box QP in the lateral shifts) by a KKT-verified
                                         active-set solve, one wavefront per scenario (any count of layers / points: up to 1024 in the
                                         wavefront's registers, beyond with the lane state in an HBM workspace).  With polish == 2 QPs with inequality rows run the plain ADMM.  The path QP treats 2
                                         like 1 */
    int32_t polish_refine_iter;       /* 4     */
    int32_t polish_every;             /* 0: only when the residual test passes; k: also try every k iterations; < 0: see adaptive_rho_interval */
    int32_t polish_warm_set;          /* 1: a warm re-linearised re-solve starts with a polish on the previous pass's active set;
                                         2: ... and keeps that pass's equilibration (D, E, c) instead of re-running Ruiz */
    int32_t polish_max_rounds;        /* 40: active-set correction rounds per polish attempt; <= 0: max(24, n/5 - 8) (40 for the smoothers) */
    int32_t polish_reseed;            /* 1: a polish attempt that gives up hands its best point (smallest KKT failure) to ADMM as the
                                         new iterate when that failure is below polish_reseed_factor x the ADMM residuals */
    int32_t polish_diverge;           /* k > 0: an attempt gives up as soon as the KKT failure exceeds k x the smallest one seen in it */
    double polish_delta;              /* 1e-6  regularisation; active rows get penalty 1/delta */
    double polish_tol;                /* 1e-7  KKT acceptance tolerance of the polished point  */
    double polish_reseed_factor;      /* 1.0   */
    double eps_prim_inf;              /* 1e-4  OSQP's primal infeasibility tolerance; <= 0: no certificate test */
    int32_t polish_patience;          /* 0: a pass ends only with an accepted polish (or max_iter).  k > 0: once ADMM has met its
                                         residual test and the gap between polish attempts has doubled k times without an accepted
                                         polish, the ADMM point is returned as PQP_STATUS_SOLVED without polish - OSQP's behaviour when
                                         its polish fails (info[4] counts the polished passes).  With polish_every = 0: at once. */
    int32_t prim_inf_after;           /* 0: OSQP's certificate on y_k - y_{k-1} at every termination check (the kernel variant with the
                                         certificate in its loop, ~12 % slower iterations).  k > 0: the lean kernel; from iteration k on every
                                         termination check that neither converged nor started a polish evaluates the SAME certificate on
                                         dy = y_now - y_at_the_previous_such_check (any dy that passes it proves infeasibility): nothing in
                                         the ADMM loop, an infeasible QP stops about two checks after iteration k */
    int32_t polish_lazy;              /* 0: every active-set round refines its point with polish_refine_iter solves before it is looked at.
                                         k > 0: a round first looks after ONE solve; during the first k full rounds of an attempt rows that
                                         fail the test by more than 10 x that solve's residual change sides at once, the remaining
                                         refinement solves only run when nothing moves that far (they are what the KKT acceptance test
                                         needs, not what the set update needs).  The accepted point is always a fully refined one. */
    int32_t polish_final_refine;      /* 0.  k: an ACCEPTED polished point gets k more refinement solves (each followed by the KKT test again).  The path QP
                                         beyond 128 waypoints runs with at least 1, beyond 256 with at least 3: residuals of 1e-9 in the transition rows - a discrete double
                                         integrator - add up to 1e-4 in l over 300 waypoints, and a refinement solve shrinks them 100-1000x.  Honoured by the kernels of
                                         paths beyond 128 waypoints only (the shorter paths' kernels do not compile the feature in: nothing to refine away there); in a
                                         launch of mixed lengths (pqp_path_solve_var) this level and the three intervals above follow the launch's n_max, not the QP's own n */
    /* smoother QP weights (src/config/planning_flags.cpp:51-61) */
    double tension2_deviation_weight;        /* 0.005 */
    double tension2_curvature_weight;        /* 1     */
    double tension2_curvature_rate_weight;   /* 10    */
    double cartesian_curvature_weight;       /* 1     */
    double cartesian_curvature_rate_weight;  /* 50    */
    double cartesian_deviation_weight;       /* 0     */
} pqp_params;

typedef struct pqp_sizes {
    int32_t n, state, control, precise, slack, vars, cons, nnz_a, nnz_p;
} pqp_sizes;

typedef struct pqp_handle pqp_handle;

void pqp_default_params(pqp_params* p);
/* the defaults with the engine's production solver setting (1e-4 + KKT-verified polish; pqp_defaults.hpp) */
void pqp_production_params(pqp_params* p);
const char* pqp_last_error(void);
const char* pqp_version(void);

/* One handle per GPU.  max_batch / max_n size the device workspaces (they grow on demand). */
int pqp_create(pqp_handle** h, const pqp_params* params, int device, int max_batch, int max_n);
int pqp_destroy(pqp_handle* h);
int pqp_set_params(pqp_handle* h, const pqp_params* params);
/* Handle options (not solver settings: results never depend on them).
 *   PQP_OPT_STORE_WARM (default 1)     keep the final primal/dual iterate of every solve on the handle: what warm == 1 and
 *                                      pqp_path_get_solution read (BaseSolver keeps its OSQP workspace the same way,
 *                                      base_solver.hpp:62).  0 saves the write when every call is a complete optimizePath.
 *   PQP_OPT_ORDER_BY_COST (default 0)  start the QPs of a batch most-expensive-first, by the reduced-KKT solves and
 *                                      factorisations each QP needed in the handle's previous solve of the same batch and n
 *                                      (a planner re-solves nearly the same scenarios every cycle).  The first solve of a shape
 *                                      runs in index order.  On the lane-per-QP kernel (batches that put a wavefront on nearly every SIMD) the same option groups
 *                                      the QPs into wavefronts of similar work - see PQP_OPT_STREAM_BATCH below.
 *   PQP_OPT_RESERVE_CUS (default 0)    compute units the path QP's persistent workgroups leave free.  Their wavefronts own a SIMD's whole
 *                                      register file, so kernels of another stream (the smoother chain of the next batch, configs[4]) only
 *                                      get onto the chip when a unit is left to them.
 *   PQP_OPT_STREAM_BATCH (default -1 = by measurement: pqp_stream_batch_default(n) - the measured crossover of the two kernels on one MI355X, one launch after the
 *                                      other: 15 360 x max(1, n / 80)^1.5 up to 128 waypoints (15 k QPs at 80, 29 k at 120), 0.75 n^2 up to 256 (where the lane-per-
 *                                      waypoint kernel runs half as many QPs at a time), 48 n beyond; profiles/r06ay_*, r06az_*; 0: never)
 *                                      cold solves (warm == 0) of at least this many QPs on a handle with polish != 0 and
 *                                      PQP_OPT_STORE_WARM off run on the lane-per-QP kernel (one QP per lane, 64 per wavefront, the
 *                                      per-waypoint state streamed through a batch-interleaved workspace of 240 n bytes per QP in HBM;
 *                                      csrc/pqp_path_lq.hpp): the path QP as a linear-quadratic control problem, interior-point rounds +
 *                                      active-set rounds whose last round is the KKT test, i.e. the same exact optimum as the lane-per-
 *                                      waypoint kernel's verified polish.  It wins where the batch fills the chip's 65 536 lanes (65 536
 *                                      QPs of 80 waypoints: 1.9x); it keeps no warm state
 *                                      (hence the PQP_OPT_STORE_WARM condition), iters[] counts its interior-point iterations and info[] =
 *                                      {row residual, complementarity, iterations of the first pass, iterations, solved passes, active-set
 *                                      rounds of the first pass, Riccati sweeps, active-set rounds}.  PQP_OPT_ORDER_BY_COST on this kernel (round 5; batches of at least
 *                                      768 wavefronts, not with PQP_OPT_CARRY_CYCLES): the wavefronts of a solve hold QPs that ran the same interior-point iterations and
 *                                      active-set rounds per pass in the handle's previous solve of the shape - a wavefront runs every phase as often as its slowest lane:
 *                                      65 536 QPs 10.5 -> 9.7 ms with the identical batch's counts, 11.2 -> 10.4 ms on jittered planning cycles, 98 304 QPs 17.5 -> 13.0 ms
 *                                      (profiles/r05g_stream_sorted_probe.txt, r05i_*); HBM traffic 1.31x -> 1.03x the algorithmic bytes.
 *                                      Round 6: in such a sorted launch (the second solve of a shape onwards) a re-linearised pass first tries the previous pass's
 *                                      active set on the new transition rows - up to three active-set rounds, a confirmed set being the KKT test of that pass's
 *                                      QP - before the interior-point rounds get their turn (98 % of the bench's QPs confirm: 14.0 instead of 17.5 sweeps per path,
 *                                      65 536 QPs 7.7 -> 9.0 M paths/s with two launches in flight, profiles/r06ae_*).  The optimum is the same one; reached through other
 *                                      arithmetic it agrees with an unsorted launch's to the rounds' tolerances (< 5e-7 in l, psi, kappa), and a sorted launch
 *                                      reproduces the previous sorted launch bit for bit.
 *   PQP_OPT_CARRY_CYCLES (default 0)   a cold call (warm == 0, lin == NULL) starts the FIRST pass of QP k from the optimum QP k had in the handle's
 *                                      previous solve of the same batch and n instead of cold (lane-per-waypoint kernel: from the warm state it then keeps
 *                                      - final iterate, equilibration, active set, kept per waypoint: counts per QP may change between the calls; lane-per-QP kernel: from its workspace) - a planner re-solves
 *                                      nearly the same scenarios cycle after cycle; the reference constructs a fresh BaseSolver every cycle
 *                                      (path_optimizer.cpp:138).  The optimum returned is the same (unique; it agrees with the cold solve to the
 *                                      1e-7 of the KKT test); on scenarios that moved by 5 %: configs[1] 4.0 M instead of 3.1 M paths/s and no stragglers
 *                                      (profiles/r03n_seed_sweep.txt, r03y_seed_sweep.txt); lane-per-QP kernel 14 -> 9 interior-point iterations per path.
 *                                      (On that kernel the option switches the sorted launches of PQP_OPT_ORDER_BY_COST off - a slot's workspace belongs to the QP that sat there - and
 *                                      since round 6 those are the faster of the two from 768 wavefronts on: 65 536 QPs 7.8 M paths/s sorted against 6.6 M carried, profiles/r06am_bench_n1.json.)
 *                                      A QP that differs wildly from its slot's previous one is still solved (the start is then merely poor).
 *                                      Value k = 2..64 ("tails", lane-per-waypoint kernel, needs PQP_OPT_ORDER_BY_COST): only the QPs that were among the most
 *                                      expensive 1 / k of the handle's previous solve of the shape start from their previous optimum, all others start
 *                                      cold - a launch lasts as long as its slowest QPs, and those are the ones a cold active-set search wanders on
 *                                      (a carried QP keeps its cold cost key, one bin less per cycle, so that it stays among the carried ones).  configs[1], one
 *                                      launch at a time: k = 8 2.47 M against 2.01 M paths/s cold, the slowest QP 29 instead of 48 reduced solves
 *                                      (profiles/r05c_*).
 *                                      The exact TensionSmoother / postSmooth kernels (polish == 1) honour it too: a line's active-set rounds start from
 *                                      the set its slot ended with in the previous solve of the shape (16-31 rounds from the cold start, 2-3 from there).
 *   PQP_OPT_STREAM_STAGED (default -1 = by launch size)
 *                                      the lane-per-QP kernel has two forms.  A launch that fills the chip (768 wavefronts = 49 152 QPs or more) is bound by HBM's throughput:
 *                                      its workspace is laid out [field][lane], every access one contiguous line, a sweep's records prefetched one waypoint ahead in registers.
 *                                      A launch that leaves SIMDs idle waits for its loads: its workspace is laid out in 16-byte chunks per lane and the sweeps' records are staged in
 *                                      LDS two waypoints ahead by LDS-direct loads (global_load_lds_dwordx4) - +29 ... 33 % at 24 576 / 32 768 QPs of 80 waypoints one launch at a
 *                                      time, -3 ... 4 % where the chip is full (profiles/r06au_*).  Same arithmetic, same paths.  0 / 1 force one form (tests, measurements).
 *   PQP_OPT_CHAIN_GRAPH (default 0)    pqp_optimize_path_device replays a captured hipGraph: the third call with the same arguments (pointers, sizes, configuration,
 *                                      parameters of both handles) captures its ~25 launches on the handles' streams, later calls with those arguments are ONE
 *                                      hipGraphLaunch - the gaps between the launches were 13 % of the chain.  Results are bit-identical; any (re)allocation inside
 *                                      the library, or other arguments, falls back to plain launches (and a new capture).  Set it on the path handle.
 *                                      A replay runs on the path handle's stream; it is fenced against the smoother handle's stream on both sides
 *                                      (work the caller enqueues there stays ordered as with plain launches).  Value 2: no fences (-3.6 % time) -
 *                                      the caller guarantees that the smoother handle's stream carries no other work while chains are in flight. */
typedef enum pqp_option { PQP_OPT_STORE_WARM = 1, PQP_OPT_ORDER_BY_COST = 2, PQP_OPT_RESERVE_CUS = 3, PQP_OPT_STREAM_BATCH = 4, PQP_OPT_CARRY_CYCLES = 5, PQP_OPT_CHAIN_GRAPH = 6, PQP_OPT_STREAM_STAGED = 7 } pqp_option;
int pqp_set_option(pqp_handle* h, int option, int value);
int pqp_get_stream(pqp_handle* h, void** hip_stream);   /* hipStream_t */
/* The handle's stream is created non-blocking: work the caller enqueued on ANOTHER stream (the inputs of a *_device call produced by
 * its own kernels or copies; NULL = the default stream) is not ordered before the handle's launches unless the caller says so:
 * pqp_stream_wait makes everything enqueued on the handle from now on wait for what is enqueued on `hip_stream` up to now. */
int pqp_stream_wait(pqp_handle* h, void* hip_stream);
/* Ordering between two handles (= two streams) without the host waiting - BASELINE configs[4]: the smoother QP of scenario batch k on
 * one handle, the path QP of batch k on another, the smoother of batch k + 1 overlapping the path QP of batch k.  pqp_mark records
 * the handle's event `slot` (0..7) behind everything enqueued on it so far; pqp_wait_mark makes everything enqueued on `h` from
 * now on wait for `other`'s mark `slot` as it was recorded last. */
int pqp_mark(pqp_handle* h, int slot);
int pqp_wait_mark(pqp_handle* h, pqp_handle* other, int slot);
int pqp_sync(pqp_handle* h);

/* BaseSolver ctor (base_solver.cpp:15-39): problem sizes.  `s` = the n arclengths of the path
 * (only read when rough_constraints_far_away), may be NULL otherwise.  Pure integer host function. */
int pqp_path_sizes(const pqp_params* params, int n, const double* s, pqp_sizes* out);

/* Value-independent sparsity of A in CSC order (17N-5 slots when precise == n) and the columns of
 * the diagonal P, in the REFERENCE variable / row numbering (base_solver.cpp:154-209,127-143).
 * Computed by a HIP kernel.  rows[nnz_a], colptr[vars+1], pcols[nnz_p] are HOST buffers. */
int pqp_path_pattern(pqp_handle* h, int n, int precise, int32_t* rows, int32_t* colptr, int32_t* pcols);

/* setCost + setConstraints for a batch: values in the pattern's order.
 *   a_val [batch][nnz_a]   p_val [batch][nnz_p]   lower/upper [batch][cons]
 * All QPs of one call share n and `precise`.  Host-pointer and device-pointer variants. */
int pqp_path_assemble(pqp_handle* h, int batch, int n, int precise, const double* ref, const double* lin,
                      const double* bounds, const double* scal,
                      double* a_val, double* p_val, double* lower, double* upper);
int pqp_path_assemble_device(pqp_handle* h, int batch, int n, int precise, const double* ref,
                             const double* lin, const double* bounds, const double* scal,
                             double* a_val, double* p_val, double* lower, double* upper);

/* Assemble + ADMM solve + unpack, `passes` re-linearised warm re-solves fused in one launch
 * (PathOptimizer::optimizePath == passes 1).
 *   warm == 0: cold start (BaseSolver::solve).  warm == 1: start from the primal/dual/rho the handle
 *   kept from its previous pqp_path_solve* call with the same batch and n
 *   (BaseSolver::updateProblemFormulationAndSolve: pass the previous output as `lin`).
 *   status[batch], iters[batch] (total ADMM iterations over all passes) may be NULL.
 *   info (may be NULL) [batch][PQP_INFO_STRIDE]: primal residual, dual residual, final rho, ADMM iterations of the
 *   last pass, number of accepted polishes, total reduced-KKT solves (ADMM iterations + polish refinement),
 *   number of factorisations, reserved.
 *   n >= 2 waypoints, any batch.  Up to 512 waypoints: one lane per waypoint (1, 2, 4 or 8 wavefronts per QP); beyond (and for large batches,
 *   PQP_OPT_STREAM_BATCH): one lane per QP, which returns exact optima in every solver setting (zero residuals meet OSQP's termination test at
 *   any eps) and keeps no warm state: beyond 512 waypoints warm == 1 solves the QP around `lin` cold - same optimum - and
 *   pqp_path_get_solution is not available.  Which solver runs is decided
 *   by the handle's pqp_params: pqp_default_params = the reference's OSQP setting (eps 2e-3, no polish, infeasibility certificate),
 *   pqp_production_params = eps 1e-4 + KKT-verified polish.  When status[q] != SOLVED (the reference's solve() returns false there and
 *   leaves its output vector untouched) out[q] is defined but kernel-specific: the lane-per-waypoint kernel leaves its last iterate there;
 *   the lane-per-QP kernel the optimum of the last pass that WAS solved, or - when none was - the reference line itself (l = dpsi = 0).
 *   iters[] and info[] differ between the two kernels as well (above; PQP_OPT_STREAM_BATCH): pqp_last_path_kernel says which one served
 *   the handle's last solve, so that a caller can tell what it received.
 *   A start curvature scal[q][2] outside the curvature box by no more than OSQP's primal tolerance eps_abs + eps_rel * bound is projected onto the
 *   box by both kernels (strictly such a QP has no feasible point; OSQP at that eps calls it solved with a point that misses the row by that
 *   little); outside by more: PQP_STATUS_PRIMAL_INFEASIBLE.
 *   A QP with a collision box whose lower bound exceeds its upper bound is refused as OSQP refuses it at setup: the host-pointer
 *   entry points do not launch it and report PQP_STATUS_PRIMAL_INFEASIBLE (out[q] = 0); the *_device entry points do not
 *   validate their inputs (there such a row is pinned to its upper bound). */
int pqp_path_solve(pqp_handle* h, int batch, int n, const double* ref, const double* lin,
                   const double* bounds, const double* scal, int passes, int warm,
                   double* out, int32_t* status, int32_t* iters, double* info);
int pqp_path_solve_device(pqp_handle* h, int batch, int n, const double* ref, const double* lin,
                          const double* bounds, const double* scal, int passes, int warm,
                          double* out, int32_t* status, int32_t* iters, double* info);
/* The same with a waypoint count per QP: n_of[batch] (device pointer, each <= n_max); every array keeps the stride n_max.
 * This is what a road cut short by an obstacle needs (ReferencePathImpl::updateBoundsImproved resizes reference_states_ to the
 * blocked waypoint, reference_path_impl.cpp:225-228; pqp_corridor_bounds reports it as n_valid).  A QP with n_of < 2 is left
 * untouched with status PQP_STATUS_UNSOLVED. */
int pqp_path_solve_var_device(pqp_handle* h, int batch, int n_max, const int32_t* n_of, const double* ref, const double* lin,
                              const double* bounds, const double* scal, int passes, int warm, double* out, int32_t* status,
                              int32_t* iters, double* info);
/* host pointers (n_of included); rows of `out` beyond a QP's own count come back as zeros */
int pqp_path_solve_var(pqp_handle* h, int batch, int n_max, const int32_t* n_of, const double* ref, const double* lin, const double* bounds,
                       const double* scal, int passes, int warm, double* out, int32_t* status, int32_t* iters, double* info);

/* Primal / dual solution of the handle's last solve in the REFERENCE numbering (OsqpEigen::Solver::
 * getSolution(), base_solver.cpp:89,112): x [batch][vars], y [batch][cons].  HOST buffers; either may be NULL. */
int pqp_path_get_solution(pqp_handle* h, int batch, int n, int precise, double* x, double* y);

/* constrainAngle (include/tools/tools.hpp:24-35: wrap to [-pi, pi], both ends inclusive) exactly as the kernels of this library evaluate
 * it for the end-heading rule (base_solver.cpp:254-258) and the unpack (base_solver.cpp:272-275): out[i] = constrainAngle(in[i]).
 * DEVICE pointers.  It exists so that the helper can be pinned bit for bit against the reference's own template
 * (oracle/_ref/libref_types.so, tests/test_ref_types.py). */
int pqp_constrain_angle_device(pqp_handle* h, int count, const double* in, double* out);

/* ---- one node, several GPUs -----------------------------------------------------------------------------------
 * The QPs of a batch are independent: the batch is cut into contiguous shards (pqp_shard_range: the first total % world shards get one
 * QP more), every shard has its own handle - GPU, stream, workspaces - and its own host thread; pqp_multi_path_solve moves each
 * shard's slice of the caller's HOST arrays to its GPU, solves it there (pqp_path_solve / pqp_path_solve_var when n_of is given) and
 * brings the paths back.  No collective: with the consumer on the host, per-GPU copies beat a device-side gather.  devices[n_shards]
 * = the device ordinal of every shard (NULL: 0, 1, ...; the same ordinal may appear twice: two handles on one GPU).  The reference
 * has no counterpart (it solves one path per call, base_solver.cpp:56-95). */
typedef struct pqp_multi pqp_multi;
void pqp_shard_range(int total, int world, int rank, int* first, int* count);
int pqp_multi_create(pqp_multi** m, const pqp_params* params, int n_shards, const int* devices, int max_batch_per_shard, int max_n);
int pqp_multi_destroy(pqp_multi* m);
int pqp_multi_shards(const pqp_multi* m);
pqp_handle* pqp_multi_handle(pqp_multi* m, int shard);       /* a shard's own handle (device-resident use, options, timing) */
/* every shard's handle gets the option (PQP_OPT_CARRY_CYCLES: a call whose batch and n equal the previous call's - no lin - starts shard by shard
 * from what the shard's handle kept: the same scenarios one planning cycle later) */
int pqp_multi_set_option(pqp_multi* m, int option, int value);
int pqp_multi_path_solve(pqp_multi* m, int batch, int n, const int32_t* n_of, const double* ref, const double* lin, const double* bounds,
                         const double* scal, int passes, double* out, int32_t* status, int32_t* iters, double* info);
/* The exchange step of SURVEY.md 8e / north_star ("RCCL over xGMI only to gather results") in the C ABI: after pqp_multi_path_solve every shard's
 * paths are still in its GPU's memory; this gathers them over RCCL so that full_out[g] - a buffer of [batch][n][PQP_OUT_STRIDE] doubles in the memory
 * of shard g's GPU - holds the whole batch in the caller's order, for a consumer that lives on the GPUs (the host copy of pqp_multi_path_solve is
 * unaffected).  batch and n as in the preceding solve.  One device per shard; librccl.so is dlopen'ed by the first call, a caller that never
 * gathers does not need it.  The reference has no counterpart (one path per call). */
int pqp_multi_gather_paths(pqp_multi* m, int batch, int n, double* const* full_out);
/* ranks of that gather's communicator as RCCL counts them (ncclCommCount); 0 before the first gather, < 0: error */
int pqp_multi_gather_ranks(pqp_multi* m);

/* ---- reference-line smoothing QPs; the solver settings of the handle apply (the reference runs
 *      them at OSQP's default eps 1e-3: tension_smoother_2.cpp:32-36, tension_smoother.cpp:61-65, reference_path_smoother.cpp:533-537).
 *      All lists are [batch][n] fp64; out_s is the cumulative chord length of the smoothed points. ------------------------- */
/* TensionSmoother2::osqpSmooth   src/reference_path_smoother/tension_smoother_2.cpp:20-72 (default smoothing_method) */
int pqp_smooth_tension2(pqp_handle* h, int batch, int n, const double* x_list, const double* y_list, const double* angle_list,
                        const double* k_list, const double* s_list, double* out_x, double* out_y, double* out_s, int32_t* status,
                        int32_t* iters);
int pqp_smooth_tension2_device(pqp_handle* h, int batch, int n, const double* x_list, const double* y_list, const double* angle_list,
                               const double* k_list, const double* s_list, double* out_x, double* out_y, double* out_s,
                               int32_t* status, int32_t* iters, double* info);
/* TensionSmoother::osqpSmooth    src/reference_path_smoother/tension_smoother.cpp:49-100; clearance = Map::getObstacleDistance
 * at each input point (the distance-map lookup itself, tension_smoother.cpp:168, stays on the caller's side; pqp_clearance_device does it).
 * n >= 4 points, no upper bound (as the reference: one point per metre of line).  Up to ~166 points a handle in the reference's ADMM setting
 * (polish == 0) runs OSQP's iteration on the 9 x 9-block core; beyond that core's LDS capacity - and for polish == 1 at any size - the QP is
 * solved exactly (iters = 0), which meets OSQP's termination test at any eps.  The same rule holds for the other two smoothers: where the
 * generic core cannot hold the QP (TensionSmoother2: more than 256 points; postSmooth: more than ~340 layers) every handle gets the exact kernel. */
int pqp_smooth_tension(pqp_handle* h, int batch, int n, const double* x_list, const double* y_list, const double* angle_list,
                       const double* clearance, double* out_x, double* out_y, double* out_s, int32_t* status, int32_t* iters);
int pqp_smooth_tension_device(pqp_handle* h, int batch, int n, const double* x_list, const double* y_list, const double* angle_list,
                              const double* clearance, double* out_x, double* out_y, double* out_s, int32_t* status, int32_t* iters,
                              double* info);
/* ReferencePathSmoother::postSmooth QP   src/reference_path_smoother/reference_path_smoother.cpp:526-558,582-636:
 * layers_s, lb, ub [batch][m] (layers_bounds_), vehicle_l [batch] (vehicle_l_wrt_smoothed_ref_); out_l [batch][m] = QPSolution(i) */
int pqp_post_smooth(pqp_handle* h, int batch, int m, const double* layers_s, const double* lb, const double* ub,
                    const double* vehicle_l, double* out_l, int32_t* status, int32_t* iters);
int pqp_post_smooth_device(pqp_handle* h, int batch, int m, const double* layers_s, const double* lb, const double* ub,
                           const double* vehicle_l, double* out_l, int32_t* status, int32_t* iters, double* info);

/* Which kernel served the handle's last pqp_path_solve* call (the dispatch depends on batch size, n, warm and the options above): a
 * pqp_path_kernel, 0 before the first solve; < 0: error. */
typedef enum pqp_path_kernel { PQP_KERNEL_NONE = 0, PQP_KERNEL_LANE_PER_WAYPOINT = 1, PQP_KERNEL_LANE_PER_QP = 2 } pqp_path_kernel;
int pqp_last_path_kernel(pqp_handle* h);
/* PQP_OPT_STREAM_BATCH's default for paths of n waypoints (above). */
int pqp_stream_batch_default(int n);
/* GPU time (ms, hipEvent) of the handle's last solve / assemble launch. */
int pqp_last_kernel_ms(pqp_handle* h, float* ms);
/* the same for the last `count` launches of this handle (oldest first; at most 256): the events are recorded on the handle's stream
 * around every launch and read here after a stream synchronise, so measuring puts nothing between the launches themselves */
int pqp_kernel_ms_history(pqp_handle* h, float* ms, int count);

/* ---- corridor bounds from the obstacle distance map ---------------------------------------------
 * ReferencePath::updateBounds(const Map&)  src/data_struct/reference_path.cpp:61  ->  ReferencePathImpl::updateBoundsImproved
 * (reference_path_impl.cpp:177-230), getClearanceWithDirectionStrict (:232-312), Map::getObstacleDistance (src/tools/Map.cpp:16-22),
 * getDirectionalProjectionByNewton (src/tools/tools.cpp:156-189).  Produces the `bounds` input of pqp_path_solve on the device. */
typedef struct pqp_grid_geometry {       /* grid_map::GridMap geometry of the "distance" layer */
    int32_t rows, cols;                  /* getSize(): cells along x, along y */
    double resolution;
    double length_x, length_y;           /* getLength() = size * resolution */
    double pos_x, pos_y;                 /* getPosition(): map centre */
} pqp_grid_geometry;

typedef struct pqp_corridor_params {
    double front_length, rear_length;    /* 3.9, -1.0   planning_flags.cpp:20,18 */
    double car_width, safety_margin;     /* 2.0, 0.3    planning_flags.cpp:10,14 */
    double epsilon;                      /* 1e-6        planning_flags.cpp:108 (isEqual) */
    double search_radius, delta_s, smaller_ds, search_range, min_space;   /* 0.5, 0.3, 0.05, 6.0, 0.2   reference_path_impl.cpp:238-304 */
    double projection_window;            /* 5.0         reference_path_impl.cpp:194 */
} pqp_corridor_params;

void pqp_corridor_default_params(pqp_corridor_params* p);
/* ref      [batch][n][5]  s, k, heading, x, y of the reference states (the layout pqp_path_solve takes)
 * spline   [batch][9][m]  row 0: knots s; rows 1-4: y, a, b, c of x(s); rows 5-8: y, a, b, c of y(s)  (tk::spline members m_y, m_a, m_b, m_c)
 * spline_ext [batch][4]   m_b0, m_c0 of x(s), then of y(s)
 * dist     [n_maps][cols][rows] float: the layer in Eigen's column-major order (grid_map::Matrix = Eigen::MatrixXf)
 * map_of   [batch] map index of each scenario, or NULL (all use map 0)
 * n_of     [batch] reference states of each scenario (<= n; n is then the array stride), or NULL (all have n)
 * bounds   [batch][n][6]  f_lb f_ub r_lb r_ub c_lb c_ub for every waypoint of the scenario;  n_valid [batch] = index of the first
 *          waypoint whose front or rear interval is empty (the reference cuts the path there and sets isBlocked(), :219-223),
 *          the scenario's own count if none.  n_valid is what pqp_path_solve_var_device takes as n_of. */
int pqp_corridor_bounds_device(pqp_handle* h, int batch, int n, int m, const double* ref, const int32_t* n_of, const double* spline,
                               const double* spline_ext, const float* dist, const int32_t* map_of, const pqp_grid_geometry* geom,
                               const pqp_corridor_params* prm, double* bounds, int32_t* n_valid);
int pqp_corridor_bounds(pqp_handle* h, int batch, int n, int m, const double* ref, const int32_t* n_of, const double* spline,
                        const double* spline_ext, const float* dist, int n_maps, const int32_t* map_of, const pqp_grid_geometry* geom,
                        const pqp_corridor_params* prm, double* bounds, int32_t* n_valid);

/* ---- reference states from the reference line's spline + the vehicle's initial error -------------------
 * ReferencePathImpl::buildReferenceFromSpline(delta_s_smaller, delta_s_larger)  src/data_struct/reference_path_impl.cpp:314-338
 *   (called with output_spacing / 2, output_spacing = 0.15, 0.3 at path_optimizer.cpp:119; dynamic = FLAGS_enable_dynamic_segmentation)
 * PathOptimizer::processInitState                                              src/path_optimizer.cpp:73-85
 * spline, spline_ext as above; max_s [batch] = ReferencePath::getLength(); start [batch][3] = vehicle start state x, y, heading (or NULL)
 * ref [batch][n_max][5] = s, k, heading, x, y;  count [batch] = states the loop produces (when it exceeds n_max only n_max rows
 * were written);  init_err [batch][2] = initial_offset, initial_heading_error (NULL or needs start): scal[0..1] of pqp_path_solve */
int pqp_reference_states_device(pqp_handle* h, int batch, int n_max, int m, const double* spline, const double* spline_ext,
                                const double* max_s, const double* start, double ds_small, double ds_large, int dynamic, double* ref,
                                int32_t* count, double* init_err);
int pqp_reference_states(pqp_handle* h, int batch, int n_max, int m, const double* spline, const double* spline_ext, const double* max_s,
                         const double* start, double ds_small, double ds_large, int dynamic, double* ref, int32_t* count,
                         double* init_err);

/* ---- lateral offsets on a line -> points with chord-length abscissae -----------------------------------------------------------
 * The tail of ReferencePathSmoother::postSmooth  src/reference_path_smoother/reference_path_smoother.cpp:559-573: x_list(i) =
 * x_s(s_i) + l_i cos(heading(s_i) + pi/2), y_list likewise, s_list = accumulated chord length - the knots of the final reference
 * line (pqp_spline_fit, :574-576; s_list.back() is its length).
 * spline [batch][9][m_spline], spline_ext [batch][4]: the smoothed line; at_s, l [batch][m] = layers_s_list_ and pqp_post_smooth's
 * out_l; m_of [batch] = points per scenario (pqp_dp_corridor's count) or NULL  ->  x, y, s [batch][m] (rows beyond m_of[q] untouched). */
int pqp_offsets_to_points_device(pqp_handle* h, int batch, int m_spline, int m, const double* spline, const double* spline_ext, const double* at_s,
                                 const double* l, const int32_t* m_of, double* x, double* y, double* s);
int pqp_offsets_to_points(pqp_handle* h, int batch, int m_spline, int m, const double* spline, const double* spline_ext, const double* at_s,
                          const double* l, const int32_t* m_of, double* x, double* y, double* s);

/* ---- length of the reference line up to the target state ----------------------------------------------------------------------
 * PathOptimizer::setReferencePathLength  src/path_optimizer.cpp:87-104 (called in processReferencePath, :117, before
 * buildReferenceFromSpline): when the target state lies behind the end of the line - x <= 0 in the frame of the line's end state,
 * global2Local tools.cpp:57-64 - the line is cut at the target's projection (getProjection tools.cpp:66-126).
 * length [batch] = ReferencePath::getLength(), target [batch][3] = x, y, heading  ->  length_out [batch], the max_s of
 * pqp_reference_states. */
int pqp_reference_length_device(pqp_handle* h, int batch, int m, const double* spline, const double* spline_ext, const double* length,
                                const double* target, double* length_out);
int pqp_reference_length(pqp_handle* h, int batch, int m, const double* spline, const double* spline_ext, const double* length,
                         const double* target, double* length_out);

/* ---- input points -> dense raw reference line ----------------------------------------------------------------------------------
 * ReferencePathSmoother::bSpline  src/reference_path_smoother/reference_path_smoother.cpp:490-521 (first step of
 * ReferencePathSmoother::solve, :31-45): the input points are the control points of a clamped B-spline of degree 3 / 4 / 5 (average
 * point spacing > 10 m / > 5 m / else), sampled at t = 0, 1/length, 2/length, ... while t < 1 and at t = 1 (about one point per
 * metre); s = accumulated chord length.  The spline itself is the third-party tinyspline (not in the reference tree): its clamped
 * knot vector and de Boor evaluation are restated, see oracle/corridor_oracle.py.
 * points [batch][p_max][2], n_points [batch] (fewer than 4: count = 0, the reference's "Few reference points")
 *   ->  x, y, s [batch][n_max] = x_list_, y_list_, s_list_;  count [batch] = points the loop produces (when it exceeds n_max only
 * n_max were written).  pqp_spline_fit of (s, x, y) then gives the splines pqp_segment_raw_reference samples. */
int pqp_bspline_resample_device(pqp_handle* h, int batch, int p_max, int n_max, const double* points, const int32_t* n_points, double* x,
                                double* y, double* s, int32_t* count);
int pqp_bspline_resample(pqp_handle* h, int batch, int p_max, int n_max, const double* points, const int32_t* n_points, double* x, double* y,
                         double* s, int32_t* count);

/* ---- raw reference line -> the input lists of the smoother QPs ---------------------------------------------------------------
 * ReferencePathSmoother::segmentRawReference  src/reference_path_smoother/reference_path_smoother.cpp:48-85 (called by
 * TensionSmoother::smooth, tension_smoother.cpp:21-26): the splines of the raw point list (pqp_spline_fit of s_list_, x_list_,
 * y_list_) sampled at 0, delta_s, 2 delta_s, ... up to AND INCLUDING the first abscissa >= max_s (the reference's loop, :64-67, with
 * delta_s = 1.0: the last sample lies beyond the line and is extrapolated), angle = atan2(dy, dx), k = (dx ddy - dy ddx) / pow(dx^2 + dy^2, 1.5).
 * x, y, s, angle, k [batch][n_max] are exactly pqp_smooth_tension2 / pqp_smooth_tension's inputs; count [batch] = samples the
 * loop produces (when it exceeds n_max only n_max were written). */
int pqp_segment_raw_reference_device(pqp_handle* h, int batch, int n_max, int m, const double* spline, const double* spline_ext,
                                     const double* max_s, double delta_s, double* x, double* y, double* s, double* angle, double* k,
                                     int32_t* count);
int pqp_segment_raw_reference(pqp_handle* h, int batch, int n_max, int m, const double* spline, const double* spline_ext, const double* max_s,
                              double delta_s, double* x, double* y, double* s, double* angle, double* k, int32_t* count);

/* ---- natural cubic spline through the knots ----------------------------------------------------------
 * tk::spline::set_points  src/tools/spline.cpp:161-249 (behaviour: the natural cubic spline, f'' = 0 at both ends, linear continuation beyond
 * them), as called on the smoothed reference line
 * (tension_smoother.cpp:36-38, reference_path_smoother.cpp:58-59,574-576, reference_path_impl.cpp:349-350).
 * s, x, y [batch][m] (s strictly increasing, m >= 3)  ->  spline [batch][9][m], spline_ext [batch][4] in the layout
 * pqp_reference_states / pqp_corridor_bounds take.  The kernel solves the textbook moment equations by a Thomas recurrence of its own (not
 * the reference's band-matrix LU): the coefficients agree with the reference's to round-off (< 1e-13 of a row's largest, tested against the
 * reference's compiled tk::spline); knots and values are copied. */
int pqp_spline_fit_device(pqp_handle* h, int batch, int m, const double* s, const double* x, const double* y, double* spline,
                          double* spline_ext);
int pqp_spline_fit(pqp_handle* h, int batch, int m, const double* s, const double* x, const double* y, double* spline, double* spline_ext);

/* ---- layered DP corridor search --------------------------------------------------------------------
 * ReferencePathSmoother::graphSearchDp  src/reference_path_smoother/reference_path_smoother.cpp:142-295 (+ calculateCostAt :107-140),
 * between the smoother QP and the postSmooth QP: its outputs layers_s / lb / ub / vehicle_l are the inputs of pqp_post_smooth. */
typedef struct pqp_dp_params {
    double lateral_range;            /* 10.0  FLAGS_search_lateral_range        planning_flags.cpp:38 */
    double longitudinal_spacing;     /* 1.5   FLAGS_search_longitudial_spacing  :40 */
    double lateral_spacing;          /* 0.6   FLAGS_search_lateral_spacing      :42 */
    double car_width;                /* 2.0   :10  (search_threshold = car_width / 2 + 0.2) */
} pqp_dp_params;
void pqp_dp_default_params(pqp_dp_params* p);
/* spline, spline_ext, dist, map_of, geom as for pqp_corridor_bounds; length [batch] = reference->getLength(); start [batch][3] = vehicle
 * start state x, y, heading.  layers_s, lb, ub [batch][max_layers]; count [batch] = layers of the corridor (0: graphSearchDp returns
 * false or no node of the first layer is reachable; -1: the line needs more than max_layers layers); vehicle_l [batch]. */
int pqp_dp_corridor_device(pqp_handle* h, int batch, int m, int max_layers, const double* spline, const double* spline_ext,
                           const double* length, const double* start, const float* dist, const int32_t* map_of,
                           const pqp_grid_geometry* geom, const pqp_dp_params* prm, double* layers_s, double* lb, double* ub,
                           int32_t* count, double* vehicle_l);
int pqp_dp_corridor(pqp_handle* h, int batch, int m, int max_layers, const double* spline, const double* spline_ext, const double* length,
                    const double* start, const float* dist, int n_maps, const int32_t* map_of, const pqp_grid_geometry* geom,
                    const pqp_dp_params* prm, double* layers_s, double* lb, double* ub, int32_t* count, double* vehicle_l);

/* ---- the whole of PathOptimizer::solve as one device-resident call -----------------------------------------------------------------
 * PathOptimizer::solve  src/path_optimizer.cpp:34-71  =  ReferencePathSmoother::solve (reference_path_smoother.cpp:31-45: bSpline,
 * TensionSmoother2 QP, graphSearchDp, postSmooth QP) + processReferencePath (:106-122) + optimizePath (:124-161), for a batch of
 * scenarios that differ in every count (input points, raw-line points, smoother samples, DP layers, waypoints).  All pointers are DEVICE
 * pointers; nothing is copied to the host between the steps; the call returns when everything is enqueued (pqp_sync(h) waits).
 *   h  : the handle whose parameters the path QP uses (pqp_production_params or the reference's pqp_default_params);
 *   hs : the handle the two smoother QPs run on - the reference solves them at OSQP's default eps 1e-3 (tension_smoother_2.cpp:32-36),
 *        i.e. pqp_default_params with eps_abs = eps_rel = 1e-3 - or NULL: h's own parameters.  The two handles must live on the same device (PQP_ERR_INVALID otherwise); they are ordered by events of the chain's own (pqp_mark's slots stay the caller's).
 *   points [batch][p_max][2], n_points [batch]  PathOptimizer::solve's reference_points (fewer than 4: "Few reference points")
 *   start, target [batch][3]                    x, y, heading of the vehicle's start and target state (the PathOptimizer constructor)
 *   start_k [batch] or NULL                     curvature of the start state (NULL: 0, State's default)
 *   dist / map_of / geom                        the obstacle distance layer(s), as for pqp_corridor_bounds
 *   out [batch][n_max][7], n_out [batch]        the path (the SlState fields getOptimizedPath fills) and its waypoint count
 *   status [batch]                              pqp_status of the path QP (PQP_STATUS_UNSOLVED when the scenario never got there)
 *   stage [batch] or NULL                       where a scenario stopped: the reference's `return false` sites
 *   iters [batch] or NULL                       ADMM iterations of the path QP */
typedef enum pqp_chain_stage {
    PQP_CHAIN_OK = 0,
    PQP_CHAIN_FEW_POINTS = 1,            /* reference_path_smoother.cpp:33-36 */
    PQP_CHAIN_SMOOTHER_FAILED = 2,       /* tension_smoother.cpp:32-35 */
    PQP_CHAIN_SEARCH_FAILED = 3,         /* graphSearchDp returned false (:164-167, no reachable node) */
    PQP_CHAIN_SHORT_REFERENCE = 4,       /* postSmooth: fewer than 4 layers (:528-531) */
    PQP_CHAIN_POST_SMOOTH_FAILED = 5,    /* :555-558 */
    PQP_CHAIN_HEADING = 6,               /* initial heading error above 75 degrees (path_optimizer.cpp:113-116) */
    PQP_CHAIN_BLOCKED = 7,               /* the road is blocked before the second waypoint */
    PQP_CHAIN_PATH_QP_FAILED = 8,        /* "Solving failed!" (path_optimizer.cpp:143-156); status says why */
    PQP_CHAIN_CAPACITY = 9               /* a line needs more points / samples / layers / waypoints than pqp_chain_config allows */
} pqp_chain_stage;
typedef enum pqp_smoothing_method { PQP_SMOOTHING_TENSION2 = 0, PQP_SMOOTHING_TENSION = 1 } pqp_smoothing_method;
typedef struct pqp_chain_config {
    int32_t raw_max, sample_max, layer_max, n_max;   /* capacities per scenario: raw-line points (bSpline, about one per metre), 1 m samples of
                                                        the smoother QP, DP layers (1.5 m), waypoints of the path.  No upper bounds (the reference
                                                        has none); a line of more than ~2000 spline knots exceeds the LDS staging of the corridor
                                                        steps (PQP_ERR_CAPACITY from that step) */
    double output_spacing;               /* 0.3   FLAGS_output_spacing, planning_flags.cpp:106 */
    int32_t dynamic_segmentation;        /* 1     FLAGS_enable_dynamic_segmentation, :110 */
    double max_steering_angle;           /* 35 degrees, :22 */
    double smoothed_length_margin;       /* 3.0   tension_smoother.cpp:40: the smoothed line is declared 3 m longer than its last point */
    pqp_corridor_params corridor;
    pqp_dp_params dp;
    int32_t smoothing_method;            /* PQP_SMOOTHING_TENSION2  FLAGS_smoothing_method, planning_flags.cpp:27 (ReferencePathSmoother::create,
                                            reference_path_smoother.cpp:18-29).  TENSION: the clearance of the raw line's samples is looked up on
                                            the device (tension_smoother.cpp:168); on a smoother handle without polish = 1 (the reference's ADMM on the 3n-variable
                                            formulation) sample_max is bounded by what one CU's LDS holds (about 190) */
} pqp_chain_config;
void pqp_chain_default_config(pqp_chain_config* c);
int pqp_optimize_path_device(pqp_handle* h, pqp_handle* hs, const pqp_chain_config* cfg, int batch, int p_max, const double* points,
                             const int32_t* n_points, const double* start, const double* target, const float* dist, const int32_t* map_of,
                             const pqp_grid_geometry* geom, const double* start_k, double* out, int32_t* n_out, int32_t* status, int32_t* stage,
                             int32_t* iters);
/* Map::getObstacleDistance (src/tools/Map.cpp:16-22) at the points x_list, y_list [batch][n]: the `clearance` input of pqp_smooth_tension
 * (tension_smoother.cpp:168), gathered on the device from the same distance layer(s) pqp_corridor_bounds takes */
int pqp_clearance_device(pqp_handle* h, int batch, int n, const double* x_list, const double* y_list, const float* dist, const int32_t* map_of,
                         const pqp_grid_geometry* geom, double* clearance);
/* the steps of the chain that share one size per launch elsewhere, with a count per scenario (device pointers): a scenario with
 * fewer points is the same QP padded with decoupled dummies (same optimum); the spline table is padded with knots far beyond the line,
 * whose cubic coefficient is zero - term by term tk::spline's right-hand extrapolation */
int pqp_smooth_tension2_var_device(pqp_handle* h, int batch, int n_max, const int32_t* n_of, const double* x_list, const double* y_list,
                                   const double* angle_list, const double* k_list, const double* s_list, double* out_x, double* out_y,
                                   double* out_s, int32_t* status, int32_t* iters, double* info);
/* TensionSmoother::osqpSmooth (tension_smoother.cpp:49-100) with a point count per scenario (at least 4); clearance as for pqp_smooth_tension */
int pqp_smooth_tension_var_device(pqp_handle* h, int batch, int n_max, const int32_t* n_of, const double* x_list, const double* y_list,
                                  const double* angle_list, const double* clearance, double* out_x, double* out_y, double* out_s, int32_t* status,
                                  int32_t* iters, double* info);
int pqp_post_smooth_var_device(pqp_handle* h, int batch, int m_max, const int32_t* m_of, const double* layers_s, const double* lb, const double* ub,
                               const double* vehicle_l, double* out_l, int32_t* status, int32_t* iters, double* info);
int pqp_spline_fit_var_device(pqp_handle* h, int batch, int m_max, const int32_t* m_of, const double* s, const double* x, const double* y,
                              double* spline, double* spline_ext);

#ifdef __cplusplus
}
#endif
#endif /* PQP_H_ */
