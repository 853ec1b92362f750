// pqp_path_lane.hpp — the per-waypoint ("lane") formulation of the batched path-QP ADMM solve.
//
// One workgroup owns one QP at a time and one thread owns one waypoint (T = 64 * NW >= N threads, NW = 2 waves for
// the N = 80 / 120 configurations).  All iterates, the problem data, the penalty metrics and the factorisation of a
// QP live in registers (86 fp64 per waypoint; with the setup-time state and temporaries the kernel takes the whole
// 512-register budget of a lane, i.e. ONE wave per SIMD), neighbouring waypoints talk through LDS (exchange buffers 21 T doubles; with
// the polish save area and the parked Ruiz vectors 79 KB per QP at T = 128) or, on the device, through DPP operands when they sit in the
// same row of 16 lanes (Ctx::kDpp), and HBM is touched only to read the scenario and to write the result.  (Round-1 history: a
// 2-waypoints-per-lane layout needed > 512 VGPRs at the factorisation and spilled 2 GB per launch; see DESIGN.md.)
//
// What is computed (reference file:line relative to LiJiangnanBit/path_optimizer_2):
//   * assemble   — BaseSolver::setCost / setConstraints / getSoftBounds
//                  (src/solver/base_solver.cpp:119-148,150-261,290-295), per waypoint:
//                  the 5 non-trivial entries of I + ds*df_x, ds, the 3 transition right-hand sides,
//                  the soft collision boxes, the curvature box, the 2 end rows.
//   * ADMM       — the OSQP-paper iteration (Stellato et al. 2020; OsqpEigen::Solver::solve() called at
//                  base_solver.cpp:88,110) in UNSCALED coordinates with diagonal penalty metrics
//                  R = rho_i E_i^2 / c and Sigma = sigma / (c D_j^2) taken from the paper's modified Ruiz
//                  equilibration — algebraically the same iterates as the scaled iteration.
//   * KKT solve  — reduced SPD system (P + Sigma + A^T R A) x = rhs.  Slack and control variables are
//                  eliminated in closed form (they touch one row each); what remains is block
//                  tridiagonal in the 3-vector (l, psi, k) per waypoint and is solved by block cyclic
//                  reduction: log2 levels instead of a 6N-long dependency chain.
//   * unpack     — BaseSolver::getOptimizedPath (base_solver.cpp:263-288).
//   * re-linearise + warm re-solve — BaseSolver::updateProblemFormulationAndSolve (:97-117).
//
// Variable grouping per waypoint i:  x = (l_i, psi_i, k_i, v_i = u_{i-1}, sf_i, sr_i)
//   (the control that LEADS INTO waypoint i is stored with i, so transition row block T_i only needs
//    the previous waypoint's state; v_0 is a decoupled dummy).
// Row grouping per waypoint i: T_i (3 equality rows: A_{i-1} X_{i-1} + ds v_i e3 - X_i), K_i, F_i, R_i;
// the two end rows belong to the last waypoint and live in LDS.
//
// The same source compiles for the device (kernels in pqp_kernels.hip) and, for tests only, for the
// host, where tests/emu runs the phases lane by lane to check the algorithm against the oracle without
// a GPU.  The host build is test infrastructure: nothing in the product links it.
#pragma once
#include <math.h>
#include <stdint.h>

#include "../../include/pqp.h"

#define PQP_SAVE_STRIDE 32   /* doubles per thread of the polish save area */

#if defined(__HIPCC__)
#define PQP_HD __host__ __device__ __forceinline__
#else
#define PQP_HD inline
#endif

namespace pqp {

constexpr double kInfty = 1e30;        // OSQP_INFTY
constexpr double kMinScaling = 1e-4;
constexpr double kMaxScaling = 1e4;
// ... and a full round moves the rows whose violation is at least this share of the largest one: the ones a hundred times smaller mostly vanish once the
// large ones have moved, and moving them along is what sends rounds wandering (round 3 sweep: 0.01 - one launch at a time +8 %, the headline
// of 15 of 16 scenario seeds within 2.5 % of each other; 0.003: no effect, 0.02-0.03: the same, 0.05-0.1: a straggler is back, 0.3: +13 % work)
constexpr double kFullMoveShare = 0.01;
constexpr int kCautiousFromRound = 8;        // active-set rounds of a polish attempt: single moves from this round on at the latest (run())
constexpr double kRhoMin = 1e-6;
constexpr double kRhoMax = 1e6;
constexpr double kRhoTol = 1e-4;
constexpr double kRhoEqFactor = 1e3;
constexpr double kPi = 3.14159265358979323846;   // M_PI
constexpr double kPi2 = 1.57079632679489661923;  // M_PI_2

// ---- device-side argument block of one solve launch ---------------------------------------------
struct PathSolveArgs {
    int batch, n, passes, warm;
    const int32_t* n_of;    // [batch] waypoints of each QP (<= n; n is then only the array stride), or nullptr: all have n
    const double* ref;      // [batch][n][5]
    const double* lin;      // [batch][n][3] or nullptr
    const double* bounds;   // [batch][n][6]
    const double* scal;     // [batch][6]
    double* out;            // [batch][n][7]
    int32_t* status;        // [batch] or nullptr
    int32_t* iters;         // [batch] or nullptr
    double* info;           // [batch][PQP_INFO_STRIDE] or nullptr
    // warm state kept by the handle, lane layout
    double* wx;             // [batch][n][6]
    double* wy;             // [batch][n][6]  (yT[3], yK, yF, yR)
    double* wye;            // [batch][2]
    double* wrho;           // [batch]
    double* wscale;         // [slots][T][18] contexts without kSaveLds: Ruiz D(6), E(6) parked between the passes of a QP, dual snapshot(6)
    double* wsave;          // [slots][T][kSaveStride] save area of the workgroup slot that runs the QP: the ADMM state while a
                            // polish is tried + the best polished point.  Per SLOT, not per QP: the few MB stay in L2
    int store_warm;         // 0: the final iterate is not written to wx / wy / wye (nobody will ask for it)
    // work distribution: persistent workgroups draw QP indices from a ticket counter; `order` (or nullptr) maps the i-th ticket to
    // a QP - most expensive first, by the cost the QPs had in the handle's previous solve (pqp_path_order_kernel)
    unsigned long long* ticket;
    unsigned long long ticket_base;
    const int32_t* order;   // [batch] or nullptr
    int32_t* cost_key;      // [batch] or nullptr: bin << 24 | rank within the bin, written at the end of every QP
    int32_t* cost_hist;     // [256] QPs per cost bin (atomically counted) + [256] = workgroups that have finished this launch
    int32_t* order_next;    // [batch] the ticket -> QP map of the NEXT launch, written by the last workgroup to finish this one
    int carry_k;            // the handle's PQP_OPT_CARRY_CYCLES (0, 1 or k >= 2) on every launch, cold ones included: the share 1 / k behind the threshold bin
    int carry_tails;        // PQP_OPT_CARRY_CYCLES = k >= 2 (with warm == 1): only the QPs whose cost in the previous launch reached the bin
                            // cost_hist[kCostBins + 1] - the most expensive 1 / k - start from their previous optimum, the others start cold
    pqp_params prm;
};
enum { kCostBins = 256 };

// ---- scalar helpers ----------------------------------------------------------------------------------------
// reciprocal and reciprocal square root: on the device the hardware seed + two Newton steps (~1 ulp) instead of the
// 30-40 instruction IEEE division / sqrt sequences -- the cold code is instruction-fetch bound, so code size is time
#if defined(__HIP_DEVICE_COMPILE__)
PQP_HD double rcp(double x) {
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.0), r, r);
    r = fma(fma(-x, r, 1.0), r, r);
    return r;
}
PQP_HD double rsq(double x) {
    double r = __builtin_amdgcn_rsq(x);
    r = r * fma(-0.5 * x * r, r, 1.5);
    r = r * fma(-0.5 * x * r, r, 1.5);
    return r;
}
// one out-of-line copy of sin/cos (their argument-reduction code is large)
__device__ __noinline__ void sincos_shared(double a, double* s, double* c) { sincos(a, s, c); }
#else
PQP_HD double rcp(double x) { return 1.0 / x; }
PQP_HD double rsq(double x) { return 1.0 / sqrt(x); }
inline void sincos_shared(double a, double* s, double* c) { *s = sin(a); *c = cos(a); }
#endif

// ---- small dense helpers: sym3 = [00,01,02,11,12,22], mat3 row-major ---------------------------------
PQP_HD double fmax3(double a, double b, double c) { return fmax(a, fmax(b, c)); }
PQP_HD double limit_scaling(double v) { v = v < kMinScaling ? 1.0 : v; return v > kMaxScaling ? kMaxScaling : v; }

// inverse of an SPD sym3 through its LDL^T factorisation
PQP_HD void sym3_inv(const double* S, double* R) {
    const double d0 = S[0];
    const double i0 = rcp(d0);
    const double l10 = S[1] * i0, l20 = S[2] * i0;
    const double d1 = S[3] - l10 * S[1];
    const double i1 = rcp(d1);
    const double t21 = S[4] - l20 * S[1];
    const double l21 = t21 * i1;
    const double d2 = S[5] - l20 * S[2] - l21 * t21;
    const double i2 = rcp(d2);
    // inv(L): [[1,0,0],[-l10,1,0],[l10*l21-l20,-l21,1]]
    const double m10 = -l10, m21 = -l21, m20 = l10 * l21 - l20;
    R[0] = i0 + m10 * m10 * i1 + m20 * m20 * i2;
    R[1] = m10 * i1 + m20 * m21 * i2;
    R[2] = m20 * i2;
    R[3] = i1 + m21 * m21 * i2;
    R[4] = m21 * i2;
    R[5] = i2;
}
// C = M * S   (M mat3, S sym3)
PQP_HD void mat3_mul_sym3(const double* M, const double* S, double* C) {
    for (int r = 0; r < 3; ++r) {
        const double a = M[3 * r], b = M[3 * r + 1], c = M[3 * r + 2];
        C[3 * r + 0] = a * S[0] + b * S[1] + c * S[2];
        C[3 * r + 1] = a * S[1] + b * S[3] + c * S[4];
        C[3 * r + 2] = a * S[2] + b * S[4] + c * S[5];
    }
}
// C = M^T * S
PQP_HD void mat3t_mul_sym3(const double* M, const double* S, double* C) {
    for (int r = 0; r < 3; ++r) {
        const double a = M[r], b = M[3 + r], c = M[6 + r];
        C[3 * r + 0] = a * S[0] + b * S[1] + c * S[2];
        C[3 * r + 1] = a * S[1] + b * S[3] + c * S[4];
        C[3 * r + 2] = a * S[2] + b * S[4] + c * S[5];
    }
}
// sym(G * M^T) for G, M mat3 with symmetric product
PQP_HD void mat3_mul_mat3t_sym(const double* G, const double* M, double* S) {
    S[0] = G[0] * M[0] + G[1] * M[1] + G[2] * M[2];
    S[1] = G[0] * M[3] + G[1] * M[4] + G[2] * M[5];
    S[2] = G[0] * M[6] + G[1] * M[7] + G[2] * M[8];
    S[3] = G[3] * M[3] + G[4] * M[4] + G[5] * M[5];
    S[4] = G[3] * M[6] + G[4] * M[7] + G[5] * M[8];
    S[5] = G[6] * M[6] + G[7] * M[7] + G[8] * M[8];
}
// sym(G * M) for G, M mat3 with symmetric product
PQP_HD void mat3_mul_mat3_sym(const double* G, const double* M, double* S) {
    S[0] = G[0] * M[0] + G[1] * M[3] + G[2] * M[6];
    S[1] = G[0] * M[1] + G[1] * M[4] + G[2] * M[7];
    S[2] = G[0] * M[2] + G[1] * M[5] + G[2] * M[8];
    S[3] = G[3] * M[1] + G[4] * M[4] + G[5] * M[7];
    S[4] = G[3] * M[2] + G[4] * M[5] + G[5] * M[8];
    S[5] = G[6] * M[2] + G[7] * M[5] + G[8] * M[8];
}
// C = G * M
PQP_HD void mat3_mul_mat3(const double* G, const double* M, double* C) {
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c)
            C[3 * r + c] = G[3 * r] * M[c] + G[3 * r + 1] * M[3 + c] + G[3 * r + 2] * M[6 + c];
}
PQP_HD void mat3_vec(const double* M, const double* v, double* o) {
    for (int r = 0; r < 3; ++r) o[r] = M[3 * r] * v[0] + M[3 * r + 1] * v[1] + M[3 * r + 2] * v[2];
}
PQP_HD void mat3t_vec(const double* M, const double* v, double* o) {
    for (int r = 0; r < 3; ++r) o[r] = M[r] * v[0] + M[3 + r] * v[1] + M[6 + r] * v[2];
}
PQP_HD void sym3_vec(const double* S, const double* v, double* o) {
    o[0] = S[0] * v[0] + S[1] * v[1] + S[2] * v[2];
    o[1] = S[1] * v[0] + S[3] * v[1] + S[4] * v[2];
    o[2] = S[2] * v[0] + S[4] * v[1] + S[5] * v[2];
}

// include/tools/tools.hpp:24-35 — recursive wrap to [-pi, pi], boundaries inclusive.
PQP_HD double constrain_angle(double a) {
    for (int it = 0; it < 64; ++it) {
        if (a > kPi) a -= 2 * kPi;
        else if (a < -kPi) a += 2 * kPi;
        else break;
    }
    return a;
}

// base_solver.cpp:290-295
PQP_HD void soft_bounds(double lb, double ub, double margin, double min_clearance, double& lo, double& up) {
    const double clearance = ub - lb;
    const double remain = fmax(min_clearance, clearance - 2 * margin);
    const double shrink = fmax(0.0, (clearance - remain) / 2.0);
    lo = lb + shrink;
    up = ub - shrink;
}

// The linearised transition (i-1 -> i), base_solver.cpp:165-186, same expression order.
//   lin_p = (l, psi, k) of waypoint i-1, k_next = lin_i.k, s_p/s_i arclengths, kref_p = ref[i-1].k
//   a[6]  = a00 a01 a10 a11 a12 ds   (A = I + ds*df_x; a02 = a20 = a21 = 0, a22 = 1, B = ds*e3)
//   c[3]  = ds * (f - df_x * x - df_u * u)     (the bounds of rows 3(i)..3(i)+2 are -c)
PQP_HD void transition_block(const double* lin_p, double k_next, double s_p, double s_i, double kref_p,
                             double* a, double* c) {
    const double l = lin_p[0], psi = lin_p[1], k = lin_p[2];
    double sn, cs;
    sincos_shared(psi, &sn, &cs);
    const double t = sn / cs;
    const double one_kl = 1 - k * l;
    const double df00 = -k * t;
    const double df01 = one_kl / (cs * cs);
    const double df10 = -k * k / cs;
    const double df11 = one_kl * k * t / cs;
    const double df12 = one_kl / cs;
    const double ds = s_i - s_p;
    a[0] = ds * df00 + 1.0;
    a[1] = ds * df01;
    a[2] = ds * df10;
    a[3] = ds * df11 + 1.0;
    a[4] = ds * df12;
    a[5] = ds;
    const double u_in = (k_next - k) / ds;
    const double f0 = one_kl * t;
    const double f1 = one_kl * k / cs - kref_p;
    const double f2 = u_in;
    c[0] = ds * (f0 - (df00 * l + df01 * psi + 0.0 * k) - 0.0 * u_in);
    c[1] = ds * (f1 - (df10 * l + df11 * psi + df12 * k) - 0.0 * u_in);
    c[2] = ds * (f2 - (0.0 * l + 0.0 * psi + 0.0 * k) - 1.0 * u_in);
}

// ---- per-thread state ---------------------------------------------------------------------------------
enum : int {
    F_REAL = 1, F_PREV = 2, F_NEXT = 4, F_PRECISE = 8, F_LAST = 16,
    F_FREE0 = 1 << 5,     // << k: inequality row k (K, F, R) has no finite bound            (rho class "free")
    F_EQ0 = 1 << 8,       // << k: inequality row k has l == u within RHO_TOL               (rho class "equality")
    F_ACTLO0 = 1 << 11,   // << k: polish: row k is taken as active at its lower bound
    F_ACTUP0 = 1 << 14,   // << k: polish: row k is taken as active at its upper bound
    F_ROWBITS = (7 << 5) | (7 << 8) | (7 << 11) | (7 << 14),
    // the scenario of this waypoint is not a number the reference could work with: NaN / Inf among its reference state, bounds or start state, or an
    // arclength that does not increase (the reference divides by ds, base_solver.cpp:174,180).  Such a QP ends PQP_STATUS_NUMERICAL at its first
    // residual check with a zero output record (solve() == false, as when OSQP returns non-finite iterates); min / max arithmetic would
    // otherwise drop a NaN bound silently and "solve" the QP without that row
    F_BADIN = 1 << 17
};
// v - v is 0 for a finite v and NaN for NaN / Inf
PQP_HD double not_finite_mark(double v) { return v - v; }

// Everything a waypoint needs inside the ADMM loop (86 doubles); kept in registers.
struct Slot {
    int flags;
    // iterates: x = (l, psi, k, v, sf, sr); duals of T rows and of the inequality rows K, F, R
    double x[6], yT[3], yI[3], zI[3];
    // problem data: incoming transition, its right-hand side, the F and R boxes (the K box is uniform per QP)
    double a[6], bT[3], lo[2], up[2];
    // penalty metrics: Sigma = sigma/(c D^2); R = rho * class * E^2 / c
    double sig[6], rhoT[3], rhoI[3], rinvI[3];
    // elimination of v, sf, sr
    double cF, cR, idsf, idsr, tu, tudc, idu;
    // cyclic-reduction factor of this waypoint's node
    double Dinv[6], GL[9], GR[9];
    // per-iteration scratch that crosses a phase boundary
    double r[3], rv, rsf, rsr, xt[6];
    // contexts with kCstAcc: the halves of the pass constants that live in the accumulator half of the register file (see PathQp::acc_get)
    int acc[24];
};
// setup-time state (dead inside the ADMM loop)
struct SlotSetup {
    double D[6], E[6];               // Ruiz
    double Dg[6], Lc[9], Rc[9];      // factor: current diagonal block, couplings (left: rows prev, cols own; right: rows own, cols next)
};

struct Lane {
    Slot s;
    SlotSetup w;
};

// end rows (owned by the thread holding waypoint n-1), kept in shared memory
struct EndRows {
    double lo[2], up[2], z[2], y[2], E[2], rb[2], rho[2], rinv[2];
    double act[2];          // polish: -1 lower-active, +1 upper-active, 0 inactive
    double sz[2], sy[2];    // ADMM state parked while a polish is tried
    double by[2];           // multipliers of the best polished point of the current attempt
    double yp[2];           // y_k - y_{k-1} of the last iteration (infeasibility certificate)
    double pad[6];          // [0..1] polish: violations of the end rows published for the local-maximum rule; [2..3] end-row duals at the
                            // previous late infeasibility check
};

// shared-memory layout in doubles, T = threads per QP = padded number of waypoints.  The per-iteration exchange
// buffers and the factor-time exchange buffer are never live at the same time and share one region.
struct ShLayout {
    int T;
    // per-iteration exchange
    PQP_HD int bufG() const { return 0; }                   // [T][3]  message to the previous waypoint
    PQP_HD int bufP() const { return 3 * T; }               // [T][3]  CR forward, to the right neighbour
    PQP_HD int bufQ() const { return 6 * T; }               // [T][3]  CR forward, to the left neighbour
    PQP_HD int xbuf() const { return 9 * T; }               // [T][3]  X~
    // X for the residuals and the polish rules behind them: the forward pass's P buffer, which nothing reads after a solve's last workgroup barrier - so
    // residuals() may write it while the other wavefront is still in the wave-local tail of iterate() (which reads xbuf): no barrier between the two
    PQP_HD int xres() const { return bufP(); }
    // factor-time exchange (aliases the above): first [T][15] M(6) Lc(9), then [T][21] SL(6) SR(6) Cnew(9)
    PQP_HD int fbuf() const { return 0; }
    // persistent
    PQP_HD int lin() const { return 21 * T; }               // [T][3] linearisation point / first-iteration dz
    PQP_HD int sk() const { return 24 * T; }                // [T][2] s, k_ref
    PQP_HD int end() const { return 26 * T; }               // EndRows (32 doubles)
    PQP_HD int red() const { return 26 * T + 32; }          // reduction scratch [8][16]
    PQP_HD int poison() const { return 26 * T + 32 + 15; }  // (a column of the reduction scratch no wavefront uses) 1.0: the scenario of this QP is not a number (F_BADIN)
                                                            // - zeroed in load(), set in assemble(), read once per pass behind it
    // per-waypoint pass constants that are read once per iteration (kCstLds contexts keep them here instead of in registers):
    // [12][T] sig(6) lo(2) up(2) idsf idsr, one array per constant (unit stride over the lanes: conflict-free)
    // 24 zeros (written once per workgroup): where a lane has no neighbour its read is redirected here, so the reads of a phase
    // need no exec-mask change and no select (a conditional LDS read costs two scalar instructions around every ds_read)
    PQP_HD int zero() const { return 26 * T + 32 + 128; }
    PQP_HD int cst() const { return 26 * T + 32 + 128 + 24; }
    // contexts with kSaveLds (T <= 256) keep the polish save area here instead of in global memory: [T][PQP_SAVE_STRIDE]
    PQP_HD int save() const { return 38 * T + 32 + 128 + 24; }
    // ... and the dual iterate of the previous late infeasibility check, [T][6] (prim_inf_after)
    PQP_HD int ysnap() const { return (38 + PQP_SAVE_STRIDE) * T + 32 + 128 + 24; }
    // (contexts whose save area lives in global memory park their scaling vectors there too: without pass constants in LDS the constants
    //  region is then unused and not allocated - 26 T doubles = 27 KB per QP at 128 waypoints)
    PQP_HD int total(bool save_in_lds = false, bool cst_in_lds = true) const { return (save_in_lds ? 44 + PQP_SAVE_STRIDE : (cst_in_lds ? 38 : 26)) * T + 32 + 128 + 24; }
    // y_k - y_{k-1} of the last iteration, [T][6] (infeasibility certificate): lives in the part of the factor-time buffer
    // the iteration does not use; every iteration rewrites it, and a check never follows a factorisation directly
    PQP_HD int yprev() const { return 12 * T; }
    // pass constants of the lanes staged for the certificate (free at a termination check: CR buffers, tail of the factor
    // buffer, linearisation point): [T][9] a(6) bT(3), [T][3] flags lo0 lo1, [T][3] up0 up1 -
    PQP_HD int stageA() const { return 3 * T; }
    PQP_HD int stageB() const { return 18 * T; }
    PQP_HD int stageC() const { return 21 * T; }
};

// diagonal of P by variable slot (base_solver.cpp:123-143); dummy variables (padding waypoints, v of waypoint 0,
// sr of a rough waypoint) get 1 so their block stays invertible
PQP_HD double cost_diag(const pqp_params& prm, int flags, int k) {
    const bool real = flags & F_REAL, prev = flags & F_PREV, precise = flags & F_PRECISE;
    switch (k) {
        case 0: return real ? prm.weight_l : 1.0;
        case 1: return real ? 0.0 : 1.0;
        case 2: return real ? prm.weight_kappa : 1.0;
        case 3: return prev ? prm.weight_dkappa : 1.0;
        case 4: return real ? prm.weight_slack : 1.0;
        default: return (real && precise) ? prm.weight_slack : 1.0;
    }
}
PQP_HD double coef_front(const pqp_params& prm, int flags) { return ((flags & F_REAL) && (flags & F_PRECISE)) ? prm.front_length : 0.0; }
PQP_HD double coef_rear(const pqp_params& prm, int flags) { return ((flags & F_REAL) && (flags & F_PRECISE)) ? prm.rear_length : 0.0; }

// ---------------------------------------------------------------------------------------------------------
// primal infeasibility certificate from LDS (see PathQp::primal_infeasible).  LC: lane-less context with T(),
// phase(f(t)), reduce_max<K>(out, f(t, v)), reduce_sum<K>(out, f(t, v)).
//   dy            sh[yprev]  [T][6]   y_k - y_{k-1} (rows T0 T1 T2 K F R), EndRows::yp for the two end rows
//   staged lanes  sh[stageA/B/C]      a(6) bT(3) | flags lo0 lo1 | up0 up1
// ---------------------------------------------------------------------------------------------------------
template <class LC>
PQP_HD bool primal_certificate(LC& c, double* sh, int T, double front_length, double rear_length, double kap, double eps, double cscale) {
    const ShLayout L{T};
    EndRows* er = reinterpret_cast<EndRows*>(sh + L.end());
    c.phase([&](int t) {        // project dy on the polar of the recession cone of [l, u] (in place); message A_in' dyT to the previous waypoint
        double* dy = sh + L.yprev() + 6 * t;
        const double* pa = sh + L.stageA() + 9 * t;
        const double* pb = sh + L.stageB() + 3 * t;
        const double* pc = sh + L.stageC() + 3 * t;
        const int flags = (int)pb[0];
        const bool real = flags & F_REAL;
        const double lo[3] = {real ? -kap : 0.0, pb[1], pb[2]}, up[3] = {real ? kap : 0.0, pc[0], pc[1]};
        for (int k = 0; k < 3; ++k) {
            const double d = dy[3 + k];
            const bool fr = flags & (F_FREE0 << k), inf_u = up[k] > 1e19, inf_l = lo[k] < -1e19;
            dy[3 + k] = (fr || (inf_u && inf_l)) ? 0.0 : (inf_u ? fmin(d, 0.0) : (inf_l ? fmax(d, 0.0) : d));
        }
        sh[L.bufG() + 3 * t + 0] = pa[0] * dy[0] + pa[2] * dy[1];
        sh[L.bufG() + 3 * t + 1] = pa[1] * dy[0] + pa[3] * dy[1];
        sh[L.bufG() + 3 * t + 2] = pa[4] * dy[1] + ((flags & F_PREV) ? dy[2] : 0.0);
        if (flags & F_LAST) {
            for (int k = 0; k < 2; ++k) {
                const double d = er->yp[k];
                const bool inf_u = er->up[k] > 1e19, inf_l = er->lo[k] < -1e19;
                er->yp[k] = (er->rb[k] < 0.0 || (inf_u && inf_l)) ? 0.0 : (inf_u ? fmin(d, 0.0) : (inf_l ? fmax(d, 0.0) : d));
            }
        }
    });
    double nm[2], lhs[1];
    c.template reduce_max<2>(nm, [&](int t, double (&v)[2]) {
        const double* dy = sh + L.yprev() + 6 * t;
        const double* pa = sh + L.stageA() + 9 * t;
        const int flags = (int)sh[L.stageB() + 3 * t];
        const bool real = flags & F_REAL, precise = flags & F_PRECISE;
        const double cf = (real && precise) ? front_length : 0.0, cr = (real && precise) ? rear_length : 0.0;
        double gn[3];
        for (int k = 0; k < 3; ++k) gn[k] = (t + 1 < T) ? sh[L.bufG() + 3 * (t + 1) + k] : 0.0;
        double de0 = 0.0, de1 = 0.0;
        if (flags & F_LAST) { de0 = er->yp[0]; de1 = er->yp[1]; }
        double at[6];
        at[0] = -dy[0] + gn[0] + dy[4] + dy[5] + de0;
        at[1] = -dy[1] + gn[1] + cf * dy[4] + cr * dy[5] + de1;
        at[2] = -dy[2] + gn[2] + dy[3];
        at[3] = pa[5] * dy[2];
        at[4] = dy[4];
        at[5] = dy[5];
        const bool colreal[6] = {real, real, real, (flags & F_PREV) != 0, real, real && precise};
        double n_dy = 0.0, n_at = 0.0;
        for (int k = 0; k < 6; ++k) {
            n_dy = fmax(n_dy, real ? fabs(dy[k]) : 0.0);
            n_at = fmax(n_at, colreal[k] ? fabs(at[k]) : 0.0);
        }
        v[0] = fmax(n_dy, fmax(fabs(de0), fabs(de1)));
        v[1] = n_at;
    });
    c.template reduce_sum<1>(lhs, [&](int t, double (&v)[1]) {
        const double* dy = sh + L.yprev() + 6 * t;
        const double* pa = sh + L.stageA() + 9 * t;
        const double* pb = sh + L.stageB() + 3 * t;
        const double* pc = sh + L.stageC() + 3 * t;
        const int flags = (int)pb[0];
        double acc = 0.0;
        if (flags & F_REAL) {
            for (int k = 0; k < 3; ++k) acc += pa[6 + k] * dy[k];                       // l = u = b on the transition rows
            const double lo[3] = {-kap, pb[1], pb[2]}, up[3] = {kap, pc[0], pc[1]};
            for (int k = 0; k < 3; ++k) { const double d = dy[3 + k]; acc += d > 0.0 ? up[k] * d : (d < 0.0 ? lo[k] * d : 0.0); }
        }
        if (flags & F_LAST)
            for (int k = 0; k < 2; ++k) { const double d = er->yp[k]; acc += d > 0.0 ? er->up[k] * d : (d < 0.0 ? er->lo[k] * d : 0.0); }
        v[0] = acc;
    });
    return cscale * nm[0] > eps && lhs[0] < -eps * nm[0] && nm[1] < eps * nm[0];
}

// ---------------------------------------------------------------------------------------------------------
// The late form of the certificate (pqp_params::prim_inf_after): dy = y_now - y_at_the_previous_late_check.  One call per lane with
// the lane's values by value; LC as for primal_certificate.  Writes dy and the staged pass constants to LDS, refreshes the snapshot,
// evaluates the test (when a snapshot of this pass existed).
// ---------------------------------------------------------------------------------------------------------
struct LateCertIn {
    double y[6], a[6], b[3], lo0, lo1, up0, up1;
    int flags;
};
template <class LC>
PQP_HD bool late_certificate(LC& c, double* sh, int T, int t, double* snap, bool have, const LateCertIn& in, double front_length, double rear_length,
                             double kap, double eps, double cscale) {
    const ShLayout L{T};
    EndRows* er = reinterpret_cast<EndRows*>(sh + L.end());
    double* dyp = sh + L.yprev() + 6 * t;
    double* pa = sh + L.stageA() + 9 * t;
    double* pb = sh + L.stageB() + 3 * t;
    double* pc = sh + L.stageC() + 3 * t;
    for (int k = 0; k < 6; ++k) { dyp[k] = in.y[k] - snap[k]; snap[k] = in.y[k]; pa[k] = in.a[k]; }
    for (int k = 0; k < 3; ++k) pa[6 + k] = in.b[k];
    pb[0] = (double)in.flags; pb[1] = in.lo0; pb[2] = in.lo1;
    pc[0] = in.up0; pc[1] = in.up1; pc[2] = 0.0;
    if (in.flags & F_LAST)
        for (int k = 0; k < 2; ++k) { er->yp[k] = er->y[k] - er->pad[2 + k]; er->pad[2 + k] = er->y[k]; }
    c.phase([](int) {});          // every lane has staged its values
    return have ? primal_certificate(c, sh, T, front_length, rear_length, kap, eps, cscale) : false;
}

// What a QP cost (about microseconds: 4 per reduced-KKT solve, 13 per factorisation), binned for the next launch's
// most-expensive-first order: key = bin << 24.  Neither the counter update nor the store returns anything the QP waits for (a returning
// atomic on device-scope memory is ~2 us of exposed latency at the end of every QP); the rank within a bin is handed out by the
// workgroup that writes the next launch's order.
PQP_HD void record_cost(const PathSolveArgs& a, int qp, int cost, int at_least_bin = -1) {
    int bin = cost >> 3;
    bin = bin < kCostBins - 1 ? bin : kCostBins - 1;
    bin = bin > at_least_bin ? bin : at_least_bin;
#if defined(__HIP_DEVICE_COMPILE__)
    (void)__hip_atomic_fetch_add(a.cost_hist + bin, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // (device-scope store: the last workgroup of the launch, on whatever XCD, reads it with a device-scope load)
    __hip_atomic_store(a.cost_key + qp, bin << 24, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
    a.cost_hist[bin]++;
    a.cost_key[qp] = bin << 24;
#endif
}

// uniform (per-QP) solver scalars; handed by value across the hot / cold boundary
struct Uni {
    double rho, cscale, kap, alpha;
    int kkt_solves;
    int factors;
    int polishing;
    int cert;
};
// cold operations (rare, register-hungry): executed out of line on a memory-resident copy of the lane state
enum ColdOp : int {
    COLD_BEGIN_PASS = 0,     // i0 = pass index, i1 = have_warm: [load, warm-load], assemble, Ruiz, start rows, factor
    COLD_REFACTOR = 1,       // i0 = RefactorKind, d0 = parameter: penalty change + factor
    COLD_END_PASS = 2,       // i0 = 1: polish accepted: [polish_end(true)], unpack; i0 = 2: the pass ended PQP_STATUS_NUMERICAL: unpack writes zeros
    COLD_FINISH = 3,         // store the warm state
    COLD_CERT = 4            // primal infeasibility certificate on the last dy -> PathQp::cert_
};
enum RefactorKind : int { RF_RESCALE = 0 /* d0 = ratio */, RF_POLISH_BEGIN = 1, RF_POLISH_UPDATE = 2 /* d0 = threshold */, RF_POLISH_REJECT = 3 };

// =======================================================================================================
// The solver.  Ctx provides:
//   int  T()                              threads per QP (power of two, >= 64)
//   double* sh()                          shared scratch of ShLayout(T).total() doubles
//   template<F> void phase(F f)           run f(t, Lane&) for every thread, then synchronise the workgroup
//   template<F> void phase_w(F f)         the same, but only the lanes of one wavefront need to see each other's LDS writes
//   template<int K,F> void reduce_max/sum(double (&out)[K], F f)   f(t, Lane&, double (&v)[K])
//   void cold(PathQp&, op, i0, i1, d0)    run do_cold() (possibly out of line)
//   static constexpr bool kCstLds         12 per-waypoint pass constants live in LDS (ShLayout::cst) instead of in Slot fields
//   static constexpr bool kParkScale      the Ruiz vectors D, E are parked (LDS or PathSolveArgs::wscale) between the passes
//   static constexpr bool kSaveLds        the polish save area (and the parked D, E) live in LDS: ShLayout::total(true) doubles
// =======================================================================================================
// CERT: compile the primal infeasibility certificate in.  It is a template parameter because its mere presence in the kernel
// (one more cold operation + six LDS stores per iteration) costs the ADMM iteration 12 % through register allocation;
// the launcher picks the variant from prm.eps_prim_inf > 0.
template <class Ctx, bool CERT = true>
struct PathQp {
    Ctx& ctx;
    const PathSolveArgs& A;
    const int qp;
    const int slot;           // workgroup slot (index of the save area)
    const int stride;         // waypoints per QP in the batch arrays
    const int n;              // waypoints of THIS QP
    const int T;
    const ShLayout L;
    double* const sh;
    // uniform per-QP scalars
    double rho, cscale, kap, alpha_;
    bool polishing_;
    bool cert_;               // result of the last COLD_CERT
    bool snap_valid_;         // late certificate: a dual snapshot of this pass exists
    int kkt_solves_;          // iterate() executions: ADMM iterations + polish refinement solves
    int factors_;             // factor() executions
#ifdef PQP_TIMING
    long long tsub_[8] = {0, 0, 0, 0, 0, 0, 0, 0};      // debug build: ticks inside the cold operations (tools/kernel_timeline.py)
// PQP_TIMING_MASK: bit k = category k of run() is timed, bit 8 + k = sub-time k of the cold operations (one category per build keeps the
// clock reads from perturbing what they measure; the total, tacc[7], is always taken)
#ifndef PQP_TIMING_MASK
#define PQP_TIMING_MASK 0xffff
#endif
#define PQP_SUB(k, stmt) do { if ((PQP_TIMING_MASK >> (8 + (k))) & 1) { const long long t0_ = ctx.clock(); stmt; tsub_[k] += ctx.clock() - t0_; } else { stmt; } } while (0)
#else
#define PQP_SUB(k, stmt) do { stmt; } while (0)
#endif
// PQP_TIMING_ITER (with PQP_TIMING): 10 ns ticks of the pieces of iterate() / factor(), accumulated per QP -> out[qp][12..27] (tools/kernel_timeline.py;
// s_memtime instead of the 100 MHz clock costs ~6 us per read on gfx950: a QP then takes 1.3 ms)
#if defined(PQP_TIMING) && defined(PQP_TIMING_ITER)
    long long tit_[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tit_last_ = 0;      // [0..7] iterate(), [8..15] factor()
#define PQP_IT_BEGIN() do { tit_last_ = ctx.clock(); } while (0)
#define PQP_IT(k) do { const long long now_ = ctx.clock(); tit_[k] += now_ - tit_last_; tit_last_ = now_; } while (0)
#else
#define PQP_IT_BEGIN() do { } while (0)
#define PQP_IT(k) do { } while (0)
#endif

    PQP_HD PathQp(Ctx& c, const PathSolveArgs& a, int q, int slot_ = -1)
        : ctx(c), A(a), qp(q), slot(slot_ < 0 ? q : slot_), stride(a.n), n(count_of(a, q)), T(c.T()), L{c.T()}, sh(c.sh()), rho(a.prm.rho), cscale(1.0), kap(0.0), alpha_(a.prm.alpha), polishing_(false), cert_(false), snap_valid_(false), kkt_solves_(0), factors_(0) {}

    // waypoints of QP q; fewer than 2: nothing to optimise (a road blocked at the first waypoints; the reference does not get
    // this far) - the caller skips the QP with status PQP_STATUS_UNSOLVED instead of constructing a solver
    PQP_HD static int count_of(const PathSolveArgs& a, int q) { return a.n_of ? (a.n_of[q] < a.n ? a.n_of[q] : a.n) : a.n; }

    PQP_HD EndRows* end_rows() const { return reinterpret_cast<EndRows*>(sh + L.end()); }
    // The fields of the two end rows by value.  The lane that owns the rows works on them while everybody else waits at the next barrier, so its
    // code loads what it needs in ONE batch up front (what a block does not use is never loaded: the copies are plain loads) and stores what it
    // changed at the end; field by field inside `for (k)` with early exits every access was an LDS round trip of its own (round 6).
    struct EndVals { double lo[2], up[2], z[2], y[2], E[2], rb[2], rho[2], rinv[2], act[2], pad[2]; };
    PQP_HD EndVals end_vals() const {
        const EndRows* er = end_rows();
        EndVals v;
        _Pragma("unroll") for (int k = 0; k < 2; ++k) {
            v.lo[k] = er->lo[k]; v.up[k] = er->up[k]; v.z[k] = er->z[k]; v.y[k] = er->y[k]; v.E[k] = er->E[k]; v.rb[k] = er->rb[k];
            v.rho[k] = er->rho[k]; v.rinv[k] = er->rinv[k]; v.act[k] = er->act[k]; v.pad[k] = er->pad[k];
        }
        return v;
    }
    // With n < T the root of the cyclic-reduction tree (tp = T, the last lane) is a padding node, and so is everything between the last
    // waypoint and it: the couplings across the first padding lane are exact zeros, hence node T/2 hands the root nothing and needs
    // nothing from it - node T/2 IS the root of the real tree.  For T >= 128 the factorisation and the solves then stop one level
    // earlier: one cross-wavefront level with its workgroup barrier less per solve (two barriers instead of four at T = 128).
    PQP_HD bool root_is_padding() const { return T >= 128 && n < T; }
    // row `other` of an exchange buffer, or the zero block when that neighbour does not exist
    PQP_HD const double* nb(bool ok, int base, int stride, int other) const { return sh + (ok ? base + stride * other : L.zero()); }
    PQP_HD static void ld3(double (&v)[3], const double* p) { v[0] = p[0]; v[1] = p[1]; v[2] = p[2]; }

    // NOTE on style: per-lane state must stay in registers, which needs every Lane field to be written through
    // one unconditional store (values chosen with selects) or under a single branch without an else-store; only
    // LDS / global stores sit freely under branches.  (if/else branches that store to different places get merged by
    // LLVM into a store through a pointer phi, which pins the whole Lane struct in scratch memory.)

    // ---------------------------------------------------------------------------------------------
    // load the scenario
    // ---------------------------------------------------------------------------------------------
    PQP_HD void load() {
        const pqp_params& prm = A.prm;
        ctx.phase([&](int t, Lane& ln) {
            Slot& S = ln.s;
            const int i = t;
            if (t < 24) sh[L.zero() + t] = 0.0;
            if (t == 0) sh[L.poison()] = 0.0;
            const bool real = i < n;
            const int ic = real ? i : n - 1;
            const double* r = A.ref + ((size_t)qp * stride + ic) * PQP_REF_STRIDE;
            const double r0 = r[0], r1 = r[1];
            // base_solver.cpp:25-34: precise planning size = lower_bound(s, precise_planning_length)
            const bool precise = !prm.rough_constraints_far_away || r0 < prm.precise_planning_length;
            double l0 = 0.0, l1 = 0.0, l2 = r1;      // path_optimizer.cpp:128-137: (0, 0, k_ref)
            if (A.lin) {
                const double* li = A.lin + ((size_t)qp * stride + ic) * PQP_LIN_STRIDE;
                l0 = li[0]; l1 = li[1]; l2 = li[2];
            }
            const double mark = not_finite_mark(r0) + not_finite_mark(r1) + not_finite_mark(r[2]) + not_finite_mark(r[3]) + not_finite_mark(r[4]) +
                                not_finite_mark(l0) + not_finite_mark(l1) + not_finite_mark(l2);
            S.flags = (real ? (F_REAL | (i > 0 ? F_PREV : 0) | (i < n - 1 ? F_NEXT : 0) | (i == n - 1 ? F_LAST : 0) | (precise ? F_PRECISE : 0)) : 0) |
                      (mark == 0.0 ? 0 : F_BADIN);
            if (real) {
                sh[L.lin() + 3 * i + 0] = l0; sh[L.lin() + 3 * i + 1] = l1; sh[L.lin() + 3 * i + 2] = l2;
                sh[L.sk() + 2 * i + 0] = r0; sh[L.sk() + 2 * i + 1] = r1;
            }
            _Pragma("unroll") for (int k = 0; k < 6; ++k) S.x[k] = 0.0;
            _Pragma("unroll") for (int k = 0; k < 3; ++k) { S.yT[k] = 0.0; S.yI[k] = 0.0; S.zI[k] = 0.0; }
        });
    }

    // ---------------------------------------------------------------------------------------------
    // assemble the per-waypoint blocks around the linearisation point in sh[lin]
    // (setCost: base_solver.cpp:119-148 -> cost_diag(); setConstraints: :150-261)
    // ---------------------------------------------------------------------------------------------
    PQP_HD void assemble() {
        const pqp_params& prm = A.prm;
        const double* sc = A.scal + (size_t)qp * PQP_SCAL_STRIDE;
        {   // curvature box (:226-231), the same for every waypoint
            double sn, cn;
            sincos_shared(sc[5], &sn, &cn);
            kap = (sn / cn) / prm.wheel_base;
        }
        // A start curvature outside its box by no more than OSQP's primal tolerance (eps_abs + eps_rel * bound) is a QP the reference calls solved: ADMM
        // meets eps with a point that misses the row by that little.  Strictly it has no feasible point, so no polish of it could ever be verified (the
        // scenario of tools/robustness_sweep.py seed 1007, QP 6640: 520 ADMM iterations, 627 reduced solves, a launch that lasts 6 ms instead of 0.66).
        // The start state is projected onto the box by that little instead, as the lane-per-QP kernel does (pqp_path_lq.hpp run()), and the QP solved exactly.
        ctx.phase([&](int t, Lane& ln) {
            Slot& S = ln.s;
            const int i = t;
            const bool real = S.flags & F_REAL, prev = S.flags & F_PREV, precise = S.flags & F_PRECISE;
            double a6[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0}, c3[3] = {0.0, 0.0, 0.0};
            if (prev)
                transition_block(sh + L.lin() + 3 * (i - 1), sh[L.lin() + 3 * i + 2], sh[L.sk() + 2 * (i - 1)],
                                 sh[L.sk() + 2 * i], sh[L.sk() + 2 * (i - 1) + 1], a6, c3);
            _Pragma("unroll") for (int k = 0; k < 6; ++k) S.a[k] = a6[k];
            // transition right-hand sides: -c (:221-224), or -x0 at the first waypoint (:216-220)
            _Pragma("unroll") for (int k = 0; k < 3; ++k) S.bT[k] = !real ? 0.0 : (prev ? -c3[k] : -sc[k]);
            if (real && !prev) {
                const double over = fabs(sc[2]) - kap;
                if (over > 0.0 && over <= prm.eps_abs + prm.eps_rel * kap) S.bT[2] = sc[2] > 0.0 ? -kap : kap;
            }
            // collision boxes (:232-248); rough: one row l + s_c on the centre box, the R row and sr are dummies
            const int ic = real ? i : n - 1;
            const double* b = A.bounds + ((size_t)qp * stride + ic) * PQP_BOUNDS_STRIDE;
            const double f_lb = precise ? b[0] : b[4], f_ub = precise ? b[1] : b[5];
            double flo, fup, rlo, rup;
            soft_bounds(f_lb, f_ub, prm.expected_safety_margin, prm.min_clearance, flo, fup);
            soft_bounds(b[2], b[3], prm.expected_safety_margin, prm.min_clearance, rlo, rup);
            const double lo0 = real ? flo : 0.0, up0 = real ? fup : 0.0, lo1 = (real && precise) ? rlo : 0.0, up1 = (real && precise) ? rup : 0.0;
            {   // F_BADIN: the bounds and the start state as numbers, the arclength increasing
                double mark = 0.0;
                _Pragma("unroll") for (int k = 0; k < 6; ++k) mark += not_finite_mark(b[k]) + not_finite_mark(sc[k]);
                const bool ds_ok = !prev || sh[L.sk() + 2 * i] - sh[L.sk() + 2 * (i - 1)] > 0.0;
                // (the QP's verdict lives in one LDS word - zeroed in load(), read by residuals() - not in a lane field of the ADMM loop)
                if (!(mark == 0.0) || !ds_ok || (S.flags & F_BADIN)) sh[L.poison()] = 1.0;
            }
            if constexpr (kAcc) { acc_set(S, C_LO, lo0); acc_set(S, C_LO + 1, lo1); acc_set(S, C_UP, up0); acc_set(S, C_UP + 1, up1); }
            else if constexpr (kCst) { cst_set(t, C_LO, lo0); cst_set(t, C_LO + 1, lo1); cst_set(t, C_UP, up0); cst_set(t, C_UP + 1, up1); }
            else { S.lo[0] = lo0; S.up[0] = up0; S.lo[1] = lo1; S.up[1] = up1; }
            if (S.flags & F_LAST) {                                         // :250-259
                EndRows* er = end_rows();
                double elo = -kInfty, eup = kInfty;
                if (prm.constraint_end_heading && sc[4] == 0.0) {
                    const double heading = A.ref[((size_t)qp * stride + i) * PQP_REF_STRIDE + 2];
                    const double end_psi = constrain_angle(sc[3] - heading);
                    if (end_psi < prm.end_psi_max) {     // signed compare, no fabs (:256)
                        elo = end_psi - prm.end_psi_tol;
                        eup = end_psi + prm.end_psi_tol;
                    }
                }
                er->lo[0] = -prm.end_l_bound; er->up[0] = prm.end_l_bound;
                er->lo[1] = elo; er->up[1] = eup;
            }
        });
    }

    // the box of inequality row k (0: curvature, 1: front, 2: rear) as assembled ...
    // Pass constants of a waypoint that the iteration reads once: in registers (Slot fields) or, in contexts with kCstLds, in LDS -
    // 24 registers less in the ADMM loop, which is what lets two wavefronts share a SIMD.
    static constexpr bool kCst = Ctx::kCstLds;
    // ... or (kCstAcc, the device) in the ACCUMULATOR half of the register file, by their own v_accvgpr_write / _read at the one definition and the one
    // use per solve: gfx950 gives a one-wavefront-per-SIMD kernel 256 + 256 registers but VALU operands come from the first 256 only, and the compiler,
    // left to park what does not fit, shuffles hundreds of registers between the halves around every phase (tools/isa_mix.py).  What is used once
    // per solve is parked by hand: one read per use, no shuffles.
    static constexpr bool kAcc = Ctx::kCstAcc;
    enum : int { C_SIG = 0, C_LO = 6, C_UP = 8, C_IDSF = 10, C_IDSR = 11 };
    PQP_HD double cst_get(int t, int c) const { return sh[L.cst() + c * T + t]; }
    PQP_HD void cst_set(int t, int c, double v) const { sh[L.cst() + c * T + t] = v; }
    PQP_HD double acc_get(const Slot& S, int c) const { return Ctx::acc_read(S.acc[2 * c], S.acc[2 * c + 1]); }
    PQP_HD void acc_set(Slot& S, int c, double v) const { Ctx::acc_write(v, S.acc[2 * c], S.acc[2 * c + 1]); }
    PQP_HD double sig_of(const Slot& S, int t, int k) const { if constexpr (kAcc) return acc_get(S, C_SIG + k); else if constexpr (kCst) return cst_get(t, C_SIG + k); else return S.sig[k]; }
    PQP_HD void set_sig(Slot& S, int t, int k, double v) const { if constexpr (kAcc) acc_set(S, C_SIG + k, v); else if constexpr (kCst) cst_set(t, C_SIG + k, v); else S.sig[k] = v; }
    PQP_HD double idsf_of(const Slot& S, int t) const { if constexpr (kAcc) return acc_get(S, C_IDSF); else if constexpr (kCst) return cst_get(t, C_IDSF); else return S.idsf; }
    PQP_HD double idsr_of(const Slot& S, int t) const { if constexpr (kAcc) return acc_get(S, C_IDSR); else if constexpr (kCst) return cst_get(t, C_IDSR); else return S.idsr; }
    PQP_HD double lo_of(const Slot& S, int t, int j) const { if constexpr (kAcc) return acc_get(S, C_LO + j); else if constexpr (kCst) return cst_get(t, C_LO + j); else return S.lo[j]; }
    PQP_HD double up_of(const Slot& S, int t, int j) const { if constexpr (kAcc) return acc_get(S, C_UP + j); else if constexpr (kCst) return cst_get(t, C_UP + j); else return S.up[j]; }
    PQP_HD double raw_lo(const Slot& S, int t, int k) const { return k == 0 ? ((S.flags & F_REAL) ? -kap : 0.0) : lo_of(S, t, k - 1); }
    PQP_HD double raw_up(const Slot& S, int t, int k) const { return k == 0 ? ((S.flags & F_REAL) ? kap : 0.0) : up_of(S, t, k - 1); }
    // ... and as the iteration sees it: while polishing, an active row is pinned to its bound, an inactive row is free
    PQP_HD double box_lo(const Slot& S, int t, int k) const {
        const double lo = raw_lo(S, t, k), up = raw_up(S, t, k);
        if (!polishing_) return lo;
        return (S.flags & (F_ACTLO0 << k)) ? lo : ((S.flags & (F_ACTUP0 << k)) ? up : -kInfty);
    }
    PQP_HD double box_up(const Slot& S, int t, int k) const {
        const double lo = raw_lo(S, t, k), up = raw_up(S, t, k);
        if (!polishing_) return up;
        return (S.flags & (F_ACTLO0 << k)) ? lo : ((S.flags & (F_ACTUP0 << k)) ? up : kInfty);
    }

    // column message of a waypoint's T rows to the previous waypoint (Ruiz)
    PQP_HD static void col_msg(const Slot& S, const SlotSetup& W, double* m) {
        m[0] = fmax(W.E[0] * fabs(S.a[0]), W.E[1] * fabs(S.a[2]));
        m[1] = fmax(W.E[0] * fabs(S.a[1]), W.E[1] * fabs(S.a[3]));
        m[2] = fmax(W.E[1] * fabs(S.a[4]), (S.flags & F_PREV) ? W.E[2] : 0.0);
    }

    // ---------------------------------------------------------------------------------------------
    // modified Ruiz equilibration (OSQP paper Alg. 2) on the structured KKT -> D, E, c, then the
    // penalty metrics Sigma = sigma/(c D^2) and R = rho * class * E^2 / c for the current rho
    // ---------------------------------------------------------------------------------------------
    // reuse = true keeps D, E, c of the previous pass of this QP (whose matrix differs only by the re-linearisation) and
    // only rebuilds the metrics; any positive diagonal scaling is a valid metric, OSQP's own update path re-equilibrates
    PQP_HD void ruiz(bool reuse = false) {
        if (!reuse) ruiz_equilibrate();
        // kParkScale: D, E are only read here, once per pass - between the passes they live in the workgroup slot's memory instead
        // of in 24 registers of the ADMM loop
        if constexpr (Ctx::kParkScale) {
            ctx.phase([&](int t, Lane& ln) {
                double* w = scale_slot(t);
                const int ws = scale_stride();
                if (!reuse) { _Pragma("unroll") for (int k = 0; k < 6; ++k) { w[k * ws] = ln.w.D[k]; w[(6 + k) * ws] = ln.w.E[k]; } }
                double d[6], e[6];
                _Pragma("unroll") for (int k = 0; k < 6; ++k) { d[k] = reuse ? w[k * ws] : ln.w.D[k]; e[k] = reuse ? w[(6 + k) * ws] : ln.w.E[k]; }
                _Pragma("unroll") for (int k = 0; k < 6; ++k) { ln.w.D[k] = d[k]; ln.w.E[k] = e[k]; }
            });
        }
        ruiz_metrics();
    }
    // scaling < 0: |scaling| passes of the same equilibration on ONE interior waypoint's blocks as if every waypoint carried them (the path
    // QP's matrix repeats from waypoint to waypoint up to the transition's curvature terms): every lane computes the same twelve numbers in
    // registers - no exchange, no reduction, ~1 us instead of 2.1 us per pass - and takes them as its D, E.  Any positive diagonal scaling is a
    // valid metric; this one keeps what the equilibration buys the ADMM iterations before a polish.  (The reference's setting keeps OSQP's own.)
    // the recurrence itself: `passes` passes on the blocks a[6] (absolute values of the transition into one waypoint) -> D[6], E[6], the two end rows' E, c
    PQP_HD static void nominal_scaling(const pqp_params& prm, const double* a, int passes, double* D, double* E, double* Ee, double& c) {
        const double acf = fabs(prm.front_length), acr = fabs(prm.rear_length);
        const double pk[6] = {prm.weight_l, 0.0, prm.weight_kappa, prm.weight_dkappa, prm.weight_slack, prm.weight_slack};
        _Pragma("unroll") for (int k = 0; k < 6; ++k) { D[k] = 1.0; E[k] = 1.0; }
        Ee[0] = 1.0; Ee[1] = 1.0; c = 1.0;
        for (int pass = 0; pass < passes; ++pass) {
            const double m0 = fmax(E[0] * a[0], E[1] * a[2]), m1 = fmax(E[0] * a[1], E[1] * a[3]), m2 = fmax(E[1] * a[4], E[2]);
            double cn[6], rn[6];
            cn[0] = fmax(fmax3(E[0], m0, E[4]), E[5]);
            cn[1] = fmax(fmax3(E[1], m1, E[4] * acf), E[5] * acr);
            cn[2] = fmax3(E[2], m2, E[3]);
            cn[3] = E[2] * a[5];
            cn[4] = E[4];
            cn[5] = E[5];
            rn[0] = fmax3(D[0], D[0] * a[0], D[1] * a[1]);
            rn[1] = fmax(fmax3(D[1], D[0] * a[2], D[1] * a[3]), D[2] * a[4]);
            rn[2] = fmax3(D[2], D[2], D[3] * a[5]);
            rn[3] = D[2];
            rn[4] = fmax3(D[0], D[1] * acf, D[4]);
            rn[5] = fmax3(D[0], D[1] * acr, D[5]);
            const double ren0 = D[0] * Ee[0], ren1 = D[1] * Ee[1];
            double acc = 0.0;
            _Pragma("unroll") for (int k = 0; k < 6; ++k) {
                const double cnk = fmax(cn[k] * D[k], c * D[k] * D[k] * pk[k]);
                const double rnk = rn[k] * E[k];
                D[k] = D[k] * rsq(limit_scaling(cnk));
                E[k] = E[k] * rsq(limit_scaling(rnk));
                acc += fabs(c * D[k] * D[k] * pk[k]);
            }
            Ee[0] = Ee[0] * rsq(limit_scaling(ren0));
            Ee[1] = Ee[1] * rsq(limit_scaling(ren1));
            double ct = acc * (1.0 / 6.0);
            ct = fmax(ct, 1.0);
            ct = limit_scaling(ct);
            c = c / ct;
        }
    }
    PQP_HD void ruiz_equilibrate_nominal() {
        const pqp_params& prm = A.prm;
        const int m = n / 2 > 0 ? n / 2 : 0;
        ctx.phase([&](int t, Lane& ln) {
            if (t == m) { _Pragma("unroll") for (int k = 0; k < 6; ++k) sh[L.xbuf() + k] = ln.s.a[k]; }
        });
        double c_out = 1.0;
        ctx.phase([&](int, Lane& ln) {
            double a[6], D[6], E[6], Ee[2], c;
            _Pragma("unroll") for (int k = 0; k < 6; ++k) a[k] = fabs(sh[L.xbuf() + k]);
            nominal_scaling(prm, a, -prm.scaling, D, E, Ee, c);
            const int f = ln.s.flags;
            const bool real = f & F_REAL, prev = f & F_PREV, precise = f & F_PRECISE;
            const bool colreal[6] = {real, real, real, prev, real, real && precise};
            const bool rowreal[6] = {real, real, real, real, real, real && precise};
            _Pragma("unroll") for (int k = 0; k < 6; ++k) { ln.w.D[k] = colreal[k] ? D[k] : 1.0; ln.w.E[k] = rowreal[k] ? E[k] : 1.0; }
            if (f & F_LAST) { end_rows()->E[0] = Ee[0]; end_rows()->E[1] = Ee[1]; }
            c_out = c;
        });
        cscale = ctx.uni(c_out);
    }
    PQP_HD void ruiz_equilibrate() {
        const pqp_params& prm = A.prm;
        if (prm.scaling < 0) { ruiz_equilibrate_nominal(); return; }
        cscale = 1.0;
        ctx.phase([&](int, Lane& ln) {
            _Pragma("unroll") for (int k = 0; k < 6; ++k) { ln.w.D[k] = 1.0; ln.w.E[k] = 1.0; }
            if (ln.s.flags & F_LAST) { end_rows()->E[0] = 1.0; end_rows()->E[1] = 1.0; }
        });
        int nreal_cols = 0;   // 3n + (n-1) + precise + n
        if (prm.scaling > 0) {
            double cnt[1];
            ctx.template reduce_sum<1>(cnt, [&](int, Lane& ln, double (&v)[1]) {
                const int f = ln.s.flags;
                v[0] = (f & F_REAL) ? 4.0 + ((f & F_PREV) ? 1.0 : 0.0) + ((f & F_PRECISE) ? 1.0 : 0.0) : 0.0;
            });
            nreal_cols = (int)(cnt[0] + 0.5);
        }
        for (int pass = 0; pass < prm.scaling; ++pass) {
            // exchange: column contributions of T rows go to the previous waypoint, D of X goes to the next
            ctx.phase([&](int t, Lane& ln) {
                double m[3];
                col_msg(ln.s, ln.w, m);
                _Pragma("unroll") for (int k = 0; k < 3; ++k) { sh[L.bufG() + 3 * t + k] = m[k]; sh[L.xbuf() + 3 * t + k] = ln.w.D[k]; }
            });
            double csum[1];
            const double c_now = cscale;
            ctx.template reduce_sum<1>(csum, [&](int t, Lane& ln, double (&v)[1]) {
                const Slot& S = ln.s;
                SlotSetup& W = ln.w;
                double Dp[3], mn[3];
                { const double* dp_ = nb(t > 0, L.xbuf(), 3, t - 1); _Pragma("unroll") for (int k = 0; k < 3; ++k) Dp[k] = dp_[k]; }
                { const double* mn_ = nb(t + 1 < T, L.bufG(), 3, t + 1); _Pragma("unroll") for (int k = 0; k < 3; ++k) mn[k] = mn_[k]; }
                const bool real = S.flags & F_REAL, last = S.flags & F_LAST, precise = S.flags & F_PRECISE, prev = S.flags & F_PREV;
                EndRows* er = end_rows();
                const double ee0 = last ? er->E[0] : 0.0, ee1 = last ? er->E[1] : 0.0;
                const double acf = fabs(coef_front(prm, S.flags)), acr = fabs(coef_rear(prm, S.flags));
                double cn[6], rn[6];
                // column norms of [P; A] (scaled)
                cn[0] = fmax(fmax3(W.E[0], mn[0], W.E[4]), fmax(precise ? W.E[5] : 0.0, ee0));
                cn[1] = fmax(fmax3(W.E[1], mn[1], W.E[4] * acf), fmax(precise ? W.E[5] * acr : 0.0, ee1));
                cn[2] = fmax3(W.E[2], mn[2], W.E[3]);
                cn[3] = W.E[2] * fabs(S.a[5]);
                cn[4] = W.E[4];
                cn[5] = W.E[5];
                // row norms of A (scaled)
                rn[0] = fmax3(W.D[0], Dp[0] * fabs(S.a[0]), Dp[1] * fabs(S.a[1]));
                rn[1] = fmax(fmax3(W.D[1], Dp[0] * fabs(S.a[2]), Dp[1] * fabs(S.a[3])), Dp[2] * fabs(S.a[4]));
                rn[2] = fmax3(W.D[2], prev ? Dp[2] : 0.0, W.D[3] * fabs(S.a[5]));
                rn[3] = W.D[2];
                rn[4] = fmax3(W.D[0], W.D[1] * acf, W.D[4]);
                rn[5] = fmax3(W.D[0], W.D[1] * acr, W.D[5]);
                const double ren0 = W.D[0] * ee0, ren1 = W.D[1] * ee1;
                const bool colreal[6] = {real, real, real, prev, real, real && precise};
                const bool rowreal[6] = {real, real, real, real, real, real && precise};
                double acc = 0.0;
                _Pragma("unroll") for (int k = 0; k < 6; ++k) {
                    const double pk = cost_diag(prm, S.flags, k);
                    const double cnk = fmax(cn[k] * W.D[k], c_now * W.D[k] * W.D[k] * pk);
                    const double rnk = rn[k] * W.E[k];
                    const double dnew = W.D[k] * rsq(limit_scaling(cnk));
                    const double enew = W.E[k] * rsq(limit_scaling(rnk));
                    W.D[k] = colreal[k] ? dnew : 1.0;
                    W.E[k] = rowreal[k] ? enew : 1.0;
                    acc += colreal[k] ? fabs(c_now * W.D[k] * W.D[k] * pk) : 0.0;
                }
                if (last) {
                    er->E[0] = ee0 * rsq(limit_scaling(ren0));
                    er->E[1] = ee1 * rsq(limit_scaling(ren1));
                }
                v[0] = acc;
            });
            // cost scaling: c <- c / max(mean column norm of P, ||q||_inf -> 1 when q == 0)
            double ct = csum[0] / (double)nreal_cols;
            ct = fmax(ct, 1.0);
            ct = limit_scaling(ct);
            cscale = cscale / ct;
        }
    }
    PQP_HD void ruiz_metrics() {
        const pqp_params& prm = A.prm;
        const double c = cscale, rho_now = rho;
        const double ic = 1.0 / c;
        ctx.phase([&](int t, Lane& ln) {
            Slot& S = ln.s;
            const SlotSetup& W = ln.w;
            const bool real = S.flags & F_REAL, precise = S.flags & F_PRECISE;
            _Pragma("unroll") for (int k = 0; k < 6; ++k) set_sig(S, t, k, prm.sigma * rcp(c * W.D[k] * W.D[k]));
            _Pragma("unroll") for (int k = 0; k < 3; ++k) S.rhoT[k] = real ? rho_now * kRhoEqFactor * W.E[k] * W.E[k] * ic : 0.0;
            int fl = S.flags & ~((7 * F_FREE0) | (7 * F_EQ0));      // the active-set bits of a previous polish survive
            _Pragma("unroll") for (int k = 0; k < 3; ++k) {
                const bool rowreal = real && (k < 2 || precise);
                const double e = W.E[3 + k], e2 = e * e * ic;
                const double sl = e * raw_lo(S, t, k), su = e * raw_up(S, t, k);
                const bool free_row = sl < -kInfty * kMinScaling && su > kInfty * kMinScaling;
                const bool eq_row = !free_row && (su - sl < kRhoTol);
                const double r = !rowreal ? 0.0 : (free_row ? kRhoMin * e2 : (eq_row ? rho_now * kRhoEqFactor * e2 : rho_now * e2));
                S.rhoI[k] = r;
                S.rinvI[k] = r > 0.0 ? rcp(r) : 0.0;
                if (rowreal && free_row) fl |= (F_FREE0 << k);
                if (rowreal && eq_row) fl |= (F_EQ0 << k);
            }
            S.flags = fl;
            if (S.flags & F_LAST) {
                EndRows* er = end_rows();
                const EndVals ev = end_vals();
                _Pragma("unroll") for (int k = 0; k < 2; ++k) {
                    const double e = ev.E[k], e2 = e * e * ic;
                    const double sl = e * fmax(ev.lo[k], -kInfty), su = e * fmin(ev.up[k], kInfty);
                    double rb;
                    if (sl < -kInfty * kMinScaling && su > kInfty * kMinScaling) rb = -kRhoMin * e2;
                    else if (su - sl < kRhoTol) rb = kRhoEqFactor * e2;
                    else rb = e2;
                    er->rb[k] = rb;
                    const double r = rb < 0.0 ? -rb : rho_now * rb;
                    er->rho[k] = r; er->rinv[k] = rcp(r);
                }
            }
        });
    }

    // rho changed by `ratio`: rescale every penalty that is proportional to rho
    PQP_HD void rescale_rho(double ratio) {
        const double rho_now = rho;
        ctx.phase([&](int, Lane& ln) {
            Slot& S = ln.s;
            _Pragma("unroll") for (int k = 0; k < 3; ++k) S.rhoT[k] *= ratio;
            _Pragma("unroll") for (int k = 0; k < 3; ++k) {
                const bool fr = S.flags & (F_FREE0 << k);
                const double r = fr ? S.rhoI[k] : S.rhoI[k] * ratio;
                S.rhoI[k] = r;
                S.rinvI[k] = r > 0.0 ? rcp(r) : 0.0;
            }
            if (S.flags & F_LAST) {
                EndRows* er = end_rows();
                for (int k = 0; k < 2; ++k) {
                    const double rb = er->rb[k];
                    const double r = rb < 0.0 ? -rb : rho_now * rb;
                    er->rho[k] = r; er->rinv[k] = rcp(r);
                }
            }
        });
    }

    // ---------------------------------------------------------------------------------------------
    // Solution polishing (OSQP paper section 4.2) with a KKT acceptance test.
    //   1. park the ADMM state, guess the active set from (z, y) with OSQP's rule (or keep the previous pass's set)
    //   2. equality-constrained QP on that set = the same reduced system with penalty 1/delta on active rows,
    //      0 on inactive rows, Sigma scaled to delta; `polish_refine_iter` proximal-multiplier iterations
    //      (alpha = 1) are the iterative refinement
    //   3. accept iff the polished point is primal feasible on the inactive rows, has the right dual signs on
    //      the active rows and is stationary -> it then IS the optimum of the QP; otherwise rows that fail the test
    //      change sides (primal-dual active-set rounds) or, when that stalls, the ADMM state is restored.
    // ---------------------------------------------------------------------------------------------
    static constexpr int kSaveStride = PQP_SAVE_STRIDE;   // x6 yT3 yI3 zI3 rhoI3 (+2 pad) | best polished point: x6 yT3 yI3
    static constexpr int kPolishRounds = 40; // active-set correction rounds per polish attempt

    PQP_HD double* save_slot(int t) const {
        if constexpr (Ctx::kSaveLds) return sh + L.save() + t * kSaveStride;
        else return A.wsave + ((size_t)slot * T + t) * kSaveStride;
    }
    // parked Ruiz vectors of lane t, D(6) E(6): element k at scale_slot(t)[k * scale_stride()] - the (otherwise unused) constants region
    // of the LDS layout, one array per element, or the workgroup slot's global memory
    PQP_HD double* scale_slot(int t) const {
        if constexpr (Ctx::kSaveLds && !Ctx::kCstLds) return sh + L.cst() + t;
        else return A.wscale + ((size_t)slot * T + t) * 18;
    }
    PQP_HD int scale_stride() const { return (Ctx::kSaveLds && !Ctx::kCstLds) ? T : 1; }

    // how badly inequality row k fails the KKT test at the polished point: violation of its true box when it is
    // treated as inactive, wrong-signed multiplier when it is treated as active (0 for rows that do not exist)
    PQP_HD double row_violation(const Slot& S, int t, int k, double ax) const {
        const bool rowreal = (S.flags & F_REAL) && (k < 2 || (S.flags & F_PRECISE)) && !(S.flags & (F_FREE0 << k));
        const bool alo = S.flags & (F_ACTLO0 << k), aup = S.flags & (F_ACTUP0 << k);
        const double pv = fmax(raw_lo(S, t, k) - ax, ax - raw_up(S, t, k));
        const double dv = alo ? S.yI[k] : (aup ? -S.yI[k] : 0.0);
        return rowreal ? fmax(fmax(pv, dv), 0.0) : 0.0;
    }
    PQP_HD double end_violation(const EndRows* er, int k, double ax) const {
        return end_violation_of(er->rb[k], er->lo[k], er->up[k], er->act[k], er->y[k], ax);
    }
    // ... from values (the hot callers load every field of the end rows first: no LDS round trip per field)
    PQP_HD static double end_violation_of(double rb, double lo, double up, double act, double y, double ax) {
        const double pv = fmax(lo - ax, ax - up);
        const double dv = act < 0.0 ? y : (act > 0.0 ? -y : 0.0);
        return rb < 0.0 ? 0.0 : fmax(fmax(pv, dv), 0.0);      // (rb < 0: a free row)
    }

    // --- polish piece 1: park the ADMM state; first guess of the active set by OSQP's rule
    //     (z - l < -y  /  u - z < y, in scaled units); switch the penalties of T rows and Sigma to polish values
    PQP_HD void polish_begin(bool keep_set) {
        const pqp_params& prm = A.prm;
        const double rho_now = rho;
        const double gain = 1.0 / prm.polish_delta;          // penalty of an active row (times E^2/c)
        const double sgain = prm.polish_delta / prm.sigma;    // Sigma -> delta / (c D^2)
        const double tgain = gain / (rho_now * kRhoEqFactor);
        const double irho = 1.0 / rho_now, irho_eq = 1.0 / (rho_now * kRhoEqFactor);
        ctx.phase([&](int t, Lane& ln) {
            Slot& S = ln.s;
            double* w = save_slot(t);
            if (S.flags & F_REAL) {      // (padding lanes never read their slot back)
                _Pragma("unroll") for (int k = 0; k < 6; ++k) w[k] = S.x[k];
                _Pragma("unroll") for (int k = 0; k < 3; ++k) { w[6 + k] = S.yT[k]; w[9 + k] = S.yI[k]; w[12 + k] = S.zI[k]; w[15 + k] = S.rhoI[k]; }
            }
            int fl = S.flags & ~((7 * F_ACTLO0) | (7 * F_ACTUP0));
            _Pragma("unroll") for (int k = 0; k < 3; ++k) S.rhoT[k] *= tgain;
            _Pragma("unroll") for (int k = 0; k < 3; ++k) {
                const bool fr = S.flags & (F_FREE0 << k), eq = S.flags & (F_EQ0 << k);
                const double e2 = S.rhoI[k] * (eq ? irho_eq : irho);     // E^2 / c of the row
                const bool can = !fr && S.rhoI[k] > 0.0;
                const bool act_lo = can && ((S.zI[k] - raw_lo(S, t, k)) * e2 < -S.yI[k]);
                const bool act_up = can && !act_lo && ((raw_up(S, t, k) - S.zI[k]) * e2 < S.yI[k]);
                if (act_lo) fl |= (F_ACTLO0 << k);
                if (act_up) fl |= (F_ACTUP0 << k);
            }
            S.flags = keep_set ? S.flags : fl;      // keep_set: start from the active set of the previous pass
            _Pragma("unroll") for (int k = 0; k < 6; ++k) set_sig(S, t, k, sig_of(S, t, k) * sgain);
            if (S.flags & F_LAST) {
                EndRows* er = end_rows();
                const EndVals e = end_vals();
                _Pragma("unroll") for (int k = 0; k < 2; ++k) {
                    er->sz[k] = e.z[k]; er->sy[k] = e.y[k];
                    const bool fr = e.rb[k] < 0.0;
                    const double e2 = e.E[k] * e.E[k] / cscale;
                    const bool act_lo = !fr && ((e.z[k] - e.lo[k]) * e2 < -e.y[k]);
                    const bool act_up = !fr && !act_lo && ((e.up[k] - e.z[k]) * e2 < e.y[k]);
                    er->act[k] = keep_set ? (fr ? 0.0 : e.act[k]) : (act_lo ? -1.0 : (act_up ? 1.0 : 0.0));
                }
            }
        });
    }

    // --- polish piece 2: penalties, multipliers and z of the current active set
    PQP_HD void polish_apply_set() {
        const pqp_params& prm = A.prm;
        const double rho_now = rho;
        const double gain = 1.0 / prm.polish_delta;
        const double irho = 1.0 / rho_now, irho_eq = 1.0 / (rho_now * kRhoEqFactor);
        ctx.phase([&](int t, Lane& ln) {
            Slot& S = ln.s;
            const bool real = S.flags & F_REAL;
            const double* w = save_slot(t);
            const EndVals e = end_vals();                                   // (every lane, with the phase's first loads: see iterate())
            const double w15[3] = {w[15], w[16], w[17]};                    // (one batch: read inside the loop, each was an LDS round trip of its own)
            _Pragma("unroll") for (int k = 0; k < 3; ++k) {
                const bool eq = S.flags & (F_EQ0 << k);
                const bool alo = S.flags & (F_ACTLO0 << k), aup = S.flags & (F_ACTUP0 << k);
                const bool act = real && (alo || aup);
                const double e2 = w15[k] * (eq ? irho_eq : irho);
                const double r = act ? gain * e2 : 0.0;
                // (read every candidate into a value first: a select between two lane FIELDS becomes an address select,
                //  i.e. dynamic indexing of the lane struct, which would push the whole struct into scratch memory)
                const double lo_k = raw_lo(S, t, k), up_k = raw_up(S, t, k), z_k = S.zI[k], y_k = S.yI[k];
                S.rhoI[k] = r;
                S.rinvI[k] = act ? rcp(r) : 0.0;
                S.yI[k] = act ? y_k : 0.0;
                S.zI[k] = alo ? lo_k : (aup ? up_k : z_k);
            }
            if (S.flags & F_LAST) {
                EndRows* er = end_rows();
                _Pragma("unroll") for (int k = 0; k < 2; ++k) {
                    const bool act = e.act[k] != 0.0;
                    const double r = act ? gain * e.E[k] * e.E[k] / cscale : 0.0;
                    er->rho[k] = r; er->rinv[k] = act ? rcp(r) : 0.0;
                    er->y[k] = act ? e.y[k] : 0.0;
                    er->z[k] = e.act[k] < 0.0 ? e.lo[k] : (e.act[k] > 0.0 ? e.up[k] : e.z[k]);
                }
            }
        });
    }

    // --- polish piece 3: worst KKT failure of the polished point over all inequality rows: res[5] of residuals() (one reduction
    //     for the solve's residuals and the KKT test)

    // --- polish piece 3b: the inactive rows that fail the test by more than thr, published per waypoint and row kind (0: none).
    //     A run of violated neighbouring rows of one kind is ONE bump of the path over its bound: pinning its peak removes it,
    //     pinning the whole run over-constrains the path and the surplus rows then have to be peeled off one end at a time.
    //     So only the local maxima of a run are added (ties: both).
    PQP_HD void polish_publish_adds(double thr) {
        ctx.phase([&](int t, Lane& ln) {
            const Slot& S = ln.s;
            const EndVals e = end_vals();
            double Xp[3], aT[3], aI[3];
            { const double* xp_ = nb(t > 0, L.xres(), 3, t - 1); _Pragma("unroll") for (int k = 0; k < 3; ++k) Xp[k] = xp_[k]; }
            rows_of(S, Xp, S.x, aT, aI);
            _Pragma("unroll") for (int k = 0; k < 3; ++k) {
                const bool inactive = !(S.flags & ((F_ACTLO0 << k) | (F_ACTUP0 << k)));
                const double v = row_violation(S, t, k, aI[k]);
                sh[L.bufQ() + 3 * t + k] = (inactive && v > thr) ? v : 0.0;
            }
            if (S.flags & F_LAST) {
                EndRows* er = end_rows();
                _Pragma("unroll") for (int k = 0; k < 2; ++k) {
                    const double v = end_violation_of(e.rb[k], e.lo[k], e.up[k], e.act[k], e.y[k], S.x[k]);
                    er->pad[k] = (e.act[k] == 0.0 && v > thr) ? v : 0.0;
                }
            }
        });
    }
    // (from values: vl / vo / vr = the published violations of the previous, this and the next waypoint, pad = those of the end rows; the caller loads
    //  them in one batch)
    PQP_HD static bool polish_is_peak(int k, double v, bool last, const double (&vl)[3], const double (&vo)[3], const double (&vr)[3], const double (&pad)[2]) {
        // the front and the rear circle of one waypoint (rows 1, 2) over the same bound are one bump too: pinning both fixes
        // offset and heading there, and the two then push each other out again round after round
        bool other = k == 0 ? true : (k == 1 ? v >= vo[2] : v > vo[1]);
        // ... and so are the end-state rows (offset, heading of the last waypoint) together with its two circle rows
        if (last && k >= 1) other = other && v >= pad[0] && v >= pad[1];
        return v >= vl[k] && v >= vr[k] && other;
    }

    // --- polish piece 4: primal-dual active-set step: rows failing the test by more than thr change sides
    PQP_HD void polish_update_set(double thr) {
        polish_publish_adds(thr);
        ctx.phase([&](int t, Lane& ln) {
            Slot& S = ln.s;
            double Xp[3], aT[3], aI[3];
            { const double* xp_ = nb(t > 0, L.xres(), 3, t - 1); _Pragma("unroll") for (int k = 0; k < 3; ++k) Xp[k] = xp_[k]; }
            // everything the rules below read from LDS, in one batch: the published violations of the three waypoints around this one, the end rows
            double vl[3], vo[3], vr[3];
            ld3(vl, nb(t > 0, L.bufQ(), 3, t - 1)); ld3(vo, sh + L.bufQ() + 3 * t); ld3(vr, nb(t + 1 < T, L.bufQ(), 3, t + 1));
            const EndVals e = end_vals();
            rows_of(S, Xp, S.x, aT, aI);
            int fl = S.flags;
            _Pragma("unroll") for (int k = 0; k < 3; ++k) {
                const bool alo = S.flags & (F_ACTLO0 << k), aup = S.flags & (F_ACTUP0 << k);
                const double w = row_violation(S, t, k, aI[k]);
                const bool move = w > thr && (alo || aup || polish_is_peak(k, w, S.flags & F_LAST, vl, vo, vr, e.pad));
                const bool add_lo = move && !alo && !aup && (raw_lo(S, t, k) - aI[k] > aI[k] - raw_up(S, t, k));
                const bool add_up = move && !alo && !aup && !add_lo;
#ifdef PQP_EMU_DEBUG
                if (move) printf("      row t=%d k=%d %s viol %.3e (ax %.5f lo %.5f up %.5f y %.4e)\n", t, k, (alo || aup) ? "RELEASE" : (add_lo ? "ADD_LO" : "ADD_UP"),
                                 row_violation(S, t, k, aI[k]), aI[k], raw_lo(S, t, k), raw_up(S, t, k), S.yI[k]);
#endif
                if (move && (alo || aup)) fl &= ~((F_ACTLO0 << k) | (F_ACTUP0 << k));
                if (add_lo) fl |= (F_ACTLO0 << k);
                if (add_up) fl |= (F_ACTUP0 << k);
            }
            S.flags = fl;
            if (S.flags & F_LAST) {
                EndRows* er = end_rows();
                const double vq1 = vo[1], vq2 = vo[2];      // this waypoint's own published circle-row violations
                _Pragma("unroll") for (int k = 0; k < 2; ++k) {
                    const double w = end_violation_of(e.rb[k], e.lo[k], e.up[k], e.act[k], e.y[k], S.x[k]);
                    // (polish_end_is_peak: the end rows and the last waypoint's circle rows over one bound are one bump)
                    const bool peak = (k == 0 ? w >= e.pad[1] : w > e.pad[0]) && w > vq1 && w > vq2;
                    const bool move = w > thr && (e.act[k] != 0.0 || peak);
#ifdef PQP_EMU_DEBUG
                    if (move) printf("      end row k=%d %s viol %.3e (x %.5f lo %.5f up %.5f y %.4e)\n", k, e.act[k] != 0.0 ? "RELEASE" : "ADD", w, S.x[k], e.lo[k], e.up[k], e.y[k]);
#endif
                    const double moved = e.act[k] != 0.0 ? 0.0 : ((e.lo[k] - S.x[k] > S.x[k] - e.up[k]) ? -1.0 : 1.0);
                    er->act[k] = move ? moved : e.act[k];
                }
            }
        });
    }

    // --- polish piece 4b: remember the polished point with the smallest KKT failure seen in this attempt
    PQP_HD void polish_save_best() {
        ctx.phase([&](int t, Lane& ln) {
            const Slot& S = ln.s;
            double* w = save_slot(t) + 20;
            if (S.flags & F_REAL) {
                _Pragma("unroll") for (int k = 0; k < 6; ++k) w[k] = S.x[k];
                _Pragma("unroll") for (int k = 0; k < 3; ++k) { w[6 + k] = S.yT[k]; w[9 + k] = S.yI[k]; }
            }
            if (S.flags & F_LAST) { end_rows()->by[0] = end_rows()->y[0]; end_rows()->by[1] = end_rows()->y[1]; }
        });
    }

    // --- polish piece 5: leave polish mode.  accept: keep (x, y).  reject: restore the ADMM iterate, or (reseed) continue
    //     ADMM from the best polished point of the attempt: (x, y) from there, z = clip(A x) as in osqp_warm_start.
    //     Either way put the ADMM penalties back.
    PQP_HD void polish_end(bool ok, bool reseed = false) {
        const pqp_params& prm = A.prm;
        const double rho_now = rho;
        const double gain = 1.0 / prm.polish_delta;
        const double sgain = prm.polish_delta / prm.sigma;
        const double tgain = gain / (rho_now * kRhoEqFactor);
        const double itgain = 1.0 / tgain, isgain = 1.0 / sgain;
        ctx.phase([&](int t, Lane& ln) {
            Slot& S = ln.s;
            const bool real = S.flags & F_REAL;
            const double* w = save_slot(t);
            const double* wb = w + (reseed ? 20 : 0);
            const bool rest = real && !ok;
            _Pragma("unroll") for (int k = 0; k < 6; ++k) S.x[k] = rest ? wb[k] : S.x[k];
            _Pragma("unroll") for (int k = 0; k < 3; ++k) {
                S.yT[k] = rest ? wb[6 + k] : S.yT[k];
                S.yI[k] = rest ? wb[9 + k] : S.yI[k];
                S.zI[k] = rest ? w[12 + k] : S.zI[k];
                const double r = real ? w[15 + k] : 0.0;
                S.rhoI[k] = r;
                S.rinvI[k] = r > 0.0 ? rcp(r) : 0.0;
                S.rhoT[k] *= itgain;
            }
            _Pragma("unroll") for (int k = 0; k < 6; ++k) set_sig(S, t, k, sig_of(S, t, k) * isgain);
            if (S.flags & F_LAST) {
                EndRows* er = end_rows();
                double e_rb[2], e_sz[2], e_sy[2], e_by[2];
                _Pragma("unroll") for (int k = 0; k < 2; ++k) { e_rb[k] = er->rb[k]; e_sz[k] = er->sz[k]; e_sy[k] = er->sy[k]; e_by[k] = er->by[k]; }
                _Pragma("unroll") for (int k = 0; k < 2; ++k) {
                    if (!ok) { er->z[k] = e_sz[k]; er->y[k] = reseed ? e_by[k] : e_sy[k]; }
                    const double rb = e_rb[k];
                    const double r = rb < 0.0 ? -rb : rho_now * rb;
                    er->rho[k] = r; er->rinv[k] = rcp(r);
                }
            }
            _Pragma("unroll") for (int k = 0; k < 3; ++k) sh[L.xbuf() + 3 * t + k] = S.x[k];
        });
        if (!ok && reseed) {
            ctx.phase([&](int t, Lane& ln) {
                Slot& S = ln.s;
                double Xp[3], aT[3], aI[3];
                { const double* xp_ = nb(t > 0, L.xbuf(), 3, t - 1); _Pragma("unroll") for (int k = 0; k < 3; ++k) Xp[k] = xp_[k]; }
                rows_of(S, Xp, S.x, aT, aI);
                const bool real = S.flags & F_REAL;
                _Pragma("unroll") for (int k = 0; k < 3; ++k) {
                    const double lo_k = raw_lo(S, t, k), up_k = raw_up(S, t, k);
                    S.zI[k] = real ? fmin(fmax(aI[k], lo_k), up_k) : 0.0;
                }
                if (S.flags & F_LAST) {
                    EndRows* er = end_rows();
                    for (int k = 0; k < 2; ++k) er->z[k] = fmin(fmax(S.x[k], er->lo[k]), er->up[k]);
                }
            });
        }
    }

    // ---------------------------------------------------------------------------------------------
    // factorisation for the current penalties: closed-form eliminations, block cyclic reduction over the T nodes
    // ---------------------------------------------------------------------------------------------
    PQP_HD void factor() {
        const pqp_params& prm = A.prm;
        factors_ += 1;
        PQP_IT_BEGIN();
        // F1: own diagonal block + message (M, Lc) to the previous waypoint
        ctx.phase([&](int t, Lane& ln) {
            Slot& S = ln.s;
            SlotSetup& W = ln.w;
            double re0 = 0.0, re1 = 0.0;
            if (S.flags & F_LAST) { re0 = end_rows()->rho[0]; re1 = end_rows()->rho[1]; }
            const double rK = S.rhoI[0], rF = S.rhoI[1], rR = S.rhoI[2];
            const double dsf = cost_diag(prm, S.flags, 4) + sig_of(S, t, 4) + rF, dsr = cost_diag(prm, S.flags, 5) + sig_of(S, t, 5) + rR;
            const double idsf = rcp(dsf), idsr = rcp(dsr);
            if constexpr (kAcc) { acc_set(S, C_IDSF, idsf); acc_set(S, C_IDSR, idsr); }
            else if constexpr (kCst) { cst_set(t, C_IDSF, idsf); cst_set(t, C_IDSR, idsr); } else { S.idsf = idsf; S.idsr = idsr; }
            S.cF = rF * idsf; S.cR = rR * idsr;
            const double gf = rF - rF * S.cF, gr = rR - rR * S.cR;
            const double ds = S.a[5];
            S.tu = S.rhoT[2] * ds;
            const double du = cost_diag(prm, S.flags, 3) + sig_of(S, t, 3) + S.tu * ds;
            S.idu = rcp(du);
            S.tudc = S.tu * S.idu;
            const double gu = S.rhoT[2] - S.tu * S.tudc;
            const double cf = coef_front(prm, S.flags), cr = coef_rear(prm, S.flags);
            W.Dg[0] = cost_diag(prm, S.flags, 0) + sig_of(S, t, 0) + S.rhoT[0] + gf + gr + re0;
            W.Dg[1] = gf * cf + gr * cr;
            W.Dg[2] = 0.0;
            W.Dg[3] = cost_diag(prm, S.flags, 1) + sig_of(S, t, 1) + S.rhoT[1] + gf * cf * cf + gr * cr * cr + re1;
            W.Dg[4] = 0.0;
            W.Dg[5] = cost_diag(prm, S.flags, 2) + sig_of(S, t, 2) + gu + rK;
            const double r0 = S.rhoT[0], r1 = S.rhoT[1];
            const double a00 = S.a[0], a01 = S.a[1], a10 = S.a[2], a11 = S.a[3], a12 = S.a[4];
            const double gup = (S.flags & F_PREV) ? gu : 0.0;
            // coupling block: rows = previous waypoint's (l,psi,k), cols = own.  The factorisation carries the NEGATED couplings C' = -C:
            // the coupling a level creates between the two neighbours of an eliminated node, -(Lc D^-1) Rc, is then the plain product
            // Lc' D^-1 Rc' (no negation of nine entries per level), the Schur updates Lc D^-1 Lc^T do not see the sign, and the solves
            // add GL' r / GL'^T x where they subtracted GL r / GL^T x - the same values bit for bit.
            W.Lc[0] = r0 * a00; W.Lc[1] = r1 * a10; W.Lc[2] = 0.0;
            W.Lc[3] = r0 * a01; W.Lc[4] = r1 * a11; W.Lc[5] = 0.0;
            W.Lc[6] = 0.0;      W.Lc[7] = r1 * a12; W.Lc[8] = gup;
            // contribution of this waypoint's T rows to the previous waypoint's diagonal block
            double* f = sh + L.fbuf() + 15 * t;
            f[0] = r0 * a00 * a00 + r1 * a10 * a10;
            f[1] = r0 * a00 * a01 + r1 * a10 * a11;
            f[2] = r1 * a10 * a12;
            f[3] = r0 * a01 * a01 + r1 * a11 * a11;
            f[4] = r1 * a11 * a12;
            f[5] = r1 * a12 * a12 + gup;
            _Pragma("unroll") for (int k = 0; k < 9; ++k) f[6 + k] = W.Lc[k];
        });
        PQP_IT(8);       // F1
        // F2: receive from the next waypoint
        ctx.phase([&](int t, Lane& ln) {
            SlotSetup& W = ln.w;
            const double* f = nb(t + 1 < T, L.fbuf(), 15, t + 1);
            _Pragma("unroll") for (int k = 0; k < 6; ++k) W.Dg[k] += f[k];
            _Pragma("unroll") for (int k = 0; k < 9; ++k) W.Rc[k] = f[6 + k];
        });
        PQP_IT(9);       // F2
        // Cyclic-reduction tree over tp = t + 1 in [1, T]: level h eliminates tp = h (mod 2h), the root is tp = T (the last
        // thread).  With this numbering the only node of a wavefront that talks to the next wavefront during the levels
        // h < 64 is its LAST lane (tp = 64m), and that node only RECEIVES until its own elimination at a level >= 64 — which
        // is what lets iterate() run the 6 in-wave levels without workgroup barriers.
        // (root_is_padding(): the tree ends one level earlier, see iterate())
        const int h_last = root_is_padding() ? (T >> 1) : T;
        _Pragma("nounroll") for (int h = 1; h <= h_last; h <<= 1) {
            ctx.phase([&](int t, Lane& ln) {
                Slot& S = ln.s;
                SlotSetup& W = ln.w;
                const int tp = t + 1;
                if (h > 1) {
                    const int hp = h >> 1;   // stride of the level just eliminated
                    const bool surv = (tp & (h - 1)) == 0;
                    const bool hr = surv && (t + hp < T), hl = surv && (t - hp >= 0);
                    const double* fr = nb(hr, L.fbuf(), 21, t + hp);       // (no neighbour: 21 zeros)
                    const double* fl = nb(hl, L.fbuf(), 21, t - hp);
                    _Pragma("unroll") for (int k = 0; k < 6; ++k) W.Dg[k] -= fr[k] + fl[6 + k];
                    // (ONE branch around all eighteen loads: written as selects, every load gets its own exec-mask change)
                    if (surv) { _Pragma("unroll") for (int k = 0; k < 9; ++k) { W.Rc[k] = fr[12 + k]; W.Lc[k] = fl[12 + k]; } }
                }
                const bool elim = (h < T) ? ((tp & (2 * h - 1)) == h) : (tp == T);
                if (elim) {      // every thread is eliminated at exactly one level (the last thread: the root)
                    sym3_inv(W.Dg, S.Dinv);
                    mat3_mul_sym3(W.Lc, S.Dinv, S.GL);
                    mat3t_mul_sym3(W.Rc, S.Dinv, S.GR);
                    if (h < T) {
                        double* f = sh + L.fbuf() + 21 * t;
                        mat3_mul_mat3t_sym(S.GL, W.Lc, f);
                        mat3_mul_mat3_sym(S.GR, W.Rc, f + 6);
                        mat3_mul_mat3(S.GL, W.Rc, f + 12);
                    }
                }
            });
            // level 1 | 2 .. 8 | 16, 32 | 64 ...  (constant indices: a computed one would put the solver object in scratch memory)
            if (h == 1) PQP_IT(10); else if (h <= 8) PQP_IT(11); else if (h <= 32) PQP_IT(12); else PQP_IT(13);
        }
    }

    // ---------------------------------------------------------------------------------------------
    // A x for the rows of a waypoint (T rows need the previous waypoint's X)
    // ---------------------------------------------------------------------------------------------
    PQP_HD void rows_of(const Slot& S, const double* Xp, const double* x, double* aT, double* aI) const {
        const double cf = coef_front(A.prm, S.flags), cr = coef_rear(A.prm, S.flags);
        aT[0] = S.a[0] * Xp[0] + S.a[1] * Xp[1] - x[0];
        aT[1] = S.a[2] * Xp[0] + S.a[3] * Xp[1] + S.a[4] * Xp[2] - x[1];
        aT[2] = ((S.flags & F_PREV) ? Xp[2] : 0.0) + S.a[5] * x[3] - x[2];
        aI[0] = x[2];
        aI[1] = x[0] + cf * x[1] + x[4];
        aI[2] = (S.flags & F_PRECISE) ? x[0] + cr * x[1] + x[5] : 0.0;
    }

    // message a waypoint sends to the previous one: A_in^T w restricted to (l,psi,k)_{i-1}
    PQP_HD static void back_msg(const Slot& S, const double* wT, double* g) {
        g[0] = S.a[0] * wT[0] + S.a[2] * wT[1];
        g[1] = S.a[1] * wT[0] + S.a[3] * wT[1];
        g[2] = S.a[4] * wT[1] + ((S.flags & F_PREV) ? wT[2] : 0.0);
    }

    // ---------------------------------------------------------------------------------------------
    // one ADMM iteration
    // ---------------------------------------------------------------------------------------------
    // Contexts with kDpp (the device): the exchanges between lanes of one row of 16 - levels h <= 8 of both passes, the message of the
    // next waypoint, the X~ of the previous one - move through DPP operands inside VALU instructions instead of an LDS write + read
    // (9 of the 15 LDS round trips of an iteration, each ~100 exposed cycles at one wavefront per SIMD).  A row's last lane (tp = 16m)
    // survives every level below 16 and only RECEIVES from the next row: it takes those messages from LDS at level 16, all at once -
    // the same deferral the wavefront's last lane uses across the workgroup barrier.
    template <int H>
    PQP_HD void forward_level_dpp(double (&gm)[3], double (&qm)[3], double (&pm)[3]) {
        ctx.phase_w([&](int t, Lane& ln) {
            Slot& S = ln.s;
            const int tp = t + 1;
            if constexpr (H == 1) {
                _Pragma("unroll") for (int k = 0; k < 3; ++k) S.r[k] += ctx.template lane_above<1>(gm[k]);
            } else {
                double qv[3], pv[3];
                _Pragma("unroll") for (int k = 0; k < 3; ++k) { qv[k] = ctx.template lane_above<(H >> 1)>(qm[k]); pv[k] = ctx.template lane_below<(H >> 1)>(pm[k]); }
                if ((tp & (H - 1)) == 0) { _Pragma("unroll") for (int k = 0; k < 3; ++k) S.r[k] += qv[k] + pv[k]; }
            }
            if ((tp & (2 * H - 1)) == H) {
                mat3_vec(S.GL, S.r, qm);
                _Pragma("unroll") for (int k = 0; k < 3; ++k) sh[L.bufQ() + 3 * t + k] = qm[k];       // (for the row's / wavefront's last lane)
                mat3_vec(S.GR, S.r, pm);
                if constexpr (H >= 8) { _Pragma("unroll") for (int k = 0; k < 3; ++k) sh[L.bufP() + 3 * t + k] = pm[k]; }
            }
        });
    }
    template <int H>
    PQP_HD void backward_level_dpp(const double (&xpv)[3]) {
        ctx.phase_w([&](int t, Lane& ln) {
            Slot& S = ln.s;
            const int tp = t + 1;
            double xl[3], xr[3];
            _Pragma("unroll") for (int k = 0; k < 3; ++k) { xl[k] = ctx.template lane_below<H>(S.xt[k]); xr[k] = ctx.template lane_above<H>(S.xt[k]); }
            if ((t & 15) < H) { _Pragma("unroll") for (int k = 0; k < 3; ++k) xl[k] = xpv[k]; }       // t - H is the previous row's last lane
            if ((tp & (2 * H - 1)) == H) {
                double x3[3], p[3], pr[3];
                sym3_vec(S.Dinv, S.r, x3);
                mat3t_vec(S.GL, xl, p);
                mat3t_vec(S.GR, xr, pr);
                _Pragma("unroll") for (int k = 0; k < 3; ++k) S.xt[k] = x3[k] + p[k] + pr[k];
            }
        });
    }

    PQP_HD void iterate() {
        const pqp_params& prm = A.prm;
        kkt_solves_ += 1;
        const double alpha = alpha_;
        constexpr bool D = Ctx::kDpp;
        double gm[3] = {0.0, 0.0, 0.0}, qm[3] = {0.0, 0.0, 0.0}, pm[3] = {0.0, 0.0, 0.0}, xpv[3] = {0.0, 0.0, 0.0};    // kDpp: per-lane transients
        PQP_IT_BEGIN();
        // I1: w = R z - y, reduced right-hand side pieces, message to the previous waypoint.  Wave-local: the message is read
        // by the previous lane (the last lane of a wavefront reads its neighbour's after the barrier below).
        ctx.phase_w([&](int t, Lane& ln) {
            Slot& S = ln.s;
            const double cf = coef_front(prm, S.flags), cr = coef_rear(prm, S.flags);
            double wT[3], wI[3];
            _Pragma("unroll") for (int k = 0; k < 3; ++k) wT[k] = S.rhoT[k] * S.bT[k] - S.yT[k];
            _Pragma("unroll") for (int k = 0; k < 3; ++k) wI[k] = S.rhoI[k] * S.zI[k] - S.yI[k];
            // (the end rows are loaded by EVERY lane, not under `if (F_LAST)`: the loads then leave with the phase's first instructions and their
            //  latency is over before the one lane that uses them gets there; under the branch that lane - and everybody behind it - waits for them)
            const EndRows* er1 = end_rows();
            const double e_rho0 = er1->rho[0], e_rho1 = er1->rho[1], e_z0 = er1->z[0], e_z1 = er1->z[1], e_y0 = er1->y[0], e_y1 = er1->y[1];
            const bool last1 = S.flags & F_LAST;
            const double we0 = last1 ? e_rho0 * e_z0 - e_y0 : 0.0, we1 = last1 ? e_rho1 * e_z1 - e_y1 : 0.0;
            S.rv = sig_of(S, t, 3) * S.x[3] + S.a[5] * wT[2];
            S.rsf = sig_of(S, t, 4) * S.x[4] + wI[1];
            S.rsr = sig_of(S, t, 5) * S.x[5] + wI[2];
            const double tud = S.tudc * S.rv;
            double g[3];
            back_msg(S, wT, g);
            g[2] -= tud;
            const double eF = S.cF * S.rsf, eR = S.cR * S.rsr;
            S.r[0] = sig_of(S, t, 0) * S.x[0] - wT[0] + wI[1] + wI[2] + we0 - eF - eR;
            S.r[1] = sig_of(S, t, 1) * S.x[1] - wT[1] + cf * wI[1] + cr * wI[2] + we1 - cf * eF - cr * eR;
            S.r[2] = sig_of(S, t, 2) * S.x[2] - wT[2] + wI[0] + tud;
            _Pragma("unroll") for (int k = 0; k < 3; ++k) sh[L.bufG() + 3 * t + k] = g[k];
            if constexpr (D) { _Pragma("unroll") for (int k = 0; k < 3; ++k) gm[k] = g[k]; }
        });
        // Forward pass.  Levels h < 64 stay inside a wavefront (wave-local phases: no workgroup barrier); a wave's last lane
        // (tp = 64m, a survivor of all of them) defers what it would receive from the next wavefront - the message g of
        // waypoint t+1 and the level messages of the nodes t + hp - and adds them after ONE barrier.  Levels h >= 64 and the
        // root use workgroup barriers.  (T = 64: no barrier at all.)
        const int hw = T < 64 ? T : 64;
        PQP_IT(0);       // I1
        if constexpr (D) {
            forward_level_dpp<1>(gm, qm, pm);
            forward_level_dpp<2>(gm, qm, pm);
            forward_level_dpp<4>(gm, qm, pm);
            forward_level_dpp<8>(gm, qm, pm);
            PQP_IT(1);   // forward levels 1 .. 8 (DPP)
            // level 16: its survivors are the rows' last lanes; everything they have deferred arrives now (not the wavefront's last
            // lane: it waits for the barrier below)
            ctx.phase_w([&](int t, Lane& ln) {
                Slot& S = ln.s;
                const int tp = t + 1;
                const bool surv = (tp & 15) == 0, wedge = (tp & 63) == 0;
                const bool hr = surv && !wedge, hl = surv && (t - 8 >= 0);
                double q_[3], p_[3], g_[3], q1[3], q2[3], q4[3];
                ld3(q_, nb(hr, L.bufQ(), 3, t + 8)); ld3(p_, nb(hl, L.bufP(), 3, t - 8)); ld3(g_, nb(hr, L.bufG(), 3, t + 1));
                ld3(q1, nb(hr, L.bufQ(), 3, t + 1)); ld3(q2, nb(hr, L.bufQ(), 3, t + 2)); ld3(q4, nb(hr, L.bufQ(), 3, t + 4));
                ctx.join(q_, p_, g_, q1, q2, q4);
                _Pragma("unroll") for (int k = 0; k < 3; ++k) S.r[k] += g_[k] + (q1[k] + q2[k] + q4[k]) + (q_[k] + p_[k]);
                if ((tp & 31) == 16) {
                    double p[3];
                    mat3_vec(S.GL, S.r, p);
                    _Pragma("unroll") for (int k = 0; k < 3; ++k) sh[L.bufQ() + 3 * t + k] = p[k];
                    mat3_vec(S.GR, S.r, p);
                    _Pragma("unroll") for (int k = 0; k < 3; ++k) sh[L.bufP() + 3 * t + k] = p[k];
                }
            });
        }
        for (int h = D ? 32 : 1; h < hw; h <<= 1) {
            ctx.phase_w([&](int t, Lane& ln) {
                Slot& S = ln.s;
                const int tp = t + 1;
                const bool edge = (tp & 63) == 0;            // last lane of its wavefront
                if (h == 1) {
                    const bool has = (t + 1 < T) && !edge;
                    const double* g_ = nb(has, L.bufG(), 3, t + 1);
                    _Pragma("unroll") for (int k = 0; k < 3; ++k) S.r[k] += g_[k];
                } else {
                    const int hp = h >> 1;
                    const bool surv = (tp & (h - 1)) == 0;
                    const bool hr = surv && (t + hp < T) && !edge, hl = surv && (t - hp >= 0);
                    double q_[3], p_[3];
                    ld3(q_, nb(hr, L.bufQ(), 3, t + hp)); ld3(p_, nb(hl, L.bufP(), 3, t - hp));
                    _Pragma("unroll") for (int k = 0; k < 3; ++k) S.r[k] += q_[k] + p_[k];
                }
                if ((tp & (2 * h - 1)) == h) {
                    double p[3];
                    mat3_vec(S.GL, S.r, p);
                    _Pragma("unroll") for (int k = 0; k < 3; ++k) sh[L.bufQ() + 3 * t + k] = p[k];
                    mat3_vec(S.GR, S.r, p);
                    _Pragma("unroll") for (int k = 0; k < 3; ++k) sh[L.bufP() + 3 * t + k] = p[k];
                }
            });
        }
        PQP_IT(2);       // forward levels 16, 32 (LDS inside the wavefront)
        ctx.phase([&](int, Lane&) {});      // the one barrier between the in-wave levels and the cross-wave part
        const bool short_tree = root_is_padding();
        for (int h = hw; h <= T; h <<= 1) {
            if (short_tree && h == T) break;
            const bool root_here = h == T || (short_tree && h == (T >> 1));       // (h is a compile-time value once the loop is unrolled)
            ctx.phase([&](int t, Lane& ln) {
                Slot& S = ln.s;
                const int tp = t + 1;
                const bool edge = (tp & 63) == 0;
                // this level's own messages (h > 1: T >= 64 always) - loaded BEFORE the deferred sums below, so that they travel with that batch
                double lq[3], lp[3];
                {
                    const int hp = h >> 1;
                    const bool surv = (tp & (h - 1)) == 0;
                    const bool hr = surv && (t + hp < T), hl = surv && (t - hp >= 0);
                    ld3(lq, nb(hr, L.bufQ(), 3, t + hp)); ld3(lp, nb(hl, L.bufP(), 3, t - hp));
                }
                if (h == hw && T > hw) {
                    // deferred (the wavefront's last lane; every other lane reads the zero block): message of waypoint t+1 and the right-hand level
                    // messages of all in-wave levels.  ALL loads first, then the sums in the old order: written as "if (in range) r += sh[..]" per
                    // source, every source was a load -> wait -> add of its own - six LDS latencies in a row on the path both wavefronts wait for
                    // (round 6: 0.9 us of a 3.2 us solve between the two barriers around the root)
                    double dv[6][3];
                    ld3(dv[0], nb(edge && t + 1 < T, L.bufG(), 3, t + 1));
                    _Pragma("unroll") for (int j = 0; j < 5; ++j) {
                        const int hp = 1 << j;
                        ld3(dv[1 + j], nb(edge && hp < (hw >> 1) && t + hp < T, L.bufQ(), 3, t + hp));
                    }
                    ctx.join(dv[0], dv[1], dv[2], dv[3], dv[4], dv[5], lq, lp);
                    _Pragma("unroll") for (int j = 0; j < 6; ++j) { _Pragma("unroll") for (int k = 0; k < 3; ++k) S.r[k] += dv[j][k]; }
                }

                _Pragma("unroll") for (int k = 0; k < 3; ++k) S.r[k] += lq[k] + lp[k];
                if (!root_here) {
                    if ((tp & (2 * h - 1)) == h) {
                        double p[3];
                        mat3_vec(S.GL, S.r, p);
                        _Pragma("unroll") for (int k = 0; k < 3; ++k) sh[L.bufQ() + 3 * t + k] = p[k];
                        mat3_vec(S.GR, S.r, p);
                        _Pragma("unroll") for (int k = 0; k < 3; ++k) sh[L.bufP() + 3 * t + k] = p[k];
                    }
                } else {
                    double x3[3];
                    sym3_vec(S.Dinv, S.r, x3);
                    // only the root's value survives (every other node is overwritten at its backward level); with the short tree the
                    // padding root tp = T has no backward level and no factor: its x~ is 0
                    _Pragma("unroll") for (int k = 0; k < 3; ++k) S.xt[k] = (h != T && tp == T) ? 0.0 : x3[k];
                    if (tp == h) { _Pragma("unroll") for (int k = 0; k < 3; ++k) sh[L.xbuf() + 3 * t + k] = x3[k]; }
                }
            });
        }
        PQP_IT(3);       // barrier + cross-wave levels / root
        // Backward pass: cross-wave levels with barriers, then the in-wave levels wave-locally (what they read from another
        // wavefront - the x of its last lane - was written before the last barrier).
        for (int h = T >> 1; h >= (D ? 16 : 1); h >>= 1) {
            if (short_tree && h == (T >> 1)) continue;          // that node was the root
            auto body = [&](int t, Lane& ln) {
                Slot& S = ln.s;
                const int tp = t + 1;
                const bool act = (tp & (2 * h - 1)) == h;
                const bool hl = act && (t - h >= 0), hr = act && (t + h < T);
                double xl[3], xr[3];
                ld3(xl, nb(hl, L.xbuf(), 3, t - h));         // no neighbour: x = 0 from the zero block
                ld3(xr, nb(hr, L.xbuf(), 3, t + h));
                double x3[3], p[3], pr[3];
                sym3_vec(S.Dinv, S.r, x3);
                mat3t_vec(S.GL, xl, p);
                mat3t_vec(S.GR, xr, pr);
                _Pragma("unroll") for (int k = 0; k < 3; ++k) {
                    const double v = x3[k] + p[k] + pr[k];
                    S.xt[k] = act ? v : S.xt[k];
                    if (act) sh[L.xbuf() + 3 * t + k] = v;
                }
            };
            if (h >= 64) ctx.phase(body); else ctx.phase_w(body);
        }
        PQP_IT(4);       // backward levels >= 16
        if constexpr (D) {
            // X~ of the previous row's last lane (solved at a level >= 16): row_bcast inside the wavefront, LDS across wavefronts
            ctx.phase_w([&](int t, Lane& ln) {
                const int w0 = t & ~63;
                const double* xw = nb(w0 > 0, L.xbuf(), 3, w0 - 1);
                _Pragma("unroll") for (int k = 0; k < 3; ++k) {
                    const double b = ctx.prev_row_last(ln.s.xt[k]);
                    xpv[k] = (t & 63) < 16 ? xw[k] : b;
                }
            });
            backward_level_dpp<8>(xpv);
            backward_level_dpp<4>(xpv);
            backward_level_dpp<2>(xpv);
            backward_level_dpp<1>(xpv);
        }
        PQP_IT(5);       // backward levels 8 .. 1 (DPP)
        // I3: back-substitute v, sf, sr; z~ = A x~; relaxed updates, projection, dual update.  Wave-local: a lane reads the X~ of the
        // previous one; the first lane of a wavefront reads the last lane of the previous wavefront, which wrote its X~ before
        // the last workgroup barrier of the backward pass.  iterate() therefore ENDS WITHOUT A BARRIER: the next iterate() may
        // follow directly, anything else (residuals, cold operations) must synchronise first - sync_after_iterate().
        ctx.phase_w([&](int t, Lane& ln) {
            Slot& S = ln.s;
            // the two end rows' fields, loaded by every lane at the top of the phase (see I1); used by the lane that owns the rows at the bottom
            double e_lo[2], e_up[2], e_z[2], e_y[2], e_rho[2], e_rinv[2], e_act[2];
            {
                const EndRows* er = end_rows();
                _Pragma("unroll") for (int k = 0; k < 2; ++k) {
                    e_lo[k] = er->lo[k]; e_up[k] = er->up[k]; e_z[k] = er->z[k]; e_y[k] = er->y[k];
                    e_rho[k] = er->rho[k]; e_rinv[k] = er->rinv[k]; e_act[k] = er->act[k];
                }
            }
            double Xp[3];
            if constexpr (D) {
                _Pragma("unroll") for (int k = 0; k < 3; ++k) { const double b = ctx.template lane_below<1>(S.xt[k]); Xp[k] = (t & 15) ? b : xpv[k]; }
            } else {
                const double* xp_ = nb(t > 0, L.xbuf(), 3, t - 1); _Pragma("unroll") for (int k = 0; k < 3; ++k) Xp[k] = xp_[k];
            }
            const double cf = coef_front(prm, S.flags), cr = coef_rear(prm, S.flags);
            double xt[6];
            xt[0] = S.xt[0]; xt[1] = S.xt[1]; xt[2] = S.xt[2];
            xt[3] = (S.rv - S.tu * (((S.flags & F_PREV) ? Xp[2] : 0.0) - xt[2])) * S.idu;
            xt[4] = (S.rsf - S.rhoI[1] * (xt[0] + cf * xt[1])) * idsf_of(S, t);
            xt[5] = (S.rsr - S.rhoI[2] * (xt[0] + cr * xt[1])) * idsr_of(S, t);
            double zT[3], zI[3];
            rows_of(S, Xp, xt, zT, zI);
            _Pragma("unroll") for (int k = 0; k < 6; ++k) S.x[k] = alpha * xt[k] + (1.0 - alpha) * S.x[k];
            double* dyp = sh + L.yprev() + 6 * t;       // y_k - y_{k-1} of this iteration, for the infeasibility certificate
            _Pragma("unroll") for (int k = 0; k < 3; ++k) {
                const double d = S.rhoT[k] * alpha * (zT[k] - S.bT[k]);
                S.yT[k] += d;
                if (CERT) dyp[k] = d;
            }
            _Pragma("unroll") for (int k = 0; k < 3; ++k) {
                const double zh = alpha * zI[k] + (1.0 - alpha) * S.zI[k];
                const double v = zh + S.yI[k] * S.rinvI[k];
                const double zn = fmin(fmax(v, box_lo(S, t, k)), box_up(S, t, k));
                const double d = S.rhoI[k] * (zh - zn);
                S.yI[k] += d;
                if (CERT) dyp[3 + k] = d;
                S.zI[k] = zn;
            }
            if (S.flags & F_LAST) {
                // The two end rows, by the one lane that owns them while its wavefront (and, at the next barrier, the other one) waits: every field
                // was loaded at the top of the phase and nothing branches - written row by row with loads and `if (polishing_)` inside, this was a
                // chain of seven LDS round trips per solve (round 6: 0.25 us of 3.2)
                EndRows* er = end_rows();
                const bool pol = polishing_;
                _Pragma("unroll") for (int k = 0; k < 2; ++k) {
                    const double zh = alpha * xt[k] + (1.0 - alpha) * e_z[k];
                    const double v = zh + e_y[k] * e_rinv[k];
                    const double bnd = e_act[k] < 0.0 ? e_lo[k] : e_up[k];
                    const bool pin = pol && e_act[k] != 0.0;
                    const double elo = pin ? bnd : (pol ? -kInfty : e_lo[k]);
                    const double eup = pin ? bnd : (pol ? kInfty : e_up[k]);
                    const double zn = fmin(fmax(v, elo), eup);
                    const double d = e_rho[k] * (zh - zn);
                    er->y[k] = e_y[k] + d;
                    if (CERT) er->yp[k] = d;
                    er->z[k] = zn;
                }
            }
        });
        PQP_IT(6);       // I3
    }

    PQP_HD void sync_after_iterate() { ctx.phase([&](int, Lane&) {}); }

    // ---------------------------------------------------------------------------------------------
    // residuals (unscaled inf-norms, OSQP termination quantities):
    //   res[0] = ||Ax - z||, res[1] = ||Px + A^T y|| (q == 0), res[2] = max(||Ax||, ||z||), res[3] = max(||Px||, ||A^T y||)
    //   res[4] = 1 if an iterate is not finite
    //   res[5] = worst KKT failure over the inequality rows for the current active set (only meaningful while polishing)
    // ---------------------------------------------------------------------------------------------
    PQP_HD void residuals(double (&res)[6]) {
        const pqp_params& prm = A.prm;
        ctx.phase([&](int t, Lane& ln) {
            double g[3];
            back_msg(ln.s, ln.s.yT, g);
            _Pragma("unroll") for (int k = 0; k < 3; ++k) { sh[L.xres() + 3 * t + k] = ln.s.x[k]; sh[L.bufG() + 3 * t + k] = g[k]; }
        });
        ctx.template reduce_max<6>(res, [&](int t, Lane& ln, double (&v)[6]) {
            const Slot& S = ln.s;
            double e_z[2], e_y[2], e_rb[2], e_lo[2], e_up[2], e_act[2];       // the end rows, loaded by every lane with the phase's first loads (see iterate())
            { const EndRows* er = end_rows(); _Pragma("unroll") for (int k = 0; k < 2; ++k) { e_z[k] = er->z[k]; e_y[k] = er->y[k]; e_rb[k] = er->rb[k]; e_lo[k] = er->lo[k]; e_up[k] = er->up[k]; e_act[k] = er->act[k]; } }
            double Xp[3], gn[3];
            { const double* xp_ = nb(t > 0, L.xres(), 3, t - 1); _Pragma("unroll") for (int k = 0; k < 3; ++k) Xp[k] = xp_[k]; }
            { const double* gn_ = nb(t + 1 < T, L.bufG(), 3, t + 1); _Pragma("unroll") for (int k = 0; k < 3; ++k) gn[k] = gn_[k]; }
            const bool real = S.flags & F_REAL;
            const double cf = coef_front(prm, S.flags), cr = coef_rear(prm, S.flags);
            double aT[3], aI[3];
            rows_of(S, Xp, S.x, aT, aI);
            double pr = 0.0, nz = 0.0;
            _Pragma("unroll") for (int k = 0; k < 3; ++k) {
                pr = fmax(pr, fmax(fabs(aT[k] - S.bT[k]), fabs(aI[k] - S.zI[k])));
                nz = fmax(nz, fmax(fmax(fabs(aT[k]), fabs(S.bT[k])), fmax(fabs(aI[k]), fabs(S.zI[k]))));
            }
            double ye0 = 0.0, ye1 = 0.0, w = 0.0;
            _Pragma("unroll") for (int k = 0; k < 3; ++k) w = fmax(w, row_violation(S, t, k, aI[k]));
            if (S.flags & F_LAST) {
                pr = fmax(pr, fmax(fabs(S.x[0] - e_z[0]), fabs(S.x[1] - e_z[1])));
                nz = fmax(nz, fmax(fmax(fabs(S.x[0]), fabs(e_z[0])), fmax(fabs(S.x[1]), fabs(e_z[1]))));
                ye0 = e_y[0]; ye1 = e_y[1];
                _Pragma("unroll") for (int k = 0; k < 2; ++k) w = fmax(w, end_violation_of(e_rb[k], e_lo[k], e_up[k], e_act[k], e_y[k], S.x[k]));
            }
            double aty[6];
            aty[0] = -S.yT[0] + gn[0] + S.yI[1] + S.yI[2] + ye0;
            aty[1] = -S.yT[1] + gn[1] + cf * S.yI[1] + cr * S.yI[2] + ye1;
            aty[2] = -S.yT[2] + gn[2] + S.yI[0];
            aty[3] = S.a[5] * S.yT[2];
            aty[4] = S.yI[1];
            aty[5] = S.yI[2];
            const bool colreal[6] = {real, real, real, (S.flags & F_PREV) != 0, real, real && (S.flags & F_PRECISE)};
            double du = 0.0, nd = 0.0, xsum = 0.0;
            _Pragma("unroll") for (int k = 0; k < 6; ++k) {
                const double px = cost_diag(prm, S.flags, k) * S.x[k];
                du = fmax(du, colreal[k] ? fabs(px + aty[k]) : 0.0);
                nd = fmax(nd, colreal[k] ? fmax(fabs(px), fabs(aty[k])) : 0.0);
                xsum += S.x[k];
            }
            v[0] = real ? pr : 0.0;
            v[1] = du;
            v[2] = real ? nz : 0.0;
            v[3] = nd;
            v[4] = (fabs(xsum) <= 1e300) ? 0.0 : 1.0;   // NaN / Inf guard: a sum of six iterates is finite iff (up to overflow near 1e300) each of them is
            v[5] = w;
        });
    }

    // ---------------------------------------------------------------------------------------------
    // primal infeasibility certificate (OSQP paper section 3.4; osqp/src/auxil.c is_primal_infeasible), evaluated at a
    // termination check on dy = y_k - y_{k-1}:   ||A' dy|| <= eps ||dy||   and   u'(dy)+ + l'(dy)- <= -eps ||dy||.
    // In the scaled variables OSQP uses all three quantities carry the same factor c, so the test reads the same in
    // unscaled ones; only "||dy|| is not zero" needs c.  The path QP has q = 0, so the dual certificate (q'dx < 0) can
    // never fire and is not evaluated.
    // ---------------------------------------------------------------------------------------------
    // The certificate itself is evaluated by primal_certificate() below from LDS only: the lanes first stage the pass
    // constants it needs.  Keeping the lane struct out of that code (on the device it is an out-of-line function) keeps its
    // register needs out of the ADMM loop - inline, it cost the loop 100 spilled VGPRs and 40 % of its speed.
    // The late form (prim_inf_after, lean kernel): dy = y_now - y_at_the_previous_late_check instead of y_k - y_{k-1}.  The test itself
    // is a Farkas certificate - whatever dy passes it proves the QP infeasible to the tolerance - so it needs no consecutive
    // iterates, and nothing has to be stored inside the ADMM loop.  The first call of a pass only takes the snapshot.
    PQP_HD double* snap_slot(int t) const {
        if constexpr (Ctx::kSaveLds) return sh + L.ysnap() + 6 * t;
        else return A.wscale + ((size_t)slot * T + t) * 18 + 12;
    }
    PQP_HD bool primal_infeasible_late() {
        const bool have = snap_valid_;
        snap_valid_ = true;
        // (everything happens in the context's function - out of line on the device - from values handed over by value: the lane struct
        // never has its address taken, and the register needs of the test stay out of this kernel's allocation)
        bool res = false;
        ctx.phase([&](int t, Lane& ln) {
            const Slot& S = ln.s;
            LateCertIn in;
            _Pragma("unroll") for (int k = 0; k < 3; ++k) { in.y[k] = S.yT[k]; in.y[3 + k] = S.yI[k]; in.b[k] = S.bT[k]; }
            _Pragma("unroll") for (int k = 0; k < 6; ++k) in.a[k] = S.a[k];
            in.flags = S.flags; in.lo0 = lo_of(S, t, 0); in.lo1 = lo_of(S, t, 1); in.up0 = up_of(S, t, 0); in.up1 = up_of(S, t, 1);
            res = ctx.late_certificate(sh, t, snap_slot(t), have, in, A.prm.front_length, A.prm.rear_length, kap, A.prm.eps_prim_inf, cscale);
        });
        return res;
    }
    PQP_HD bool primal_infeasible() {
        ctx.phase([&](int t, Lane& ln) {
            const Slot& S = ln.s;
            double* pa = sh + L.stageA() + 9 * t;
            double* pb = sh + L.stageB() + 3 * t;
            double* pc = sh + L.stageC() + 3 * t;
            _Pragma("unroll") for (int k = 0; k < 6; ++k) pa[k] = S.a[k];
            _Pragma("unroll") for (int k = 0; k < 3; ++k) pa[6 + k] = S.bT[k];
            pb[0] = (double)S.flags; pb[1] = lo_of(S, t, 0); pb[2] = lo_of(S, t, 1);
            pc[0] = up_of(S, t, 0); pc[1] = up_of(S, t, 1); pc[2] = 0.0;
        });
        return ctx.certificate(sh, T, A.prm.front_length, A.prm.rear_length, kap, A.prm.eps_prim_inf, cscale);
    }

    // ---------------------------------------------------------------------------------------------
    // start of a solve (osqp_warm_start semantics: x, y given, z = A x; cold: all zero)
    // ---------------------------------------------------------------------------------------------
    // OSQP starts an equality row from z_0 (0 cold, (A x)_i warm), not from its bound b; afterwards z == b
    // forever.  The solve keeps z_T implicit (== bT), so the first iteration is made exact by shifting the
    // dual: y' = y - rho (z_0 - b) before it, y += rho (2 - alpha) (z_0 - b) after it.  dz lives in sh[lin]
    // (free between assemble and unpack).
    PQP_HD void start_transition_rows(bool have_warm) {
        ctx.phase([&](int t, Lane& ln) {
            _Pragma("unroll") for (int k = 0; k < 3; ++k) sh[L.xbuf() + 3 * t + k] = ln.s.x[k];
        });
        ctx.phase([&](int t, Lane& ln) {
            Slot& S = ln.s;
            double Xp[3], aT[3], aI[3];
            { const double* xp_ = nb(t > 0, L.xbuf(), 3, t - 1); _Pragma("unroll") for (int k = 0; k < 3; ++k) Xp[k] = xp_[k]; }
            rows_of(S, Xp, S.x, aT, aI);
            const bool real = S.flags & F_REAL;
            _Pragma("unroll") for (int k = 0; k < 3; ++k) {
                const double dz = real ? (have_warm ? aT[k] : 0.0) - S.bT[k] : 0.0;
                sh[L.lin() + 3 * t + k] = dz;
                S.yT[k] -= S.rhoT[k] * dz;
                S.zI[k] = (real && have_warm) ? aI[k] : 0.0;
            }
            if (S.flags & F_LAST) {
                end_rows()->z[0] = have_warm ? S.x[0] : 0.0;
                end_rows()->z[1] = have_warm ? S.x[1] : 0.0;
            }
        });
    }
    PQP_HD void finish_first_iteration() {
        const double f = 2.0 - A.prm.alpha;
        ctx.phase([&](int t, Lane& ln) {
            Slot& S = ln.s;
            _Pragma("unroll") for (int k = 0; k < 3; ++k) S.yT[k] += S.rhoT[k] * f * sh[L.lin() + 3 * t + k];
        });
    }

    // (A warm state is a starting point, not data: with PQP_OPT_CARRY_CYCLES it is whatever the slot's previous QP left - possibly a QP with another
    //  waypoint count, or one that ended non-finite.  Entries that are not finite start from 0, so that one bad cycle does not poison the slot for good.)
    PQP_HD void load_warm() {
        auto fin = [](double v) { return fabs(v) <= 1e300 ? v : 0.0; };
        ctx.phase([&](int t, Lane& ln) {
            Slot& S = ln.s;
            const bool real = S.flags & F_REAL;
            const size_t o = ((size_t)qp * stride + (real ? t : n - 1)) * 6;
            _Pragma("unroll") for (int k = 0; k < 6; ++k) S.x[k] = real ? fin(A.wx[o + k]) : 0.0;
            _Pragma("unroll") for (int k = 0; k < 3; ++k) { S.yT[k] = real ? fin(A.wy[o + k]) : 0.0; S.yI[k] = real ? fin(A.wy[o + 3 + k]) : 0.0; }
            if (S.flags & F_LAST) { end_rows()->y[0] = fin(A.wye[2 * (size_t)qp]); end_rows()->y[1] = fin(A.wye[2 * (size_t)qp + 1]); }
        });
    }

    PQP_HD void store_warm() {
        ctx.phase([&](int t, Lane& ln) {
            const Slot& S = ln.s;
            if (S.flags & F_REAL) {
                double* wx = A.wx + ((size_t)qp * stride + t) * 6;
                double* wy = A.wy + ((size_t)qp * stride + t) * 6;
                _Pragma("unroll") for (int k = 0; k < 6; ++k) wx[k] = S.x[k];
                _Pragma("unroll") for (int k = 0; k < 3; ++k) { wy[k] = S.yT[k]; wy[3 + k] = S.yI[k]; }
                if (S.flags & F_LAST) { A.wye[2 * (size_t)qp] = end_rows()->y[0]; A.wye[2 * (size_t)qp + 1] = end_rows()->y[1]; }
            }
        });
    }

    // ---------------------------------------------------------------------------------------------
    // unpack (base_solver.cpp:263-288) and publish the next linearisation point
    // ---------------------------------------------------------------------------------------------
    // final = false (a re-linearised pass follows): only the linearisation point - the output record of this pass would be overwritten
    // by the next one, and its sincos and seven stores per waypoint are 2 % of a QP's time
    PQP_HD void unpack(bool final, bool zero = false) {
        if (!final) {
            ctx.phase([&](int t, Lane& ln) {
                const Slot& S = ln.s;
                if (S.flags & F_REAL) { sh[L.lin() + 3 * t + 0] = S.x[0]; sh[L.lin() + 3 * t + 1] = S.x[1]; sh[L.lin() + 3 * t + 2] = S.x[2]; }
            });
            return;
        }
        ctx.phase([&](int t, Lane& ln) {
            sh[L.xbuf() + t] = ln.s.x[3];    // v of waypoint t = u of waypoint t-1
        });
        ctx.phase([&](int t, Lane& ln) {
            const Slot& S = ln.s;
            if (S.flags & F_REAL) {
                const int i = t;
                const double* r = A.ref + ((size_t)qp * stride + i) * PQP_REF_STRIDE;
                double* o = A.out + ((size_t)qp * stride + i) * PQP_OUT_STRIDE;
                const double angle = r[2];
                const double l = S.x[0], dpsi = S.x[1];
                const double new_angle = constrain_angle(angle + kPi2);
                double sn, cn;
                sincos_shared(new_angle, &sn, &cn);
                // (zero: the record of a QP that ended PQP_STATUS_NUMERICAL - its iterates are not numbers)
                o[0] = zero ? 0.0 : r[3] + l * cn;
                o[1] = zero ? 0.0 : r[4] + l * sn;
                o[2] = zero ? 0.0 : constrain_angle(angle + dpsi);
                o[3] = zero ? 0.0 : l;
                o[4] = zero ? 0.0 : dpsi;
                o[5] = zero ? 0.0 : S.x[2];
                o[6] = (!zero && (S.flags & F_NEXT)) ? sh[L.xbuf() + t + 1] : 0.0;
                // input_path_ = first solution (base_solver.cpp:100): l, d_heading, k
                sh[L.lin() + 3 * i + 0] = l; sh[L.lin() + 3 * i + 1] = dpsi; sh[L.lin() + 3 * i + 2] = S.x[2];
            }
        });
    }

    PQP_HD Uni get_uni() const { return Uni{rho, cscale, kap, alpha_, kkt_solves_, factors_, polishing_ ? 1 : 0, (cert_ ? 1 : 0) | (snap_valid_ ? 2 : 0)}; }
    PQP_HD void set_uni(const Uni& u) { rho = u.rho; cscale = u.cscale; kap = u.kap; alpha_ = u.alpha; kkt_solves_ = u.kkt_solves; factors_ = u.factors; polishing_ = u.polishing != 0; cert_ = (u.cert & 1) != 0; snap_valid_ = (u.cert & 2) != 0; }

    // The cold side of the solver.  On the device this runs inside a __noinline__ function on a copy of the lane
    // state that lives in memory (DevCtx::cold), so its register needs never leak into the ADMM loop.
    PQP_HD void do_cold(int op, int i0, int i1, double d0) {
        const pqp_params& prm = A.prm;
        if (op == COLD_BEGIN_PASS) {
            const bool have_warm = (i1 & 1) != 0;
            if (i0 == 0) {
                PQP_SUB(0, load());
                rho = prm.rho;
                if (have_warm) {
                    load_warm();
                    rho = ctx.uni(A.wrho[qp]);       // (a vector load: told to be the same in every lane, see DevCtx::uni)
                    if (!(rho >= kRhoMin && rho <= kRhoMax)) rho = prm.rho;
                } else {
                    ctx.phase([&](int, Lane& ln) {
                        if (ln.s.flags & F_LAST) { end_rows()->y[0] = 0.0; end_rows()->y[1] = 0.0; }
                    });
                }
            }
            PQP_SUB(1, assemble());
            cert_ = ctx.uni(sh[L.poison()]) != 0.0;       // (F_BADIN: the scenario is not a number - run() ends the QP before its first iteration)
            PQP_SUB(2, ruiz(/*reuse=*/(i1 & 2) != 0 && prm.polish_warm_set >= 2));
            PQP_SUB(3, start_transition_rows(have_warm));
            if (i1 & 2) {      // warm re-solve: go straight to a polish on the active set the previous pass ended with
                PQP_SUB(4, polish_begin(true));
                polishing_ = true; alpha_ = 1.0;
                PQP_SUB(4, polish_apply_set());
            }
            PQP_SUB(5, factor());
        } else if (op == COLD_REFACTOR) {
            if (i0 == RF_RESCALE) {
                rescale_rho(d0);
            } else if (i0 == RF_POLISH_BEGIN) {
                PQP_SUB(4, polish_begin(false));
                polishing_ = true; alpha_ = 1.0;
                PQP_SUB(4, polish_apply_set());
            } else if (i0 == RF_POLISH_UPDATE) {
                PQP_SUB(6, polish_update_set(d0));
                PQP_SUB(4, polish_apply_set());
            } else {   // RF_POLISH_REJECT
                PQP_SUB(7, polish_end(false, d0 != 0.0));
                polishing_ = false; alpha_ = prm.alpha;
            }
            PQP_SUB(5, factor());
        } else if (op == COLD_CERT) {
            if (CERT) cert_ = primal_infeasible();
            else cert_ = primal_infeasible_late();
        } else if (op == COLD_END_PASS) {
            if (i0 == 1) {
                polish_end(true);
                polishing_ = false; alpha_ = prm.alpha;
            }
            unpack(i1 != 0, /*zero=*/i0 == 2);
        } else {
            if (A.store_warm) store_warm();
        }
    }

    // ---------------------------------------------------------------------------------------------
    // the whole path: (warm) solve + `passes` re-linearised warm re-solves.
    // Hot side: one loop around iterate() / residuals() / the KKT test.  Everything else is a cold operation, issued
    // from ONE call site (the lane state crosses the hot/cold boundary through memory there, once per operation).
    //   mode ADMM  : OSQP iterations; every check_termination iterations residuals -> stop / adapt rho / start a polish
    //   mode POLISH: polish_refine_iter solves, then the KKT test -> accept / next active-set round / give up
    // ---------------------------------------------------------------------------------------------
    PQP_HD void run() {
        const pqp_params& prm = A.prm;
        int total_iters = 0, last_iters = 0, status = PQP_STATUS_UNSOLVED, polished = 0;
        // active-set rounds per polish attempt; <= 0: sized to the path (long paths need more rounds, short ones pay for them)
        const int auto_rounds = n / 5 - 8;
        const int max_rounds = prm.polish_max_rounds > 0 ? prm.polish_max_rounds : (auto_rounds > 24 ? auto_rounds : 24);
        // (the cautious switch by round count applies to the FIRST attempt of a pass only: later attempts get half the rounds, and switched at 8 they spend
        //  them on single moves - one QP in ~16 000 of 300 waypoints then never completed a polish, 868 reduced solves; first attempt only: 83.  A switch
        //  at round n / 16 on long paths cures that QP and a 368-solve one at 200 waypoints too, but BASELINE configs[4]'s batch holds a QP that then
        //  wanders: 551 k instead of 690 k scenarios/s - round 3 records)
        const int cautious_from = kCautiousFromRound;
        double res[6] = {0, 0, 0, 0, 0, 0};
        int pass = 0;
        // per-pass state of the hot loop
        bool polish_mode = false, conservative = false, end_after_reject = false;
        double eps_scale = 1.0, best = 1e300, best_any = 1e300, admm_merit = 1e300;
        int it = 0, refine_left = 0, round = 0, stall = 0, polish_gap = 0, next_polish = 0;
        int extra_refine = 0;            // extra pairs of refinement solves spent on the current polish round
        int final_refine = 0;            // pqp_params::polish_final_refine: refinement solves an accepted point of the current round has had (contexts with kFinalRefine)
        bool lazy_look = false;          // polish_lazy: the next look at the polished point follows a single solve
        bool direct_polish = false, last_accepted = false;
        // the pending cold operation
        // (carry_tails: the QP's cost bin in the previous launch - its key is rewritten at the end of this solve - against the launch's threshold bin)
        // A carried QP keeps its cold cost key minus one bin per cycle instead of its (cheap) carried cost: with the carried cost it would drop out of
        // the expensive eighth at once, start cold - and late, by a key that says "cheap" - in the next cycle, and every hard QP would alternate
        // between the two (measured: one launch at a time 1.75 M against the cold 1.98 M paths/s, profiles/r05b_*)
        bool qp_warm = A.warm != 0;
        int keep_bin = -1;
        if (A.carry_tails && A.warm) {
            const int prev_bin = ctx.uni_int((A.cost_key[qp] >> 24) & 0xff);       // (the key is bin << 24 in an int32: bins from 128 on are negative numbers)
            qp_warm = prev_bin >= ctx.uni_int(A.cost_hist[kCostBins + 1]);
            if (qp_warm) keep_bin = prev_bin - 1;
        }
        int op = COLD_BEGIN_PASS, i0 = 0, i1 = qp_warm ? 1 : 0;
        double d0 = 0.0;
#ifdef PQP_TIMING
        // debug build only (tools/kernel_timeline.py): wall-clock ticks per category, written over the info record
        long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        const long long t_begin = ctx.clock();
#define PQP_TIC(m) const long long tic_ = (PQP_TIMING_MASK & (m)) ? ctx.clock() : 0
#define PQP_TOC(k) do { if ((PQP_TIMING_MASK >> (k)) & 1) tacc[k] += ctx.clock() - tic_; } while (0)
#else
#define PQP_TIC(m)
#define PQP_TOC(k)
#endif
        for (;;) {
            // (the last pass of the call - the reference stops after a solve that fails - writes the output record)
            if (op == COLD_END_PASS) i1 = (status != PQP_STATUS_SOLVED || pass == A.passes) ? 1 : 0;
            {
                PQP_TIC(0x2d);
                ctx.cold(*this, op, i0, i1, d0);
                PQP_TOC(op == COLD_BEGIN_PASS ? 0 : op == COLD_REFACTOR ? 2 : op == COLD_CERT ? 5 : 3);
            }
            if (op == COLD_FINISH) break;
            if (op == COLD_CERT && cert_) { status = PQP_STATUS_PRIMAL_INFEASIBLE; op = COLD_END_PASS; i0 = 0; continue; }
            if (op == COLD_END_PASS) {
                last_iters = it;
                total_iters += it;
                if (status != PQP_STATUS_SOLVED || pass == A.passes) { op = COLD_FINISH; continue; }   // reference: solve() false -> stop
                pass += 1;
                direct_polish = prm.polish && prm.polish_warm_set && last_accepted;
                op = COLD_BEGIN_PASS; i0 = pass; i1 = 1 | (direct_polish ? 2 : 0);
                continue;
            }
            if (op == COLD_BEGIN_PASS) {
                if (cert_) { cert_ = false; status = PQP_STATUS_NUMERICAL; op = COLD_END_PASS; i0 = 2; continue; }      // (F_BADIN; i0 = 2: a zero output record)
                status = PQP_STATUS_MAX_ITER;
                snap_valid_ = false;
                polish_mode = false; conservative = false; end_after_reject = false;
                eps_scale = 1.0; best = 1e300;
                it = 0; refine_left = 0; round = 0; stall = 0;
                polish_gap = prm.polish_every; next_polish = prm.polish_every;
                last_accepted = false;
                if (direct_polish) { polish_mode = true; refine_left = prm.polish_lazy ? 1 : prm.polish_refine_iter; lazy_look = prm.polish_lazy != 0; best_any = 1e300; admm_merit = 1e300; }
            } else if (end_after_reject) {       // a rejected polish at max_iter
                op = COLD_END_PASS; i0 = 0;
                continue;
            }
            // ---- hot loop: runs until the next cold operation is due
#ifdef PQP_TIMING
            const long long tic_hot_ = (PQP_TIMING_MASK & 0x02) ? ctx.clock() : 0;      // category 1: the whole hot loop (iterate + residuals + policy)
#endif
            for (;;) {
                { PQP_TIC(0x10); iterate(); PQP_TOC(4); }
                bool want_res, check = false, adapt = false;
                if (!polish_mode) {
                    it += 1;
                    if (it == 1) finish_first_iteration();
                    // (OSQP also tests the residuals when it stops at max_iter: a pass that converges between two checks is SOLVED)
                    check = prm.check_termination > 0 && ((it % prm.check_termination) == 0 || it >= prm.max_iter);
                    adapt = prm.adaptive_rho && prm.adaptive_rho_interval > 0 && (it % prm.adaptive_rho_interval) == 0;
                    want_res = check || adapt;
                } else {
                    refine_left -= 1;
                    want_res = refine_left <= 0;
                }
                if (!want_res) {
                    if (!polish_mode && it >= prm.max_iter) { sync_after_iterate(); op = COLD_END_PASS; i0 = 0; break; }
                    continue;
                }
                // (no barrier between the solve and the residuals: residuals() publishes into buffers the tail of iterate() does not touch - ShLayout::xres)
                { PQP_TIC(0x20); residuals(res); PQP_TOC(5); }
                if (!polish_mode) {
                    bool start_polish = false;
                    if (res[4] != 0.0) { status = PQP_STATUS_NUMERICAL; op = COLD_END_PASS; i0 = 2; break; }      // (i0 = 2: a zero output record)
                    if (check) {
                        const double eps_p = eps_scale * (prm.eps_abs + prm.eps_rel * res[2]);
                        const double eps_d = eps_scale * (prm.eps_abs + prm.eps_rel * res[3]);
                        // the polish of this QP cannot be verified (typically: infeasible by less than eps, so that the tightened
                        // residual tests below are never met): what OSQP does when its polish fails - return the ADMM point
                        if (prm.polish_patience > 0 && eps_scale < 1.0 && polish_gap >= (prm.polish_every << (prm.polish_patience < 16 ? prm.polish_patience : 16)) &&
                            res[0] <= prm.eps_abs + prm.eps_rel * res[2] && res[1] <= prm.eps_abs + prm.eps_rel * res[3]) {
                            status = PQP_STATUS_SOLVED; op = COLD_END_PASS; i0 = 0; break;
                        }
                        if (res[0] <= eps_p && res[1] <= eps_d) {
                            if (!prm.polish || eps_scale * fmax(prm.eps_abs, prm.eps_rel) < 1e-10) {
                                status = PQP_STATUS_SOLVED; op = COLD_END_PASS; i0 = 0; break;
                            }
                            start_polish = true;
                            eps_scale *= 0.1;      // if this polish is rejected ADMM resumes one decade tighter
                        } else if (prm.polish && prm.polish_every > 0 && it >= next_polish) {
                            // a slow ADMM tail: the active set is often already right long before the residuals say so
                            start_polish = true;
                            polish_gap *= 2;       // back off if it is rejected
                            next_polish = it + polish_gap;
                        }
                    }
                    if (start_polish) {
#ifdef PQP_EMU_DEBUG
                        printf("  START qp %d it %d ratio_p %.3e ratio_d %.3e\n", qp, it, res[0] / (prm.eps_abs + prm.eps_rel * res[2]), res[1] / (prm.eps_abs + prm.eps_rel * res[3]));
#endif
                        polish_mode = true;
                        refine_left = prm.polish_lazy ? 1 : prm.polish_refine_iter; lazy_look = prm.polish_lazy != 0;
                        round = 0; stall = 0; best = 1e300; conservative = false;
                        best_any = 1e300; admm_merit = fmax(res[0], res[1]);
                        op = COLD_REFACTOR; i0 = RF_POLISH_BEGIN; break;
                    }
                    bool refactor = false;
                    if (adapt) {
                        const double pn = res[0] / (res[2] + 1e-10);
                        const double dn = res[1] / (res[3] + 1e-10);
                        double rn = rho * sqrt(pn / (dn + 1e-10));
                        rn = fmin(fmax(rn, kRhoMin), kRhoMax);
                        if (rn > rho * prm.adaptive_rho_tolerance || rn < rho / prm.adaptive_rho_tolerance) {
                            d0 = rn / rho;
                            rho = rn;
                            refactor = true;
                        }
                    }
                    if (it >= prm.max_iter) { op = COLD_END_PASS; i0 = 0; break; }
                    if (refactor) { op = COLD_REFACTOR; i0 = RF_RESCALE; break; }
                    // not converged, nothing else due at this check: is the problem infeasible?  (a cold operation: evaluated
                    // inside this loop the test costs the ADMM iteration 14 % through register pressure alone)
                    if (CERT && check && prm.eps_prim_inf > 0.0 && it > 1) { op = COLD_CERT; break; }
                    // the lean kernel: the same certificate between two late checks (see primal_infeasible_late)
                    if (!CERT && check && prm.prim_inf_after > 0 && prm.eps_prim_inf > 0.0 && it >= prm.prim_inf_after) { op = COLD_CERT; break; }
                } else {
                    // KKT acceptance test of the polished point (OSQP paper 4.2 + verification)
                    const double tol = prm.polish_tol;
                    const double viol = res[5];
                    const bool solve_ok = res[4] == 0.0 && res[0] <= tol * (1.0 + res[2]) && res[1] <= tol * (1.0 + res[3]);
                    const bool ok = solve_ok && viol <= tol;
                    // polish_lazy = k: the first look of a round comes after one solve.  Its point is not accurate enough for the acceptance
                    // test, but rows that fail by far more than the solve's own residual fail at the refined point too: during the first
                    // k full rounds of an attempt (where many rows move at once) they move now and the round is over; otherwise the
                    // remaining refinement solves follow.  (Unbounded k: the late rounds, which move single rows, cycle on unrefined points.)
                    if (lazy_look) {
                        lazy_look = false;
                        const double noise = 10.0 * fmax(res[0], res[1]);
                        if (res[4] == 0.0 && !ok && viol > fmax(10.0 * tol, noise) && round + 1 < max_rounds && round < prm.polish_lazy && !conservative) {
                            round += 1;
                            refine_left = 1; lazy_look = true;
                            op = COLD_REFACTOR; i0 = RF_POLISH_UPDATE; d0 = fmax(fmax(tol, noise), kFullMoveShare * viol); break;
                        }
                        // (nothing moves: the point is only looked at again, and only ever accepted, fully refined)
                        refine_left = prm.polish_refine_iter > 1 ? prm.polish_refine_iter - 1 : 1; continue;
                    }
                    // the refinement solves start from the ADMM iterate; from a distant one the budgeted number of them may leave the
                    // polished point short of the accuracy the KKT test needs: up to 3 more pairs of solves instead of throwing the
                    // attempt away
                    // (long paths, pqp_params::polish_final_refine = k: a point that passes is refined k times more - and tested again - before it is returned)
                    // (only in the contexts of long paths - Ctx::kFinalRefine: more than 128 lanes per QP -, so that the kernels of shorter paths compile
                    //  to what they were: their register allocation is one source line away from 20 more spilled registers and -1.5 %)
                    if constexpr (Ctx::kFinalRefine) {
                        // (the final refinements have a count of their own: a point that needed extra pairs to pass still gets all of them)
                        const bool hold = ok && final_refine < prm.polish_final_refine;
                        if (hold && res[4] == 0.0) { final_refine += 1; refine_left = 1; continue; }
                        if (!solve_ok && extra_refine < 3 && res[4] == 0.0) { extra_refine += 1; refine_left = 2; continue; }
                    } else {
                        if (!solve_ok && res[4] == 0.0 && extra_refine < 3) { extra_refine += 1; refine_left = 2; continue; }
                    }
                    extra_refine = 0; final_refine = 0;
#ifdef PQP_EMU_DEBUG
                    printf("  polish qp %d it %d round %d: pri %.3e dua %.3e viol %.3e %s -> %s\n", qp, it, round, res[0], res[1], viol, conservative ? "(cons)" : "", ok ? "ACCEPT" : "reject");
#endif
                    if (ok) { status = PQP_STATUS_SOLVED; polished += 1; polish_mode = false; last_accepted = true; op = COLD_END_PASS; i0 = 1; break; }
                    // the active-set rounds diverge: this attempt will not get there, stop paying for it
                    const bool diverged = prm.polish_diverge > 0 && viol > (double)prm.polish_diverge * best_any;
                    bool give_up = !solve_ok || diverged;
                    if (solve_ok && viol < best_any) { best_any = viol; if (prm.polish_reseed) polish_save_best(); }
                    if (!give_up) {
                        // primal-dual active-set step.  A full update can cycle: when the violation stops improving only
                        // the worst offenders (>= 90 % of the maximum) move.
                        if (viol < 0.7 * best) { best = viol; stall = 0; } else { stall += 1; }
                        // (progress of the cautious rounds is measured from where they start: the full rounds before them may
                        // have passed through a smaller violation on their way out)
                        // (round 3: ... or when the attempt reaches its 8th round - nearly every QP is done by its 5th, one still moving rows
                        //  wholesale in its 8th is wandering and would do so until three stalls in a row, e.g. through eight turns of a period-7
                        //  cycle: over 16 scenario seeds the slowest QP of a batch of 1024 went from 46-121 reduced solves to 39-66, the headline of the
                        //  two straggler seeds from 1.91 / 2.72 M to 2.79 / 3.12 M paths/s, one launch at a time from 1.03-1.96 M to 1.68-2.19 M, the
                        //  mean cost stayed - profiles/r03x_seed_sweep.txt; 7 and 9-12 instead of 8: seed 6's straggler survives from 9 on, 7 costs 2 %)
                        if ((stall >= 3 || (round + 1 >= cautious_from && !(prm.polish_every > 0 && it > prm.polish_every))) && !conservative) { conservative = true; best = viol; stall = 0; }
                        round += 1;
                        // (the attempts after a pass's first periodic one start from a better ADMM iterate and get half the rounds:
                        // when those are not enough the rounds are usually cycling, and every further one is wasted.  A quarter
                        // is too few: long paths with contact segments then fail attempts they would have completed)
                        give_up = (conservative && stall >= 8) || round >= ((prm.polish_every > 0 && it > prm.polish_every) ? max_rounds / 2 : max_rounds);
                    }
                    if (give_up) {
                        polish_mode = false;
                        end_after_reject = it >= prm.max_iter;
                        // continue ADMM from the best polished point when it is closer to the optimum than the parked iterate
                        d0 = (prm.polish_reseed && best_any < prm.polish_reseed_factor * admm_merit) ? 1.0 : 0.0;
                        op = COLD_REFACTOR; i0 = RF_POLISH_REJECT; break;
                    }
                    refine_left = prm.polish_lazy ? 1 : prm.polish_refine_iter; lazy_look = prm.polish_lazy != 0;
                    op = COLD_REFACTOR; i0 = RF_POLISH_UPDATE; d0 = conservative ? fmax(tol, 0.9 * viol) : fmax(tol, kFullMoveShare * viol); break;
                }
            }
#ifdef PQP_TIMING
            if (PQP_TIMING_MASK & 0x02) tacc[1] += ctx.clock() - tic_hot_;
#endif
        }
        const double rho_final = rho;
        const int kkt_total = kkt_solves_, fac_total = factors_;
        PQP_TIC(0x40);
        ctx.phase([&](int t, Lane&) {
            if (t == 0) {
                if (A.cost_key) record_cost(A, qp, 4 * kkt_total + 13 * fac_total, keep_bin);
                A.wrho[qp] = rho_final;
                if (A.status) A.status[qp] = status;
                if (A.iters) A.iters[qp] = total_iters;
                if (A.info) {
                    double* f = A.info + PQP_INFO_STRIDE * (size_t)qp;
                    f[0] = res[0]; f[1] = res[1]; f[2] = rho_final; f[3] = (double)last_iters;
                    f[4] = (double)polished; f[5] = (double)kkt_total; f[6] = (double)fac_total; f[7] = 0.0;
#ifdef PQP_TIMING
                    tacc[7] = ctx.clock() - t_begin;
                    for (int k = 0; k < 8; ++k) f[k] = (double)tacc[k];
                    for (int k = 0; k < 8; ++k) A.out[(size_t)qp * stride * PQP_OUT_STRIDE + k] = (double)tsub_[k];     // debug build only
#ifdef PQP_TIMING_ITER
                    for (int k = 0; k < 16; ++k) A.out[(size_t)qp * stride * PQP_OUT_STRIDE + 12 + k] = (double)tit_[k];
#endif
#endif
                }
            }
        });
#ifdef PQP_TIMING
        // the finish phase itself (category 6) and what the kernel's ticket loop spent before this QP (out[..][8], set by the kernel)
        if ((PQP_TIMING_MASK >> 6) & 1) { const long long d_ = ctx.clock() - tic_; ctx.phase([&](int t, Lane&) { if (t == 0 && A.info) A.info[PQP_INFO_STRIDE * (size_t)qp + 6] = (double)d_; }); }
#endif
    }
};

}  // namespace pqp
