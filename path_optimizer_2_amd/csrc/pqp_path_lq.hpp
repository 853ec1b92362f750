// pqp_path_lq.hpp — the path QP as a linear-quadratic control problem, ONE LANE PER QP, every per-waypoint quantity streamed through a
// batch-interleaved workspace in HBM: the solver behind pqp_path_solve* for large batches (PQP_OPT_STREAM_BATCH) and for paths of more than
// 512 waypoints.
//
// reference                                                   here
//   BaseSolver::setCost          src/solver/base_solver.cpp:119-148   the stage costs of `backward()` (weights 0 / 20 / 100, slack weight 10)
//   BaseSolver::setConstraints   :150-261                             `prep()`: the transition rows become the dynamics x_{i+1} = M_i x_i + b u_i + c_i
//                                                                     (same expressions as :165-186), the kappa / collision / end rows become boxes
//   getSoftBounds                :290-295                             soft_bounds() (pqp_path_lane.hpp)
//   OsqpEigen solve              :88,110                              `solve_pass()`: interior-point rounds + active-set rounds (NOT OSQP's ADMM: see "Algorithm")
//   updateProblemFormulationAndSolve :97-117                          the pass loop of `run()`: re-linearise around the previous optimum
//   getOptimizedPath             :263-288                             `unpack()`
//
// Algorithm.  The QP's variables per waypoint are the state x_i = (l, psi, kappa)_i, the control u_i = kappa' and two slacks.  The
// transition rows (equalities) make the states an affine function of x_0 (given) and the controls; a collision row lo <= l + L psi + s <= up
// with its slack's cost (w_s / 2) s^2 is the convex penalty (w_s / 2) dist(l + L psi, [lo, up])^2; what remains are hard boxes on kappa_i
// and on the end state (l, psi)_{n-1}.  For quadratic row terms 1/2 w (a^T x - t)^2 the problem is solved EXACTLY by one backward Riccati
// sweep (3 x 3 value matrices, scalar control) and one forward roll-out: O(n) flops, a dependency chain in the waypoint index - which is why
// a QP is one LANE here (64 QPs per wavefront run the same instruction stream on different data; all loads / stores are contiguous
// across the wavefront).  Two kinds of rounds use that solve:
//   * interior-point rounds (primal-dual path following on the boxes, one Riccati solve per iteration: weights z/t per row side, fraction-
//     to-the-boundary step) bring a cold QP to complementarity 1e-6 in 8-16 iterations without any combinatorial decision - the active-set
//     rounds of the lane-per-waypoint kernel (pqp_path_lane.hpp) cycle on 1 QP in 25 when started cold (tools/lq_prototype.py);
//   * active-set rounds (rows of the predicted set as exact quadratic terms: w_s at the bound for collision rows, 1/delta at the bound
//     shifted by delta * multiplier for hard rows) verify the prediction: a round that asks for no change of the set IS the KKT test
//     (stationarity holds by construction, primal feasibility and dual signs are what the round checks), so the returned point is
//     the exact optimum of the QP - the same guarantee as the KKT-verified polish of the lane-per-waypoint kernel.
// The re-linearised pass starts its interior-point rounds from the previous pass's optimum (complementarity reset to 1e-3).
//
// The same source compiles for the device (path_stream_kernel, pqp_path_stream.hip) and, for tests and the bench's CPU line only, for the
// host (tests/emu/lq_emu.cpp).
#pragma once
#include "pqp_path_lane.hpp"
#include "pqp_path_lq_abi.hpp"
#if defined(PQP_LQ_DEBUG) && !defined(__HIP_DEVICE_COMPILE__)
#include <cstdio>
#endif

namespace pqp {
namespace lq {

constexpr double kBig = 1e29;           // a bound beyond this is no bound (OSQP_INFTY = 1e30)
constexpr double kDelta = 1e-9;         // hard active rows: penalty 1 / delta around the bound shifted by delta * multiplier
constexpr double kInvDelta = 1e9;
constexpr double kSetTol = 1e-7;        // a row changes sides when it fails its test by more than this (pqp_params.polish_tol's role)
constexpr double kPinTol = 1e-9;        // accepted points hold their hard active rows to this
constexpr double kMuStop = 1e-6;        // complementarity at which the interior-point rounds hand over to the active-set rounds
constexpr double kMuWarm = 1e-3;        // complementarity a re-linearised pass starts from
constexpr double kEqWidth = 1e-6;       // a collision box narrower than this is an equality row (weight w_s at its upper bound)
constexpr int kIpmMaxIter = 100;
constexpr int kPolishMaxRounds = 12;
// A re-linearised pass of a launch whose wavefronts are sorted by their phase counts (Args::order) first tries the previous pass's active set on the new
// transition rows: this many active-set rounds before the interior-point rounds get their turn (round 6).  Of the bench's QPs 60 % confirm their set in
// the first round, 93 % within two, 98 % within three (tools/lq_direct_probe.py) - 2 x 328 bytes per waypoint instead of 3.9 interior-point iterations of
// 448 and the hand-over.  Only in sorted launches: a wavefront runs its phases as often as its slowest lane, and of 64 unsorted lanes one nearly always
// falls back (0.977^64 = 0.22) - the rounds would be paid on top of the iterations.
constexpr int kDirectRounds = 3;
// where the direct rounds keep the previous pass's optimum (point, set, multiplier: five doubles per waypoint) for the interior-point rounds to start
// from should the set not be confirmed: the fp32 fields of the waypoint, which only the interior-point rounds use - and initialise
constexpr int kStash = kFieldsD;
static_assert(kFieldsF / 2 >= 5 && D_X1 == D_X0 + 1 && D_X2 == D_X0 + 2, "five doubles per waypoint fit the fp32 fields; the point's fields are consecutive");

// reciprocal: the hardware seed (4.6e-8, tools/probes/rcp_probe.hip) + ONE Newton step = 2.2e-15 relative - a third fewer instructions than
// pqp::rcp's two steps in a kernel whose row arithmetic is mostly reciprocals (two steps: -2 %, profiles/r03a_stream_first.txt)
#if defined(__HIP_DEVICE_COMPILE__)
PQP_HD double rcpq(double x) {
    const double r = __builtin_amdgcn_rcp(x);
    return fma(fma(-x, r, 1.0), r, r);
}
#else
PQP_HD double rcpq(double x) { return rcp(x); }
#endif

// The sweeps of the solver are inlined into one kernel body (as functions of their own - the solver object then lives in the lane's private
// memory between them - 280 registers instead of 512 and 1.25x slower: profiles/r03a_stream_first.txt)
#define PQP_SWEEP PQP_HD

// a two-sided row of the interior-point rounds: value, slacks to the two bounds, multipliers of the two bounds
struct Row { double g, tl, tu, zl, zu; };
struct RowStep { double dg, dtl, dtu, dzl, dzu; };

struct Acc {          // what a roll-out accumulates for the step length and the next complementarity
    double rho;       // max over row sides of -delta / value (the step to the boundary is 1 / rho)
    double s0, s1, s2, cnt;   // sum t z, sum (t dz + z dt), sum dt dz, number of row sides
    double res;       // largest primal residual of the rows
};

// Newton target of a row: weight d and target of the quadratic term that replaces the barrier (sm = sigma * mu)
PQP_HD void row_weight(const Row& r, double lo, double up, double sm, double& d, double& tgt) {
    const double itl = rcpq(r.tl), itu = rcpq(r.tu);
    const double rl = r.g - lo - r.tl, ru = up - r.g - r.tu;
    d = r.zu * itu + r.zl * itl;
    const double e = sm * (itu - itl) - r.zu * itu * ru + r.zl * itl * rl;
    tgt = r.g - e * rcpq(d);
}
// the step of a row's state towards the Newton point whose row value is g + dg
PQP_HD RowStep row_step(const Row& r, double lo, double up, double sm, double dg) {
    const double itl = rcpq(r.tl), itu = rcpq(r.tu);
    const double rl = r.g - lo - r.tl, ru = up - r.g - r.tu;
    RowStep s;
    s.dg = dg;
    s.dtl = s.dg + rl;
    s.dtu = -s.dg + ru;
    s.dzl = (sm - r.zl * s.dtl) * itl - r.zl;
    s.dzu = (sm - r.zu * s.dtu) * itu - r.zu;
    return s;
}
PQP_HD void row_accumulate(const Row& r, const RowStep& s, double lo, double up, Acc& a, bool hard = true) {
    const double itl = rcpq(r.tl), itu = rcpq(r.tu), izl = rcpq(r.zl), izu = rcpq(r.zu);
#if defined(PQP_LQ_DEBUG) && !defined(__HIP_DEVICE_COMPILE__)
    {
        const double m = fmax(fmax(-s.dtl * itl, -s.dtu * itu), fmax(-s.dzl * izl, -s.dzu * izu));
        if (m > a.rho && m > 100.0) std::fprintf(stderr, "    blocking row (hard %d): g %.9g lo %.9g up %.9g | tl %.3e tu %.3e zl %.3e zu %.3e | dg %.3e dtl %.3e dtu %.3e dzl %.3e dzu %.3e\n", (int)hard, r.g, lo, up, r.tl, r.tu, r.zl, r.zu, s.dg, s.dtl, s.dtu, s.dzl, s.dzu);
    }
#endif
    a.rho = fmax(fmax(a.rho, -s.dtl * itl), fmax(-s.dtu * itu, fmax(-s.dzl * izl, -s.dzu * izu)));
    a.s0 += r.tl * r.zl + r.tu * r.zu;
    a.s1 += r.tl * s.dzl + r.zl * s.dtl + r.tu * s.dzu + r.zu * s.dtu;
    a.s2 += s.dtl * s.dzl + s.dtu * s.dzu;
    a.cnt += 2.0;
    if (hard) a.res = fmax(a.res, fmax(fabs(r.g - lo - r.tl), fabs(up - r.g - r.tu)));      // (a collision row's residual is the rounding of its fp32 slacks)
}
// what the roll-out hands to the next backward sweep goes through an fp32 field: both sweeps use the value as stored
PQP_HD double as_stored(double v) { return (double)(float)v; }
// Centrality safeguard after a step (the role of IPOPT's kappa_Sigma): a multiplier is kept within [mu / (kappa t), kappa mu / t].  Without it a
// row that blocks several consecutive 0.995-steps loses its slack 200x per step while its multiplier stands still, the pair leaves every
// neighbourhood of the central path and the steps shrink to nothing (seen on QPs whose end-heading box lies 1 rad off the line's heading).
constexpr double kCentral = 1e3;
PQP_HD void row_centre(Row& r, double mu) {
    const double itl = rcpq(r.tl), itu = rcpq(r.tu);
    r.zl = fmin(fmax(r.zl, mu * itl * (1.0 / kCentral)), kCentral * mu * itl);
    r.zu = fmin(fmax(r.zu, mu * itu * (1.0 / kCentral)), kCentral * mu * itu);
}
PQP_HD void row_apply(Row& r, const RowStep& s, double alpha) {
    r.g += alpha * s.dg; r.tl += alpha * s.dtl; r.tu += alpha * s.dtu; r.zl += alpha * s.dzl; r.zu += alpha * s.dzu;
}

// value function 1/2 x^T P x + p^T x of the backward sweep: P symmetric as (00, 01, 02, 11, 12, 22)
struct Value { double P[6], p[3]; };
struct Stage { double m00, m01, m10, m11, m12, c0, c1, ds; };

// One backward step across the transition `s`: the control's feedback law out of the successor's value function, and that value
// function pulled back to waypoint i (without waypoint i's own stage cost).
PQP_HD void riccati_step(const Stage& s, double w_u, Value& v, double* K, double& kk) {
    const double g0 = v.P[2], g1 = v.P[4], g2 = v.P[5];
    const double S = w_u + s.ds * s.ds * g2;
    const double iS = rcpq(S);
    const double r = s.ds * s.ds * iS, wS = w_u * iS, f = s.ds * iS;
    // Pb = P - r g g^T, its last row / column in the cancellation-free form g w_u / S; pb likewise
    const double b00 = v.P[0] - r * g0 * g0, b01 = v.P[1] - r * g0 * g1, b11 = v.P[3] - r * g1 * g1;
    const double b02 = g0 * wS, b12 = g1 * wS, b22 = g2 * wS;
    const double q0 = v.p[0] - r * v.p[2] * g0, q1 = v.p[1] - r * v.p[2] * g1, q2 = v.p[2] * wS;
    K[0] = f * (s.m00 * g0 + s.m10 * g1);
    K[1] = f * (s.m01 * g0 + s.m11 * g1);
    K[2] = f * (s.m12 * g1 + g2);
    kk = f * (g0 * s.c0 + g1 * s.c1 + v.p[2]);
    const double h0 = b00 * s.c0 + b01 * s.c1 + q0;
    const double h1 = b01 * s.c0 + b11 * s.c1 + q1;
    const double h2 = b02 * s.c0 + b12 * s.c1 + q2;
    const double t00 = b00 * s.m00 + b01 * s.m10, t01 = b00 * s.m01 + b01 * s.m11, t02 = b01 * s.m12 + b02;
    const double t10 = b01 * s.m00 + b11 * s.m10, t11 = b01 * s.m01 + b11 * s.m11, t12 = b11 * s.m12 + b12;
    const double t22 = b12 * s.m12 + b22;
    v.P[0] = s.m00 * t00 + s.m10 * t10;
    v.P[1] = s.m00 * t01 + s.m10 * t11;
    v.P[2] = s.m00 * t02 + s.m10 * t12;
    v.P[3] = s.m01 * t01 + s.m11 * t11;
    v.P[4] = s.m01 * t02 + s.m11 * t12;
    v.P[5] = s.m12 * t12 + t22;
    v.p[0] = s.m00 * h0 + s.m10 * h1;
    v.p[1] = s.m01 * h0 + s.m11 * h1;
    v.p[2] = s.m12 * h1 + h2;
}
// + 1/2 w (x_l + L x_psi - t)^2
PQP_HD void add_lpsi_term(Value& v, double w, double L, double t) {
    v.P[0] += w; v.P[1] += w * L; v.P[3] += w * L * L;
    v.p[0] -= w * t; v.p[1] -= w * t * L;
}

enum Mode { MODE_INIT = 0, MODE_IPM = 1, MODE_GUESS = 2, MODE_SET = 3, MODE_SET_GUARDED = 4 };      // (the last: roll-outs only, forward_set)

// The solver of one QP.  WS: the lane's view of the workspace, ld(field, waypoint) / st(field, waypoint, value).
template <class WS>
struct Solver {
    const Args& a;
    WS ws;
    int qp, n;
    double Lf, Lr, w_l, w_k, w_u, w_s;
    double x0[3];
    double kl;                    // curvature limit
    double psi_lo, psi_hi;        // end-heading box (psi_hi >= kBig: none)
    // end rows: interior-point state and active-set state
    Row el, ep;
    int act_el, act_ep;
    double lam_el, lam_ep;
    // interior-point scalars
    double alpha, sm_prev, mu, res;
    int ipm_iters, set_rounds, fac, ipm_iters_first, set_rounds_first;
    bool lin0;                    // this pass linearises around (0, 0, k_ref): M = [[1, ds, 0], [m10, 1, ds], [0, 0, 1]], c = (0, c1)

    PQP_HD Solver(const Args& a_, int qp_, WS ws_) : a(a_), ws(ws_), qp(qp_) {}

    // where a sweep's per-waypoint record is read from: the workspace itself, or the wavefront's LDS slot the record was staged into ahead of time
    // (WS::kStageDepth > 0: pqp_path_stream.hip).  A slot is laid out like a workspace block, so a field keeps its index whatever waypoint it came from.
    struct FromWs {
        const WS w;
        PQP_HD double ld(int f, int i) const { return w.ld(f, i); }
        PQP_HD float ldf(int f, int i) const { return w.ldf(f, i); }
    };
    struct FromSlot {
        const WS w;
        int slot;
        PQP_HD double ld(int f, int) const { return w.slot_ld(slot, f); }
        PQP_HD float ldf(int f, int) const { return w.slot_ldf(slot, f); }
    };
    template <class Src>
    PQP_HD Stage load_stage(const Src& src, int i) const {
        Stage s;
        s.m10 = src.ld(D_M10, i); s.c1 = src.ld(D_C1, i); s.ds = src.ld(D_DS, i);
        if (lin0) { s.m00 = 1.0; s.m01 = s.ds; s.m11 = 1.0; s.m12 = s.ds; s.c0 = 0.0; }      // (3 of the 8 doubles: the first pass's share of the traffic)
        else { s.m00 = src.ld(D_M00, i); s.m01 = src.ld(D_M01, i); s.m11 = src.ld(D_M11, i); s.m12 = src.ld(D_M12, i); s.c0 = src.ld(D_C0, i); }
        return s;
    }
    PQP_HD Stage load_stage(int i) const { return load_stage(FromWs{ws}, i); }

    // ---- stage data of a pass: the transition rows around the linearisation point (base_solver.cpp:165-186) -----------------------
    // src 0: (0, 0, k_ref) (path_optimizer.cpp:128-137), 1: a.lin, 2: the previous pass's optimum (F_X*)
    PQP_HD void lin_at(int src, int i, double& l, double& psi, double& k) const {
        if (src == 2) { l = ws.ld(D_X0, i); psi = ws.ld(D_X1, i); k = ws.ld(D_X2, i); }
        else if (src == 1) { const double* p = a.lin + ((size_t)qp * a.n + i) * PQP_LIN_STRIDE; l = p[0]; psi = p[1]; k = p[2]; }
        else { l = 0.0; psi = 0.0; k = a.ref[((size_t)qp * a.n + i) * PQP_REF_STRIDE + 1]; }
    }
    struct PrepIn { double l, psi, k, s, kref, b[6], mark; };
    double poison;        // 0 while every number of the scenario is finite and the arclength increases; NaN otherwise (first prep of a QP)
    PQP_SWEEP void prep(int src, bool with_bounds) {
        const double* rq = a.ref + (size_t)qp * a.n * PQP_REF_STRIDE;
        const double* bq = a.bounds + (size_t)qp * a.n * PQP_BOUNDS_STRIDE;
        PrepIn prev;
        lin0 = src == 0;
        sweep_up<kDepth, PrepIn>(0, n, [&](int i) {
            PrepIn in;
            lin_at(src, i, in.l, in.psi, in.k);
            in.s = rq[PQP_REF_STRIDE * i]; in.kref = rq[PQP_REF_STRIDE * i + 1];
            if (with_bounds) for (int k = 0; k < 6; ++k) in.b[k] = bq[PQP_BOUNDS_STRIDE * i + k];
            // (first preparation of a QP: is every number of the scenario one? v - v is 0 for a finite v, NaN otherwise; the pose columns too - unpack() reads them)
            if (with_bounds) in.mark = not_finite_mark(rq[PQP_REF_STRIDE * i + 2]) + not_finite_mark(rq[PQP_REF_STRIDE * i + 3]) + not_finite_mark(rq[PQP_REF_STRIDE * i + 4]);
            return in;
        }, [&](int i, const PrepIn& in) {
            if (with_bounds) {
                double m = in.mark + not_finite_mark(in.s) + not_finite_mark(in.kref) + not_finite_mark(in.l) + not_finite_mark(in.psi) + not_finite_mark(in.k);
                for (int k = 0; k < 6; ++k) m += not_finite_mark(in.b[k]);
                if (i > 0 && !(in.s - prev.s > 0.0)) m = __builtin_nan("");       // (an arclength that does not increase: the reference divides by ds)
                poison += m;
            }
            if (i > 0) {            // transition i - 1 -> i around the linearisation point of waypoint i - 1
                const double l = prev.l, psi = prev.psi, k = prev.k;
                const double t = tan(psi), cs = cos(psi);
                const double df00 = -k * t, df01 = (1 - k * l) / (cs * cs);
                const double df10 = -k * k / cs, df11 = (1 - k * l) * k * t / cs, df12 = (1 - k * l) / cs;
                const double ds = in.s - prev.s;
                const double f0 = (1 - k * l) * t, f1 = (1 - k * l) * k / cs - prev.kref;
                ws.st2(D_M00, i - 1, ds * df00 + 1.0, ds * df01);
                ws.st2(D_M10, i - 1, ds * df10, ds * df11 + 1.0);
                ws.st2(D_M12, i - 1, ds * df12, ds * (f0 - (df00 * l + df01 * psi)));
                ws.st2(D_C1, i - 1, ds * (f1 - (df10 * l + df11 * psi + df12 * k)), ds);
            }
            if (with_bounds) {
                const bool rough = a.prm.rough_constraints_far_away && !(in.s < a.prm.precise_planning_length);
                double lo, up;
                if (!rough) {
                    soft_bounds(in.b[0], in.b[1], a.prm.expected_safety_margin, a.prm.min_clearance, lo, up);
                    ws.st2(D_LOF, i, lo, up);
                    soft_bounds(in.b[2], in.b[3], a.prm.expected_safety_margin, a.prm.min_clearance, lo, up);
                    ws.st2(D_LOR, i, lo, up);
                } else {            // base_solver.cpp:201-205,241-247: one row on l alone with the centre circle's box
                    soft_bounds(in.b[4], in.b[5], a.prm.expected_safety_margin, a.prm.min_clearance, lo, up);
                    ws.st2(D_LOF, i, lo, up);
                    ws.st2(D_LOR, i, -kInfty, kInfty);
                }
            }
            prev = in;
        });
    }

    // ---- software pipelining ---------------------------------------------------------------------------------------------------
    // Every sweep is a dependency chain in the waypoint index whose loads depend on nothing: with one wavefront per SIMD nothing else
    // hides their latency (~1.5 us per waypoint measured, profiles/r03a_stream_first.txt), so the loads of waypoint i -+ D are issued before
    // waypoint i is computed.  D buffers in registers (the D-step inner loops unroll: static indices), D * 26 doubles at most.
    template <int D, class In, class Load, class Body>
    PQP_HD void sweep_down(int i0, int i_last, Load load, Body body) {
        In buf[D];
#pragma unroll
        for (int k = 0; k < D; ++k) if (i0 - k >= i_last) buf[k] = load(i0 - k);
        for (int i = i0; i >= i_last; i -= D) {
#pragma unroll
            for (int k = 0; k < D; ++k) {
                const int ii = i - k;
                if (ii >= i_last) {
                    const In cur = buf[k];
                    if (ii - D >= i_last) buf[k] = load(ii - D);
                    body(ii, cur);
                }
            }
        }
    }
    template <int D, class In, class Load, class Body>
    PQP_HD void sweep_up(int i0, int i_end, Load load, Body body) {          // i0 <= i < i_end
        In buf[D];
#pragma unroll
        for (int k = 0; k < D; ++k) if (i0 + k < i_end) buf[k] = load(i0 + k);
        for (int i = i0; i < i_end; i += D) {
#pragma unroll
            for (int k = 0; k < D; ++k) {
                const int ii = i + k;
                if (ii < i_end) {
                    const In cur = buf[k];
                    if (ii + D < i_end) buf[k] = load(ii + D);
                    body(ii, cur);
                }
            }
        }
    }
    static constexpr int kDepth = 1;         // (2 / 3 / 4 / 6 waypoints ahead: slower at every depth - spills; profiles/r03a_stream_first.txt)
    // The same pipelines with the records staged in LDS (round 6): `stage(slot, i)` issues the copies of waypoint i's record into a slot, `load(slot, i)`
    // reads it from there when its turn has come - D waypoints ahead without a register held for them.  Before a record is read the wavefront
    // waits until at most the copies issued AFTER that record's are outstanding (vector memory returns in order): C per later record.
    template <int D, int C, class In, class Stg, class Load, class Body>
    PQP_HD void sweep_down_staged(int i0, int i_last, Stg stage, Load load, Body body) {
#pragma unroll
        for (int k = 0; k < D; ++k) if (i0 - k >= i_last) stage(k, i0 - k);
        for (int i = i0; i >= i_last; i -= D) {
#pragma unroll
            for (int k = 0; k < D; ++k) {
                const int ii = i - k;
                if (ii >= i_last) {
                    ws.template staged_wait<C>(ii - i_last < D - 1 ? ii - i_last : D - 1);
                    const In cur = load(k, ii);
                    ws.reads_done();          // (the copies below overwrite the slot just read)
                    if (ii - D >= i_last) stage(k, ii - D);
                    body(ii, cur);
                }
            }
        }
    }
    template <int D, int C, class In, class Stg, class Load, class Body>
    PQP_HD void sweep_up_staged(int i0, int i_end, Stg stage, Load load, Body body) {          // i0 <= i < i_end
#pragma unroll
        for (int k = 0; k < D; ++k) if (i0 + k < i_end) stage(k, i0 + k);
        for (int i = i0; i < i_end; i += D) {
#pragma unroll
            for (int k = 0; k < D; ++k) {
                const int ii = i + k;
                if (ii < i_end) {
                    ws.template staged_wait<C>(i_end - 1 - ii < D - 1 ? i_end - 1 - ii : D - 1);
                    const In cur = load(k, ii);
                    ws.reads_done();
                    if (ii + D < i_end) stage(k, ii + D);
                    body(ii, cur);
                }
            }
        }
    }

    // what a sweep reads per waypoint
    struct Box { double lof, upf, lor, upr; };
    struct IpmRows { float tlf, tuf, zlf, zuf, tlr, tur, zlr, zur, tlk, tuk, zlk, zuk; double gk, x2; };      // (x2: the other half of GK's chunk, staged form only)
    struct BackIn { Stage s; Box b; IpmRows r; float dgf, dgr, dgk; double act, lam; };
    struct FwdIn { Stage s; double K0, K1, K2, kk; Box b; IpmRows r; double act, lam; double xo[3]; };     // stage / gains of transition i, rows of waypoint i + 1

    template <class Src>
    PQP_HD Box load_box(const Src& src, int i) const { Box b; b.lof = src.ld(D_LOF, i); b.upf = src.ld(D_UPF, i); b.lor = src.ld(D_LOR, i); b.upr = src.ld(D_UPR, i); return b; }
    PQP_HD Box load_box(int i) const { return load_box(FromWs{ws}, i); }
    template <class Src>
    PQP_HD IpmRows load_rows(const Src& src, int i) const {
        IpmRows r;
        r.tlf = src.ldf(S_TLF, i); r.tuf = src.ldf(S_TUF, i); r.zlf = src.ldf(S_ZLF, i); r.zuf = src.ldf(S_ZUF, i);
        r.tlr = src.ldf(S_TLR, i); r.tur = src.ldf(S_TUR, i); r.zlr = src.ldf(S_ZLR, i); r.zur = src.ldf(S_ZUR, i);
        r.tlk = src.ldf(S_TLK, i); r.tuk = src.ldf(S_TUK, i); r.zlk = src.ldf(S_ZLK, i); r.zuk = src.ldf(S_ZUK, i);
        r.gk = src.ld(D_GK, i);
        if constexpr (WS::kStageDepth > 0) r.x2 = src.ld(D_X2, i);          // (the same 16 bytes of the slot)
        return r;
    }
    // a collision row's state: its value is lo + t_l by definition (the slack absorbs the rest), t_u carries its own rounding
    PQP_HD static Row soft_row(float tl, float tu, float zl, float zu, double lo) { Row r; r.tl = tl; r.tu = tu; r.zl = zl; r.zu = zu; r.g = lo + r.tl; return r; }
    PQP_HD static Row hard_row(double g, float tl, float tu, float zl, float zu) { Row r; r.g = g; r.tl = tl; r.tu = tu; r.zl = zl; r.zu = zu; return r; }
    PQP_HD void store_f(int i, const Row& r) const { ws.stf4(S_TLF, i, r.tl, r.tu, r.zl, r.zu); }
    PQP_HD void store_r(int i, const Row& r) const { ws.stf4(S_TLR, i, r.tl, r.tu, r.zl, r.zu); }
    PQP_HD void store_k(int i, const Row& r) const { ws.st(D_GK, i, r.g); ws.stf4(S_TLK, i, r.tl, r.tu, r.zl, r.zu); }
    template <int MODE, class Src>
    PQP_HD BackIn load_back(const Src& src, int i) const {           // transition i (i < n - 1) and the rows of waypoint i (i > 0)
        BackIn in;
        if (i < n - 1) in.s = load_stage(src, i);
        if (i > 0) {
            in.b = load_box(src, i);
            if (MODE == MODE_IPM || MODE == MODE_GUESS) { in.r = load_rows(src, i); in.dgf = src.ldf(S_DGF, i); in.dgr = src.ldf(S_DGR, i); in.dgk = src.ldf(S_DGK, i); }
            if (MODE == MODE_SET) { in.act = src.ld(D_ACT, i); in.lam = src.ld(D_LAM, i); }
        }
        return in;
    }
    template <int MODE, class Src>
    PQP_HD FwdIn load_fwd(const Src& src, int i) const {
        FwdIn in;
        in.s = load_stage(src, i);
        in.K0 = src.ld(D_K0, i); in.K1 = src.ld(D_K1, i); in.K2 = src.ld(D_K2, i); in.kk = src.ld(D_KK, i);
        in.b = load_box(src, i + 1);
        if (MODE == MODE_IPM) in.r = load_rows(src, i + 1);
        if (MODE == MODE_SET || MODE == MODE_SET_GUARDED) { in.act = src.ld(D_ACT, i + 1); in.lam = src.ld(D_LAM, i + 1); }
        if (MODE == MODE_SET_GUARDED) { in.xo[0] = src.ld(D_X0, i + 1); in.xo[1] = src.ld(D_X1, i + 1); in.xo[2] = src.ld(D_X2, i + 1); }      // the point the set was taken from
        return in;
    }
    // ---- staging (WS::kStageDepth > 0) ----------------------------------------------------------------------------------------------
    // What load_back / load_fwd read, as 16-byte-per-lane chunks of a workspace block (chunk c = double fields 2c, 2c + 1; the float fields fill
    // doubles 22 .. 29) copied into LDS slot `slot` by LDS-direct loads: no register is tied up while they are in flight, so the records of
    // kStageDepth waypoints ahead can be.  A forward record takes its transition and gains from block i and its rows from block i + 1 -
    // different fields, so one slot holds both.  kBackChunks / kFwdChunks: copies per record (a lower bound: it sizes the wait).
    //   chunks: 0-3 transition (the first pass: 1 and 3), 4-5 boxes, 6-7 gains, 8 X0 X1, 9 X2 GK, 10 ACT LAM, 11-13 the three rows' slacks and multipliers, 14 their steps
    // (chunks of one waypoint go out in groups around a centre chunk - StagedWs::stage_group: one address, one M0 per group)
    template <int MODE> static constexpr int back_chunks() { return 2 + 2 + ((MODE == MODE_IPM || MODE == MODE_GUESS) ? 5 : 0) + (MODE == MODE_SET ? 1 : 0); }
    template <int MODE>
    PQP_HD void stage_back(int slot, int i) const {
        if (!lin0) ws.template stage_group<4, 0, 2>(slot, i);
        ws.template stage_group<4, 1, 3, 4, 5>(slot, i);                                                                   // transition, boxes
        if (MODE == MODE_IPM || MODE == MODE_GUESS) ws.template stage_group<12, 9, 11, 12, 13, 14>(slot, i);              // GK, the rows' states and steps
        if (MODE == MODE_SET) ws.template stage_group<12, 10>(slot, i);
    }
    template <int MODE> static constexpr int fwd_chunks() { return 2 + 2 + 2 + (MODE == MODE_IPM ? 4 : 0) + ((MODE == MODE_SET || MODE == MODE_SET_GUARDED) ? 1 : 0) + (MODE == MODE_SET_GUARDED ? 2 : 0); }
    template <int MODE>
    PQP_HD void stage_fwd(int slot, int i) const {
        if (!lin0) ws.template stage_group<4, 0, 2>(slot, i);
        ws.template stage_group<4, 1, 3, 6, 7>(slot, i);                                                                   // transition, gains
        if (MODE == MODE_IPM) { ws.template stage_group<4, 4, 5>(slot, i + 1); ws.template stage_group<12, 9, 11, 12, 13>(slot, i + 1); }       // boxes | GK, the rows' states
        else if (MODE == MODE_SET) ws.template stage_group<8, 4, 5, 10>(slot, i + 1);
        else if (MODE == MODE_SET_GUARDED) ws.template stage_group<8, 4, 5, 8, 9, 10>(slot, i + 1);
        else ws.template stage_group<4, 4, 5>(slot, i + 1);
    }
    // a sweep over staged records: the one function the sweeps below call
    template <int MODE, class Body>
    PQP_HD void sweep_back(Body body) {
        if constexpr (WS::kStageDepth > 0)
            sweep_down_staged<WS::kStageDepth, back_chunks<MODE>(), BackIn>(n - 1, 0, [&](int slot, int i) { stage_back<MODE>(slot, i); },
                                                                           [&](int slot, int i) { return load_back<MODE>(FromSlot{ws, slot}, i); }, body);
        else
            sweep_down<kDepth, BackIn>(n - 1, 0, [&](int i) { return load_back<MODE>(FromWs{ws}, i); }, body);
    }
    template <int MODE, class Body>
    PQP_HD void sweep_fwd(Body body) {
        if constexpr (WS::kStageDepth > 0)
            sweep_up_staged<WS::kStageDepth, fwd_chunks<MODE>(), FwdIn>(0, n - 1, [&](int slot, int i) { stage_fwd<MODE>(slot, i); },
                                                                       [&](int slot, int i) { return load_fwd<MODE>(FromSlot{ws, slot}, i); }, body);
        else
            sweep_up<kDepth, FwdIn>(0, n - 1, [&](int i) { return load_fwd<MODE>(FromWs{ws}, i); }, body);
    }

    // ---- one backward sweep -------------------------------------------------------------------------------------------------------
    // row terms of waypoint i in the given mode; for MODE_IPM / MODE_GUESS the waypoint's interior-point state first takes the step of
    // the previous roll-out (alpha, sm_prev), which is where that state is updated
    template <int MODE>
    PQP_HD void stage_cost(int i, const BackIn& in, double sm, Value& v) {
        v.P[0] += w_l; v.P[5] += w_k;
        if (i == 0) return;                                   // x_0 is given: its rows are constants
        const double lof = in.b.lof, upf = in.b.upf, lor = in.b.lor, upr = in.b.upr;
        const double L0 = lor <= -kBig ? 0.0 : Lf;            // a rough waypoint: one row on l alone
        const bool live_f = upf - lof > kEqWidth, live_r = upr < kBig && upr - lor > kEqWidth, on_r = upr < kBig;
        if (MODE == MODE_INIT) {
            const double w0 = 1.0, ws0 = w_s * w0 / (w_s + w0);
            add_lpsi_term(v, ws0, L0, live_f ? 0.5 * (lof + upf) : upf);
            if (on_r) add_lpsi_term(v, ws0, Lr, live_r ? 0.5 * (lor + upr) : upr);
            v.P[5] += w0;                                     // kappa towards 0, the middle of its box
            return;
        }
        if (MODE == MODE_SET) {
            const int code = (int)in.act;
            const int af = code % 3 - 1, ar = (code / 3) % 3 - 1, ak = code / 9 - 1;
            if (af != 0) add_lpsi_term(v, w_s, L0, af > 0 ? upf : lof);
            if (ar != 0) add_lpsi_term(v, w_s, Lr, ar > 0 ? upr : lor);
            if (ak != 0) { const double w = kInvDelta; v.P[5] += w; v.p[2] -= w * (ak * kl - kDelta * in.lam); }
            return;
        }
        // interior-point state of the waypoint: previous step applied, then this iteration's weights (or the set it predicts)
        Row rf = soft_row(in.r.tlf, in.r.tuf, in.r.zlf, in.r.zuf, lof), rr = soft_row(in.r.tlr, in.r.tur, in.r.zlr, in.r.zur, lor);
        Row rk = hard_row(in.r.gk, in.r.tlk, in.r.tuk, in.r.zlk, in.r.zuk);
        if (alpha > 0.0) {
            if (live_f) { row_apply(rf, row_step(rf, lof, upf, sm_prev, in.dgf), alpha); row_centre(rf, mu); }
            if (live_r) { row_apply(rr, row_step(rr, lor, upr, sm_prev, in.dgr), alpha); row_centre(rr, mu); }
            row_apply(rk, row_step(rk, -kl, kl, sm_prev, in.dgk), alpha); row_centre(rk, mu);
        }
        if (MODE == MODE_IPM) {
            double d, tgt;
            // (weights from the state AS STORED: the roll-out recomputes them from what it reads back)
            if (live_f) { store_f(i, rf); rf = soft_row((float)rf.tl, (float)rf.tu, (float)rf.zl, (float)rf.zu, lof); row_weight(rf, lof, upf, sm, d, tgt); add_lpsi_term(v, w_s * d * rcpq(w_s + d), L0, tgt); }
            else add_lpsi_term(v, w_s, L0, upf);
            if (live_r) { store_r(i, rr); rr = soft_row((float)rr.tl, (float)rr.tu, (float)rr.zl, (float)rr.zu, lor); row_weight(rr, lor, upr, sm, d, tgt); add_lpsi_term(v, w_s * d * rcpq(w_s + d), Lr, tgt); }
            else if (on_r) add_lpsi_term(v, w_s, Lr, upr);
            // (staged form: GK goes out with the X2 it shares its 16-byte chunk with - read from the slot, written back as it was: a store of 8 bytes per lane at a
            //  stride of 16 leaves every line half written)
            if constexpr (WS::kStageDepth > 0) { static_assert(D_GK == D_X2 + 1 && D_X2 % 2 == 0, "X2 and GK share a chunk"); ws.st2(D_X2, i, in.r.x2, rk.g); ws.stf4(S_TLK, i, rk.tl, rk.tu, rk.zl, rk.zu); }
            else store_k(i, rk);
            rk = hard_row(rk.g, (float)rk.tl, (float)rk.tu, (float)rk.zl, (float)rk.zu);
            row_weight(rk, -kl, kl, sm, d, tgt);
            v.P[5] += d; v.p[2] -= d * tgt;
            return;
        }
        // MODE_GUESS: a side is active when its multiplier outweighs its slack
        const int af = !live_f ? 1 : (rf.zu > rf.tu ? 1 : (rf.zl > rf.tl ? -1 : 0));
        const int ar = !on_r ? 0 : (!live_r ? 1 : (rr.zu > rr.tu ? 1 : (rr.zl > rr.tl ? -1 : 0)));
        const int ak = rk.zu > rk.tu ? 1 : (rk.zl > rk.tl ? -1 : 0);
        const double lam = ak > 0 ? rk.zu : (ak < 0 ? -rk.zl : 0.0);
        ws.st2(D_ACT, i, (double)((af + 1) + 3 * (ar + 1) + 9 * (ak + 1)), lam);
        if (af != 0) add_lpsi_term(v, w_s, L0, af > 0 ? upf : lof);
        if (ar != 0) add_lpsi_term(v, w_s, Lr, ar > 0 ? upr : lor);
        if (ak != 0) { const double w = kInvDelta; v.P[5] += w; v.p[2] -= w * (ak * kl - kDelta * lam); }
    }
    // the two end rows (base_solver.cpp:208-209,250-259), part of waypoint n - 1
    PQP_HD void end_cost(int mode, double sm, Value& v) {
        const bool has_ep = psi_hi < kBig;
        const double L = a.prm.end_l_bound;
        if (mode == MODE_INIT) {
            // the initial point reaches for the middle of the end boxes with a weight the controls cannot ignore: an end heading a radian off
            // the line's would otherwise start the hard rows a radian infeasible, and the infeasible start then crawls (40 iterations)
            const double we = 1e3;
            v.P[0] += we;
            if (has_ep) { v.P[3] += we; v.p[1] -= we * 0.5 * (psi_lo + psi_hi); }
            return;
        }
        if (mode == MODE_IPM || mode == MODE_GUESS) {
            if (alpha > 0.0) {
                row_apply(el, row_step(el, -L, L, sm_prev, gp_el - el.g), alpha); row_centre(el, mu);
                if (has_ep) { row_apply(ep, row_step(ep, psi_lo, psi_hi, sm_prev, gp_ep - ep.g), alpha); row_centre(ep, mu); }
            }
            if (mode == MODE_IPM) {
                double d, tgt;
                row_weight(el, -L, L, sm, d, tgt); v.P[0] += d; v.p[0] -= d * tgt;
                if (has_ep) { row_weight(ep, psi_lo, psi_hi, sm, d, tgt); v.P[3] += d; v.p[1] -= d * tgt; }
                return;
            }
            act_el = el.zu > el.tu ? 1 : (el.zl > el.tl ? -1 : 0);
            lam_el = act_el > 0 ? el.zu : (act_el < 0 ? -el.zl : 0.0);
            act_ep = !has_ep ? 0 : (ep.zu > ep.tu ? 1 : (ep.zl > ep.tl ? -1 : 0));
            lam_ep = act_ep > 0 ? ep.zu : (act_ep < 0 ? -ep.zl : 0.0);
        }
        const double w = kInvDelta;
        if (act_el != 0) { v.P[0] += w; v.p[0] -= w * (act_el * L - kDelta * lam_el); }
        if (act_ep != 0) { v.P[3] += w; v.p[1] -= w * ((act_ep > 0 ? psi_hi : psi_lo) - kDelta * lam_ep); }
    }
    double gp_el, gp_ep;          // end-row values of the last interior-point roll-out

    template <int MODE>
    PQP_SWEEP void backward(double sm) {
        Value v;
        for (int k = 0; k < 6; ++k) v.P[k] = 0.0;
        v.p[0] = v.p[1] = v.p[2] = 0.0;
        sweep_back<MODE>([&](int i, const BackIn& in) {
            if (i < n - 1) {
                double K[3], kk;
                riccati_step(in.s, w_u, v, K, kk);
                ws.st2(D_K0, i, K[0], K[1]); ws.st2(D_K2, i, K[2], kk);
                if (i > 0) stage_cost<MODE>(i, in, sm, v);
            } else {
                stage_cost<MODE>(i, in, sm, v);
                end_cost(MODE, sm, v);
            }
        });
        fac += 1;
    }

    // ---- roll-outs ------------------------------------------------------------------------------------------------------------------
    PQP_HD void advance(const FwdIn& in, double* x) const {          // x_i -> x_{i+1}
        const double u = -(in.K0 * x[0] + in.K1 * x[1] + in.K2 * x[2]) - in.kk;
        const double y0 = in.s.m00 * x[0] + in.s.m01 * x[1] + in.s.c0;
        const double y1 = in.s.m10 * x[0] + in.s.m11 * x[1] + in.s.m12 * x[2] + in.s.c1;
        x[2] = x[2] + in.s.ds * u; x[0] = y0; x[1] = y1;
    }
    // after the initial solve: the interior-point state of every row, strictly inside its box where the row has a slack
    PQP_SWEEP void forward_init() {
        // (theta: how far inside its box a row with a slack starts, as a share of the box's width.  0.05 until round 6; 0.2 costs a bench QP 9.2 instead of 10.2
        //  iterations, -6 ... -9 % of a sorted wavefront's bytes at 60 ... 1000 waypoints - profiles/r06ag_lq_constant_sweep_emulation.txt)
        const double theta = 0.2, mu0 = 0.1;
        double x[3] = {x0[0], x0[1], x0[2]};
        Acc acc = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
        auto start = [&](double v, double lo, double up, bool slack) {
            Row r;
            const double wd = up - lo;
            r.g = slack ? fmin(fmax(v, lo + theta * wd), up - theta * wd) : v;
            // (a hard row that starts outside its box gets slacks as large as its violation: with small ones the steps of the infeasible
            //  start are cut by those very slacks and the residual falls by a few per cent per iteration)
            const double viol = fmax(fmax(lo - r.g, r.g - up), 0.0);
            r.tl = fmax(fmax(r.g - lo, theta * wd), viol); r.tu = fmax(fmax(up - r.g, theta * wd), viol);
            r.zl = mu0 * rcpq(r.tl); r.zu = mu0 * rcpq(r.tu);
            acc.s0 += r.tl * r.zl + r.tu * r.zu; acc.cnt += 2.0;
            acc.res = fmax(acc.res, fmax(fabs(r.g - lo - r.tl), fabs(up - r.g - r.tu)));
            return r;
        };
        sweep_fwd<MODE_INIT>([&](int i, const FwdIn& in) {
            advance(in, x);
            const int j = i + 1;
            const double lof = in.b.lof, upf = in.b.upf, lor = in.b.lor, upr = in.b.upr;
            const double L0 = lor <= -kBig ? 0.0 : Lf;
            if (upf - lof > kEqWidth) store_f(j, start(x[0] + L0 * x[1], lof, upf, true));
            if (upr < kBig && upr - lor > kEqWidth) store_r(j, start(x[0] + Lr * x[1], lor, upr, true));
            store_k(j, start(x[2], -kl, kl, false));
        });
        el = start(x[0], -a.prm.end_l_bound, a.prm.end_l_bound, false);
        if (psi_hi < kBig) ep = start(x[1], psi_lo, psi_hi, false);
        mu = acc.s0 / acc.cnt; res = acc.res; alpha = 0.0;
    }
    // a re-linearised pass: the interior-point state out of the previous pass's optimum, its active set and multipliers
    struct WarmIn { double xl, xp, xk, act, lam; Box b; };
    // before the direct rounds of a re-linearised pass overwrite them: the previous pass's point, set and multiplier of every waypoint into the stash
    struct StashIn { double v[5]; };
    PQP_SWEEP void stash() {
        sweep_up<kDepth, StashIn>(1, n, [&](int j) {
            StashIn in;
            in.v[0] = ws.ld(D_X0, j); in.v[1] = ws.ld(D_X1, j); in.v[2] = ws.ld(D_X2, j); in.v[3] = ws.ld(D_ACT, j); in.v[4] = ws.ld(D_LAM, j);
            return in;
        }, [&](int j, const StashIn& in) {
            ws.st2(kStash, j, in.v[0], in.v[1]); ws.st2(kStash + 2, j, in.v[2], in.v[3]); ws.st(kStash + 4, j, in.v[4]);
        });
    }
    PQP_SWEEP void warm_init(bool stashed) {
        const double mu_w = kMuWarm, sq = sqrt(kMuWarm);
        Acc acc = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
        auto start = [&](double v, double y, double lo, double up, bool slack) {
            Row r;
            r.zu = fmax(y, 0.0); r.zl = fmax(-y, 0.0);
            const double tl_min = mu_w * rcpq(fmax(r.zl, sq)), tu_min = mu_w * rcpq(fmax(r.zu, sq));
            if (slack) { r.g = fmin(fmax(fmin(fmax(v, lo), up), lo + tl_min), up - tu_min); r.tl = r.g - lo; r.tu = up - r.g; }
            else { const double viol = fmax(fmax(lo - v, v - up), 0.0); r.g = v; r.tl = fmax(fmax(v - lo, tl_min), viol); r.tu = fmax(fmax(up - v, tu_min), viol); }
            r.zl = fmax(r.zl, mu_w * rcpq(fmax(r.tl, sq))); r.zu = fmax(r.zu, mu_w * rcpq(fmax(r.tu, sq)));
            acc.s0 += r.tl * r.zl + r.tu * r.zu; acc.cnt += 2.0;
            acc.res = fmax(acc.res, fmax(fabs(r.g - lo - r.tl), fabs(up - r.g - r.tu)));
            return r;
        };
        const int fx = stashed ? kStash : D_X0, fa = stashed ? kStash + 3 : D_ACT, fl = stashed ? kStash + 4 : D_LAM;
        const double x_end_l = ws.ld(fx, n - 1), x_end_p = ws.ld(fx + 1, n - 1);
        sweep_up<kDepth, WarmIn>(1, n, [&](int j) {
            WarmIn in;
            in.xl = ws.ld(fx, j); in.xp = ws.ld(fx + 1, j); in.xk = ws.ld(fx + 2, j); in.act = ws.ld(fa, j); in.lam = ws.ld(fl, j); in.b = load_box(j);
            return in;
        }, [&](int j, const WarmIn& in) {
            const int code = (int)in.act;
            const int af = code % 3 - 1, ar = (code / 3) % 3 - 1, ak = code / 9 - 1;
            const double lof = in.b.lof, upf = in.b.upf, lor = in.b.lor, upr = in.b.upr;
            const double L0 = lor <= -kBig ? 0.0 : Lf;
            if (upf - lof > kEqWidth) {
                const double v = in.xl + L0 * in.xp;
                store_f(j, start(v, af > 0 ? w_s * (v - upf) : (af < 0 ? w_s * (v - lof) : 0.0), lof, upf, true));
            }
            if (upr < kBig && upr - lor > kEqWidth) {
                const double v = in.xl + Lr * in.xp;
                store_r(j, start(v, ar > 0 ? w_s * (v - upr) : (ar < 0 ? w_s * (v - lor) : 0.0), lor, upr, true));
            }
            store_k(j, start(in.xk, ak != 0 ? in.lam : 0.0, -kl, kl, false));
        });
        el = start(x_end_l, act_el != 0 ? lam_el : 0.0, -a.prm.end_l_bound, a.prm.end_l_bound, false);
        if (psi_hi < kBig) ep = start(x_end_p, act_ep != 0 ? lam_ep : 0.0, psi_lo, psi_hi, false);
        mu = acc.s0 / acc.cnt; res = acc.res; alpha = 0.0;
    }
    // roll-out of an interior-point iteration: row values of the Newton point, the step to the boundary, next complementarity
    PQP_SWEEP void forward_ipm(double sm) {
        double x[3] = {x0[0], x0[1], x0[2]};
        Acc acc = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
        sweep_fwd<MODE_IPM>([&](int i, const FwdIn& in) {
            advance(in, x);
            const int j = i + 1;
            const double lof = in.b.lof, upf = in.b.upf, lor = in.b.lor, upr = in.b.upr;
            const double L0 = lor <= -kBig ? 0.0 : Lf;
            double dgf = 0.0, dgr = 0.0;          // (the step of a row that is not live is never read)
            if (upf - lof > kEqWidth) {
                const Row r = soft_row(in.r.tlf, in.r.tuf, in.r.zlf, in.r.zuf, lof);
                double d, tgt;
                row_weight(r, lof, upf, sm, d, tgt);
                const double v = x[0] + L0 * x[1];
                dgf = as_stored(v - d * rcpq(w_s + d) * (v - tgt) - r.g);           // (+ the slack of the Newton point)
                row_accumulate(r, row_step(r, lof, upf, sm, dgf), lof, upf, acc, false);
            }
            if (upr < kBig && upr - lor > kEqWidth) {
                const Row r = soft_row(in.r.tlr, in.r.tur, in.r.zlr, in.r.zur, lor);
                double d, tgt;
                row_weight(r, lor, upr, sm, d, tgt);
                const double v = x[0] + Lr * x[1];
                dgr = as_stored(v - d * rcpq(w_s + d) * (v - tgt) - r.g);
                row_accumulate(r, row_step(r, lor, upr, sm, dgr), lor, upr, acc, false);
            }
            const Row r = hard_row(in.r.gk, in.r.tlk, in.r.tuk, in.r.zlk, in.r.zuk);
            const double dg = as_stored(x[2] - r.g);
            ws.stf4(S_DGF, j, dgf, dgr, dg, 0.0);          // (one chunk: the three steps; S_PAD only means something at waypoint 0, and j >= 1)
            row_accumulate(r, row_step(r, -kl, kl, sm, dg), -kl, kl, acc);
        });
        gp_el = x[0]; gp_ep = x[1];
        row_accumulate(el, row_step(el, -a.prm.end_l_bound, a.prm.end_l_bound, sm, gp_el - el.g), -a.prm.end_l_bound, a.prm.end_l_bound, acc);
        if (psi_hi < kBig) row_accumulate(ep, row_step(ep, psi_lo, psi_hi, sm, gp_ep - ep.g), psi_lo, psi_hi, acc);
        alpha = acc.rho > 0.995 ? 0.995 * rcpq(acc.rho) : 1.0;
        sm_prev = sm;
        mu = (acc.s0 + alpha * (acc.s1 + alpha * acc.s2)) / acc.cnt;      // complementarity after the step
        res = (1.0 - alpha) * acc.res;
    }
    // roll-out of an active-set round: the point, the set it asks for, the multipliers of its hard rows.  Returns true when the point
    // confirms its set (the KKT test) and holds its hard rows.
    //
    // The rounds are Newton steps on a convex, piecewise quadratic function F (a collision row that is out of its box costs w_s / 2 times
    // the square of what it is out by): the set S is the piece the current point x lies on, the round's point x_N minimises that piece's
    // quadratic q_S.  Undamped - the next point is x_N - they can cycle on paths of several hundred waypoints, where a handful of rows that
    // are barely active move the far end of the path by decimetres (1 QP in 16 384 at 512 waypoints: a cycle of 28 sets, the same one
    // from all three attempts; 5 in 131 072 at 1000).  GUARDED rounds step to x + t (x_N - x) with the largest t of 1, 1/2, 1/4, 1/8, 1/16
    // at which F still falls along the segment: F' (t) = (t - 1) d^T H_S d + (what the rows that have left their piece at x + t d add),
    // d^T H_S d and the second term for the candidate steps summed in this roll-out, D_X* keeping x and the step fields of the
    // interior-point rounds d; settle(t) then moves x and takes the collision rows' sides from where it arrives.  F falls with every
    // guarded round, so no set comes back.  (Hard rows - curvature, end boxes - change sides by their own test as before: they are
    // multiplier estimates, not part of F.)  A set is only ever accepted when its own point x_N confirms it.
    template <bool GUARDED>
    PQP_SWEEP bool forward_set() {
        double x[3] = {x0[0], x0[1], x0[2]};
        bool asks = false;          // x_N asks for another set than the one it was computed with
        double pin = 0.0;
        double dhd = 0.0, slope[5] = {0.0, 0.0, 0.0, 0.0, 0.0}, dk_prev = 0.0, dl = 0.0, dp = 0.0;
        if (!GUARDED) { ws.st2(D_X0, 0, x[0], x[1]); ws.st(D_X2, 0, x[2]); }
        auto soft = [&](int act, double v, double lo, double up) {
            // consistent within the tolerance: keep; else what the point asks for
            const bool keep = (act == 1 && v >= up - kSetTol) || (act == -1 && v <= lo + kSetTol) || (act == 0 && v <= up + kSetTol && v >= lo - kSetTol);
            return keep ? act : (v > up ? 1 : (v < lo ? -1 : 0));
        };
        auto hard = [&](int act, double& lam, double v, double lo, double up) {
            if (act != 0) {
                const double bnd = act > 0 ? up : lo;
                const double y = lam + (v - bnd) * kInvDelta;              // multiplier of the active row
                if (act * y < -kSetTol) { lam = 0.0; return 0; }       // wrong sign: released
                pin = fmax(pin, fabs(v - bnd));
                lam = y;
                return act;
            }
            lam = 0.0;
            return v > up + kSetTol ? 1 : (v < lo - kSetTol ? -1 : 0);
        };
        // a collision row along the segment: its share of d^T H_S d and of the slopes at the candidate steps (v = o + t dv; side a in S)
        auto along = [&](int a, double o, double dv, double lo, double up) {
            if (a != 0) dhd += w_s * dv * dv;
            double t = 1.0;
#pragma unroll
            for (int c = 0; c < 5; ++c, t *= 0.5) {
                const double v = o + t * dv;
                const double truth = v > up ? v - up : (v < lo ? v - lo : 0.0);
                const double model = a > 0 ? v - up : (a < 0 ? v - lo : 0.0);
                slope[c] += w_s * (truth - model) * dv;
            }
        };
        sweep_fwd<GUARDED ? MODE_SET_GUARDED : MODE_SET>([&](int i, const FwdIn& in) {
            advance(in, x);
            const int j = i + 1;
            if (!GUARDED) { ws.st2(D_X0, j, x[0], x[1]); ws.st(D_X2, j, x[2]); }
            const double lof = in.b.lof, upf = in.b.upf, lor = in.b.lor, upr = in.b.upr;
            const double L0 = lor <= -kBig ? 0.0 : Lf;
            const int code = (int)in.act;
            const int af = code % 3 - 1, ar = (code / 3) % 3 - 1, ak = code / 9 - 1;
            const bool live_f = upf - lof > kEqWidth, on_r = upr < kBig, live_r = on_r && upr - lor > kEqWidth;
            int nf = live_f ? soft(af, x[0] + L0 * x[1], lof, upf) : 1;
            int nr = !on_r ? 0 : (live_r ? soft(ar, x[0] + Lr * x[1], lor, upr) : 1);
            double lam = in.lam;
            const int nk = hard(ak, lam, x[2], -kl, kl);
            asks = asks || nf != af || nr != ar || nk != ak;
#if defined(PQP_LQ_DEBUG) && !defined(__HIP_DEVICE_COMPILE__)
            if (nf != af || nr != ar || nk != ak)
                std::fprintf(stderr, "    qp %d set round %d waypoint %d: front %d -> %d (%.3e in [%.6f, %.6f]) rear %d -> %d (%.3e in [%.6f, %.6f]) kappa %d -> %d (%.9f, lam %.3e)\n", qp, set_rounds, j,
                             af, nf, x[0] + L0 * x[1], lof, upf, ar, nr, x[0] + Lr * x[1], lor, upr, ak, nk, x[2], lam);
#endif
            if (GUARDED) {
                dl = x[0] - in.xo[0]; dp = x[1] - in.xo[1];
                const double dk = x[2] - in.xo[2], du = (dk - dk_prev) * rcpq(in.s.ds);
                dk_prev = dk;
                ws.stf4(S_DGF, j, dl, dp, dk, 0.0);
                dhd += w_l * dl * dl + (w_k + (ak != 0 ? kInvDelta : 0.0)) * dk * dk + w_u * du * du;
                if (live_f) along(af, in.xo[0] + L0 * in.xo[1], dl + L0 * dp, lof, upf); else dhd += w_s * (dl + L0 * dp) * (dl + L0 * dp);
                if (live_r) along(ar, in.xo[0] + Lr * in.xo[1], dl + Lr * dp, lor, upr); else if (on_r) dhd += w_s * (dl + Lr * dp) * (dl + Lr * dp);
                nf = af; nr = ar;           // (settle() takes the collision rows' sides from where the step arrives)
            }
            ws.st2(D_ACT, j, (double)((nf + 1) + 3 * (nr + 1) + 9 * (nk + 1)), lam);
        });
        if (GUARDED) dhd += kInvDelta * ((act_el != 0 ? dl * dl : 0.0) + (act_ep != 0 ? dp * dp : 0.0));
        const int ne = hard(act_el, lam_el, x[0], -a.prm.end_l_bound, a.prm.end_l_bound);
        asks = asks || ne != act_el; act_el = ne;
        if (psi_hi < kBig) { const int np = hard(act_ep, lam_ep, x[1], psi_lo, psi_hi); asks = asks || np != act_ep; act_ep = np; }
        set_rounds += 1;
        if (GUARDED) {
            step = 1.0 / 16.0;
            double t = 1.0;
            for (int c = 0; c < 5; ++c, t *= 0.5) if ((t - 1.0) * dhd + slope[c] <= 0.0) { step = t; break; }
        }
#if defined(PQP_LQ_DEBUG) && !defined(__HIP_DEVICE_COMPILE__)
        std::fprintf(stderr, "  qp %d set round %d: asks %d pin %.3e end rows %d %d", qp, set_rounds, (int)asks, pin, act_el, act_ep);
        if (GUARDED) std::fprintf(stderr, " | guarded: d^T H d %.3e slopes %.3e %.3e %.3e %.3e %.3e -> step %.4f", dhd, slope[0], (0.5 - 1.0) * dhd + slope[1], (0.25 - 1.0) * dhd + slope[2], (0.125 - 1.0) * dhd + slope[3], (0.0625 - 1.0) * dhd + slope[4], step);
        std::fprintf(stderr, "\n");
#endif
        return !asks && pin <= kPinTol;
    }
    double step;          // of the last guarded round
    // after a guarded round: x <- x + step d, the collision rows on the sides that point asks for
    struct SettleIn { double x0, x1, x2, act; float d0, d1, d2; Box b; };
    PQP_SWEEP void settle() {
        const double t = step;
        sweep_up<kDepth, SettleIn>(1, n, [&](int j) {
            SettleIn in;
            in.x0 = ws.ld(D_X0, j); in.x1 = ws.ld(D_X1, j); in.x2 = ws.ld(D_X2, j); in.act = ws.ld(D_ACT, j);
            in.d0 = ws.ldf(S_DGF, j); in.d1 = ws.ldf(S_DGR, j); in.d2 = ws.ldf(S_DGK, j);
            in.b = load_box(j);
            return in;
        }, [&](int j, const SettleIn& in) {
            const double l = in.x0 + t * (double)in.d0, psi = in.x1 + t * (double)in.d1;
            ws.st2(D_X0, j, l, psi); ws.st(D_X2, j, in.x2 + t * (double)in.d2);
            const double lof = in.b.lof, upf = in.b.upf, lor = in.b.lor, upr = in.b.upr;
            const double L0 = lor <= -kBig ? 0.0 : Lf;
            const int code = (int)in.act;
            const int ak = code / 9 - 1;
            const double vf = l + L0 * psi, vr = l + Lr * psi;
            // (no hysteresis here: the side is the piece of F the point lies on)
            const int nf = upf - lof > kEqWidth ? (vf > upf ? 1 : (vf < lof ? -1 : 0)) : 1;
            const int nr = upr >= kBig ? 0 : (upr - lor > kEqWidth ? (vr > upr ? 1 : (vr < lor ? -1 : 0)) : 1);
            ws.st(D_ACT, j, (double)((nf + 1) + 3 * (nr + 1) + 9 * (ak + 1)));
        });
    }

    // Returns PQP_STATUS_SOLVED, or why not.  A QP whose hard rows cannot all hold (an end box out of the controls' reach, say) shows in
    // the interior-point rounds as steps that shrink to nothing while the row residual stays: PQP_STATUS_PRIMAL_INFEASIBLE - the
    // verdict OSQP's certificate gives the lane-per-waypoint kernel on the same QP.
    PQP_HD int solve_pass(bool warm, int direct = 0) {
        // attempts: (from the previous optimum,) cold to complementarity 1e-6, cold to 1e-9.  A warm start that fails - the slot's previous QP may
        // have nothing to do with this one - is no verdict on the QP: the cold attempts follow.
        // In a sorted launch a re-linearised pass begins with attempt -1: the previous pass's set on this pass's transition rows (kDirectRounds) - a round that asks
        // for no change is the KKT test of THIS pass's QP.  What those rounds overwrite - point, sets, multipliers, the end rows' state - is what the
        // interior-point rounds of attempt 0 start from should the set not be confirmed: kept aside (stash()).
        int verdict = PQP_STATUS_MAX_ITER;
        bool stashed = false;
        int k_el = 0, k_ep = 0;
        double k_lel = 0.0, k_lep = 0.0;
        for (int attempt = (warm && direct > 0) ? -1 : (warm ? 0 : 1); attempt < 3; ++attempt) {
            const bool from_previous = attempt == 0;
            bool rounds = false;          // the attempt has a set for the active-set rounds
            if (attempt < 0) {
                k_el = act_el; k_ep = act_ep; k_lel = lam_el; k_lep = lam_ep;
                stash();
                rounds = true;
            } else {
                const double mu_stop = attempt < 2 ? kMuStop : 1e-9;
                if (from_previous) warm_init(stashed);
                else { backward<MODE_INIT>(0.0); forward_init(); }
                bool first = true;
                int it = 0, stall = 0;
                int slow = 0;          // iterations in a row with a step below 1e-3 although the rows are feasible: complementarity has stopped falling
                const double res0 = res;
                // (a QP whose hard rows start far outside their boxes - an end heading a radian off what the curvature limit can reach early on the
                //  line - crawls: 2 % of residual per iteration.  Every QP of the bench distributions is feasible within 12 iterations; one that
                //  has not shed 90 % of its initial residual after 30 gives up as PQP_STATUS_MAX_ITER instead of holding its wavefront for 100)
                while (!(mu < mu_stop && res < 1e-6) && it < kIpmMaxIter && stall < 6 && slow < 3 && !(it >= 30 && res > 0.1 * res0 && res > 1e-6)) {
                    const double sigma = (first || alpha <= 0.9) ? 0.2 : 0.05;
                    const double sm = sigma * mu;
                    backward<MODE_IPM>(sm);
                    forward_ipm(sm);
                    stall = (alpha < 1e-3 && res > 1e-6) ? stall + 1 : 0;
                    slow = (alpha < 1e-3 && res <= 1e-6) ? slow + 1 : 0;
#if defined(PQP_LQ_DEBUG) && !defined(__HIP_DEVICE_COMPILE__)
                    std::fprintf(stderr, "  qp %d attempt %d it %d: sigma %.2f alpha %.3e mu %.3e res %.3e\n", qp, attempt, it, sigma, alpha, mu, res);
#endif
                    first = false;
                    it += 1;
                }
                ipm_iters += it;
                if (!(mu == mu)) verdict = PQP_STATUS_NUMERICAL;
                else if (!(res < 1e-6)) verdict = stall >= 6 ? PQP_STATUS_PRIMAL_INFEASIBLE : PQP_STATUS_MAX_ITER;
                else if (!(mu < 1e-3)) verdict = PQP_STATUS_MAX_ITER;      // (short of mu_stop but below 1e-3: the active-set rounds get their chance)
                else rounds = true;
            }
            if (rounds) {
                // (the last attempt: guarded rounds once the plain ones have had their chance)
                const int max_rounds = attempt < 0 ? direct : (attempt == 2 ? 5 * kPolishMaxRounds : kPolishMaxRounds);
                for (int r = 0; ; ++r) {
                    if (attempt < 0 && r == max_rounds) break;
                    // (one call site per sweep: each is inlined into the kernel body)
                    if (r == 0 && attempt >= 0) backward<MODE_GUESS>(0.0); else backward<MODE_SET>(0.0);
                    if (r == max_rounds) break;
                    if (r < kPolishMaxRounds) { if (forward_set<false>()) return PQP_STATUS_SOLVED; }
                    else {
                        // (x_N confirms its set: one plain roll-out of the same gains puts it into D_X*, which a guarded one leaves at x)
                        if (forward_set<true>()) { if (forward_set<false>()) return PQP_STATUS_SOLVED; }
                        else settle();
                    }
                }
                if (attempt < 0) { act_el = k_el; act_ep = k_ep; lam_el = k_lel; lam_ep = k_lep; stashed = true; }
                verdict = PQP_STATUS_MAX_ITER;
                continue;                                              // the set was not confirmed: the next attempt
            }
            if (!from_previous) return verdict;                        // a cold attempt's verdict on the rows stands
        }
        return verdict;
    }

    // BaseSolver::getOptimizedPath (base_solver.cpp:263-288)
    struct OutIn { double l, dpsi, k, K0, K1, K2, kk, angle, rx, ry; };
    PQP_SWEEP void unpack() {
        const double* rq = a.ref + (size_t)qp * a.n * PQP_REF_STRIDE;
        double* oq = a.out + (size_t)qp * a.n * PQP_OUT_STRIDE;
        sweep_up<kDepth, OutIn>(0, n, [&](int i) {
            OutIn in;
            in.l = ws.ld(D_X0, i); in.dpsi = ws.ld(D_X1, i); in.k = ws.ld(D_X2, i);
            in.K0 = in.K1 = in.K2 = in.kk = 0.0;
            if (i < n - 1) { in.K0 = ws.ld(D_K0, i); in.K1 = ws.ld(D_K1, i); in.K2 = ws.ld(D_K2, i); in.kk = ws.ld(D_KK, i); }
            in.angle = rq[PQP_REF_STRIDE * i + 2]; in.rx = rq[PQP_REF_STRIDE * i + 3]; in.ry = rq[PQP_REF_STRIDE * i + 4];
            return in;
        }, [&](int i, const OutIn& in) {
            const double dk = i < n - 1 ? -(in.K0 * in.l + in.K1 * in.dpsi + in.K2 * in.k) - in.kk : 0.0;
            const double new_angle = constrain_angle(in.angle + kPi2);
            double* o = oq + PQP_OUT_STRIDE * i;
            o[0] = in.rx + in.l * cos(new_angle);
            o[1] = in.ry + in.l * sin(new_angle);
            o[2] = constrain_angle(in.angle + in.dpsi);
            o[3] = in.l; o[4] = in.dpsi; o[5] = in.k; o[6] = dk;
        });
    }

    PQP_HD void zero_point() {
        for (int i = 0; i < n; ++i) { ws.st(D_X0, i, 0.0); ws.st(D_X1, i, 0.0); ws.st(D_X2, i, 0.0); ws.st(D_K0, i, 0.0); ws.st(D_K1, i, 0.0); ws.st(D_K2, i, 0.0); ws.st(D_KK, i, 0.0); }
    }
    PQP_HD void finish(int status, int solved_passes) {
        if (a.status) a.status[qp] = status;
        if (a.iters) a.iters[qp] = ipm_iters;
        if (a.info) {
            double* f = a.info + (size_t)qp * PQP_INFO_STRIDE;
            f[0] = res; f[1] = mu; f[2] = (double)ipm_iters_first; f[3] = (double)ipm_iters; f[4] = (double)solved_passes; f[5] = (double)set_rounds_first;
            f[6] = (double)fac; f[7] = (double)set_rounds;
        }
    }

    PQP_HD void run() {
        n = a.n_of ? a.n_of[qp] : a.n;
        ipm_iters = 0; set_rounds = 0; fac = 0; ipm_iters_first = 0; set_rounds_first = 0; mu = 0.0; res = 0.0; alpha = 0.0; sm_prev = 0.0;
        if (n > a.n) n = a.n;
        if (n < 2) { finish(PQP_STATUS_UNSOLVED, 0); return; }
        const pqp_params& p = a.prm;
        Lf = p.front_length; Lr = p.rear_length; w_l = p.weight_l; w_k = p.weight_kappa; w_u = p.weight_dkappa; w_s = p.weight_slack;
        const double* sc = a.scal + (size_t)qp * PQP_SCAL_STRIDE;
        x0[0] = sc[0]; x0[1] = sc[1]; x0[2] = sc[2];                      // base_solver.cpp:216-220
        kl = tan(sc[5]) / p.wheel_base;                                   // :226
        // A start curvature outside its box by no more than OSQP's primal tolerance (eps_abs + eps_rel * bound) is a QP the reference calls solved - ADMM
        // meets eps with a point that misses the row by that little.  The start state is projected onto the box by that little, and the QP solved exactly (the
        // lane-per-waypoint kernel does the same in assemble()); beyond the tolerance the QP is PRIMAL_INFEASIBLE (below).
        if (fabs(x0[2]) > kl && fabs(x0[2]) - kl <= p.eps_abs + p.eps_rel * kl) x0[2] = x0[2] > 0.0 ? kl : -kl;
        psi_lo = -kInfty; psi_hi = kInfty;
        if (p.constraint_end_heading && sc[4] == 0.0) {                   // :254-258
            const double end_psi = constrain_angle(sc[3] - a.ref[((size_t)qp * a.n + n - 1) * PQP_REF_STRIDE + 2]);
            if (end_psi < p.end_psi_max) { psi_lo = end_psi - p.end_psi_tol; psi_hi = end_psi + p.end_psi_tol; }
        }
        act_el = act_ep = 0; lam_el = lam_ep = 0.0; gp_el = gp_ep = 0.0;
        // what the slot's previous QP left behind (waypoint 0's row fields are otherwise unused: x_0 is given): its end rows and its size
        bool carried = false;
        if (a.carry) {
            const int code = (int)ws.ld(D_ACT, 0);
            if (code >= 100 && code < 109 && (int)ws.ldf(S_PAD, 0) == n) {
                carried = true;
                act_el = (code - 100) % 3 - 1; act_ep = (code - 100) / 3 - 1;
                lam_el = ws.ld(D_LAM, 0); lam_ep = ws.ld(D_GK, 0);
                if (psi_hi >= kBig) { act_ep = 0; lam_ep = 0.0; }
            }
        }
        ws.st(D_ACT, 0, 0.0);
        poison = 0.0;
        for (int k = 0; k < 6; ++k) poison += not_finite_mark(sc[k]);
        prep(a.lin ? 1 : 0, true);
        if (!(poison == 0.0)) {
            // NaN / Inf in the scenario, or an arclength that does not increase: not a QP the reference could solve (OSQP would return non-finite iterates and
            // BaseSolver::solve false) - and min / max arithmetic would drop a NaN bound silently.  PQP_STATUS_NUMERICAL, a zero output record; the
            // other lanes of the wavefront are not held up by interior-point iterations on NaNs
            double* oq = a.out + (size_t)qp * a.n * PQP_OUT_STRIDE;
            for (int i = 0; i < n; ++i) for (int k = 0; k < PQP_OUT_STRIDE; ++k) oq[PQP_OUT_STRIDE * i + k] = 0.0;
            finish(PQP_STATUS_NUMERICAL, 0);
            return;
        }
        if (!(fabs(x0[2]) <= kl)) {
            // the start curvature violates its own box (kappa row 0 against the fixed x_0): no point satisfies the rows
            zero_point();
            unpack();
            finish(PQP_STATUS_PRIMAL_INFEASIBLE, 0);
            return;
        }
        int solved = 0;
        int st = solve_pass(carried);
        ipm_iters_first = ipm_iters; set_rounds_first = set_rounds;
        if (st == PQP_STATUS_SOLVED) solved += 1;
        for (int pass = 0; st == PQP_STATUS_SOLVED && pass < a.passes; ++pass) {
            prep(2, false);
            st = solve_pass(true, a.order ? kDirectRounds : 0);
            if (st == PQP_STATUS_SOLVED) solved += 1;
        }
        if (solved == 0) zero_point();          // (no active-set round ever ran: F_X* hold nothing)
        unpack();
        if (st == PQP_STATUS_SOLVED) {          // for the next planning cycle (Args::carry): the end rows' state and the size, in waypoint 0's free fields
            ws.st(D_ACT, 0, (double)(100 + (act_el + 1) + 3 * (act_ep + 1))); ws.st(D_LAM, 0, lam_el); ws.st(D_GK, 0, lam_ep); ws.stf(S_PAD, 0, (double)n);
        }
        finish(st, solved);
    }
};

// a lane's view of its wavefront's workspace block, in either of the two layouts of pqp_path_lq_abi.hpp
// [field][lane]: every access of a wavefront is one contiguous line of 8 (4) bytes per lane - the launches that fill the chip (and the host emulation)
struct StridedWs {
    static constexpr int kStageDepth = 0;      // records are prefetched into registers, one waypoint ahead
    double* block;          // the wavefront's block
    int lane;
    int lanes;              // lanes per block: 64 on the device, 1 in the host emulation
    PQP_HD double ld(int f, int i) const { return block[((size_t)i * kBlockDoubles + f) * lanes + lane]; }
    PQP_HD void st(int f, int i, double v) const { block[((size_t)i * kBlockDoubles + f) * lanes + lane] = v; }
    PQP_HD float* floats(int i) const { return reinterpret_cast<float*>(block + ((size_t)i * kBlockDoubles + kFieldsD) * lanes); }
    PQP_HD float ldf(int f, int i) const { return floats(i)[(size_t)f * lanes + lane]; }
    PQP_HD void stf(int f, int i, double v) const { floats(i)[(size_t)f * lanes + lane] = (float)v; }
    // (fields that are written together: one access per field here, one per 16-byte chunk in ChunkWs)
    PQP_HD void st2(int f, int i, double v0, double v1) const { st(f, i, v0); st(f + 1, i, v1); }
    PQP_HD void stf4(int f, int i, double v0, double v1, double v2, double v3) const { stf(f, i, v0); stf(f + 1, i, v1); stf(f + 2, i, v2); stf(f + 3, i, v3); }
};
// [chunk][lane][16 bytes]: what LDS-direct loads can stage (StagedWs, pqp_path_stream.hip) - the launches that leave SIMDs idle, whose sweeps wait for
// their loads rather than for HBM's throughput
struct ChunkWs {
    static constexpr int kStageDepth = 0;
    double* block;
    int lane;
    int lanes;
    PQP_HD size_t chunk_at(int c, int i) const { return (((size_t)i * kBlockChunks + c) * lanes + lane) * 2; }       // this lane's 16 bytes of chunk c, in doubles
    PQP_HD double ld(int f, int i) const { return block[chunk_at(f >> 1, i) + (f & 1)]; }
    PQP_HD void st(int f, int i, double v) const { block[chunk_at(f >> 1, i) + (f & 1)] = v; }
    PQP_HD float* floats(int f, int i) const { return reinterpret_cast<float*>(block + chunk_at(kFieldsD / 2 + (f >> 2), i)) + (f & 3); }
    PQP_HD float ldf(int f, int i) const { return *floats(f, i); }
    PQP_HD void stf(int f, int i, double v) const { *floats(f, i) = (float)v; }
    // a whole chunk in one 16-byte access: the two doubles of an even field pair, the four floats of a row
    struct alignas(16) D2 { double a, b; };
    struct alignas(16) F4 { float a, b, c, d; };
    PQP_HD void st2(int f, int i, double v0, double v1) const { *reinterpret_cast<D2*>(block + chunk_at(f >> 1, i)) = D2{v0, v1}; }
    PQP_HD void stf4(int f, int i, double v0, double v1, double v2, double v3) const {
        *reinterpret_cast<F4*>(block + chunk_at(kFieldsD / 2 + (f >> 2), i)) = F4{(float)v0, (float)v1, (float)v2, (float)v3};
    }
};

}  // namespace lq
}  // namespace pqp
