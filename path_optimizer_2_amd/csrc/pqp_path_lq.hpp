// pqp_path_lq.hpp — the path QP as a linear-quadratic control problem, ONE LANE PER QP, every per-waypoint quantity streamed through a
// batch-interleaved workspace in HBM: the large-batch solver behind pqp_path_solve_device (PQP_OPT_STREAM_BATCH).
//
// reference                                                   here
//   BaseSolver::setCost          src/solver/base_solver.cpp:119-148   the stage costs of `backward()` (weights 0 / 20 / 100, slack weight 10)
//   BaseSolver::setConstraints   :150-261                             `prep()`: the transition rows become the dynamics x_{i+1} = M_i x_i + b u_i + c_i
//                                                                     (same expressions as :165-186), the kappa / collision / end rows become boxes
//   getSoftBounds                :290-295                             soft_bounds() (pqp_path_lane.hpp)
//   OsqpEigen solve              :88,110                              `ipm()` + `polish()` below (NOT OSQP's ADMM: see "algorithm")
//   updateProblemFormulationAndSolve :97-117                          the pass loop of solve_one(): re-linearise around the previous optimum
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
// The same source compiles for the device (path_stream_kernel, pqp_path_stream.inc) and, for tests only, for the host (tests/emu).
#pragma once
#include "pqp_path_lane.hpp"

namespace pqp {
namespace lq {

// workspace fields per waypoint (doubles); element (waypoint i, field f) of a lane sits at base[(i * kFields + f) * lane_stride]
enum Field {
    F_M00 = 0, F_M01, F_M10, F_M11, F_M12, F_C0, F_C1, F_DS,      // transition i -> i + 1 (i < n - 1)
    F_LOF, F_UPF, F_LOR, F_UPR,                                    // soft boxes of the collision rows (rear off: up = +inf)
    F_K0, F_K1, F_K2, F_KK,                                        // feedback law u_i = -K x_i - k
    F_X0, F_X1, F_X2,                                              // the point of the last active-set round
    F_GPF, F_GPR, F_GPK,                                           // row values of the last interior-point roll-out
    F_GF, F_ZLF, F_ZUF, F_GR, F_ZLR, F_ZUR,                        // interior-point state of the two collision rows (t = g - lo, up - g)
    F_GK, F_TLK, F_TUK, F_ZLK, F_ZUK,                              // ... of the kappa row (infeasible start: its own t)
    kFields
};
constexpr int F_ACT = F_GF;       // active-set rounds: the three rows' states packed as f + 3 r + 9 k + 13 (aliases the interior-point state)
constexpr int F_LAM = F_GK;       //                    multiplier of the kappa row

constexpr double kBig = 1e29;           // a bound beyond this is no bound (OSQP_INFTY = 1e30)
constexpr double kDelta = 1e-9;         // hard active rows: penalty 1 / delta around the bound shifted by delta * multiplier
constexpr double kSetTol = 1e-7;        // a row changes sides when it fails its test by more than this (pqp_params.polish_tol's role)
constexpr double kPinTol = 1e-9;        // accepted points hold their hard active rows to this
constexpr double kMuStop = 1e-6;        // complementarity at which the interior-point rounds hand over to the active-set rounds
constexpr double kMuWarm = 1e-3;        // complementarity a re-linearised pass starts from
constexpr double kEqWidth = 1e-6;       // a collision box narrower than this is an equality row (weight w_s at its upper bound)
constexpr int kIpmMaxIter = 60;
constexpr int kPolishMaxRounds = 12;

struct Args {
    int batch, n, passes;
    const int32_t* n_of;        // [batch] or nullptr
    const double* ref;          // [batch][n][5]
    const double* lin;          // [batch][n][3] or nullptr
    const double* bounds;       // [batch][n][6]
    const double* scal;         // [batch][6]
    double* out;                // [batch][n][7]
    int32_t* status;            // [batch] or nullptr
    int32_t* iters;             // [batch] or nullptr: interior-point iterations over all passes
    double* info;               // [batch][PQP_INFO_STRIDE] or nullptr
    double* ws;                 // [ceil(batch / 64)][n][kFields][64]
    pqp_params prm;
};

// a two-sided row of the interior-point rounds
struct Row { double g, tl, tu, zl, zu; };
struct RowStep { double dg, dtl, dtu, dzl, dzu; };

struct Acc {          // what a roll-out accumulates for the step length and the next complementarity
    double rho;       // max over row sides of -delta / value (the step to the boundary is 1 / rho)
    double s0, s1, s2, cnt;   // sum t z, sum (t dz + z dt), sum dt dz, number of row sides
    double res;       // largest primal residual of the rows
};

// Newton target of a row: weight d and target of the quadratic term that replaces the barrier (sm = sigma * mu)
PQP_HD void row_weight(const Row& r, double lo, double up, double sm, double& d, double& tgt) {
    const double itl = rcp(r.tl), itu = rcp(r.tu);
    const double rl = r.g - lo - r.tl, ru = up - r.g - r.tu;
    d = r.zu * itu + r.zl * itl;
    const double e = sm * (itu - itl) - r.zu * itu * ru + r.zl * itl * rl;
    tgt = r.g - e * rcp(d);
}
// the step of a row's state towards the Newton point whose row value is gp
PQP_HD RowStep row_step(const Row& r, double lo, double up, double sm, double gp) {
    const double itl = rcp(r.tl), itu = rcp(r.tu);
    const double rl = r.g - lo - r.tl, ru = up - r.g - r.tu;
    RowStep s;
    s.dg = gp - r.g;
    s.dtl = s.dg + rl;
    s.dtu = -s.dg + ru;
    s.dzl = (sm - r.zl * s.dtl) * itl - r.zl;
    s.dzu = (sm - r.zu * s.dtu) * itu - r.zu;
    return s;
}
PQP_HD void row_accumulate(const Row& r, const RowStep& s, double lo, double up, Acc& a) {
    const double itl = rcp(r.tl), itu = rcp(r.tu), izl = rcp(r.zl), izu = rcp(r.zu);
    a.rho = fmax(fmax(a.rho, -s.dtl * itl), fmax(-s.dtu * itu, fmax(-s.dzl * izl, -s.dzu * izu)));
    a.s0 += r.tl * r.zl + r.tu * r.zu;
    a.s1 += r.tl * s.dzl + r.zl * s.dtl + r.tu * s.dzu + r.zu * s.dtu;
    a.s2 += s.dtl * s.dzl + s.dtu * s.dzu;
    a.cnt += 2.0;
    a.res = fmax(a.res, fmax(fabs(r.g - lo - r.tl), fabs(up - r.g - r.tu)));
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
    const double iS = rcp(S);
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

enum Mode { MODE_INIT = 0, MODE_IPM = 1, MODE_GUESS = 2, MODE_SET = 3 };

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
    int ipm_iters, set_rounds, fac;

    PQP_HD Solver(const Args& a_, int qp_, WS ws_) : a(a_), ws(ws_), qp(qp_) {}

    PQP_HD Stage load_stage(int i) const {
        Stage s;
        s.m00 = ws.ld(F_M00, i); s.m01 = ws.ld(F_M01, i); s.m10 = ws.ld(F_M10, i); s.m11 = ws.ld(F_M11, i); s.m12 = ws.ld(F_M12, i);
        s.c0 = ws.ld(F_C0, i); s.c1 = ws.ld(F_C1, i); s.ds = ws.ld(F_DS, i);
        return s;
    }

    // ---- stage data of a pass: the transition rows around the linearisation point (base_solver.cpp:165-186) -----------------------
    // src 0: (0, 0, k_ref) (path_optimizer.cpp:128-137), 1: a.lin, 2: the previous pass's optimum (F_X*)
    PQP_HD void lin_at(int src, int i, double& l, double& psi, double& k) const {
        if (src == 2) { l = ws.ld(F_X0, i); psi = ws.ld(F_X1, i); k = ws.ld(F_X2, i); }
        else if (src == 1) { const double* p = a.lin + ((size_t)qp * a.n + i) * PQP_LIN_STRIDE; l = p[0]; psi = p[1]; k = p[2]; }
        else { l = 0.0; psi = 0.0; k = a.ref[((size_t)qp * a.n + i) * PQP_REF_STRIDE + 1]; }
    }
    PQP_HD void prep(int src, bool with_bounds) {
        const double* rq = a.ref + (size_t)qp * a.n * PQP_REF_STRIDE;
        const double* bq = a.bounds + (size_t)qp * a.n * PQP_BOUNDS_STRIDE;
        double l, psi, k;
        lin_at(src, 0, l, psi, k);
        for (int i = 0; i < n; ++i) {
            if (i < n - 1) {
                double ln, pn, kn;
                lin_at(src, i + 1, ln, pn, kn);
                const double t = tan(psi), cs = cos(psi);
                const double df00 = -k * t, df01 = (1 - k * l) / (cs * cs);
                const double df10 = -k * k / cs, df11 = (1 - k * l) * k * t / cs, df12 = (1 - k * l) / cs;
                const double ds = rq[PQP_REF_STRIDE * (i + 1)] - rq[PQP_REF_STRIDE * i];
                const double f0 = (1 - k * l) * t, f1 = (1 - k * l) * k / cs - rq[PQP_REF_STRIDE * i + 1];
                ws.st(F_M00, i, ds * df00 + 1.0); ws.st(F_M01, i, ds * df01);
                ws.st(F_M10, i, ds * df10); ws.st(F_M11, i, ds * df11 + 1.0); ws.st(F_M12, i, ds * df12);
                ws.st(F_C0, i, ds * (f0 - (df00 * l + df01 * psi)));
                ws.st(F_C1, i, ds * (f1 - (df10 * l + df11 * psi + df12 * k)));
                ws.st(F_DS, i, ds);
                l = ln; psi = pn; k = kn;
            }
            if (with_bounds) {
                const double* b = bq + PQP_BOUNDS_STRIDE * i;
                const bool rough = a.prm.rough_constraints_far_away && !(rq[PQP_REF_STRIDE * i] < a.prm.precise_planning_length);
                double lo, up;
                if (!rough) {
                    soft_bounds(b[0], b[1], a.prm.expected_safety_margin, a.prm.min_clearance, lo, up);
                    ws.st(F_LOF, i, lo); ws.st(F_UPF, i, up);
                    soft_bounds(b[2], b[3], a.prm.expected_safety_margin, a.prm.min_clearance, lo, up);
                    ws.st(F_LOR, i, lo); ws.st(F_UPR, i, up);
                } else {            // base_solver.cpp:201-205,241-247: one row on l alone with the centre circle's box
                    soft_bounds(b[4], b[5], a.prm.expected_safety_margin, a.prm.min_clearance, lo, up);
                    ws.st(F_LOF, i, lo); ws.st(F_UPF, i, up);
                    ws.st(F_LOR, i, -kInfty); ws.st(F_UPR, i, kInfty);
                }
            }
        }
    }

    // ---- one backward sweep -------------------------------------------------------------------------------------------------------
    // row terms of waypoint i in the given mode; for MODE_IPM / MODE_GUESS the waypoint's interior-point state first takes the step of
    // the previous roll-out (alpha, sm_prev), which is where that state is updated
    PQP_HD void stage_cost(int mode, int i, double sm, Value& v) {
        v.P[0] += w_l; v.P[5] += w_k;
        if (i == 0) return;                                   // x_0 is given: its rows are constants
        const double lof = ws.ld(F_LOF, i), upf = ws.ld(F_UPF, i), lor = ws.ld(F_LOR, i), upr = ws.ld(F_UPR, i);
        const double L0 = lor <= -kBig ? 0.0 : Lf;            // a rough waypoint: one row on l alone
        const bool live_f = upf - lof > kEqWidth, live_r = upr < kBig && upr - lor > kEqWidth, on_r = upr < kBig;
        if (mode == MODE_INIT) {
            const double w0 = 1.0, ws0 = w_s * w0 / (w_s + w0);
            add_lpsi_term(v, ws0, L0, live_f ? 0.5 * (lof + upf) : upf);
            if (on_r) add_lpsi_term(v, ws0, Lr, live_r ? 0.5 * (lor + upr) : upr);
            v.P[5] += w0;                                     // kappa towards 0, the middle of its box
            return;
        }
        if (mode == MODE_SET) {
            const int code = (int)ws.ld(F_ACT, i);
            const int af = code % 3 - 1, ar = (code / 3) % 3 - 1, ak = code / 9 - 1;
            if (af != 0) add_lpsi_term(v, w_s, L0, af > 0 ? upf : lof);
            if (ar != 0) add_lpsi_term(v, w_s, Lr, ar > 0 ? upr : lor);
            if (ak != 0) { const double w = 1.0 / kDelta; v.P[5] += w; v.p[2] -= w * (ak * kl - kDelta * ws.ld(F_LAM, i)); }
            return;
        }
        // interior-point state of the waypoint: previous step applied, then this iteration's weights (or the set it predicts)
        Row rf, rr, rk;
        rf.g = ws.ld(F_GF, i); rf.zl = ws.ld(F_ZLF, i); rf.zu = ws.ld(F_ZUF, i); rf.tl = rf.g - lof; rf.tu = upf - rf.g;
        rr.g = ws.ld(F_GR, i); rr.zl = ws.ld(F_ZLR, i); rr.zu = ws.ld(F_ZUR, i); rr.tl = rr.g - lor; rr.tu = upr - rr.g;
        rk.g = ws.ld(F_GK, i); rk.tl = ws.ld(F_TLK, i); rk.tu = ws.ld(F_TUK, i); rk.zl = ws.ld(F_ZLK, i); rk.zu = ws.ld(F_ZUK, i);
        if (alpha > 0.0) {
            if (live_f) { row_apply(rf, row_step(rf, lof, upf, sm_prev, ws.ld(F_GPF, i)), alpha); rf.tl = rf.g - lof; rf.tu = upf - rf.g; }
            if (live_r) { row_apply(rr, row_step(rr, lor, upr, sm_prev, ws.ld(F_GPR, i)), alpha); rr.tl = rr.g - lor; rr.tu = upr - rr.g; }
            row_apply(rk, row_step(rk, -kl, kl, sm_prev, ws.ld(F_GPK, i)), alpha);
        }
        if (mode == MODE_IPM) {
            double d, tgt;
            if (live_f) { row_weight(rf, lof, upf, sm, d, tgt); add_lpsi_term(v, w_s * d * rcp(w_s + d), L0, tgt); ws.st(F_GF, i, rf.g); ws.st(F_ZLF, i, rf.zl); ws.st(F_ZUF, i, rf.zu); }
            else add_lpsi_term(v, w_s, L0, upf);
            if (live_r) { row_weight(rr, lor, upr, sm, d, tgt); add_lpsi_term(v, w_s * d * rcp(w_s + d), Lr, tgt); ws.st(F_GR, i, rr.g); ws.st(F_ZLR, i, rr.zl); ws.st(F_ZUR, i, rr.zu); }
            else if (on_r) add_lpsi_term(v, w_s, Lr, upr);
            row_weight(rk, -kl, kl, sm, d, tgt);
            v.P[5] += d; v.p[2] -= d * tgt;
            ws.st(F_GK, i, rk.g); ws.st(F_TLK, i, rk.tl); ws.st(F_TUK, i, rk.tu); ws.st(F_ZLK, i, rk.zl); ws.st(F_ZUK, i, rk.zu);
            return;
        }
        // MODE_GUESS: a side is active when its multiplier outweighs its slack
        const int af = !live_f ? 1 : (rf.zu > rf.tu ? 1 : (rf.zl > rf.tl ? -1 : 0));
        const int ar = !on_r ? 0 : (!live_r ? 1 : (rr.zu > rr.tu ? 1 : (rr.zl > rr.tl ? -1 : 0)));
        const int ak = rk.zu > rk.tu ? 1 : (rk.zl > rk.tl ? -1 : 0);
        const double lam = ak > 0 ? rk.zu : (ak < 0 ? -rk.zl : 0.0);
        ws.st(F_ACT, i, (double)((af + 1) + 3 * (ar + 1) + 9 * (ak + 1)));
        ws.st(F_LAM, i, lam);
        if (af != 0) add_lpsi_term(v, w_s, L0, af > 0 ? upf : lof);
        if (ar != 0) add_lpsi_term(v, w_s, Lr, ar > 0 ? upr : lor);
        if (ak != 0) { const double w = 1.0 / kDelta; v.P[5] += w; v.p[2] -= w * (ak * kl - kDelta * lam); }
    }
    // the two end rows (base_solver.cpp:208-209,250-259), part of waypoint n - 1
    PQP_HD void end_cost(int mode, double sm, Value& v) {
        const bool has_ep = psi_hi < kBig;
        const double L = a.prm.end_l_bound;
        if (mode == MODE_INIT) {
            v.P[0] += 1.0;
            if (has_ep) { v.P[3] += 1.0; v.p[1] -= 0.5 * (psi_lo + psi_hi); }
            return;
        }
        if (mode == MODE_IPM || mode == MODE_GUESS) {
            if (alpha > 0.0) {
                row_apply(el, row_step(el, -L, L, sm_prev, gp_el), alpha);
                if (has_ep) row_apply(ep, row_step(ep, psi_lo, psi_hi, sm_prev, gp_ep), alpha);
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
        const double w = 1.0 / kDelta;
        if (act_el != 0) { v.P[0] += w; v.p[0] -= w * (act_el * L - kDelta * lam_el); }
        if (act_ep != 0) { v.P[3] += w; v.p[1] -= w * ((act_ep > 0 ? psi_hi : psi_lo) - kDelta * lam_ep); }
    }
    double gp_el, gp_ep;          // end-row values of the last interior-point roll-out

    PQP_HD void backward(int mode, double sm) {
        Value v;
        for (int k = 0; k < 6; ++k) v.P[k] = 0.0;
        v.p[0] = v.p[1] = v.p[2] = 0.0;
        stage_cost(mode, n - 1, sm, v);
        end_cost(mode, sm, v);
        for (int i = n - 2; i >= 0; --i) {
            const Stage s = load_stage(i);
            double K[3], kk;
            riccati_step(s, w_u, v, K, kk);
            ws.st(F_K0, i, K[0]); ws.st(F_K1, i, K[1]); ws.st(F_K2, i, K[2]); ws.st(F_KK, i, kk);
            if (i > 0) stage_cost(mode, i, sm, v);
        }
        fac += 1;
    }

    // ---- roll-outs ------------------------------------------------------------------------------------------------------------------
    PQP_HD void advance(int i, double* x) const {          // x_i -> x_{i+1}
        const Stage s = load_stage(i);
        const double u = -(ws.ld(F_K0, i) * x[0] + ws.ld(F_K1, i) * x[1] + ws.ld(F_K2, i) * x[2]) - ws.ld(F_KK, i);
        const double y0 = s.m00 * x[0] + s.m01 * x[1] + s.c0;
        const double y1 = s.m10 * x[0] + s.m11 * x[1] + s.m12 * x[2] + s.c1;
        x[2] = x[2] + s.ds * u; x[0] = y0; x[1] = y1;
    }
    // after the initial solve: the interior-point state of every row, strictly inside its box where the row has a slack
    PQP_HD void forward_init() {
        const double theta = 0.05, mu0 = 0.1;
        double x[3] = {x0[0], x0[1], x0[2]};
        Acc acc = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
        auto start = [&](double v, double lo, double up, bool slack) {
            Row r;
            const double wd = up - lo;
            r.g = slack ? fmin(fmax(v, lo + theta * wd), up - theta * wd) : v;
            r.tl = fmax(r.g - lo, theta * wd); r.tu = fmax(up - r.g, theta * wd);
            r.zl = mu0 * rcp(r.tl); r.zu = mu0 * rcp(r.tu);
            acc.s0 += r.tl * r.zl + r.tu * r.zu; acc.cnt += 2.0;
            acc.res = fmax(acc.res, fmax(fabs(r.g - lo - r.tl), fabs(up - r.g - r.tu)));
            return r;
        };
        for (int i = 0; i < n - 1; ++i) {
            advance(i, x);
            const int j = i + 1;
            const double lof = ws.ld(F_LOF, j), upf = ws.ld(F_UPF, j), lor = ws.ld(F_LOR, j), upr = ws.ld(F_UPR, j);
            const double L0 = lor <= -kBig ? 0.0 : Lf;
            if (upf - lof > kEqWidth) { const Row r = start(x[0] + L0 * x[1], lof, upf, true); ws.st(F_GF, j, r.g); ws.st(F_ZLF, j, r.zl); ws.st(F_ZUF, j, r.zu); }
            if (upr < kBig && upr - lor > kEqWidth) { const Row r = start(x[0] + Lr * x[1], lor, upr, true); ws.st(F_GR, j, r.g); ws.st(F_ZLR, j, r.zl); ws.st(F_ZUR, j, r.zu); }
            const Row r = start(x[2], -kl, kl, false);
            ws.st(F_GK, j, r.g); ws.st(F_TLK, j, r.tl); ws.st(F_TUK, j, r.tu); ws.st(F_ZLK, j, r.zl); ws.st(F_ZUK, j, r.zu);
        }
        el = start(x[0], -a.prm.end_l_bound, a.prm.end_l_bound, false);
        if (psi_hi < kBig) ep = start(x[1], psi_lo, psi_hi, false);
        mu = acc.s0 / acc.cnt; res = acc.res; alpha = 0.0;
    }
    // a re-linearised pass: the interior-point state out of the previous pass's optimum, its active set and multipliers
    PQP_HD void warm_init() {
        const double mu_w = kMuWarm, sq = sqrt(kMuWarm);
        Acc acc = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
        auto start = [&](double v, double y, double lo, double up, bool slack) {
            Row r;
            r.zu = fmax(y, 0.0); r.zl = fmax(-y, 0.0);
            const double tl_min = mu_w * rcp(fmax(r.zl, sq)), tu_min = mu_w * rcp(fmax(r.zu, sq));
            if (slack) { r.g = fmin(fmax(fmin(fmax(v, lo), up), lo + tl_min), up - tu_min); r.tl = r.g - lo; r.tu = up - r.g; }
            else { r.g = v; r.tl = fmax(v - lo, tl_min); r.tu = fmax(up - v, tu_min); }
            r.zl = fmax(r.zl, mu_w * rcp(fmax(r.tl, sq))); r.zu = fmax(r.zu, mu_w * rcp(fmax(r.tu, sq)));
            acc.s0 += r.tl * r.zl + r.tu * r.zu; acc.cnt += 2.0;
            acc.res = fmax(acc.res, fmax(fabs(r.g - lo - r.tl), fabs(up - r.g - r.tu)));
            return r;
        };
        for (int j = 1; j < n; ++j) {
            const double xl = ws.ld(F_X0, j), xp = ws.ld(F_X1, j), xk = ws.ld(F_X2, j);
            const int code = (int)ws.ld(F_ACT, j);
            const int af = code % 3 - 1, ar = (code / 3) % 3 - 1, ak = code / 9 - 1;
            const double lam = ws.ld(F_LAM, j);
            const double lof = ws.ld(F_LOF, j), upf = ws.ld(F_UPF, j), lor = ws.ld(F_LOR, j), upr = ws.ld(F_UPR, j);
            const double L0 = lor <= -kBig ? 0.0 : Lf;
            // (the three reads above alias what is written below: all reads of waypoint j come first)
            Row rf, rr, rk;
            const bool live_f = upf - lof > kEqWidth, live_r = upr < kBig && upr - lor > kEqWidth;
            if (live_f) { const double v = xl + L0 * xp; rf = start(v, af > 0 ? w_s * (v - upf) : (af < 0 ? w_s * (v - lof) : 0.0), lof, upf, true); }
            if (live_r) { const double v = xl + Lr * xp; rr = start(v, ar > 0 ? w_s * (v - upr) : (ar < 0 ? w_s * (v - lor) : 0.0), lor, upr, true); }
            rk = start(xk, ak != 0 ? lam : 0.0, -kl, kl, false);
            if (live_f) { ws.st(F_GF, j, rf.g); ws.st(F_ZLF, j, rf.zl); ws.st(F_ZUF, j, rf.zu); }
            if (live_r) { ws.st(F_GR, j, rr.g); ws.st(F_ZLR, j, rr.zl); ws.st(F_ZUR, j, rr.zu); }
            ws.st(F_GK, j, rk.g); ws.st(F_TLK, j, rk.tl); ws.st(F_TUK, j, rk.tu); ws.st(F_ZLK, j, rk.zl); ws.st(F_ZUK, j, rk.zu);
        }
        el = start(ws.ld(F_X0, n - 1), act_el != 0 ? lam_el : 0.0, -a.prm.end_l_bound, a.prm.end_l_bound, false);
        if (psi_hi < kBig) ep = start(ws.ld(F_X1, n - 1), act_ep != 0 ? lam_ep : 0.0, psi_lo, psi_hi, false);
        mu = acc.s0 / acc.cnt; res = acc.res; alpha = 0.0;
    }
    // roll-out of an interior-point iteration: row values of the Newton point, the step to the boundary, next complementarity
    PQP_HD void forward_ipm(double sm) {
        double x[3] = {x0[0], x0[1], x0[2]};
        Acc acc = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
        for (int i = 0; i < n - 1; ++i) {
            advance(i, x);
            const int j = i + 1;
            const double lof = ws.ld(F_LOF, j), upf = ws.ld(F_UPF, j), lor = ws.ld(F_LOR, j), upr = ws.ld(F_UPR, j);
            const double L0 = lor <= -kBig ? 0.0 : Lf;
            if (upf - lof > kEqWidth) {
                Row r; r.g = ws.ld(F_GF, j); r.zl = ws.ld(F_ZLF, j); r.zu = ws.ld(F_ZUF, j); r.tl = r.g - lof; r.tu = upf - r.g;
                double d, tgt;
                row_weight(r, lof, upf, sm, d, tgt);
                const double v = x[0] + L0 * x[1];
                const double gp = v - d * rcp(w_s + d) * (v - tgt);           // + the slack of the Newton point
                ws.st(F_GPF, j, gp);
                row_accumulate(r, row_step(r, lof, upf, sm, gp), lof, upf, acc);
            }
            if (upr < kBig && upr - lor > kEqWidth) {
                Row r; r.g = ws.ld(F_GR, j); r.zl = ws.ld(F_ZLR, j); r.zu = ws.ld(F_ZUR, j); r.tl = r.g - lor; r.tu = upr - r.g;
                double d, tgt;
                row_weight(r, lor, upr, sm, d, tgt);
                const double v = x[0] + Lr * x[1];
                const double gp = v - d * rcp(w_s + d) * (v - tgt);
                ws.st(F_GPR, j, gp);
                row_accumulate(r, row_step(r, lor, upr, sm, gp), lor, upr, acc);
            }
            Row r; r.g = ws.ld(F_GK, j); r.tl = ws.ld(F_TLK, j); r.tu = ws.ld(F_TUK, j); r.zl = ws.ld(F_ZLK, j); r.zu = ws.ld(F_ZUK, j);
            ws.st(F_GPK, j, x[2]);
            row_accumulate(r, row_step(r, -kl, kl, sm, x[2]), -kl, kl, acc);
        }
        gp_el = x[0]; gp_ep = x[1];
        row_accumulate(el, row_step(el, -a.prm.end_l_bound, a.prm.end_l_bound, sm, gp_el), -a.prm.end_l_bound, a.prm.end_l_bound, acc);
        if (psi_hi < kBig) row_accumulate(ep, row_step(ep, psi_lo, psi_hi, sm, gp_ep), psi_lo, psi_hi, acc);
        alpha = acc.rho > 0.995 ? 0.995 * rcp(acc.rho) : 1.0;
        sm_prev = sm;
        mu = (acc.s0 + alpha * (acc.s1 + alpha * acc.s2)) / acc.cnt;      // complementarity after the step
        res = (1.0 - alpha) * acc.res;
    }
    // roll-out of an active-set round: the point, the set it asks for, the multipliers of its hard rows.  Returns true when the point
    // confirms its set (the KKT test) and holds its hard rows.
    PQP_HD bool forward_set() {
        double x[3] = {x0[0], x0[1], x0[2]};
        bool changed = false;
        double pin = 0.0;
        ws.st(F_X0, 0, x[0]); ws.st(F_X1, 0, x[1]); ws.st(F_X2, 0, x[2]);
        auto soft = [&](int act, double v, double lo, double up) {
            // consistent within the tolerance: keep; else what the point asks for
            const bool keep = (act == 1 && v >= up - kSetTol) || (act == -1 && v <= lo + kSetTol) || (act == 0 && v <= up + kSetTol && v >= lo - kSetTol);
            return keep ? act : (v > up ? 1 : (v < lo ? -1 : 0));
        };
        auto hard = [&](int act, double& lam, double v, double lo, double up) {
            if (act != 0) {
                const double bnd = act > 0 ? up : lo;
                const double y = lam + (v - bnd) / kDelta;              // multiplier of the active row
                if (act * y < -kSetTol) { lam = 0.0; return 0; }       // wrong sign: released
                pin = fmax(pin, fabs(v - bnd));
                lam = y;
                return act;
            }
            lam = 0.0;
            return v > up + kSetTol ? 1 : (v < lo - kSetTol ? -1 : 0);
        };
        for (int i = 0; i < n - 1; ++i) {
            advance(i, x);
            const int j = i + 1;
            ws.st(F_X0, j, x[0]); ws.st(F_X1, j, x[1]); ws.st(F_X2, j, x[2]);
            const double lof = ws.ld(F_LOF, j), upf = ws.ld(F_UPF, j), lor = ws.ld(F_LOR, j), upr = ws.ld(F_UPR, j);
            const double L0 = lor <= -kBig ? 0.0 : Lf;
            const int code = (int)ws.ld(F_ACT, j);
            const int af = code % 3 - 1, ar = (code / 3) % 3 - 1, ak = code / 9 - 1;
            const int nf = upf - lof > kEqWidth ? soft(af, x[0] + L0 * x[1], lof, upf) : 1;
            const int nr = upr >= kBig ? 0 : (upr - lor > kEqWidth ? soft(ar, x[0] + Lr * x[1], lor, upr) : 1);
            double lam = ws.ld(F_LAM, j);
            const int nk = hard(ak, lam, x[2], -kl, kl);
            changed = changed || nf != af || nr != ar || nk != ak;
            ws.st(F_ACT, j, (double)((nf + 1) + 3 * (nr + 1) + 9 * (nk + 1)));
            ws.st(F_LAM, j, lam);
        }
        const int ne = hard(act_el, lam_el, x[0], -a.prm.end_l_bound, a.prm.end_l_bound);
        changed = changed || ne != act_el; act_el = ne;
        if (psi_hi < kBig) { const int np = hard(act_ep, lam_ep, x[1], psi_lo, psi_hi); changed = changed || np != act_ep; act_ep = np; }
        set_rounds += 1;
        return !changed && pin <= kPinTol;
    }

    // ---- one pass: interior-point rounds to complementarity mu_stop, then active-set rounds until the set is confirmed ------------
    PQP_HD bool solve_pass(bool warm) {
        for (int attempt = 0; attempt < 2; ++attempt) {
            const double mu_stop = attempt == 0 ? kMuStop : 1e-9;
            if (attempt > 0 || !warm) { backward(MODE_INIT, 0.0); forward_init(); }
            else warm_init();
            bool first = true;
            int it = 0;
            while (!(mu < mu_stop && res < 1e-6) && it < kIpmMaxIter) {
                const double sigma = (first || alpha <= 0.9) ? 0.2 : 0.05;
                const double sm = sigma * mu;
                backward(MODE_IPM, sm);
                forward_ipm(sm);
                first = false;
                it += 1;
            }
            ipm_iters += it;
            if (!(mu == mu)) return false;                    // NaN: numerical failure
            backward(MODE_GUESS, 0.0);
            for (int r = 0; r < kPolishMaxRounds; ++r) {
                if (forward_set()) return true;
                backward(MODE_SET, 0.0);
            }
        }
        return false;
    }

    // BaseSolver::getOptimizedPath (base_solver.cpp:263-288)
    PQP_HD void unpack() {
        const double* rq = a.ref + (size_t)qp * a.n * PQP_REF_STRIDE;
        double* oq = a.out + (size_t)qp * a.n * PQP_OUT_STRIDE;
        for (int i = 0; i < n; ++i) {
            const double l = ws.ld(F_X0, i), dpsi = ws.ld(F_X1, i), k = ws.ld(F_X2, i);
            double dk = 0.0;
            if (i < n - 1) dk = -(ws.ld(F_K0, i) * l + ws.ld(F_K1, i) * dpsi + ws.ld(F_K2, i) * k) - ws.ld(F_KK, i);
            const double angle = rq[PQP_REF_STRIDE * i + 2];
            const double new_angle = constrain_angle(angle + kPi2);
            double* o = oq + PQP_OUT_STRIDE * i;
            o[0] = rq[PQP_REF_STRIDE * i + 3] + l * cos(new_angle);
            o[1] = rq[PQP_REF_STRIDE * i + 4] + l * sin(new_angle);
            o[2] = constrain_angle(angle + dpsi);
            o[3] = l; o[4] = dpsi; o[5] = k; o[6] = dk;
        }
    }

    PQP_HD void finish(int status, int solved_passes) {
        if (a.status) a.status[qp] = status;
        if (a.iters) a.iters[qp] = ipm_iters;
        if (a.info) {
            double* f = a.info + (size_t)qp * PQP_INFO_STRIDE;
            f[0] = res; f[1] = mu; f[2] = 0.0; f[3] = (double)ipm_iters; f[4] = (double)solved_passes; f[5] = (double)(fac); f[6] = (double)fac;
            f[7] = (double)set_rounds;
        }
    }

    PQP_HD void run() {
        n = a.n_of ? a.n_of[qp] : a.n;
        ipm_iters = 0; set_rounds = 0; fac = 0; mu = 0.0; res = 0.0; alpha = 0.0; sm_prev = 0.0;
        if (n > a.n) n = a.n;
        if (n < 2) { finish(PQP_STATUS_UNSOLVED, 0); return; }
        const pqp_params& p = a.prm;
        Lf = p.front_length; Lr = p.rear_length; w_l = p.weight_l; w_k = p.weight_kappa; w_u = p.weight_dkappa; w_s = p.weight_slack;
        const double* sc = a.scal + (size_t)qp * PQP_SCAL_STRIDE;
        x0[0] = sc[0]; x0[1] = sc[1]; x0[2] = sc[2];                      // base_solver.cpp:216-220
        kl = tan(sc[5]) / p.wheel_base;                                   // :226
        psi_lo = -kInfty; psi_hi = kInfty;
        if (p.constraint_end_heading && sc[4] == 0.0) {                   // :254-258
            const double end_psi = constrain_angle(sc[3] - a.ref[((size_t)qp * a.n + n - 1) * PQP_REF_STRIDE + 2]);
            if (end_psi < p.end_psi_max) { psi_lo = end_psi - p.end_psi_tol; psi_hi = end_psi + p.end_psi_tol; }
        }
        act_el = act_ep = 0; lam_el = lam_ep = 0.0; gp_el = gp_ep = 0.0;
        prep(a.lin ? 1 : 0, true);
        if (!(fabs(x0[2]) <= kl)) {
            // the start curvature violates its own box (kappa row 0 against the fixed x_0): no point satisfies the rows
            for (int i = 0; i < n; ++i) { ws.st(F_X0, i, 0.0); ws.st(F_X1, i, 0.0); ws.st(F_X2, i, 0.0); ws.st(F_K0, i, 0.0); ws.st(F_K1, i, 0.0); ws.st(F_K2, i, 0.0); ws.st(F_KK, i, 0.0); }
            unpack();
            finish(PQP_STATUS_PRIMAL_INFEASIBLE, 0);
            return;
        }
        int solved = 0;
        bool ok = solve_pass(false);
        if (ok) solved += 1;
        for (int pass = 0; ok && pass < a.passes; ++pass) {
            prep(2, false);
            ok = solve_pass(true);
            if (ok) solved += 1;
        }
        unpack();
        finish(ok ? PQP_STATUS_SOLVED : (mu == mu ? PQP_STATUS_MAX_ITER : PQP_STATUS_NUMERICAL), solved);
    }
};

// workspace views
struct StridedWs {
    double* p;
    size_t stride;          // doubles between consecutive (waypoint, field) elements of this lane
    PQP_HD double ld(int f, int i) const { return p[((size_t)i * kFields + f) * stride]; }
    PQP_HD void st(int f, int i, double v) const { p[((size_t)i * kFields + f) * stride] = v; }
};

}  // namespace lq
}  // namespace pqp
