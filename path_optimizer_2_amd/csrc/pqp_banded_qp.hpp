// pqp_banded_qp.hpp — generic banded-QP ADMM core for the reference-line smoothing QPs:
//   S1 TensionSmoother2::osqpSmooth   src/reference_path_smoother/tension_smoother_2.cpp:20-158   (default smoother)
//   S2 TensionSmoother::osqpSmooth    src/reference_path_smoother/tension_smoother.cpp:49-177
//   S3 ReferencePathSmoother::postSmooth  src/reference_path_smoother/reference_path_smoother.cpp:526-636
// Each is  min 1/2 x'Px + q'x  s.t.  l <= Ax <= u  with a banded P and rows that touch two neighbouring points; the
// assemble kernels emit the variables point-interleaved, which makes the reduced KKT matrix S = P + Sigma + A'RA banded
// (half-bandwidth 4 / 9 / 3).  This core is the OSQP-paper ADMM (the same iteration as pqp_path_lane.hpp: unscaled
// coordinates, Ruiz metrics, adaptive rho, KKT-verified polish) on that band:
//   one workgroup (one wavefront) per QP, band factor + all vectors resident in LDS, HBM touched to read the QP and
//   write the solution.  The band Cholesky / triangular solves are sequential in the variable index and parallel over
//   the band; everything else is parallel over rows / columns.
//
// QP data layout (per QP unless marked shared; doubles; sparsity shared by the whole batch):
//   pband [pbw+1][nv]   P[j+d][j] at [d][j]                      q [nv]   l, u [nc]
//   acol  [nc][RMAX] shared   column of entry s of row r (-1: none)      aval [nc][RMAX]
//   trow  [nv][CMAX] shared   row of the c-th entry of column j (-1)      tslot [nv][CMAX] shared   its slot in that row
// The same source runs on the host under tests/emu (test infrastructure only).
#pragma once
#include "pqp_path_lane.hpp"

namespace pqp {

constexpr int kRMax = 4;   // entries per row
constexpr int kCMax = 6;   // entries per column

struct BandedQpArgs {
    int batch, nv, nc, bw, pbw;
    const double* pband;     // [batch][pbw+1][nv]
    const double* q;         // [batch][nv]
    const int* acol;         // [nc][kRMax]
    const double* aval;      // [batch][nc][kRMax]
    const int* trow;         // [nv][kCMax]
    const int* tslot;        // [nv][kCMax]
    const double* lo;        // [batch][nc]
    const double* up;        // [batch][nc]
    double* x;               // [batch][nv]   solution (interleaved variable order)
    double* y;               // [batch][nc]
    int32_t* status;         // [batch]
    int32_t* iters;          // [batch]
    double* info;            // [batch][PQP_INFO_STRIDE] or nullptr
    pqp_params prm;
};

// LDS layout (doubles)
struct BqLayout {
    int nv, nc, bw;
    PQP_HD int band() const { return 0; }                          // [bw+1][nv] factor / S
    PQP_HD int x() const { return (bw + 1) * nv; }
    PQP_HD int xt() const { return x() + nv; }
    PQP_HD int rhs() const { return xt() + nv; }
    PQP_HD int sig() const { return rhs() + nv; }                  // sigma / (c D^2)
    PQP_HD int dsc() const { return sig() + nv; }                  // D
    PQP_HD int xs() const { return dsc() + nv; }                   // polish: saved x
    PQP_HD int z() const { return xs() + nv; }
    PQP_HD int y() const { return z() + nc; }
    PQP_HD int zt() const { return y() + nc; }
    PQP_HD int rv() const { return zt() + nc; }                    // rho vector
    PQP_HD int e2() const { return rv() + nc; }                    // class * E^2 / c   (<0: free row, |.| absolute rho)
    PQP_HD int esc() const { return e2() + nc; }                   // E
    PQP_HD int lo() const { return esc() + nc; }
    PQP_HD int up() const { return lo() + nc; }
    PQP_HD int act() const { return up() + nc; }                   // polish: -1 / 0 / +1
    PQP_HD int zs() const { return act() + nc; }                   // polish: saved z
    PQP_HD int ys() const { return zs() + nc; }                    // polish: saved y
    PQP_HD int red() const { return ys() + nc; }                   // [64] scratch
    PQP_HD int total() const { return red() + 64; }
};

// Ctx: T(), sh(), phase(f(t)), reduce_max/sum<K>(out, f(t, v[K]))
template <class Ctx>
struct BandedQp {
    Ctx& ctx;
    const BandedQpArgs& A;
    const int qp, nv, nc, bw, T;
    const BqLayout L;
    double* const sh;
    double rho, cscale, alpha_;
    bool polishing_;
    int kkt_solves_, factors_;

    PQP_HD BandedQp(Ctx& c, const BandedQpArgs& a, int q_)
        : ctx(c), A(a), qp(q_), nv(a.nv), nc(a.nc), bw(a.bw), T(c.T()), L{a.nv, a.nc, a.bw}, sh(c.sh()), rho(a.prm.rho), cscale(1.0),
          alpha_(a.prm.alpha), polishing_(false), kkt_solves_(0), factors_(0) {}

    PQP_HD const double* aval() const { return A.aval + (size_t)qp * nc * kRMax; }
    PQP_HD const double* pband() const { return A.pband + (size_t)qp * (A.pbw + 1) * nv; }
    PQP_HD double pdiag(int j) const { return pband()[j]; }

    template <class F> PQP_HD void rows(F f) { ctx.phase([&](int t) { for (int r = t; r < nc; r += T) f(r); }); }
    template <class F> PQP_HD void cols(F f) { ctx.phase([&](int t) { for (int j = t; j < nv; j += T) f(j); }); }

    PQP_HD double row_dot(int r, const double* v) const {      // (A v)_r
        const double* av = aval() + (size_t)r * kRMax;
        const int* ac = A.acol + (size_t)r * kRMax;
        double s = 0.0;
        for (int k = 0; k < kRMax; ++k) { const int c = ac[k]; if (c >= 0) s += av[k] * v[c]; }
        return s;
    }
    PQP_HD double col_dot(int j, const double* w) const {      // (A' w)_j
        const int* tr = A.trow + (size_t)j * kCMax;
        const int* ts = A.tslot + (size_t)j * kCMax;
        double s = 0.0;
        for (int k = 0; k < kCMax; ++k) { const int r = tr[k]; if (r >= 0) s += aval()[(size_t)r * kRMax + ts[k]] * w[r]; }
        return s;
    }
    PQP_HD double p_times(int j, const double* v) const {      // (P v)_j, symmetric band
        const double* pb = pband();
        double s = pb[j] * v[j];
        for (int d = 1; d <= A.pbw; ++d) {
            if (j + d < nv) s += pb[(size_t)d * nv + j] * v[j + d];
            if (j - d >= 0) s += pb[(size_t)d * nv + j - d] * v[j - d];
        }
        return s;
    }

    // ---- setup: load, Ruiz (paper Alg. 2 in metric form), rho vector ---------------------------------------------
    PQP_HD void load() {
        const double* lo = A.lo + (size_t)qp * nc;
        const double* up = A.up + (size_t)qp * nc;
        rows([&](int r) {
            sh[L.lo() + r] = fmax(lo[r], -kInfty); sh[L.up() + r] = fmin(up[r], kInfty);
            sh[L.z() + r] = 0.0; sh[L.y() + r] = 0.0; sh[L.esc() + r] = 1.0; sh[L.act() + r] = 0.0;
        });
        cols([&](int j) { sh[L.x() + j] = 0.0; sh[L.dsc() + j] = 1.0; });
    }

    PQP_HD void ruiz() {
        const pqp_params& prm = A.prm;
        const double* qv = A.q + (size_t)qp * nv;
        cscale = 1.0;
        double* D = sh + L.dsc();
        double* E = sh + L.esc();
        double* dn = sh + L.rhs();     // scratch: new D
        double* en = sh + L.zt();      // scratch: new E
        for (int pass = 0; pass < prm.scaling; ++pass) {
            const double c_now = cscale;
            cols([&](int j) {
                const double* pb = pband();
                double m = fabs(pb[j]) * c_now * D[j] * D[j];
                for (int d = 1; d <= A.pbw; ++d) {
                    if (j + d < nv) m = fmax(m, fabs(pb[(size_t)d * nv + j]) * c_now * D[j] * D[j + d]);
                    if (j - d >= 0) m = fmax(m, fabs(pb[(size_t)d * nv + j - d]) * c_now * D[j] * D[j - d]);
                }
                const int* tr = A.trow + (size_t)j * kCMax;
                const int* ts = A.tslot + (size_t)j * kCMax;
                for (int k = 0; k < kCMax; ++k) {
                    const int r = tr[k];
                    if (r >= 0) m = fmax(m, fabs(aval()[(size_t)r * kRMax + ts[k]]) * E[r] * D[j]);
                }
                dn[j] = D[j] * rsq(limit_scaling(m));
            });
            rows([&](int r) {
                const double* av = aval() + (size_t)r * kRMax;
                const int* ac = A.acol + (size_t)r * kRMax;
                double m = 0.0;
                for (int k = 0; k < kRMax; ++k) { const int c = ac[k]; if (c >= 0) m = fmax(m, fabs(av[k]) * D[c] * E[r]); }
                en[r] = E[r] * rsq(limit_scaling(m));
            });
            cols([&](int j) { D[j] = dn[j]; });
            rows([&](int r) { E[r] = en[r]; });
            // cost scaling: c <- c / max(mean_j ||P_j||_inf, ||q||_inf)   (both after the D update, limited)
            double acc0[1], acc1[1];
            ctx.template reduce_sum<1>(acc0, [&](int t, double (&v)[1]) {
                const double* pb = pband();
                double s = 0.0;
                for (int j = t; j < nv; j += T) {
                    double m = fabs(pb[j]) * D[j] * D[j];
                    for (int d = 1; d <= A.pbw; ++d) {
                        if (j + d < nv) m = fmax(m, fabs(pb[(size_t)d * nv + j]) * D[j] * D[j + d]);
                        if (j - d >= 0) m = fmax(m, fabs(pb[(size_t)d * nv + j - d]) * D[j] * D[j - d]);
                    }
                    s += m * c_now;
                }
                v[0] = s;
            });
            ctx.template reduce_max<1>(acc1, [&](int t, double (&v)[1]) {
                double m = 0.0;
                for (int j = t; j < nv; j += T) m = fmax(m, fabs(qv[j]) * D[j] * c_now);
                v[0] = m;
            });
            double qn = acc1[0];
            qn = qn < kMinScaling ? 1.0 : fmin(qn, kMaxScaling);
            double ct = fmax(acc0[0] / (double)nv, qn);
            ct = limit_scaling(ct);
            cscale = cscale / ct;
        }
        const double c = cscale;
        cols([&](int j) { sh[L.sig() + j] = prm.sigma / (c * D[j] * D[j]); });
        rows([&](int r) {
            const double e = E[r], e2 = e * e / c;
            const double sl = e * sh[L.lo() + r], su = e * sh[L.up() + r];
            double v;
            if (sl < -kInfty * kMinScaling && su > kInfty * kMinScaling) v = -kRhoMin * e2;
            else if (su - sl < kRhoTol) v = kRhoEqFactor * e2;
            else v = e2;
            sh[L.e2() + r] = v;
        });
        set_rho();
    }
    PQP_HD void set_rho() {
        const double rho_now = rho;
        rows([&](int r) { const double b = sh[L.e2() + r]; sh[L.rv() + r] = b < 0.0 ? -b : rho_now * b; });
    }

    // ---- factorisation: S = P + Sigma + A' R A in band storage, banded Cholesky in place --------------------------
    PQP_HD void factor() {
        factors_ += 1;
        double* B = sh + L.band();
        const double* rv = sh + L.rv();
        cols([&](int j) {      // column j of the lower band of S; deterministic: every lane owns whole columns
            const double* pb = pband();
            for (int d = 0; d <= bw; ++d) B[(size_t)d * nv + j] = (d <= A.pbw) ? pb[(size_t)d * nv + j] : 0.0;
            B[j] += sh[L.sig() + j];
            const int* tr = A.trow + (size_t)j * kCMax;
            const int* ts = A.tslot + (size_t)j * kCMax;
            for (int k = 0; k < kCMax; ++k) {
                const int r = tr[k];
                if (r < 0) continue;
                const double* av = aval() + (size_t)r * kRMax;
                const int* ac = A.acol + (size_t)r * kRMax;
                const double v = rv[r] * av[ts[k]];
                for (int s = 0; s < kRMax; ++s) {
                    const int c2 = ac[s];
                    if (c2 >= j) B[(size_t)(c2 - j) * nv + j] += v * av[s];
                }
            }
        });
        // right-looking banded Cholesky: column j is finished, then the trailing (bw x bw) window is updated in parallel
        for (int j = 0; j < nv; ++j) {
            ctx.phase([&](int t) { if (t == 0) B[j] = sqrt(B[j]); });
            ctx.phase([&](int t) { if (t >= 1 && t <= bw && j + t < nv) B[(size_t)t * nv + j] *= rcp(B[j]); });   // lane d scales L[j+d][j]
            ctx.phase([&](int t) {
                // pairs (d1 >= d2 >= 1): S[j+d1][j+d2] -= L[j+d1][j] * L[j+d2][j]
                const int npair = bw * (bw + 1) / 2;
                for (int p = t; p < npair; p += T) {
                    int d2 = 1, rem = p;
                    while (rem >= bw - d2 + 1) { rem -= bw - d2 + 1; ++d2; }
                    const int d1 = d2 + rem;
                    if (j + d1 < nv) B[(size_t)(d1 - d2) * nv + j + d2] -= B[(size_t)d1 * nv + j] * B[(size_t)d2 * nv + j];
                }
            });
        }
    }

    // solve S v = b in place on sh[rhs]
    PQP_HD void band_solve() {
        const double* B = sh + L.band();
        double* b = sh + L.rhs();
        for (int j = 0; j < nv; ++j) {       // forward, column oriented
            ctx.phase([&](int t) { if (t == 0) b[j] = b[j] / B[j]; });
            ctx.phase([&](int t) { if (t >= 1 && t <= bw && j + t < nv) b[j + t] -= B[(size_t)t * nv + j] * b[j]; });
        }
        for (int j = nv - 1; j >= 0; --j) {  // backward: row oriented dot product over the band, reduced by lane 0
            ctx.phase([&](int t) {
                if (t == 0) {
                    double s = b[j];
                    for (int d = 1; d <= bw; ++d) if (j + d < nv) s -= B[(size_t)d * nv + j] * b[j + d];
                    b[j] = s / B[j];
                }
            });
        }
    }

    // ---- one ADMM iteration ------------------------------------------------------------------------------------------
    PQP_HD void iterate() {
        kkt_solves_ += 1;
        const double alpha = alpha_;
        const double* qv = A.q + (size_t)qp * nv;
        double* x = sh + L.x(); double* z = sh + L.z(); double* y = sh + L.y();
        double* zt = sh + L.zt(); double* xt = sh + L.xt(); double* b = sh + L.rhs();
        const double* rv = sh + L.rv();
        rows([&](int r) { zt[r] = rv[r] * z[r] - y[r]; });
        cols([&](int j) { b[j] = sh[L.sig() + j] * x[j] - qv[j] + col_dot(j, zt); });
        band_solve();
        cols([&](int j) { xt[j] = b[j]; });
        rows([&](int r) {
            const double ztr = row_dot(r, xt);
            const double zh = alpha * ztr + (1.0 - alpha) * z[r];
            double lo = sh[L.lo() + r], up = sh[L.up() + r];
            if (polishing_) {
                const double a = sh[L.act() + r];
                const double bnd = a < 0.0 ? lo : up;
                lo = a != 0.0 ? bnd : -kInfty; up = a != 0.0 ? bnd : kInfty;
            }
            const double rr = rv[r];
            const double v = zh + (rr > 0.0 ? y[r] * rcp(rr) : 0.0);
            const double zn = fmin(fmax(v, lo), up);
            y[r] += rr * (zh - zn);
            z[r] = zn;
        });
        cols([&](int j) { x[j] = alpha * xt[j] + (1.0 - alpha) * x[j]; });
    }

    // residuals as OSQP tests them (unscaled inf-norms); res[4] != 0: non-finite iterate
    PQP_HD void residuals(double (&res)[5]) {
        const double* qv = A.q + (size_t)qp * nv;
        const double* x = sh + L.x(); const double* z = sh + L.z(); const double* y = sh + L.y();
        double a[2], b[3];
        ctx.template reduce_max<2>(a, [&](int t, double (&v)[2]) {
            v[0] = 0.0; v[1] = 0.0;
            for (int r = t; r < nc; r += T) {
                const double ax = row_dot(r, x);
                v[0] = fmax(v[0], fabs(ax - z[r]));
                v[1] = fmax(v[1], fmax(fabs(ax), fabs(z[r])));
            }
        });
        ctx.template reduce_max<3>(b, [&](int t, double (&v)[3]) {
            v[0] = 0.0; v[1] = 0.0; v[2] = 0.0;
            for (int j = t; j < nv; j += T) {
                const double px = p_times(j, x), aty = col_dot(j, y);
                v[0] = fmax(v[0], fabs(px + qv[j] + aty));
                v[1] = fmax(v[1], fmax(fmax(fabs(px), fabs(aty)), fabs(qv[j])));
                if (!(fabs(x[j]) <= 1e300)) v[2] = 1.0;
            }
        });
        res[0] = a[0]; res[2] = a[1]; res[1] = b[0]; res[3] = b[1]; res[4] = b[2];
    }

    // ---- polish (same scheme as pqp_path_lane.hpp) -------------------------------------------------------------
    PQP_HD void polish_begin() {
        const pqp_params& prm = A.prm;
        const double gain = 1.0 / prm.polish_delta, sgain = prm.polish_delta / prm.sigma;
        rows([&](int r) {
            sh[L.zs() + r] = sh[L.z() + r]; sh[L.ys() + r] = sh[L.y() + r];
            const double b = sh[L.e2() + r];
            const bool fr = b < 0.0;
            const double e = sh[L.esc() + r], e2 = e * e / cscale;
            const double z = sh[L.z() + r], y = sh[L.y() + r];
            const bool alo = !fr && ((z - sh[L.lo() + r]) * e2 < -y);
            const bool aup = !fr && !alo && ((sh[L.up() + r] - z) * e2 < y);
            sh[L.act() + r] = alo ? -1.0 : (aup ? 1.0 : 0.0);
        });
        cols([&](int j) { sh[L.xs() + j] = sh[L.x() + j]; sh[L.sig() + j] *= sgain; });
        (void)gain;
    }
    PQP_HD void polish_apply_set() {
        const double gain = 1.0 / A.prm.polish_delta;
        rows([&](int r) {
            const double a = sh[L.act() + r];
            const double e = sh[L.esc() + r], e2 = e * e / cscale;
            sh[L.rv() + r] = a != 0.0 ? gain * e2 : 0.0;
            if (a == 0.0) sh[L.y() + r] = 0.0;
            else sh[L.z() + r] = a < 0.0 ? sh[L.lo() + r] : sh[L.up() + r];
        });
    }
    PQP_HD double row_violation(int r, double ax) const {
        if (sh[L.e2() + r] < 0.0) return 0.0;
        const double a = sh[L.act() + r], y = sh[L.y() + r];
        const double pv = fmax(sh[L.lo() + r] - ax, ax - sh[L.up() + r]);
        const double dv = a < 0.0 ? y : (a > 0.0 ? -y : 0.0);
        return fmax(fmax(pv, dv), 0.0);
    }
    PQP_HD double polish_violation() {
        double v1[1];
        const double* x = sh + L.x();
        ctx.template reduce_max<1>(v1, [&](int t, double (&v)[1]) {
            v[0] = 0.0;
            for (int r = t; r < nc; r += T) v[0] = fmax(v[0], row_violation(r, row_dot(r, x)));
        });
        return v1[0];
    }
    PQP_HD void polish_update_set(double thr) {
        const double* x = sh + L.x();
        rows([&](int r) {
            const double ax = row_dot(r, x);
            if (!(row_violation(r, ax) > thr)) return;
            const double a = sh[L.act() + r];
            if (a != 0.0) sh[L.act() + r] = 0.0;
            else sh[L.act() + r] = (sh[L.lo() + r] - ax > ax - sh[L.up() + r]) ? -1.0 : 1.0;
        });
    }
    PQP_HD void polish_end(bool ok) {
        const double isgain = A.prm.sigma / A.prm.polish_delta;
        rows([&](int r) {
            if (!ok) { sh[L.z() + r] = sh[L.zs() + r]; sh[L.y() + r] = sh[L.ys() + r]; }
        });
        cols([&](int j) { if (!ok) sh[L.x() + j] = sh[L.xs() + j]; sh[L.sig() + j] *= isgain; });
        set_rho();
    }

    // ---- driver ----------------------------------------------------------------------------------------------------
    PQP_HD void run() {
        const pqp_params& prm = A.prm;
        load();
        if (prm.scaling > 0) ruiz();
        else {
            cols([&](int j) { sh[L.sig() + j] = prm.sigma; });
            rows([&](int r) {
                const double sl = sh[L.lo() + r], su = sh[L.up() + r];
                sh[L.e2() + r] = (sl < -kInfty * kMinScaling && su > kInfty * kMinScaling) ? -kRhoMin : (su - sl < kRhoTol ? kRhoEqFactor : 1.0);
            });
            set_rho();
        }
        factor();
        int status = PQP_STATUS_MAX_ITER, it = 0, polished = 0;
        double res[5] = {0, 0, 0, 0, 0};
        double eps_scale = 1.0;
        int polish_gap = prm.polish_every, next_polish = prm.polish_every;
        for (it = 1; it <= prm.max_iter; ++it) {
            iterate();
            const bool check = prm.check_termination > 0 && (it % prm.check_termination) == 0;
            const bool adapt = prm.adaptive_rho && prm.adaptive_rho_interval > 0 && (it % prm.adaptive_rho_interval) == 0;
            if (!check && !adapt) continue;
            residuals(res);
            if (res[4] != 0.0) { status = PQP_STATUS_NUMERICAL; break; }
            bool start_polish = false;
            if (check) {
                const double eps_p = eps_scale * (prm.eps_abs + prm.eps_rel * res[2]);
                const double eps_d = eps_scale * (prm.eps_abs + prm.eps_rel * res[3]);
                if (res[0] <= eps_p && res[1] <= eps_d) {
                    if (!prm.polish || eps_scale * fmax(prm.eps_abs, prm.eps_rel) < 1e-10) { status = PQP_STATUS_SOLVED; break; }
                    start_polish = true;
                    eps_scale *= 0.1;
                } else if (prm.polish && prm.polish_every > 0 && it >= next_polish) {
                    start_polish = true;
                    polish_gap *= 2;
                    next_polish = it + polish_gap;
                }
            }
            if (start_polish) {
                polish_begin();
                polishing_ = true; alpha_ = 1.0;
                bool ok = false, conservative = false;
                double best = 1e300;
                int stall = 0;
                for (int round = 0; round < (prm.polish_max_rounds > 0 ? prm.polish_max_rounds : 40); ++round) {
                    polish_apply_set();
                    factor();
                    for (int k = 0; k < prm.polish_refine_iter; ++k) iterate();
                    residuals(res);
                    const double tol = prm.polish_tol;
                    // ill-conditioned Hessians (the 3rd-difference weights of S2) need more refinement: keep going while the
                    // residual is above the absolute tolerance and still shrinking
                    double prev = fmax(res[0], res[1]);
                    for (int extra = 0; extra < 24 && prev > tol; extra += 2) {
                        iterate(); iterate();
                        residuals(res);
                        const double cur = fmax(res[0], res[1]);
                        if (cur > 0.7 * prev) break;
                        prev = cur;
                    }
                    const double viol = polish_violation();
                    const bool solve_ok = res[4] == 0.0 && res[0] <= tol * (1.0 + res[2]) && res[1] <= tol * (1.0 + res[3]);
                    ok = solve_ok && viol <= tol;
                    if (ok || !solve_ok) break;
                    if (viol < 0.7 * best) { best = viol; stall = 0; } else { stall += 1; }
                    if (stall >= 3) conservative = true;
                    if (conservative && stall >= 16) break;
                    polish_update_set(conservative ? fmax(tol, 0.9 * viol) : tol);
                }
                polishing_ = false; alpha_ = prm.alpha;
                polish_end(ok);
                if (ok) { status = PQP_STATUS_SOLVED; polished = 1; break; }
                factor();
                continue;
            }
            if (adapt) {
                const double pn = res[0] / (res[2] + 1e-10), dn = res[1] / (res[3] + 1e-10);
                double rn = rho * sqrt(pn / (dn + 1e-10));
                rn = fmin(fmax(rn, kRhoMin), kRhoMax);
                if (rn > rho * prm.adaptive_rho_tolerance || rn < rho / prm.adaptive_rho_tolerance) {
                    rho = rn;
                    set_rho();
                    factor();
                }
            }
        }
        if (it > prm.max_iter) it = prm.max_iter;
        double* xo = A.x + (size_t)qp * nv;
        double* yo = A.y + (size_t)qp * nc;
        cols([&](int j) { xo[j] = sh[L.x() + j]; });
        rows([&](int r) { yo[r] = sh[L.y() + r]; });
        const int kk = kkt_solves_, ff = factors_;
        const double rho_final = rho;
        ctx.phase([&](int t) {
            if (t == 0) {
                if (A.status) A.status[qp] = status;
                if (A.iters) A.iters[qp] = it;
                if (A.info) {
                    double* f = A.info + PQP_INFO_STRIDE * (size_t)qp;
                    f[0] = res[0]; f[1] = res[1]; f[2] = rho_final; f[3] = (double)it; f[4] = (double)polished; f[5] = (double)kk; f[6] = (double)ff; f[7] = 0.0;
                }
            }
        });
    }
};

}  // namespace pqp
