// pqp_banded_qp.hpp — generic banded-QP ADMM core for the reference-line smoothing QPs:
//   S1 TensionSmoother2::osqpSmooth   src/reference_path_smoother/tension_smoother_2.cpp:20-158   (default smoother)
//   S2 TensionSmoother::osqpSmooth    src/reference_path_smoother/tension_smoother.cpp:49-177
//   S3 ReferencePathSmoother::postSmooth  src/reference_path_smoother/reference_path_smoother.cpp:526-636
// Each is  min 1/2 x'Px + q'x  s.t.  l <= Ax <= u  with a banded P and rows that touch two neighbouring points; the
// assemble kernels emit the variables point-interleaved, which makes the reduced KKT matrix S = P + Sigma + A'RA banded
// (half-bandwidth B = 4 / 9 / 3), i.e. block tridiagonal with B x B blocks.  This core is the OSQP-paper ADMM (the same
// iteration as pqp_path_lane.hpp: unscaled coordinates, Ruiz metrics, adaptive rho, KKT-verified polish) on that structure:
//   one workgroup per QP, ONE LANE PER VARIABLE (lane t = variable t = row t%B of block t/B).  S is factorised and solved by
//   block cyclic reduction over the blocks: log2(#blocks) levels instead of a chain as long as the variable count.  A lane
//   keeps the rows of its block's factor (D^-1, D^-1 S_left, D^-1 S_right and the transposed rows) in registers; vectors and
//   the row data of A live in LDS; HBM is touched to read the QP and write the solution.
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
// Ctx::kStage: the row data of A, the index lists and q are copied to LDS once per QP (BqLayout::total(true) doubles of LDS); otherwise
// they are read from global memory (L2) at every use - two dependent global round trips per ADMM iteration.

// per-lane register state: the factor rows of variable t (valid at the elimination level of its block) and the
// factorisation workspace
template <int B>
struct BqLane {
    double Dinv[B], HL[B], HR[B];    // row i of D^-1, D^-1 S_{e,left}, D^-1 S_{e,right}
    double HLt[B], HRt[B];           // column i of the latter two (rows of their transposes)
    double Dr[B], Lr[B], Rr[B], Ag[B];   // factorisation: row i of S_ee, S_{e,left}, S_{e,right}, Gauss-Jordan augment
    double r;                        // right-hand side entry / solution entry
    double b0, x0;                   // polish: right-hand side and accumulated solution during iterative refinement
};

// LDS layout (doubles).  nbb = padded variable count (#blocks x B)
struct BqLayout {
    int nv, nc, bw;
    PQP_HD int nb() const { return (nv + bw - 1) / bw; }
    PQP_HD int nbb() const { return nb() * bw; }
    PQP_HD int piv() const { return 0; }                           // [nb][4B] Gauss-Jordan pivot rows
    PQP_HD int hm() const { return piv() + nb() * 4 * bw; }        // [nb/2+1][2][B][B] D^-1 S_left / D^-1 S_right of the level's blocks
    PQP_HD int x() const { return hm() + (nb() / 2 + 1) * 2 * bw * bw; }
    PQP_HD int xt() const { return x() + nv; }                     // [nbb] solution of the reduced system
    PQP_HD int rhs() const { return xt() + nbb(); }                // [nbb] right-hand side at elimination time (and Ruiz scratch)
    PQP_HD int pl() const { return rhs() + nbb(); }                // [nbb] message of a block to its left neighbour
    PQP_HD int pr() const { return pl() + nbb(); }                 // [nbb] ... right neighbour
    PQP_HD int sig() const { return pr() + nbb(); }                // sigma / (c D^2)
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
    PQP_HD int yp() const { return ys() + nc; }                    // multipliers one iteration before a termination check
    PQP_HD int red() const { return yp() + nc; }                   // [5][16] reduction scratch
    PQP_HD int sav() const { return red() + 128; }                 // staged: aval [nc][kRMax]
    PQP_HD int sq() const { return sav() + nc * 4; }               // staged: q [nv]
    PQP_HD int sidx() const { return sq() + nv; }                  // staged (int32): acol [nc][kRMax], trow [nv][kCMax], tslot [nv][kCMax]
    PQP_HD int total(bool staged = false) const { return staged ? sidx() + (nc * 4 + nv * 12 + 1) / 2 : red() + 128; }
};

// Ctx: T(), sh(), phase(f(t, BqLane<B>&)), reduce_max/sum<K>(out, f(t, v[K]))
template <class Ctx, int B>
struct BandedQp {
    typedef BqLane<B> Lane;
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

    template <class F> PQP_HD void rows(F f) { ctx.phase([&](int t, Lane&) { for (int r = t; r < nc; r += T) f(r); }); }
    template <class F> PQP_HD void cols(F f) { ctx.phase([&](int t, Lane&) { for (int j = t; j < nv; j += T) f(j); }); }

    // staged copies (LDS) of the per-QP row values, of q and of the batch's index lists
    PQP_HD const double* s_aval() const { return sh + L.sav(); }
    PQP_HD const int* s_acol() const { return reinterpret_cast<const int*>(sh + L.sidx()); }
    PQP_HD const int* s_trow() const { return s_acol() + nc * kRMax; }
    PQP_HD const int* s_tslot() const { return s_trow() + nv * kCMax; }
    PQP_HD double q_of(int j) const { return Ctx::kStage ? sh[L.sq() + j] : A.q[(size_t)qp * nv + j]; }

    PQP_HD double row_dot(int r, const double* v) const {      // (A v)_r
        double s = 0.0;
        if constexpr (Ctx::kStage) {
            const double* av = s_aval() + r * kRMax;
            const int* ac = s_acol() + r * kRMax;
            for (int k = 0; k < kRMax; ++k) { const int c = ac[k]; if (c >= 0) s += av[k] * v[c]; }
        } else {
            const double* av = aval() + (size_t)r * kRMax;
            const int* ac = A.acol + (size_t)r * kRMax;
            for (int k = 0; k < kRMax; ++k) { const int c = ac[k]; if (c >= 0) s += av[k] * v[c]; }
        }
        return s;
    }
    PQP_HD double col_dot(int j, const double* w) const {      // (A' w)_j
        double s = 0.0;
        if constexpr (Ctx::kStage) {
            const int* tr = s_trow() + j * kCMax;
            const int* ts = s_tslot() + j * kCMax;
            const double* av = s_aval();
            for (int k = 0; k < kCMax; ++k) { const int r = tr[k]; if (r >= 0) s += av[r * kRMax + ts[k]] * w[r]; }
        } else {
            const int* tr = A.trow + (size_t)j * kCMax;
            const int* ts = A.tslot + (size_t)j * kCMax;
            for (int k = 0; k < kCMax; ++k) { const int r = tr[k]; if (r >= 0) s += aval()[(size_t)r * kRMax + ts[k]] * w[r]; }
        }
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
        if constexpr (Ctx::kStage) {
            ctx.phase([&](int t, Lane&) {
                double* av = sh + L.sav();
                int* ix = reinterpret_cast<int*>(sh + L.sidx());
                const double* ga = aval();
                for (int k = t; k < nc * kRMax; k += T) { av[k] = ga[k]; ix[k] = A.acol[k]; }
                for (int k = t; k < nv * kCMax; k += T) { ix[nc * kRMax + k] = A.trow[k]; ix[nc * kRMax + nv * kCMax + k] = A.tslot[k]; }
                for (int j = t; j < nv; j += T) sh[L.sq() + j] = A.q[(size_t)qp * nv + j];
            });
        }
    }

    PQP_HD void ruiz() {
        const pqp_params& prm = A.prm;
        const double* qv = A.q + (size_t)qp * nv;
        cscale = 1.0;
        double* D = sh + L.dsc();
        double* E = sh + L.esc();
        double* dn = sh + L.rhs();     // scratch: new D
        double* en = sh + L.zt();      // scratch: new E
        // (scaling < 0 is the path QP's one-waypoint shortcut: the smoother QPs run |scaling| ordinary passes)
        for (int pass = 0; pass < (prm.scaling < 0 ? -prm.scaling : prm.scaling); ++pass) {
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

    // ---- factorisation: S = P + Sigma + A' R A, block cyclic reduction over the B x B blocks ----------------------------
    // Block e = t / B, row i = t % B.  Tree over tp = e + 1: a block is eliminated at level h = lowest set bit of tp into the
    // blocks e - h and e + h (those that exist).  Elimination of e: Gauss-Jordan on [S_ee | S_e,left | S_e,right | I] with one
    // row per lane gives D^-1 S_e,left, D^-1 S_e,right and D^-1 row by row; the surviving neighbours then update their own
    // rows:  S_aa -= S_ae (D^-1 S_ea),  S_a,c = -S_ae (D^-1 S_ec)  (a = e - h, c = e + h) and the mirror image for c.
    PQP_HD static int level_of(int e) { const int tp = e + 1; return tp & (-tp); }
    PQP_HD void factor() {
        factors_ += 1;
        const int nb = L.nb();
        const double* rv = sh + L.rv();
        ctx.phase([&](int t, Lane& ln) {      // row t of S, split by block column: left / own / right block
            const int e = t / B;
            _Pragma("unroll") for (int k = 0; k < B; ++k) { ln.Dr[k] = 0.0; ln.Lr[k] = 0.0; ln.Rr[k] = 0.0; }
            if (t < nv) {
                const int j = t;
                const double* pb = pband();
                for (int d = -A.pbw; d <= A.pbw; ++d) {
                    const int c = j + d;
                    if (c < 0 || c >= nv) continue;
                    const double v = d >= 0 ? pb[(size_t)d * nv + j] : pb[(size_t)(-d) * nv + c];
                    add_entry(ln, e, c, v);
                }
                add_entry(ln, e, j, sh[L.sig() + j]);
                const int* tr = A.trow + (size_t)j * kCMax;
                const int* ts = A.tslot + (size_t)j * kCMax;
                for (int k = 0; k < kCMax; ++k) {
                    const int r = tr[k];
                    if (r < 0) continue;
                    const double* av = aval() + (size_t)r * kRMax;
                    const int* ac = A.acol + (size_t)r * kRMax;
                    const double v = rv[r] * av[ts[k]];
                    for (int s2 = 0; s2 < kRMax; ++s2) { const int c2 = ac[s2]; if (c2 >= 0) add_entry(ln, e, c2, v * av[s2]); }
                }
            } else {
                add_entry(ln, e, t, 1.0);     // padding variable of the last block: unit row
            }
        });
        for (int h = 1; h <= nb; h <<= 1) {
            // Gauss-Jordan, pivot k: the pivot lane publishes its row; then it normalises it and the other rows of the block
            // eliminate.  (No lane field is indexed by k under a test `i == k`: there the compiler rewrites Dr[k] as Dr[i], a
            // dynamic index into the lane struct, which would keep the whole struct in scratch memory.)
            _Pragma("unroll") for (int k = 0; k < B; ++k) {
                ctx.phase([&](int t, Lane& ln) {
                    const int e = t / B, i = t - e * B;
                    if (e < nb && level_of(e) == h) {
                        if (k == 0) { _Pragma("unroll") for (int c = 0; c < B; ++c) ln.Ag[c] = (c == i) ? 1.0 : 0.0; }
                        if (i == k) {
                            double* pv = sh + L.piv() + (size_t)e * 4 * B;
                            _Pragma("unroll") for (int c = 0; c < B; ++c) {
                                pv[c] = ln.Dr[c]; pv[B + c] = ln.Lr[c]; pv[2 * B + c] = ln.Rr[c]; pv[3 * B + c] = ln.Ag[c];
                            }
                        }
                    }
                });
                ctx.phase([&](int t, Lane& ln) {
                    const int e = t / B, i = t - e * B;
                    if (e < nb && level_of(e) == h) {
                        const double* pv = sh + L.piv() + (size_t)e * 4 * B;
                        const bool pivot = (i == k);
                        const double inv = rcp(pv[k]);
                        const double f = pivot ? 0.0 : ln.Dr[k] * inv;      // multiple of the (raw) pivot row to subtract
                        const double sc = pivot ? inv : 1.0;               // the pivot row itself is normalised
                        _Pragma("unroll") for (int c = 0; c < B; ++c) {
                            ln.Dr[c] = ln.Dr[c] * sc - f * pv[c];
                            ln.Lr[c] = ln.Lr[c] * sc - f * pv[B + c];
                            ln.Rr[c] = ln.Rr[c] * sc - f * pv[2 * B + c];
                            ln.Ag[c] = ln.Ag[c] * sc - f * pv[3 * B + c];
                        }
                    }
                });
            }
            // keep the factor rows, publish D^-1 S_left / D^-1 S_right of the eliminated blocks
            ctx.phase([&](int t, Lane& ln) {
                const int e = t / B, i = t - e * B;
                if (e < nb && level_of(e) == h) {
                    double* m = sh + L.hm() + (size_t)(e / (2 * h)) * 2 * B * B;
                    _Pragma("unroll") for (int c = 0; c < B; ++c) {
                        ln.Dinv[c] = ln.Ag[c]; ln.HL[c] = ln.Lr[c]; ln.HR[c] = ln.Rr[c];
                        m[i * B + c] = ln.Lr[c]; m[B * B + i * B + c] = ln.Rr[c];
                    }
                }
            });
            // (one unconditional store per lane field with selected values: if/else branches that end in stores to different
            //  fields are merged by LLVM into a store through a pointer phi, which pushes the lane struct into scratch memory)
            ctx.phase([&](int t, Lane& ln) {
                const int e = t / B, i = t - e * B;
                if (e >= nb) return;
                const int lv = level_of(e);
                const bool el = lv == h, sv = lv > h;
                if (el) {   // eliminated block: transposed rows for the forward sweep.  (A plain `if` without else: selecting between
                            //  an LDS load and a lane field would again become an address select.)
                    const double* m = sh + L.hm() + (size_t)(e / (2 * h)) * 2 * B * B;
                    _Pragma("unroll") for (int c = 0; c < B; ++c) { ln.HLt[c] = m[c * B + i]; ln.HRt[c] = m[B * B + c * B + i]; }
                }
                // survivor: fold the eliminated neighbours into the own row, couple to the next neighbours
                double nd[B], nl[B], nr[B];
                _Pragma("unroll") for (int c = 0; c < B; ++c) { nd[c] = ln.Dr[c]; nl[c] = 0.0; nr[c] = 0.0; }
                if (sv && e + h < nb) {     // right neighbour e + h: own Rr is row i of S_{a,e}
                    const double* m = sh + L.hm() + (size_t)((e + h) / (2 * h)) * 2 * B * B;
                    _Pragma("unroll") for (int k2 = 0; k2 < B; ++k2) {
                        const double f = ln.Rr[k2];
                        _Pragma("unroll") for (int c = 0; c < B; ++c) { nd[c] -= f * m[k2 * B + c]; nr[c] -= f * m[B * B + k2 * B + c]; }
                    }
                }
                if (sv && e - h >= 0) {     // left neighbour e - h: own Lr is row i of S_{c,e}
                    const double* m = sh + L.hm() + (size_t)((e - h) / (2 * h)) * 2 * B * B;
                    _Pragma("unroll") for (int k2 = 0; k2 < B; ++k2) {
                        const double f = ln.Lr[k2];
                        _Pragma("unroll") for (int c = 0; c < B; ++c) { nd[c] -= f * m[B * B + k2 * B + c]; nl[c] -= f * m[k2 * B + c]; }
                    }
                }
                _Pragma("unroll") for (int c = 0; c < B; ++c) {
                    ln.Dr[c] = sv ? nd[c] : ln.Dr[c]; ln.Lr[c] = sv ? nl[c] : ln.Lr[c]; ln.Rr[c] = sv ? nr[c] : ln.Rr[c];
                }
            });
        }
    }
    // S[row of lane][c] += v, routed to the left / own / right block of the row (static register indices only)
    PQP_HD static void add_entry(Lane& ln, int e, int c, double v) {
        const int base = e * B;
        _Pragma("unroll") for (int k = 0; k < B; ++k) {
            ln.Dr[k] += (c == base + k) ? v : 0.0;
            ln.Lr[k] += (c == base - B + k) ? v : 0.0;
            ln.Rr[k] += (c == base + B + k) ? v : 0.0;
        }
    }

    // solve S v = r (r in the lanes' registers); the solution ends in sh[xt] and in ln.r
    PQP_HD void band_solve() {
        const int nb = L.nb();
        double* rb = sh + L.rhs(); double* pl = sh + L.pl(); double* pr = sh + L.pr(); double* xb = sh + L.xt();
        int hmax = 1;
        while (2 * hmax <= nb) hmax <<= 1;
        for (int h = 1; h <= hmax; h <<= 1) {
            ctx.phase([&](int t, Lane& ln) {      // receive from the level just eliminated; blocks of this level publish r
                const int e = t / B, i = t - e * B;
                if (e >= nb) return;
                const int lv = level_of(e);
                if (h > 1 && lv >= h) {
                    const int hp = h >> 1;
                    if (e + hp < nb) ln.r -= pl[(e + hp) * B + i];
                    if (e - hp >= 0) ln.r -= pr[(e - hp) * B + i];
                }
                if (lv == h) rb[t] = ln.r;
            });
            ctx.phase([&](int t, Lane& ln) {      // messages of the blocks eliminated at this level
                const int e = t / B;
                if (e < nb && level_of(e) == h) {
                    double a = 0.0, b2 = 0.0;
                    _Pragma("unroll") for (int k = 0; k < B; ++k) { const double re = rb[e * B + k]; a += ln.HLt[k] * re; b2 += ln.HRt[k] * re; }
                    pl[t] = a; pr[t] = b2;
                }
            });
        }
        for (int h = hmax; h >= 1; h >>= 1) {
            ctx.phase([&](int t, Lane& ln) {
                const int e = t / B;
                if (e < nb && level_of(e) == h) {
                    double v = 0.0;
                    _Pragma("unroll") for (int k = 0; k < B; ++k) v += ln.Dinv[k] * rb[e * B + k];
                    if (e - h >= 0) { _Pragma("unroll") for (int k = 0; k < B; ++k) v -= ln.HL[k] * xb[(e - h) * B + k]; }
                    if (e + h < nb) { _Pragma("unroll") for (int k = 0; k < B; ++k) v -= ln.HR[k] * xb[(e + h) * B + k]; }
                    xb[t] = v;
                    ln.r = v;
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
        double* zt = sh + L.zt(); double* xt = sh + L.xt();
        const double* rv = sh + L.rv();
        rows([&](int r) { zt[r] = rv[r] * z[r] - y[r]; });
        ctx.phase([&](int t, Lane& ln) { ln.r = t < nv ? sh[L.sig() + t] * x[t] - q_of(t) + col_dot(t, zt) : 0.0; ln.b0 = ln.r; ln.x0 = 0.0; });
        band_solve();
        // While polishing (penalties 1/delta next to delta: condition ~1e12) the cyclic-reduction solve alone is not accurate
        // enough (S2: |b - S x| ~ 25); iterative refinement against the exactly applied S reaches the round-off floor
        // eps |S| |x| in one step, a second one is insurance.
        for (int step = 0; step < (polishing_ ? (B > 4 ? 2 : 1) : 0); ++step) {
            rows([&](int r) { zt[r] = rv[r] * row_dot(r, xt); });
            ctx.phase([&](int t, Lane& ln) {
                ln.x0 += ln.r;
                ln.r = t < nv ? ln.b0 - (p_times(t, xt) + sh[L.sig() + t] * xt[t] + col_dot(t, zt)) : 0.0;
            });
            // xt still holds the previous solve's output; the correction overwrites it, so fold it in afterwards
            band_solve();
            ctx.phase([&](int t, Lane& ln) { if (t < L.nbb()) { ln.r += ln.x0; ln.x0 = 0.0; xt[t] = ln.r; } });
        }
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

    // ---- primal infeasibility certificate on dy = y_k - y_{k-1} (OSQP paper 3.4; see pqp_path_lane.hpp::primal_infeasible) ----
    PQP_HD bool primal_infeasible() {
        double* dy = sh + L.yp();
        const double* y = sh + L.y();
        rows([&](int r) {
            const double d = y[r] - dy[r];
            const bool fr = sh[L.e2() + r] < 0.0, inf_u = sh[L.up() + r] > 1e19, inf_l = sh[L.lo() + r] < -1e19;
            dy[r] = (fr || (inf_u && inf_l)) ? 0.0 : (inf_u ? fmin(d, 0.0) : (inf_l ? fmax(d, 0.0) : d));
        });
        double nm[2], lhs[1];
        ctx.template reduce_max<2>(nm, [&](int t, double (&v)[2]) {
            v[0] = 0.0; v[1] = 0.0;
            for (int r = t; r < nc; r += T) v[0] = fmax(v[0], fabs(dy[r]));
            for (int j = t; j < nv; j += T) v[1] = fmax(v[1], fabs(col_dot(j, dy)));
        });
        ctx.template reduce_sum<1>(lhs, [&](int t, double (&v)[1]) {
            double acc = 0.0;
            for (int r = t; r < nc; r += T) { const double d = dy[r]; acc += d > 0.0 ? sh[L.up() + r] * d : (d < 0.0 ? sh[L.lo() + r] * d : 0.0); }
            v[0] = acc;
        });
        const double eps = A.prm.eps_prim_inf;
        return cscale * nm[0] > eps && lhs[0] < -eps * nm[0] && nm[1] < eps * nm[0];
    }

    // ---- polish (same scheme as pqp_path_lane.hpp) -------------------------------------------------------------
    // delta of the polish for these QPs: their equality rows carry the whole problem, and the multiplier update y += R (Ax - z)
    // has a round-off floor of eps/delta * |Ax|; at the path QP's 1e-6 that floor sits above the acceptance tolerance
    PQP_HD double polish_delta() const { return B > 4 ? A.prm.polish_delta : fmax(A.prm.polish_delta, 1e-4); }
    PQP_HD void polish_begin() {
        const pqp_params& prm = A.prm;
        const double gain = 1.0 / polish_delta(), sgain = polish_delta() / prm.sigma;
        rows([&](int r) {
            sh[L.zs() + r] = sh[L.z() + r]; sh[L.ys() + r] = sh[L.y() + r];
            const double b = sh[L.e2() + r];
            const bool fr = b < 0.0;
            const double e = sh[L.esc() + r], e2 = e * e / cscale;
            const double z = sh[L.z() + r], y = sh[L.y() + r];
            const bool eq = !fr && is_equality_row(r);           // an equality row is active whatever (z, y) say
            const bool alo = !fr && (eq || (z - sh[L.lo() + r]) * e2 < -y);
            const bool aup = !fr && !alo && ((sh[L.up() + r] - z) * e2 < y);
            sh[L.act() + r] = alo ? -1.0 : (aup ? 1.0 : 0.0);
        });
        cols([&](int j) { sh[L.xs() + j] = sh[L.x() + j]; sh[L.sig() + j] *= sgain; });
        (void)gain;
    }
    PQP_HD void polish_apply_set() {
        const double gain = 1.0 / polish_delta();
        rows([&](int r) {
            const double a = sh[L.act() + r];
            const double e = sh[L.esc() + r], e2 = e * e / cscale;
            sh[L.rv() + r] = a != 0.0 ? gain * e2 : 0.0;
            if (a == 0.0) sh[L.y() + r] = 0.0;
            else sh[L.z() + r] = a < 0.0 ? sh[L.lo() + r] : sh[L.up() + r];
        });
    }
    PQP_HD bool is_equality_row(int r) const { return sh[L.up() + r] - sh[L.lo() + r] < kRhoTol; }
    PQP_HD double row_violation(int r, double ax) const {
        if (sh[L.e2() + r] < 0.0) return 0.0;
        const double a = sh[L.act() + r], y = sh[L.y() + r];
        const double pv = fmax(sh[L.lo() + r] - ax, ax - sh[L.up() + r]);
        // (the multiplier of an equality row may have either sign)
        const double dv = is_equality_row(r) ? 0.0 : (a < 0.0 ? y : (a > 0.0 ? -y : 0.0));
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
        const double isgain = A.prm.sigma / polish_delta();
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
        // polish != 0: a QP WITHOUT inequality rows (TensionSmoother2's: every row of tension_smoother_2.cpp:119-145 has l == u) is an
        // equality-constrained QP - one KKT system.  It is solved as the polish solves it (all rows active, penalty 1/delta, proximal
        // multiplier iterations as refinement, KKT test) at iteration 0: no equilibration, no ADMM iterations, `iters` = 0, the exact optimum
        // where the reference's ADMM stops within eps of it.  QPs with inequality rows: polish == 2 the plain ADMM, == 1 see below.
        const bool polish_on = prm.polish == 1;
        bool direct = false;
        if (prm.polish != 0) {
            double ineq[1];
            ctx.template reduce_max<1>(ineq, [&](int t, double (&v)[1]) {
                v[0] = 0.0;
                for (int r = t; r < nc; r += T) {
                    const double sl = sh[L.lo() + r], su = sh[L.up() + r];
                    const bool free_row = sl < -kInfty * kMinScaling && su > kInfty * kMinScaling;
                    v[0] = fmax(v[0], (free_row || su - sl < kRhoTol) ? 0.0 : 1.0);
                }
            });
            // (polish == 1: QPs with inequality rows start the same way - the first active set is OSQP's rule applied to the cold start, the
            //  active-set rounds do the rest: postSmooth's boxes are found in 1-4 rounds, 4-22 solves instead of 50-75 ADMM iterations -, and
            //  fall back to equilibration + ADMM + periodic polish attempts when that first attempt is rejected)
            direct = ineq[0] == 0.0 || prm.polish == 1;
        }
        if (prm.scaling != 0 && !direct) ruiz();
        else {
            cols([&](int j) { sh[L.sig() + j] = prm.sigma; });
            rows([&](int r) {
                const double sl = sh[L.lo() + r], su = sh[L.up() + r];
                sh[L.e2() + r] = (sl < -kInfty * kMinScaling && su > kInfty * kMinScaling) ? -kRhoMin : (su - sl < kRhoTol ? kRhoEqFactor : 1.0);
            });
            set_rho();
        }
        if (!direct) factor();
        int status = PQP_STATUS_MAX_ITER, it = 0, polished = 0;
        double res[5] = {0, 0, 0, 0, 0};
        double eps_scale = 1.0;
        int polish_gap = prm.polish_every, next_polish = prm.polish_every;
        for (it = direct ? 0 : 1; it <= prm.max_iter; ++it) {
            // (prim_inf_after > 0: the certificate only from that iteration on - the production setting's feasible QPs never get there)
            const bool cert_now = prm.eps_prim_inf > 0.0 && it >= prm.prim_inf_after;
            bool start_polish = it == 0;
            bool check = false, adapt = false;
            if (it > 0) {
            if (cert_now && prm.check_termination > 0 && (it % prm.check_termination) == 0)
                rows([&](int r) { sh[L.yp() + r] = sh[L.y() + r]; });
            iterate();
            check = prm.check_termination > 0 && (it % prm.check_termination) == 0;
            adapt = prm.adaptive_rho && prm.adaptive_rho_interval > 0 && (it % prm.adaptive_rho_interval) == 0;
            if (!check && !adapt) continue;
            residuals(res);
            if (res[4] != 0.0) { status = PQP_STATUS_NUMERICAL; break; }
            }
            if (check) {
                const double eps_p = eps_scale * (prm.eps_abs + prm.eps_rel * res[2]);
                const double eps_d = eps_scale * (prm.eps_abs + prm.eps_rel * res[3]);
                const bool converged = res[0] <= eps_p && res[1] <= eps_d;
                if (converged) {
                    if (!polish_on || eps_scale * fmax(prm.eps_abs, prm.eps_rel) < 1e-10) { status = PQP_STATUS_SOLVED; break; }
                    start_polish = true;
                } else if (cert_now && it > 1 && primal_infeasible()) {
                    status = PQP_STATUS_PRIMAL_INFEASIBLE; break;
                } else if (polish_on && prm.polish_every > 0 && it >= next_polish) {
                    start_polish = true;
                    polish_gap *= 2;
                    next_polish = it + polish_gap;
                }
            }
            if (start_polish) {
                const bool was_converged = it > 0 && res[0] <= eps_scale * (prm.eps_abs + prm.eps_rel * res[2]) && res[1] <= eps_scale * (prm.eps_abs + prm.eps_rel * res[3]);
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
                    double prev = fmax(res[0], res[1]);
                    double viol = polish_violation();
                    // A row that fails the KKT test by far more than the solve's own residual fails it at the converged point too: those rows
                    // change sides now (the path QP's lazy refinement); only a point that might be accepted is refined further first.
                    const bool lazy_move = res[4] == 0.0 && !conservative && viol > 10.0 * fmax(tol, prev);
                    bool solve_ok = false;
                    if (!lazy_move) {
                        // ill-conditioned Hessians (the 3rd-difference weights of S2) need more refinement: keep going while the
                        // residual is above the absolute tolerance and still shrinking
                        for (int extra = 0; extra < 24 && prev > tol; extra += 2) {
                            iterate(); iterate();
                            residuals(res);
                            const double cur = fmax(res[0], res[1]);
                            if (cur > 0.7 * prev) break;
                            prev = cur;
                        }
                        viol = polish_violation();
                        solve_ok = res[4] == 0.0 && res[0] <= tol * (1.0 + res[2]) && res[1] <= tol * (1.0 + res[3]);
                        ok = solve_ok && viol <= tol;
#ifdef PQP_EMU_DEBUG
                        printf("  bq polish qp %d it %d round %d: pri %.3e (norm %.2e) dua %.3e (norm %.2e) viol %.3e -> %s\n", qp, it, round, res[0], res[2], res[1], res[3], viol, ok ? "ACCEPT" : (solve_ok ? "next" : "solve failed"));
#endif
                        if (ok || !solve_ok) break;
                    }
                    if (viol < 0.7 * best) { best = viol; stall = 0; } else { stall += 1; }
                    if (stall >= 3) conservative = true;
                    if (conservative && stall >= 16) break;
                    polish_update_set(lazy_move ? 10.0 * fmax(tol, prev) : (conservative ? fmax(tol, 0.9 * viol) : tol));
                }
                polishing_ = false; alpha_ = prm.alpha;
                polish_end(ok);
                if (ok) { status = PQP_STATUS_SOLVED; polished = 1; break; }
                if (was_converged) eps_scale *= 0.1;     // rejected: ADMM resumes one decade tighter
                if (it == 0 && prm.scaling != 0) ruiz();  // (the direct attempt ran without equilibration)
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
        ctx.phase([&](int t, Lane&) {
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
