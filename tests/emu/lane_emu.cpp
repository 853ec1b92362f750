// lane_emu.cpp — TEST INFRASTRUCTURE: runs the device algorithm source (pqp_path_lane.hpp) on the host,
// phase by phase, lane by lane, so the algorithm can be checked against the oracle without a GPU.
// Built only by tests/ (tests/test_lane_emulation.py); never linked into libpqp_hip.so, never timed.
#include <cstdlib>
#include <cstring>
#include <limits>
#include <vector>

#include "../../path_optimizer_2_amd/csrc/pqp_path_lane.hpp"
#include "../../path_optimizer_2_amd/csrc/pqp_defaults.hpp"
#include "../../path_optimizer_2_amd/csrc/pqp_banded_qp.hpp"

// PQP_EMU_POISON=1 in the environment: shared arrays and lane registers start as NaN instead of zero, so that anything the algorithm
// reads before it wrote it (LDS and registers are NOT cleared between the QPs of a persistent workgroup) shows up in the results
static double poison_value() {
    static const bool on = std::getenv("PQP_EMU_POISON") && std::getenv("PQP_EMU_POISON")[0] == '1';
    return on ? std::numeric_limits<double>::quiet_NaN() : 0.0;
}
template <class LaneT>
static void poison_lanes(std::vector<LaneT>& lanes) {
    if (poison_value() == 0.0) return;
    for (auto& l : lanes) std::memset(static_cast<void*>(&l), 0xff, sizeof(LaneT));      // all-ones doubles are NaN, ints are -1
}
static int g_wave_order = 1;
static const int32_t* g_n_of = nullptr;      // per-QP waypoint counts of the next pqp_emu_path_solve call (nullptr: all n)
extern "C" void pqp_emu_set_counts(const int32_t* n_of) { g_n_of = n_of; }
extern "C" void pqp_emu_set_wave_order(int o) { g_wave_order = o; }
// PQP_OPT_CARRY_CYCLES = k >= 2 in the emulation: the cost keys of the "previous launch" (bin << 24), the histogram block (its entry kCostBins + 1 = the
// threshold bin) and k for the next pqp_emu_path_solve call (nullptr: off)
static int32_t* g_cost_key = nullptr;
static int32_t* g_cost_hist = nullptr;
static int g_carry_tails = 0;
extern "C" void pqp_emu_set_carry(int32_t* cost_key, int32_t* cost_hist, int carry_tails) { g_cost_key = cost_key; g_cost_hist = cost_hist; g_carry_tails = carry_tails; }

namespace {
#ifndef PQP_EMU_DIET
#define PQP_EMU_DIET 0
#endif
struct HostCtx {
    // default: the device contexts' setting (Ruiz vectors parked, save area in the shared array up to 256 lanes); PQP_EMU_DIET: pass
    // constants in the shared array too, everything parked in global memory
    static constexpr bool kCstLds = PQP_EMU_DIET != 0, kParkScale = true, kSaveLds = PQP_EMU_DIET == 0, kDpp = false, kCstAcc = false;
    static constexpr bool kFinalRefine = true;      // (the device: contexts of more than 128 lanes per QP; the parameter is 0 below that)
    template <class... A> static void join(A&...) {}      // (the device: a scheduling fence behind a batch of LDS loads)
    int T_;
    std::vector<pqp::Lane> lanes;
    std::vector<double> shm;
    explicit HostCtx(int T) : T_(T), lanes(T), shm(pqp::ShLayout{T}.total(true), poison_value()) { poison_lanes(lanes); }
    int T() const { return T_; }
    double* sh() { return shm.data(); }
    template <class F> void phase(F f) { for (int t = 0; t < T_; ++t) f(t, lanes[t]); }
    long long clock() const { return 0; }
    // wave-local phase: the emulation runs the wavefronts one after the other, in the order g_wave_order selects (0: first
    // wavefront first, 1: last wavefront first), so that code relying on a workgroup barrier it does not have reads stale data
    // in one of the two orders and fails the tests
    template <class F> void phase_w(F f) {
        const int nw = (T_ + 63) / 64;
        for (int k = 0; k < nw; ++k) {
            const int w = g_wave_order ? nw - 1 - k : k;
            for (int t = 64 * w; t < 64 * (w + 1) && t < T_; ++t) f(t, lanes[t]);
        }
    }
    static double uni(double x) { return x; }
    static int uni_int(int x) { return x; }
    template <class PQ> void cold(PQ& pq, int op, int i0, int i1, double d0) { pq.do_cold(op, i0, i1, d0); }
    // lane-less context of the infeasibility certificate
    struct LaneLess {
        int T_;
        int T() const { return T_; }
        template <class F> void phase(F f) { for (int t = 0; t < T_; ++t) f(t); }
        template <int K, class F> void reduce_max(double (&out)[K], F f) {
            for (int k = 0; k < K; ++k) out[k] = 0.0;
            for (int t = 0; t < T_; ++t) { double v[K]; f(t, v); for (int k = 0; k < K; ++k) out[k] = out[k] > v[k] ? out[k] : v[k]; }
        }
        template <int K, class F> void reduce_sum(double (&out)[K], F f) {
            for (int k = 0; k < K; ++k) out[k] = 0.0;
            for (int t = 0; t < T_; ++t) { double v[K]; f(t, v); for (int k = 0; k < K; ++k) out[k] += v[k]; }
        }
    };
    bool certificate(double* sh, int T, double fl, double rl, double kap, double eps, double cscale) {
        LaneLess c{T};
        return pqp::primal_certificate(c, sh, T, fl, rl, kap, eps, cscale);
    }
    // The device calls this once per lane and the lanes meet at the barrier inside; lane by lane on the host: every lane stages its
    // values (the phase inside is then a no-op), the LAST lane's call evaluates the test on the complete data
    struct StageOnly {
        int T_;
        int T() const { return T_; }
        template <class F> void phase(F) {}
        template <int K, class F> void reduce_max(double (&out)[K], F) { for (int k = 0; k < K; ++k) out[k] = 0.0; }
        template <int K, class F> void reduce_sum(double (&out)[K], F) { for (int k = 0; k < K; ++k) out[k] = 0.0; }
    };
    bool late_certificate(double* sh, int t, double* snap, bool have, const pqp::LateCertIn& in, double fl, double rl, double kap, double eps,
                          double cscale) {
        StageOnly st{T_};
        (void)pqp::late_certificate(st, sh, T_, t, snap, false, in, fl, rl, kap, eps, cscale);
        if (t != T_ - 1 || !have) return false;
        LaneLess c{T_};
        return pqp::primal_certificate(c, sh, T_, fl, rl, kap, eps, cscale);
    }
    template <int K, class F> void reduce_max(double (&out)[K], F f) {
        for (int k = 0; k < K; ++k) out[k] = 0.0;
        for (int t = 0; t < T_; ++t) { double v[K]; f(t, lanes[t], v); for (int k = 0; k < K; ++k) out[k] = out[k] > v[k] ? out[k] : v[k]; }
    }
    template <int K, class F> void reduce_sum(double (&out)[K], F f) {
        for (int k = 0; k < K; ++k) out[k] = 0.0;
        for (int t = 0; t < T_; ++t) { double v[K]; f(t, lanes[t], v); for (int k = 0; k < K; ++k) out[k] += v[k]; }
    }
};
}  // namespace

extern "C" void pqp_emu_default_params(pqp_params* p) { pqp::default_params(p); }
extern "C" void pqp_emu_production_params(pqp_params* p) { pqp::production_params(p); }

extern "C" int pqp_emu_path_solve(const pqp_params* prm, int batch, int n, const double* ref, const double* lin,
                                  const double* bounds, const double* scal, int passes, int warm, double* out,
                                  int32_t* status, int32_t* iters, double* info, double* wx, double* wy,
                                  double* wye, double* wrho) {
    int T = 64;
    while (T < n) T *= 2;
    pqp::PathSolveArgs a;
    std::memset(&a, 0, sizeof(a));
    a.batch = batch; a.n = n; a.passes = passes; a.warm = warm;
    a.ref = ref; a.lin = lin; a.bounds = bounds; a.scal = scal; a.out = out;
    a.status = status; a.iters = iters; a.info = info;
    a.wx = wx; a.wy = wy; a.wye = wye; a.wrho = wrho;
    std::vector<double> wsave((size_t)batch * T * PQP_SAVE_STRIDE, 0.0);
    a.wsave = wsave.data();
    std::vector<double> wscale((size_t)batch * T * 18, 0.0);
    a.wscale = wscale.data();
    a.store_warm = 1;
    a.prm = *prm;
    pqp::resolve_path_params(&a.prm, n);          // (as the launcher does)
    a.n_of = g_n_of;
    a.cost_key = g_cost_key; a.cost_hist = g_cost_hist; a.carry_tails = g_cost_key ? g_carry_tails : 0; a.carry_k = a.carry_tails;
    for (int q = 0; q < batch; ++q) {
        if (pqp::PathQp<HostCtx, true>::count_of(a, q) < 2) {
            if (status) status[q] = PQP_STATUS_UNSOLVED;
            if (iters) iters[q] = 0;
            continue;
        }
        HostCtx ctx(T);
        if (prm->eps_prim_inf > 0.0 && prm->prim_inf_after <= 0) { pqp::PathQp<HostCtx, true> s(ctx, a, q); s.run(); }       // the two variants the launcher picks from
        else { pqp::PathQp<HostCtx, false> s(ctx, a, q); s.run(); }
    }
    return 0;
}

// Debug probe: run load/assemble/ruiz/factor (+ iterations) and dump per-waypoint state.
//   dump [n][64]: a(6) bT(3) lo(3) up(3) D(6) E(6) sig(6) rhoT(3) rhoI(3)  = 39 used
extern "C" int pqp_emu_probe(const pqp_params* prm, int n, const double* ref, const double* bounds, const double* scal,
                             double* dump, double* endrows, double* cscale, int do_iters, double* xout) {
    int T = 64;
    while (T < n) T *= 2;
    pqp::PathSolveArgs a;
    std::memset(&a, 0, sizeof(a));
    a.batch = 1; a.n = n; a.ref = ref; a.bounds = bounds; a.scal = scal; a.prm = *prm;
    pqp::resolve_path_params(&a.prm, n);
    std::vector<double> wsave((size_t)T * PQP_SAVE_STRIDE, 0.0);
    a.wsave = wsave.data();
    std::vector<double> wscale((size_t)T * 18, 0.0);
    a.wscale = wscale.data();
    HostCtx ctx(T);
    pqp::PathQp<HostCtx> s(ctx, a, 0);
    s.load();
    s.end_rows()->y[0] = s.end_rows()->y[1] = 0.0;
    s.assemble();
    s.ruiz();
    s.factor();
    s.start_transition_rows(false);
    for (int it = 0; it < do_iters; ++it) { s.iterate(); if (it == 0) s.finish_first_iteration(); }
    s.sync_after_iterate();
    for (int i = 0; i < n; ++i) {
        const pqp::Slot& S = ctx.lanes[i].s;
        const pqp::SlotSetup& W = ctx.lanes[i].w;
        double* d = dump + 64 * i;
        int o = 0;
        for (int k = 0; k < 6; ++k) d[o++] = S.a[k];
        for (int k = 0; k < 3; ++k) d[o++] = S.bT[k];
        for (int k = 0; k < 3; ++k) d[o++] = s.box_lo(S, i, k);
        for (int k = 0; k < 3; ++k) d[o++] = s.box_up(S, i, k);
        for (int k = 0; k < 6; ++k) d[o++] = W.D[k];
        for (int k = 0; k < 6; ++k) d[o++] = W.E[k];
        for (int k = 0; k < 6; ++k) d[o++] = S.sig[k];
        for (int k = 0; k < 3; ++k) d[o++] = S.rhoT[k];
        for (int k = 0; k < 3; ++k) d[o++] = S.rhoI[k];
        for (int k = 0; k < 6; ++k) xout[6 * i + k] = S.x[k];
    }
    std::memcpy(endrows, s.end_rows(), sizeof(pqp::EndRows));
    *cscale = s.cscale;
    return 0;
}

// ---- the generic banded-QP core (pqp_banded_qp.hpp) on the host -----------------------------------------------------
namespace {
template <int B>
struct BqHostCtx {
    static constexpr bool kStage = true;      // as the device's 256- and 512-lane kernels: row data, index lists and q in the shared array
    int T_;
    std::vector<double> shm;
    std::vector<pqp::BqLane<B>> lanes;
    BqHostCtx(int T, int doubles) : T_(T), shm(doubles, poison_value()), lanes(T) { poison_lanes(lanes); }
    int T() const { return T_; }
    double* sh() { return shm.data(); }
    template <class F> void phase(F f) { for (int t = 0; t < T_; ++t) f(t, lanes[t]); }
    template <int K, class F> void reduce_max(double (&out)[K], F f) {
        for (int k = 0; k < K; ++k) out[k] = 0.0;
        for (int t = 0; t < T_; ++t) { double v[K]; f(t, v); for (int k = 0; k < K; ++k) out[k] = out[k] > v[k] ? out[k] : v[k]; }
    }
    template <int K, class F> void reduce_sum(double (&out)[K], F f) {
        for (int k = 0; k < K; ++k) out[k] = 0.0;
        for (int t = 0; t < T_; ++t) { double v[K]; f(t, v); for (int k = 0; k < K; ++k) out[k] += v[k]; }
    }
};

template <int B>
void bq_run(const pqp::BandedQpArgs& a) {
    const pqp::BqLayout L{a.nv, a.nc, a.bw};
    const int T = 64 * ((L.nbb() + 63) / 64);
    for (int qp = 0; qp < a.batch; ++qp) {
        BqHostCtx<B> ctx(T, L.total(true));
        pqp::BandedQp<BqHostCtx<B>, B> s(ctx, a, qp);
        s.run();
    }
}
}  // namespace

extern "C" int pqp_emu_banded_solve(const pqp_params* prm, int batch, int nv, int nc, int bw, int pbw, const double* pband, const double* q,
                                    const int* acol, const double* aval, const int* trow, const int* tslot, const double* lo,
                                    const double* up, double* x, double* y, int32_t* status, int32_t* iters, double* info) {
    pqp::BandedQpArgs a;
    std::memset(&a, 0, sizeof(a));
    a.batch = batch; a.nv = nv; a.nc = nc; a.bw = bw; a.pbw = pbw;
    a.pband = pband; a.q = q; a.acol = acol; a.aval = aval; a.trow = trow; a.tslot = tslot; a.lo = lo; a.up = up;
    a.x = x; a.y = y; a.status = status; a.iters = iters; a.info = info; a.prm = *prm;
    pqp::resolve_banded_params(&a.prm);
    switch (bw) {
        case 3: bq_run<3>(a); break;
        case 4: bq_run<4>(a); break;
        case 9: bq_run<9>(a); break;
        default: return -1;
    }
    return 0;
}
