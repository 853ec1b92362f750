// pqp_kernels.hip — gfx950 (MI355X / CDNA4) kernels and the C ABI of include/pqp.h.
//
//   path_solve_kernel     the lane-per-waypoint path-QP kernel lives in pqp_path_solve.hip (one translation unit per workgroup width);
//                         this file holds its launcher, path_solve_impl.
//   path_assemble_kernel  BaseSolver::setCost/setConstraints in the REFERENCE numbering: CSC values of A,
//                         diagonal of P, l, u; staged through LDS and written with contiguous, coalesced
//                         stores (base_solver.cpp:119-261).
//   path_pattern_kernel   the value-independent CSC pattern (integer index maps, base_solver.cpp:154-209).
//   path_gather_solution  lane layout -> reference numbering of the primal / dual solution.
//
// No CPU fallback: every entry point needs a HIP device.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <atomic>
#include <new>
#include <string>

#include "pqp_defaults.hpp"
#include "pqp_path_lane.hpp"
#include "pqp_banded_qp.hpp"
#include "pqp_path_lq_abi.hpp"
#include <vector>

#include "pqp_wave.hpp"

namespace pqp {

// -------------------------------------------------------------------------------------------------------
// reference numbering helpers (base_solver.cpp:22-37,154-158)
// -------------------------------------------------------------------------------------------------------
struct RefIndex {
    int n, precise;
    __host__ __device__ int state() const { return 3 * n; }
    __host__ __device__ int control() const { return n - 1; }
    __host__ __device__ int vars() const { return 3 * n + n - 1 + precise + n; }
    __host__ __device__ int cons() const { return 4 * n + precise + n + 2; }
    __host__ __device__ int kappa_idx() const { return 3 * n; }
    __host__ __device__ int precise_idx() const { return 4 * n; }
    __host__ __device__ int rough_idx() const { return 4 * n + 2 * precise; }
    __host__ __device__ int end_idx() const { return 4 * n + 2 * precise + n - precise; }
    __host__ __device__ int slack_col(int i, int which) const {
        return i < precise ? 4 * n - 1 + 2 * i + which : 4 * n - 1 + 2 * precise + (i - precise);
    }
    __host__ __device__ int nnz_a() const { return 3 * n + 7 * (n - 1) + n + 6 * precise + 2 * (n - precise) + 2; }
    __host__ __device__ int nnz_p() const { return n + n - 1 + precise + n; }
    // CSC offset of the first entry of column 3i (state column block of waypoint i)
    __host__ __device__ int state_col_offset(int i) const {
        // per waypoint j < n-1: precise -> 5+5+4 = 14 entries, rough -> 4+3+4 = 11
        const int np = i < precise ? i : precise;
        return 14 * np + 11 * (i - np);
    }
    __host__ __device__ int control_col_offset() const {
        // all state columns: waypoints 0..n-2 full, last waypoint has no outgoing transition (-2 per column)
        // but two end rows (+1 on l and psi): l: 1 + (2|1) + 1, psi: 1 + (2|0) + 1, k: 1 + 1
        const bool last_precise = (n - 1) < precise;
        return state_col_offset(n - 1) + (last_precise ? 4 + 4 + 2 : 3 + 2 + 2);
    }
    __host__ __device__ int slack_col_offset() const { return control_col_offset() + (n - 1); }
};

// pattern: rows[nnz_a], colptr[vars+1], pcols[nnz_p]; one thread per waypoint
__global__ void path_pattern_kernel(RefIndex R, int32_t* rows, int32_t* colptr, int32_t* pcols) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int n = R.n;
    if (i >= n) return;
    const bool precise = i < R.precise, has_next = i < n - 1, last = i == n - 1;
    int o = R.state_col_offset(i);
    // column l_i
    colptr[3 * i] = o;
    rows[o++] = 3 * i;
    if (has_next) { rows[o++] = 3 * (i + 1); rows[o++] = 3 * (i + 1) + 1; }
    if (precise) { rows[o++] = R.precise_idx() + 2 * i; rows[o++] = R.precise_idx() + 2 * i + 1; }
    else rows[o++] = R.rough_idx() + (i - R.precise);
    if (last) rows[o++] = R.end_idx();
    // column psi_i
    colptr[3 * i + 1] = o;
    rows[o++] = 3 * i + 1;
    if (has_next) { rows[o++] = 3 * (i + 1); rows[o++] = 3 * (i + 1) + 1; }
    if (precise) { rows[o++] = R.precise_idx() + 2 * i; rows[o++] = R.precise_idx() + 2 * i + 1; }
    if (last) rows[o++] = R.end_idx() + 1;
    // column k_i
    colptr[3 * i + 2] = o;
    rows[o++] = 3 * i + 2;
    if (has_next) { rows[o++] = 3 * (i + 1) + 1; rows[o++] = 3 * (i + 1) + 2; }
    rows[o++] = R.kappa_idx() + i;
    // control column u_i
    if (has_next) {
        const int oc = R.control_col_offset() + i;
        colptr[R.state() + i] = oc;
        rows[oc] = 3 * (i + 1) + 2;
    }
    // slack columns
    const int os = R.slack_col_offset();
    if (precise) {
        colptr[R.slack_col(i, 0)] = os + 2 * i;
        colptr[R.slack_col(i, 1)] = os + 2 * i + 1;
        rows[os + 2 * i] = R.precise_idx() + 2 * i;
        rows[os + 2 * i + 1] = R.precise_idx() + 2 * i + 1;
    } else {
        const int li = i - R.precise;
        colptr[R.slack_col(i, 0)] = os + 2 * R.precise + li;
        rows[os + 2 * R.precise + li] = R.rough_idx() + li;
    }
    if (last) colptr[R.vars()] = R.nnz_a();
    // P diagonal columns (base_solver.cpp:127-143), ascending: k_i, then u_i, then slacks
    pcols[i] = 3 * i + 2;
    if (has_next) pcols[n + i] = R.state() + i;
    if (precise) { pcols[2 * n - 1 + 2 * i] = R.slack_col(i, 0); pcols[2 * n - 1 + 2 * i + 1] = R.slack_col(i, 1); }
    else pcols[2 * n - 1 + 2 * R.precise + (i - R.precise)] = R.slack_col(i, 0);
}

// assemble in the reference numbering.  One workgroup per QP; values are staged in LDS in their final
// order and then streamed out with unit-stride stores.
__global__ void __launch_bounds__(256) path_assemble_kernel(RefIndex R, int batch, const double* __restrict__ ref,
                                                            const double* __restrict__ lin, const double* __restrict__ bounds,
                                                            const double* __restrict__ scal, pqp_params prm,
                                                            double* __restrict__ a_val, double* __restrict__ p_val,
                                                            double* __restrict__ lower, double* __restrict__ upper,
                                                            int stage_in_lds) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int n = R.n, nnz_a = R.nnz_a(), nnz_p = R.nnz_p(), cons = R.cons();
    for (int qp = blockIdx.x; qp < batch; qp += gridDim.x) {
        double* va = stage_in_lds ? smem : a_val + (size_t)qp * nnz_a;
        double* vp = stage_in_lds ? smem + nnz_a : p_val + (size_t)qp * nnz_p;
        double* vl = stage_in_lds ? smem + nnz_a + nnz_p : lower + (size_t)qp * cons;
        double* vu = stage_in_lds ? smem + nnz_a + nnz_p + cons : upper + (size_t)qp * cons;
        const double* rq = ref + (size_t)qp * n * PQP_REF_STRIDE;
        const double* lq = lin ? lin + (size_t)qp * n * PQP_LIN_STRIDE : nullptr;
        const double* bq = bounds + (size_t)qp * n * PQP_BOUNDS_STRIDE;
        const double* sc = scal + (size_t)qp * PQP_SCAL_STRIDE;
        for (int i = threadIdx.x; i < n; i += blockDim.x) {
            const bool precise = i < R.precise, has_next = i < n - 1, last = i == n - 1;
            double a[6] = {0, 0, 0, 0, 0, 0}, c3[3] = {0, 0, 0};
            if (has_next) {   // outgoing transition i -> i+1 fills the columns of waypoint i
                double lp[3];
                double knext;
                if (lq) { lp[0] = lq[3 * i]; lp[1] = lq[3 * i + 1]; lp[2] = lq[3 * i + 2]; knext = lq[3 * (i + 1) + 2]; }
                else { lp[0] = 0.0; lp[1] = 0.0; lp[2] = rq[5 * i + 1]; knext = rq[5 * (i + 1) + 1]; }
                transition_block(lp, knext, rq[5 * i], rq[5 * (i + 1)], rq[5 * i + 1], a, c3);
            }
            int o = R.state_col_offset(i);
            // column l_i
            va[o++] = -1.0;
            if (has_next) { va[o++] = a[0]; va[o++] = a[2]; }
            if (precise) { va[o++] = 1.0; va[o++] = 1.0; } else va[o++] = 1.0;
            if (last) va[o++] = 1.0;
            // column psi_i
            va[o++] = -1.0;
            if (has_next) { va[o++] = a[1]; va[o++] = a[3]; }
            if (precise) { va[o++] = prm.front_length; va[o++] = prm.rear_length; }
            if (last) va[o++] = 1.0;
            // column k_i
            va[o++] = -1.0;
            if (has_next) { va[o++] = a[4]; va[o++] = 1.0; }
            va[o++] = 1.0;
            if (has_next) va[R.control_col_offset() + i] = a[5];
            const int os = R.slack_col_offset();
            if (precise) { va[os + 2 * i] = 1.0; va[os + 2 * i + 1] = 1.0; }
            else va[os + 2 * R.precise + (i - R.precise)] = 1.0;
            // P diagonal (base_solver.cpp:123-143)
            vp[i] = prm.weight_kappa;
            if (has_next) vp[n + i] = prm.weight_dkappa;
            if (precise) { vp[2 * n - 1 + 2 * i] = prm.weight_slack; vp[2 * n - 1 + 2 * i + 1] = prm.weight_slack; }
            else vp[2 * n - 1 + 2 * R.precise + (i - R.precise)] = prm.weight_slack;
            // bounds (base_solver.cpp:212-260)
            if (i == 0) {
                for (int k = 0; k < 3; ++k) { vl[k] = -sc[k]; vu[k] = -sc[k]; }
            }
            if (has_next) {
                for (int k = 0; k < 3; ++k) { vl[3 * (i + 1) + k] = -c3[k]; vu[3 * (i + 1) + k] = -c3[k]; }
            }
            const double kappa_limit = tan(sc[5]) / prm.wheel_base;
            vl[R.kappa_idx() + i] = -kappa_limit;
            vu[R.kappa_idx() + i] = kappa_limit;
            double lo, up;
            if (precise) {
                soft_bounds(bq[6 * i], bq[6 * i + 1], prm.expected_safety_margin, prm.min_clearance, lo, up);
                vl[R.precise_idx() + 2 * i] = lo; vu[R.precise_idx() + 2 * i] = up;
                soft_bounds(bq[6 * i + 2], bq[6 * i + 3], prm.expected_safety_margin, prm.min_clearance, lo, up);
                vl[R.precise_idx() + 2 * i + 1] = lo; vu[R.precise_idx() + 2 * i + 1] = up;
            } else {
                soft_bounds(bq[6 * i + 4], bq[6 * i + 5], prm.expected_safety_margin, prm.min_clearance, lo, up);
                vl[R.rough_idx() + (i - R.precise)] = lo; vu[R.rough_idx() + (i - R.precise)] = up;
            }
            if (last) {
                vl[R.end_idx()] = -prm.end_l_bound; vu[R.end_idx()] = prm.end_l_bound;
                double el = -kInfty, eu = kInfty;
                if (prm.constraint_end_heading && sc[4] == 0.0) {
                    const double end_psi = constrain_angle(sc[3] - rq[5 * i + 2]);
                    if (end_psi < prm.end_psi_max) { el = end_psi - prm.end_psi_tol; eu = end_psi + prm.end_psi_tol; }
                }
                vl[R.end_idx() + 1] = el; vu[R.end_idx() + 1] = eu;
            }
        }
        if (stage_in_lds) {
            __syncthreads();
            double* ga = a_val + (size_t)qp * nnz_a;
            double* gp = p_val + (size_t)qp * nnz_p;
            double* gl = lower + (size_t)qp * cons;
            double* gu = upper + (size_t)qp * cons;
            for (int k = threadIdx.x; k < nnz_a; k += blockDim.x) ga[k] = smem[k];
            for (int k = threadIdx.x; k < nnz_p; k += blockDim.x) gp[k] = smem[nnz_a + k];
            for (int k = threadIdx.x; k < cons; k += blockDim.x) gl[k] = smem[nnz_a + nnz_p + k];
            for (int k = threadIdx.x; k < cons; k += blockDim.x) gu[k] = smem[nnz_a + nnz_p + cons + k];
            __syncthreads();
        }
    }
}

// lane layout (handle's warm state) -> reference numbering (OsqpEigen getSolution order)
__global__ void path_gather_solution(RefIndex R, int batch, const double* __restrict__ wx, const double* __restrict__ wy,
                                     const double* __restrict__ wye, double* __restrict__ x, double* __restrict__ y) {
    const int n = R.n;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= batch * n) return;
    const int qp = idx / n, i = idx - qp * n;
    const double* sx = wx + (size_t)idx * 6;
    const double* sy = wy + (size_t)idx * 6;
    if (x) {
        double* xo = x + (size_t)qp * R.vars();
        xo[3 * i] = sx[0]; xo[3 * i + 1] = sx[1]; xo[3 * i + 2] = sx[2];
        if (i > 0) xo[R.state() + i - 1] = sx[3];
        xo[R.slack_col(i, 0)] = sx[4];
        if (i < R.precise) xo[R.slack_col(i, 1)] = sx[5];
    }
    if (y) {
        double* yo = y + (size_t)qp * R.cons();
        yo[3 * i] = sy[0]; yo[3 * i + 1] = sy[1]; yo[3 * i + 2] = sy[2];
        yo[R.kappa_idx() + i] = sy[3];
        if (i < R.precise) { yo[R.precise_idx() + 2 * i] = sy[4]; yo[R.precise_idx() + 2 * i + 1] = sy[5]; }
        else yo[R.rough_idx() + (i - R.precise)] = sy[4];
        if (i == n - 1) { yo[R.end_idx()] = wye[2 * qp]; yo[R.end_idx() + 1] = wye[2 * qp + 1]; }
    }
}

// constrainAngle (include/tools/tools.hpp:24-35) as the kernels of this library evaluate it: what pqp_constrain_angle_device exposes
__global__ void constrain_angle_kernel(int count, const double* __restrict__ in, double* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) out[i] = constrain_angle(in[i]);
}

}  // namespace pqp

#include "pqp_smoother_kernels.inc"
#include "pqp_corridor_kernels.inc"

// =========================================================================================================
// C ABI
// =========================================================================================================
namespace {
thread_local std::string g_last_error;

int fail(int code, const std::string& msg) {
    g_last_error = msg;
    return code;
}
#define PQP_HIP(call)                                                                                  \
    do {                                                                                               \
        hipError_t e_ = (call);                                                                        \
        if (e_ != hipSuccess) return fail(PQP_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_)); \
    } while (0)

// every (re)allocation of a device buffer of the library: captured hipGraphs of the chain hold device pointers and are only replayed
// while this has not moved (pqp_chain.inc)
std::atomic<unsigned long long> g_alloc_generation{0};

struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
    int ensure(size_t need) {
        if (need <= bytes) return PQP_OK;
        g_alloc_generation.fetch_add(1, std::memory_order_relaxed);
        if (p) (void)hipFree(p);
        p = nullptr; bytes = 0;
        PQP_HIP(hipMalloc(&p, need));
        bytes = need;
        // No call ever reads uninitialised device memory (warm state of skipped QPs, info rows).  hipMemset runs on the NULL stream and
        // may return before the fill has executed; the handles' streams are non-blocking, i.e. NOT ordered behind the NULL stream, so a
        // fill still queued there could land on the buffer milliseconds later, after kernels of the handle have written it (seen: a
        // whole smoother batch solved on zeroed problem data).  Allocation is rare: wait for the fill.
        PQP_HIP(hipMemset(p, 0, need));
        PQP_HIP(hipStreamSynchronize(nullptr));
        return PQP_OK;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; bytes = 0; }
    template <class T> T* as() const { return static_cast<T*>(p); }
};
}  // namespace

struct pqp_handle {
    int device = 0;
    pqp_params prm;
    hipStream_t stream = nullptr;
    // HIP events around the dominant kernel of every call, on the stream it is launched on: a ring of the last kEvRing launches,
    // read back (after the work is done) by pqp_last_kernel_ms / pqp_kernel_ms_history without putting a sync between launches
    static constexpr int kEvRing = 256;
    hipEvent_t evs0[kEvRing] = {}, evs1[kEvRing] = {};
    long long ev_count = 0;          // launches recorded so far
    hipEvent_t ev0 = nullptr, ev1 = nullptr;      // the pair of the launch being recorded
    bool timed = false;
    static constexpr int kMarks = 8;
    static constexpr int kChainMarks = 2;      // + two events of pqp_optimize_path_device's own
    hipEvent_t marks[kMarks + kChainMarks] = {};   // pqp_mark / pqp_wait_mark: ordering between the streams of two handles
    void next_event_pair() { if (capturing) return; ev0 = evs0[ev_count % kEvRing]; ev1 = evs1[ev_count % kEvRing]; ev_count += 1; }
    // PQP_OPT_CHAIN_GRAPH: pqp_optimize_path_device captured as hipGraphs (pqp_chain.inc).  capturing: the handle's stream is in capture
    // mode - no timing events, the path solve resets its ticket counter inside the graph
    bool capturing = false;
    int opt_chain_graph = 0;
    struct ChainGraph { std::vector<unsigned char> key; hipGraphExec_t exec = nullptr; bool failed = false; bool lane_launch = false; int path_kernel = 0; unsigned long long ticket_after = 0; };
    std::vector<ChainGraph> chain_graphs;
    int warm_batch = 0, warm_n = 0;
    bool warm_stored = false;                   // the last solve wrote its final iterate to wx / wy / wye
    DevBuf wx, wy, wye, wrho, wsave, wscale;    // warm state (lane layout) + polish save area, parked Ruiz vectors (per workgroup slot)
    // work distribution of the solve kernel: ticket counter (never reset: a launch uses batch + grid tickets), cost bins of the
    // last solve and the ticket -> QP order derived from them
    DevBuf ticket, cost_key, cost_hist, order;
    DevBuf chain_d, chain_i;                    // workspace of pqp_optimize_path_device
    unsigned long long ticket_next = 0;
    long long solves = 0;                       // solve launches so far (parity selects the cost histogram being filled)
    int hist_batch = 0, hist_n = 0;             // shape of the solve whose costs cost_key / cost_hist hold (0: none)
    int opt_store_warm = 1, opt_order_by_cost = 0, opt_reserve_cus = 0, opt_stream_batch = -1, opt_carry = 0, opt_stream_staged = -1;      // (opt_stream_batch < 0: stream_batch_auto(n))
    int stream_last_batch = 0, stream_last_n = 0;      // shape of the last path_stream_kernel launch (what its workspace still holds)
    int last_path_kernel = 0;                          // pqp_path_kernel of the last pqp_path_solve* launch (pqp_last_path_kernel)
    DevBuf sm_act[2];                                  // final active sets of the exact TensionSmoother / postSmooth kernels (PQP_OPT_CARRY_CYCLES)
    int sm_act_batch[2] = {0, 0}, sm_act_n[2] = {0, 0};
    DevBuf stream_ws;                           // workspace of path_stream_kernel
    DevBuf stream_key, stream_hist, stream_order;      // PQP_OPT_ORDER_BY_COST on that kernel: phase keys, key histogram, two slot -> QP maps
    long long stream_solves = 0;                // ordered launches so far (parity selects the map being read)
    int stream_order_batch = 0, stream_order_n = 0;    // shape the map being read was built for (0: none)
    int num_cu = 0;
    int blocks_per_cu[8] = {0, 0, 0, 0, 0, 0, 0, 0};    // occupancy of the solve kernel variants [log2(nw)][cert]
    DevBuf s_ref, s_lin, s_bounds, s_scal;      // staging for the host-pointer entry points
    DevBuf s_out, s_status, s_iters, s_info, s_a, s_p, s_l, s_u, s_idx;
    // smoother QPs: banded problem data + shared sparsity (cached per type and size) + staging
    DevBuf b_pband, b_q, b_aval, b_lo, b_up, b_x, b_y, b_acol, b_trow, b_tslot, b_in[5], b_out[3], c_buf[12];
    int b_struct_type = -1, b_struct_n = -1;
};

extern "C" {

void pqp_default_params(pqp_params* p) { if (p) pqp::default_params(p); }
void pqp_production_params(pqp_params* p) { if (p) pqp::production_params(p); }
const char* pqp_last_error(void) { return g_last_error.c_str(); }
void pqp_set_last_error(const char* msg) { g_last_error = msg ? msg : ""; }      // for the other translation units of the library
const char* pqp_version(void) { return "pqp-hip 0.1 (gfx950)"; }

int pqp_create(pqp_handle** out, const pqp_params* params, int device, int max_batch, int max_n) {
    if (!out) return fail(PQP_ERR_INVALID, "pqp_create: null handle pointer");
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
        return fail(PQP_ERR_NO_DEVICE, "pqp_create: no HIP device (this library has no CPU fallback)");
    if (device < 0 || device >= count) return fail(PQP_ERR_INVALID, "pqp_create: bad device ordinal");
    PQP_HIP(hipSetDevice(device));
    pqp_handle* h = new (std::nothrow) pqp_handle();
    if (!h) return fail(PQP_ERR_INVALID, "pqp_create: out of host memory");
    h->device = device;
    if (params) h->prm = *params; else pqp::default_params(&h->prm);
    // (a failure below destroys what was built so far: pqp_destroy tolerates a partially built handle)
    auto build = [&]() -> int {
        PQP_HIP(hipDeviceGetAttribute(&h->num_cu, hipDeviceAttributeMultiprocessorCount, device));
        PQP_HIP(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
        for (int k = 0; k < pqp_handle::kEvRing; ++k) { PQP_HIP(hipEventCreate(&h->evs0[k])); PQP_HIP(hipEventCreate(&h->evs1[k])); }
        for (int k = 0; k < pqp_handle::kMarks + pqp_handle::kChainMarks; ++k) PQP_HIP(hipEventCreateWithFlags(&h->marks[k], hipEventDisableTiming));
        int rc;
        if ((rc = h->ticket.ensure(8)) || (rc = h->cost_hist.ensure(2 * pqp::kCostBins * 4))) return rc;
        if (max_batch > 0 && max_n > 0) {
            const size_t bn = (size_t)max_batch * max_n;
            if ((rc = h->wx.ensure(bn * 6 * 8)) || (rc = h->wy.ensure(bn * 6 * 8)) || (rc = h->wye.ensure((size_t)max_batch * 2 * 8)) ||
                (rc = h->wrho.ensure((size_t)max_batch * 8)))
                return rc;
        }
        return PQP_OK;
    };
    const int rc = build();
    if (rc != PQP_OK) { (void)pqp_destroy(h); return rc; }
    *out = h;
    return PQP_OK;
}

int pqp_destroy(pqp_handle* h) {
    if (!h) return PQP_OK;
    (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    // (captured chains of OTHER handles may hold this handle's workspaces and events - the smoother handle of a pair: their keys carry the
    //  allocation generation, so none of them is replayed after this)
    g_alloc_generation.fetch_add(1, std::memory_order_relaxed);
    for (auto& g : h->chain_graphs) if (g.exec) (void)hipGraphExecDestroy(g.exec);
    h->chain_graphs.clear();
    for (DevBuf* b : {&h->sm_act[0], &h->sm_act[1], &h->stream_ws, &h->stream_key, &h->stream_hist, &h->stream_order, &h->chain_d, &h->chain_i, &h->wscale, &h->ticket, &h->cost_key, &h->cost_hist, &h->order, &h->wx, &h->wy, &h->wye, &h->wrho, &h->wsave, &h->s_ref, &h->s_lin, &h->s_bounds, &h->s_scal, &h->s_out,
                      &h->s_status, &h->s_iters, &h->s_info, &h->s_a, &h->s_p, &h->s_l, &h->s_u, &h->s_idx, &h->b_pband, &h->b_q, &h->b_aval,
                      &h->b_lo, &h->b_up, &h->b_x, &h->b_y, &h->b_acol, &h->b_trow, &h->b_tslot, &h->b_in[0], &h->b_in[1], &h->b_in[2],
                      &h->b_in[3], &h->b_in[4], &h->b_out[0], &h->b_out[1], &h->b_out[2], &h->c_buf[0], &h->c_buf[1], &h->c_buf[2],
                      &h->c_buf[3], &h->c_buf[4], &h->c_buf[5], &h->c_buf[6], &h->c_buf[7], &h->c_buf[8], &h->c_buf[9], &h->c_buf[10],
                      &h->c_buf[11]})
        b->release();
    for (int k = 0; k < pqp_handle::kEvRing; ++k) { if (h->evs0[k]) (void)hipEventDestroy(h->evs0[k]); if (h->evs1[k]) (void)hipEventDestroy(h->evs1[k]); }
    for (int k = 0; k < pqp_handle::kMarks + pqp_handle::kChainMarks; ++k) if (h->marks[k]) (void)hipEventDestroy(h->marks[k]);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
    return PQP_OK;
}

int pqp_set_params(pqp_handle* h, const pqp_params* params) {
    if (!h || !params) return fail(PQP_ERR_INVALID, "pqp_set_params: null argument");
    if (params->scaling < -64 || params->scaling > 64) return fail(PQP_ERR_INVALID, "pqp_set_params: |scaling| (equilibration passes) beyond 64");
    h->prm = *params;
    return PQP_OK;
}

int pqp_set_option(pqp_handle* h, int option, int value) {
    if (!h) return fail(PQP_ERR_INVALID, "pqp_set_option: null handle");
    switch (option) {
        case PQP_OPT_STORE_WARM: h->opt_store_warm = value ? 1 : 0; return PQP_OK;
        case PQP_OPT_ORDER_BY_COST: h->opt_order_by_cost = value ? 1 : 0; h->hist_batch = 0; h->stream_order_batch = 0; return PQP_OK;
        case PQP_OPT_RESERVE_CUS: h->opt_reserve_cus = value < 0 ? 0 : value; return PQP_OK;
        case PQP_OPT_STREAM_BATCH: h->opt_stream_batch = value < 0 ? -1 : value; return PQP_OK;
        case PQP_OPT_STREAM_STAGED: h->opt_stream_staged = value < 0 ? -1 : (value ? 1 : 0); h->stream_last_batch = 0; return PQP_OK;      // (another layout: nothing to carry)
        case PQP_OPT_CARRY_CYCLES: h->opt_carry = value < 0 ? 0 : (value > 64 ? 64 : value); h->stream_last_batch = 0; h->sm_act_batch[0] = h->sm_act_batch[1] = 0; return PQP_OK;
        case PQP_OPT_CHAIN_GRAPH: h->opt_chain_graph = value == 2 ? 2 : (value ? 1 : 0); return PQP_OK;
        default: return fail(PQP_ERR_INVALID, "pqp_set_option: unknown option");
    }
}

int pqp_get_stream(pqp_handle* h, void** s) {
    if (!h || !s) return fail(PQP_ERR_INVALID, "pqp_get_stream: null argument");
    *s = (void*)h->stream;
    return PQP_OK;
}

int pqp_stream_wait(pqp_handle* h, void* other_stream) {
    if (!h) return fail(PQP_ERR_INVALID, "pqp_stream_wait: null handle");
    PQP_HIP(hipSetDevice(h->device));
    hipEvent_t ev;
    PQP_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    hipError_t e = hipEventRecord(ev, (hipStream_t)other_stream);
    if (e == hipSuccess) e = hipStreamWaitEvent(h->stream, ev, 0);
    (void)hipEventDestroy(ev);                 // released once the wait has completed
    if (e != hipSuccess) return fail(PQP_ERR_HIP, std::string("pqp_stream_wait: ") + hipGetErrorString(e));
    return PQP_OK;
}

int pqp_mark(pqp_handle* h, int slot) {
    if (!h || slot < 0 || slot >= pqp_handle::kMarks) return fail(PQP_ERR_INVALID, "pqp_mark: bad handle or slot (0..7)");
    PQP_HIP(hipSetDevice(h->device));
    PQP_HIP(hipEventRecord(h->marks[slot], h->stream));
    return PQP_OK;
}

int pqp_wait_mark(pqp_handle* h, pqp_handle* other, int slot) {
    if (!h || !other || slot < 0 || slot >= pqp_handle::kMarks) return fail(PQP_ERR_INVALID, "pqp_wait_mark: bad handle or slot (0..7)");
    PQP_HIP(hipSetDevice(h->device));
    PQP_HIP(hipStreamWaitEvent(h->stream, other->marks[slot], 0));       // (a mark never recorded counts as complete)
    return PQP_OK;
}

int pqp_sync(pqp_handle* h) {
    if (!h) return fail(PQP_ERR_INVALID, "pqp_sync: null handle");
    PQP_HIP(hipSetDevice(h->device));
    PQP_HIP(hipStreamSynchronize(h->stream));
    return PQP_OK;
}

int pqp_path_sizes(const pqp_params* params, int n, const double* s, pqp_sizes* out) {
    if (!out || n < 2) return fail(PQP_ERR_INVALID, "pqp_path_sizes: need n >= 2 and an output struct");
    pqp_params def;
    if (!params) { pqp::default_params(&def); params = &def; }
    int precise = n;   // base_solver.cpp:24-34
    if (params->rough_constraints_far_away) {
        if (!s) return fail(PQP_ERR_INVALID, "pqp_path_sizes: rough_constraints_far_away needs the arclengths");
        int lo = 0, hi = n;   // std::lower_bound(s, precise_planning_length)
        while (lo < hi) { const int mid = (lo + hi) / 2; if (s[mid] < params->precise_planning_length) lo = mid + 1; else hi = mid; }
        precise = lo;
    }
    pqp::RefIndex R{n, precise};
    out->n = n; out->state = 3 * n; out->control = n - 1; out->precise = precise; out->slack = precise + n;
    out->vars = R.vars(); out->cons = R.cons(); out->nnz_a = R.nnz_a(); out->nnz_p = R.nnz_p();
    return PQP_OK;
}

int pqp_path_pattern(pqp_handle* h, int n, int precise, int32_t* rows, int32_t* colptr, int32_t* pcols) {
    if (!h || !rows || !colptr || !pcols || n < 2 || precise < 0 || precise > n)
        return fail(PQP_ERR_INVALID, "pqp_path_pattern: bad argument");
    PQP_HIP(hipSetDevice(h->device));
    pqp::RefIndex R{n, precise};
    const size_t total = (size_t)R.nnz_a() + R.vars() + 1 + R.nnz_p();
    int rc;
    if ((rc = h->s_idx.ensure(total * 4))) return rc;
    int32_t* d_rows = h->s_idx.as<int32_t>();
    int32_t* d_colptr = d_rows + R.nnz_a();
    int32_t* d_pcols = d_colptr + R.vars() + 1;
    PQP_HIP(hipMemsetAsync(d_rows, 0xff, total * 4, h->stream));
    hipLaunchKernelGGL(pqp::path_pattern_kernel, dim3((n + 127) / 128), dim3(128), 0, h->stream, R, d_rows, d_colptr, d_pcols);
    PQP_HIP(hipGetLastError());
    PQP_HIP(hipMemcpyAsync(rows, d_rows, (size_t)R.nnz_a() * 4, hipMemcpyDeviceToHost, h->stream));
    PQP_HIP(hipMemcpyAsync(colptr, d_colptr, (size_t)(R.vars() + 1) * 4, hipMemcpyDeviceToHost, h->stream));
    PQP_HIP(hipMemcpyAsync(pcols, d_pcols, (size_t)R.nnz_p() * 4, hipMemcpyDeviceToHost, h->stream));
    PQP_HIP(hipStreamSynchronize(h->stream));
    return PQP_OK;
}

int pqp_path_assemble_device(pqp_handle* h, int batch, int n, int precise, const double* ref, const double* lin,
                             const double* bounds, const double* scal, double* a_val, double* p_val, double* lower,
                             double* upper) {
    if (!h || !ref || !bounds || !scal || !a_val || !p_val || !lower || !upper || batch < 1 || n < 2 || precise < 0 || precise > n)
        return fail(PQP_ERR_INVALID, "pqp_path_assemble: bad argument");
    PQP_HIP(hipSetDevice(h->device));
    pqp::RefIndex R{n, precise};
    const size_t lds = ((size_t)R.nnz_a() + R.nnz_p() + 2 * (size_t)R.cons()) * 8;
    const int stage = lds <= 150 * 1024 ? 1 : 0;
    if (stage && lds > 64 * 1024)
        PQP_HIP(hipFuncSetAttribute((const void*)pqp::path_assemble_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int grid = batch < 4096 ? batch : 4096;
    h->next_event_pair();
    if (!h->capturing) PQP_HIP(hipEventRecord(h->ev0, h->stream));
    hipLaunchKernelGGL(pqp::path_assemble_kernel, dim3(grid), dim3(256), stage ? lds : 0, h->stream, R, batch, ref, lin, bounds,
                       scal, h->prm, a_val, p_val, lower, upper, stage);
    PQP_HIP(hipGetLastError());
    if (!h->capturing) PQP_HIP(hipEventRecord(h->ev1, h->stream));
    h->timed = true;
    return PQP_OK;
}

int pqp_path_assemble(pqp_handle* h, int batch, int n, int precise, const double* ref, const double* lin,
                      const double* bounds, const double* scal, double* a_val, double* p_val, double* lower, double* upper) {
    if (!h || !ref || !bounds || !scal || !a_val || !p_val || !lower || !upper || batch < 1 || n < 2 || precise < 0 || precise > n)
        return fail(PQP_ERR_INVALID, "pqp_path_assemble: bad argument");
    PQP_HIP(hipSetDevice(h->device));
    pqp::RefIndex R{n, precise};
    const size_t bn = (size_t)batch * n;
    int rc;
    if ((rc = h->s_ref.ensure(bn * 5 * 8)) || (rc = h->s_bounds.ensure(bn * 6 * 8)) || (rc = h->s_scal.ensure((size_t)batch * 6 * 8)) ||
        (rc = h->s_a.ensure((size_t)batch * R.nnz_a() * 8)) || (rc = h->s_p.ensure((size_t)batch * R.nnz_p() * 8)) ||
        (rc = h->s_l.ensure((size_t)batch * R.cons() * 8)) || (rc = h->s_u.ensure((size_t)batch * R.cons() * 8)))
        return rc;
    if (lin && (rc = h->s_lin.ensure(bn * 3 * 8))) return rc;
    PQP_HIP(hipMemcpyAsync(h->s_ref.p, ref, bn * 5 * 8, hipMemcpyHostToDevice, h->stream));
    PQP_HIP(hipMemcpyAsync(h->s_bounds.p, bounds, bn * 6 * 8, hipMemcpyHostToDevice, h->stream));
    PQP_HIP(hipMemcpyAsync(h->s_scal.p, scal, (size_t)batch * 6 * 8, hipMemcpyHostToDevice, h->stream));
    if (lin) PQP_HIP(hipMemcpyAsync(h->s_lin.p, lin, bn * 3 * 8, hipMemcpyHostToDevice, h->stream));
    rc = pqp_path_assemble_device(h, batch, n, precise, h->s_ref.as<double>(), lin ? h->s_lin.as<double>() : nullptr,
                                  h->s_bounds.as<double>(), h->s_scal.as<double>(), h->s_a.as<double>(), h->s_p.as<double>(),
                                  h->s_l.as<double>(), h->s_u.as<double>());
    if (rc) return rc;
    PQP_HIP(hipMemcpyAsync(a_val, h->s_a.p, (size_t)batch * R.nnz_a() * 8, hipMemcpyDeviceToHost, h->stream));
    PQP_HIP(hipMemcpyAsync(p_val, h->s_p.p, (size_t)batch * R.nnz_p() * 8, hipMemcpyDeviceToHost, h->stream));
    PQP_HIP(hipMemcpyAsync(lower, h->s_l.p, (size_t)batch * R.cons() * 8, hipMemcpyDeviceToHost, h->stream));
    PQP_HIP(hipMemcpyAsync(upper, h->s_u.p, (size_t)batch * R.cons() * 8, hipMemcpyDeviceToHost, h->stream));
    PQP_HIP(hipStreamSynchronize(h->stream));
    return PQP_OK;
}

extern "C" hipError_t pqp_stream_launch(const pqp::lq::Args* a, int waves, void* stream);

// Large batches (PQP_OPT_STREAM_BATCH): one lane per QP, state streamed through HBM (pqp_path_lq.hpp).  Same optimum as the
// lane-per-waypoint kernel's KKT-verified polish; no warm state is kept (warm == 1 and pqp_path_get_solution need the other kernel).
static int path_stream_impl(pqp_handle* h, int batch, int n, const int32_t* n_of, const double* ref, const double* lin, const double* bounds,
                            const double* scal, int passes, double* out, int32_t* status, int32_t* iters, double* info) {
    const int waves = (batch + 63) / 64;
    int rc;
    const void* ws_before = h->stream_ws.p;
    if ((rc = h->stream_ws.ensure((size_t)waves * n * pqp::lq::kBlockDoubles * 64 * 8))) return rc;
    pqp::lq::Args a;
    std::memset(&a, 0, sizeof(a));
    a.batch = batch; a.n = n; a.passes = passes; a.n_of = n_of; a.ref = ref; a.lin = lin; a.bounds = bounds; a.scal = scal; a.out = out;
    a.status = status; a.iters = iters; a.info = info; a.ws = h->stream_ws.as<double>(); a.prm = h->prm;
    // Which of the kernel's two workspace layouts (pqp_path_lq_abi.hpp): a launch that leaves SIMDs idle - fewer than 768 wavefronts - waits for its loads, not for
    // HBM's throughput: the sweeps' records staged in LDS two waypoints ahead, +29 ... 33 % at 24 576 / 32 768 QPs of 80 waypoints; a launch that fills the chip
    // loses 3-4 % with them (profiles/r06au_*) and keeps the [field][lane] layout and the register prefetch.  A function of the shape alone: what PQP_OPT_CARRY_CYCLES
    // finds in the workspace was left there in the same layout.
    a.staged = h->opt_stream_staged >= 0 ? h->opt_stream_staged : (waves < 3 * h->num_cu ? 1 : 0);
    // PQP_OPT_CARRY_CYCLES: the workspace still holds, slot by slot, the optimum of the previous launch of this very shape
    a.carry = (h->opt_carry && !lin && h->stream_last_batch == batch && h->stream_last_n == n && h->stream_ws.p == ws_before) ? 1 : 0;       // (lin == NULL: pqp.h)
    // PQP_OPT_ORDER_BY_COST: wavefronts of QPs that ran the same phases in the handle's previous solve of the shape.  Only where it pays - batches that
    // put a wavefront on (nearly) every SIMD: below that a launch lasts as long as one wavefront's sweeps whatever its lanes do (profiles/r05g_*) - and
    // not with PQP_OPT_CARRY_CYCLES (a slot's workspace then holds the previous optimum of the QP that sat there) or inside a graph capture (host-side parity).
    const bool ordered = h->opt_order_by_cost && !h->opt_carry && !h->capturing && waves >= 3 * h->num_cu;
    if (ordered) {
        if ((rc = h->stream_key.ensure((size_t)batch * 4)) || (rc = h->stream_hist.ensure(((size_t)pqp::lq::kOrderBins + 1) * 4)) ||
            (rc = h->stream_order.ensure((size_t)2 * batch * 4)))
            return rc;
        if (h->stream_order_batch != batch || h->stream_order_n != n) {          // a shape change: stale counts, no map yet
            PQP_HIP(hipMemsetAsync(h->stream_hist.p, 0, ((size_t)pqp::lq::kOrderBins + 1) * 4, h->stream));
            a.order = nullptr;
        } else {
            a.order = h->stream_order.as<int32_t>() + (size_t)(h->stream_solves & 1) * batch;
        }
        a.key_out = h->stream_key.as<int32_t>();
        a.hist = h->stream_hist.as<int32_t>();
        a.order_next = h->stream_order.as<int32_t>() + (size_t)((h->stream_solves + 1) & 1) * batch;
    }
    h->next_event_pair();
    if (!h->capturing) PQP_HIP(hipEventRecord(h->ev0, h->stream));
    PQP_HIP(pqp_stream_launch(&a, waves, (void*)h->stream));      // path_stream_kernel lives in its own translation unit (pqp_path_stream.hip)
    if (!h->capturing) PQP_HIP(hipEventRecord(h->ev1, h->stream));
    h->timed = true;
    if (ordered) { h->stream_solves += 1; h->stream_order_batch = batch; h->stream_order_n = n; }
    h->stream_last_batch = batch; h->stream_last_n = n;
    h->warm_batch = batch; h->warm_n = n;
    h->warm_stored = false;
    h->last_path_kernel = PQP_KERNEL_LANE_PER_QP;
    return PQP_OK;
}

// PQP_OPT_STREAM_BATCH "auto": the batch size from which the lane-per-QP kernel is the faster one, by measurement on one MI355X
// (profiles/r05a_crossover_n80.txt, r05a_crossover_n120.txt: 18.7 k QPs at 80 waypoints, ~45 k at 120).  A lone wavefront of that kernel
// takes sweeps x n waypoint steps whatever the batch (5.5 ms at n = 80, 10.5 ms at n = 120), the lane-per-waypoint kernel's time per QP hardly
// depends on n: the crossover grows like n^2.
// Beyond 256 waypoints the lane-per-waypoint kernel runs two wavefronts per SIMD on half the registers and takes 4.1 ... 4.9 us per QP instead
// of 1.0: the crossover is back at 16.7 k / 22.5 k / 32 k QPs at 300 / 400 / 512 waypoints (profiles/r05t_crossover_long_paths.txt): 64 n.
// the lane-per-waypoint kernels, by wavefronts per QP (pqp_path_solve.hip compiled with -DPQP_NW=1 / 2 / 4 / 8); cert: with the in-loop
// infeasibility certificate
extern "C" {
const void* pqp_path_solve_fn_nw1(int cert);
const void* pqp_path_solve_fn_nw2(int cert);
const void* pqp_path_solve_fn_nw4(int cert);
const void* pqp_path_solve_fn_nw8(int cert);
}

// PQP_OPT_STREAM_BATCH's default: where the lane-per-QP kernel overtakes the lane-per-waypoint kernel, one launch after the other on one MI355X, remeasured on
// round 6's kernels (profiles/r06ay_crossover_hybrid.txt, r06az_crossover_other_n.txt).  The lane-per-waypoint kernel's rate steps down with its workgroup width
// (3.5 M paths/s up to 128 waypoints, ~1.1 M up to 256, 0.2 M beyond), a lone wavefront of the other takes sweeps x n waypoint steps: measured crossovers
// 15 k QPs at 80 waypoints, 20 k at 100, 29 k at 120 | 13 k at 160, 34 k at 200, 49 k at 256 | 11.5 k at 300, 24.5 k at 512.
static int stream_batch_auto(int n) {
    if (n > 256) return 48 * n;
    if (n > 128) return (int)(0.75 * n * n);
    const double r = n > 80 ? (double)n / 80.0 : 1.0;
    return (int)(15360.0 * r * sqrt(r));
}
extern "C" int pqp_stream_batch_default(int n) { return n < 2 ? 0 : stream_batch_auto(n); }

static int path_solve_impl(pqp_handle* h, int batch, int n, const int32_t* n_of, const double* ref, const double* lin, const double* bounds,
                           const double* scal, int passes, int warm, double* out, int32_t* status, int32_t* iters, double* info) {
    if (!h || !ref || !bounds || !scal || !out || batch < 1 || n < 2 || passes < 0)
        return fail(PQP_ERR_INVALID, "pqp_path_solve: bad argument");
    PQP_HIP(hipSetDevice(h->device));
    // lane-per-QP kernel: large batches of a caller that keeps no warm state (PQP_OPT_STREAM_BATCH), and every path of more than 512
    // waypoints (the lane-per-waypoint kernel's workgroup ends there; a reference path of 80 m at 0.15 m spacing has 530:
    // reference_path_impl.cpp:321-336) - those with any batch size and without warm state
    // (warm == 1 beyond 512 waypoints: the QP around `lin` is solved cold - its optimum is unique, a warm start only saves iterations -
    //  so BaseSolver::solve + updateProblemFormulationAndSolve work at any size; and there also a handle in the reference's ADMM setting
    //  gets the exact optimum: zero residuals meet OSQP's termination test at any eps)
    const int stream_from = h->opt_stream_batch < 0 ? stream_batch_auto(n) : h->opt_stream_batch;
    if ((n > 512 && (!warm || lin)) || (h->prm.polish != 0 && !warm && !h->opt_store_warm && stream_from > 0 && batch >= stream_from))
        return path_stream_impl(h, batch, n, n_of, ref, lin, bounds, scal, passes, out, status, iters, info);
    // PQP_OPT_CARRY_CYCLES on this kernel: a cold call (warm == 0, lin == NULL) of the shape of the handle's previous solve starts from
    // the final iterate, equilibration and active set that solve left in the warm state - the same scenarios one planning cycle later.
    // (With waypoint counts per QP the state is kept per waypoint: a path that grew or shrank by a few waypoints since the previous cycle
    //  starts its common waypoints from where they were and the new ones from whatever the slot last held there - zero at first.)
    // (only on a handle whose polish returns the exact optimum: with polish == 0 - the reference's ADMM setting - a promoted call would end at an
    //  eps-accurate point that depends on the slot's previous QP, and pqp.h promises the cold solve's optimum to the 1e-7 of the KKT test)
    // (value k >= 2, "tails": only the QPs that were among the most expensive 1 / k of the previous launch - by the cost keys PQP_OPT_ORDER_BY_COST
    //  keeps - start from there, the others start cold: what bounds a launch is its slowest QPs)
    int carry_tails = 0;
    if (h->opt_carry && h->prm.polish != 0 && !warm && !lin && h->warm_stored && h->warm_batch == batch && h->warm_n == n &&
        (h->opt_carry < 2 || (h->opt_order_by_cost && h->hist_batch == batch && h->hist_n == n))) { warm = 1; carry_tails = h->opt_carry >= 2 ? h->opt_carry : 0; }
    if (n > 512) return fail(PQP_ERR_CAPACITY, "pqp_path_solve: warm == 1 beyond 512 waypoints needs the linearisation point (`lin`): the lane-per-QP kernel keeps no warm state");
    if (warm && (h->warm_batch != batch || h->warm_n != n || !h->warm_stored))
        return fail(PQP_ERR_INVALID, "pqp_path_solve: warm == 1 needs a previous solve with the same batch and n (with PQP_OPT_STORE_WARM on)");
    const size_t bn = (size_t)batch * n;
    int rc;
    if (!warm) {
        if ((rc = h->wx.ensure(bn * 6 * 8)) || (rc = h->wy.ensure(bn * 6 * 8)) || (rc = h->wye.ensure((size_t)batch * 2 * 8)) ||
            (rc = h->wrho.ensure((size_t)batch * 8)))
            return rc;
    }
    pqp::PathSolveArgs a;
    std::memset(&a, 0, sizeof(a));
    a.batch = batch; a.n = n; a.n_of = n_of; a.passes = passes; a.warm = warm ? 1 : 0;
    a.ref = ref; a.lin = lin; a.bounds = bounds; a.scal = scal; a.out = out;
    a.status = status; a.iters = iters; a.info = info;
    a.wx = h->wx.as<double>(); a.wy = h->wy.as<double>(); a.wye = h->wye.as<double>(); a.wrho = h->wrho.as<double>();
    a.prm = h->prm;
    // Long paths: the polished point's residuals in the transition rows add up along the path - the rows are a discrete double integrator, an error of
    // 1e-9 per row in (psi, kappa) is 1e-4 in l after 300 waypoints - so an accepted point gets one more refinement solve per pass beyond 128 waypoints, three beyond 256 (each shrinks the error ~10x: 4.6e-3 -> 4.4e-4 -> 4.1e-5 -> 3.3e-6 on
    // the worst path of 300 waypoints; the kernels of up to 128 waypoints do not compile the feature in: Ctx::kFinalRefine) (a 100x
    // tighter acceptance test instead leaves a few QPs in 30 000 unverifiable: 350-1000 solves, configs[4] halved).  Measured against the converged C oracle over 32 768 QPs per shape (profiles/r05o_*): before, 7 paths of 200 waypoints
    // and 117 of 300 were 3e-5 ... 4.6e-3 off (the lane-per-QP kernel, whose roll-out satisfies the rows exactly: 5e-6).
    // ... and the production setting's intervals by path length (pqp_defaults.hpp)
    pqp::resolve_path_params(&a.prm, n);
    int nw = 1, lg = 0;
    while (64 * nw < n) { nw *= 2; lg += 1; }           // one waypoint per lane: T = 64 * nw >= n threads per QP
    const int T_lanes = 64 * nw;
    const bool save_lds = nw <= pqp::kSaveLdsMaxNw;
    const size_t lds = (size_t)pqp::ShLayout{T_lanes}.total(save_lds, save_lds) * 8;
    // two variants of every kernel: with and without OSQP's primal infeasibility certificate (prm.eps_prim_inf > 0)
    const bool cert = h->prm.eps_prim_inf > 0.0 && h->prm.prim_inf_after <= 0;
    const void* fn = nullptr;
    switch (nw) {       // (pqp_path_solve.hip, one translation unit per width)
        case 1: fn = pqp_path_solve_fn_nw1(cert); break;
        case 2: fn = pqp_path_solve_fn_nw2(cert); break;
        case 4: fn = pqp_path_solve_fn_nw4(cert); break;
        default: fn = pqp_path_solve_fn_nw8(cert); break;
    }
    if (lds > 64 * 1024) PQP_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    // persistent workgroups: as many as the chip holds at once (a surplus one would only wait for a free slot), each with its own
    // save area; they draw the QPs from the ticket counter
    int& per_cu = h->blocks_per_cu[2 * lg + (cert ? 1 : 0)];
    if (per_cu == 0) {
        PQP_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, 64 * nw, lds));
        if (per_cu < 1) per_cu = 1;
    }
    const int cus = h->num_cu - h->opt_reserve_cus > 1 ? h->num_cu - h->opt_reserve_cus : 1;
    const long long resident = (long long)per_cu * cus;
    const int grid = (int)(batch < resident ? batch : resident);
    if (!save_lds) {          // more than 256 lanes per QP: the save area and the parked Ruiz vectors live in the workgroup slot's global memory
        if ((rc = h->wsave.ensure((size_t)grid * T_lanes * PQP_SAVE_STRIDE * 8))) return rc;
        if ((rc = h->wscale.ensure((size_t)grid * T_lanes * 18 * 8))) return rc;
    }
    a.wsave = h->wsave.as<double>();
    a.wscale = h->wscale.as<double>();
    a.store_warm = (h->opt_store_warm || h->opt_carry) ? 1 : 0;
    a.carry_tails = carry_tails;
    a.carry_k = h->opt_carry;
    a.ticket = h->ticket.as<unsigned long long>();
    a.ticket_base = h->ticket_next;
    // inside a captured graph the launch cannot take its ticket base from a host counter that moves between replays: the graph resets the
    // device counter itself and every replay starts at 0 (pqp_chain.inc puts the host counter where the replay leaves the device one)
    if (h->capturing) { PQP_HIP(hipMemsetAsync(h->ticket.p, 0, 8, h->stream)); a.ticket_base = 0; }
    // Host-side bookkeeping of the launch (ticket base of the next launch, launch parity, shape of the cost histogram) is committed
    // only after the launch has been accepted: a failing step below (allocation, memset, event, launch) leaves the device ticket
    // counter and the host's idea of it in step.
    if (h->opt_order_by_cost) {
        // most expensive QPs first, by what they cost in this handle's previous solve of the same shape (a planner re-solves
        // nearly the same scenarios cycle after cycle); results do not depend on the order
        if ((rc = h->cost_key.ensure((size_t)batch * 4)) || (rc = h->order.ensure((size_t)2 * batch * 4))) return rc;
        // two order arrays: the one this launch reads (written by the previous launch's last workgroup) and the one it writes
        int32_t* order_read = h->order.as<int32_t>() + (size_t)(h->solves & 1) * batch;
        int32_t* order_write = h->order.as<int32_t>() + (size_t)((h->solves + 1) & 1) * batch;
        if (h->hist_batch == batch && h->hist_n == n) a.order = order_read;
        else { h->hist_batch = 0; h->hist_n = 0; PQP_HIP(hipMemsetAsync(h->cost_hist.p, 0, (pqp::kCostBins + 1) * 4, h->stream)); }     // (a shape change: stale counts)
        a.cost_key = h->cost_key.as<int32_t>();
        a.cost_hist = h->cost_hist.as<int32_t>();
        a.order_next = order_write;
    }
    h->next_event_pair();
    if (!h->capturing) PQP_HIP(hipEventRecord(h->ev0, h->stream));
    void* kargs[] = {(void*)&a};
    hipError_t le = hipLaunchKernel(fn, dim3(grid), dim3(64 * nw), kargs, lds, h->stream);
    if (le == hipSuccess) le = hipGetLastError();
    if (le != hipSuccess) {
        h->hist_batch = 0; h->hist_n = 0;          // (the histogram may have been cleared for a launch that never ran)
        return fail(PQP_ERR_HIP, std::string("hipLaunchKernel(path_solve_kernel): ") + hipGetErrorString(le));
    }
    h->ticket_next = a.ticket_base + (unsigned long long)batch + (unsigned long long)grid;
    if (h->opt_order_by_cost) { h->hist_batch = batch; h->hist_n = n; }
    h->solves += 1;
    if (!h->capturing) PQP_HIP(hipEventRecord(h->ev1, h->stream));
    h->timed = true;
    h->warm_batch = batch; h->warm_n = n;
    h->warm_stored = h->opt_store_warm != 0 || h->opt_carry != 0;
    h->last_path_kernel = PQP_KERNEL_LANE_PER_WAYPOINT;
    return PQP_OK;
}

int pqp_path_solve_device(pqp_handle* h, int batch, int n, const double* ref, const double* lin, const double* bounds,
                          const double* scal, int passes, int warm, double* out, int32_t* status, int32_t* iters,
                          double* info) {
    return path_solve_impl(h, batch, n, nullptr, ref, lin, bounds, scal, passes, warm, out, status, iters, info);
}

int pqp_path_solve_var_device(pqp_handle* h, int batch, int n_max, const int32_t* n_of, const double* ref, const double* lin,
                              const double* bounds, const double* scal, int passes, int warm, double* out, int32_t* status,
                              int32_t* iters, double* info) {
    if (!n_of) return fail(PQP_ERR_INVALID, "pqp_path_solve_var: n_of is null");
    return path_solve_impl(h, batch, n_max, n_of, ref, lin, bounds, scal, passes, warm, out, status, iters, info);
}

static int path_solve_host_body(pqp_handle* h, int batch, int n, const int32_t* n_of, const double* ref, const double* lin, const double* bounds,
                                const double* scal, int passes, int warm, double* out, int32_t* status, int32_t* iters, double* info,
                                std::vector<int32_t>& counts);

static int path_solve_host(pqp_handle* h, int batch, int n, const int32_t* n_of, const double* ref, const double* lin, const double* bounds,
                           const double* scal, int passes, int warm, double* out, int32_t* status, int32_t* iters, double* info) {
    if (!h || !ref || !bounds || !scal || !out || batch < 1 || n < 2 || passes < 0)
        return fail(PQP_ERR_INVALID, "pqp_path_solve: bad argument");
    PQP_HIP(hipSetDevice(h->device));
    std::vector<int32_t> counts;            // outlives every copy enqueued from it: an early error return synchronises first
    const int rc = path_solve_host_body(h, batch, n, n_of, ref, lin, bounds, scal, passes, warm, out, status, iters, info, counts);
    if (rc != PQP_OK) (void)hipStreamSynchronize(h->stream);
    return rc;
}

static int path_solve_host_body(pqp_handle* h, int batch, int n, const int32_t* n_of, const double* ref, const double* lin, const double* bounds,
                                const double* scal, int passes, int warm, double* out, int32_t* status, int32_t* iters, double* info,
                                std::vector<int32_t>& counts) {
    const size_t bn = (size_t)batch * n;
    int rc;
    // A box with lower > upper bound: OSQP refuses such data at setup (OsqpEigen's initSolver() fails and BaseSolver::solve returns
    // false, base_solver.cpp:76-80).  The host-pointer entry points see the data anyway: such a QP is not launched (waypoint count 0)
    // and comes back PQP_STATUS_PRIMAL_INFEASIBLE.  (The device-pointer entry points do not validate: there the row would be
    // pinned to its upper bound.)
    bool any_invalid = false;
    for (int q = 0; q < batch; ++q) {
        const int cnt = n_of ? n_of[q] : n;
        bool bad = false;
        for (int i = 0; i < cnt && i < n && !bad; ++i) {
            const double* b = bounds + ((size_t)q * n + i) * PQP_BOUNDS_STRIDE;
            bad = b[0] > b[1] || b[2] > b[3] || b[4] > b[5];
        }
        if (bad && !any_invalid) {
            counts.assign(batch, n);
            if (n_of) counts.assign(n_of, n_of + batch);
            any_invalid = true;
        }
        if (bad) counts[q] = -1;
    }
    if (any_invalid) n_of = counts.data();
    if (n_of) {
        if ((rc = h->c_buf[11].ensure((size_t)batch * 4))) return rc;
        PQP_HIP(hipMemcpyAsync(h->c_buf[11].p, n_of, (size_t)batch * 4, hipMemcpyHostToDevice, h->stream));
    }
    if ((rc = h->s_ref.ensure(bn * 5 * 8)) || (rc = h->s_bounds.ensure(bn * 6 * 8)) || (rc = h->s_scal.ensure((size_t)batch * 6 * 8)) ||
        (rc = h->s_out.ensure(bn * 7 * 8)) || (rc = h->s_status.ensure((size_t)batch * 4)) || (rc = h->s_iters.ensure((size_t)batch * 4)) ||
        (rc = h->s_info.ensure((size_t)batch * PQP_INFO_STRIDE * 8)))
        return rc;
    if (lin && (rc = h->s_lin.ensure(bn * 3 * 8))) return rc;
    PQP_HIP(hipMemcpyAsync(h->s_ref.p, ref, bn * 5 * 8, hipMemcpyHostToDevice, h->stream));
    PQP_HIP(hipMemcpyAsync(h->s_bounds.p, bounds, bn * 6 * 8, hipMemcpyHostToDevice, h->stream));
    PQP_HIP(hipMemcpyAsync(h->s_scal.p, scal, (size_t)batch * 6 * 8, hipMemcpyHostToDevice, h->stream));
    if (lin) PQP_HIP(hipMemcpyAsync(h->s_lin.p, lin, bn * 3 * 8, hipMemcpyHostToDevice, h->stream));
    if (n_of) PQP_HIP(hipMemsetAsync(h->s_out.p, 0, bn * 7 * 8, h->stream));        // rows beyond a QP's own count are not written
    rc = path_solve_impl(h, batch, n, n_of ? h->c_buf[11].as<int32_t>() : nullptr, h->s_ref.as<double>(), lin ? h->s_lin.as<double>() : nullptr,
                         h->s_bounds.as<double>(), h->s_scal.as<double>(), passes, warm, h->s_out.as<double>(), h->s_status.as<int32_t>(),
                         h->s_iters.as<int32_t>(), h->s_info.as<double>());
    if (rc) return rc;
    PQP_HIP(hipMemcpyAsync(out, h->s_out.p, bn * 7 * 8, hipMemcpyDeviceToHost, h->stream));
    if (status) PQP_HIP(hipMemcpyAsync(status, h->s_status.p, (size_t)batch * 4, hipMemcpyDeviceToHost, h->stream));
    if (iters) PQP_HIP(hipMemcpyAsync(iters, h->s_iters.p, (size_t)batch * 4, hipMemcpyDeviceToHost, h->stream));
    if (info) PQP_HIP(hipMemcpyAsync(info, h->s_info.p, (size_t)batch * PQP_INFO_STRIDE * 8, hipMemcpyDeviceToHost, h->stream));
    PQP_HIP(hipStreamSynchronize(h->stream));
    if (any_invalid && status)
        for (int q = 0; q < batch; ++q)
            if (counts[q] < 0) status[q] = PQP_STATUS_PRIMAL_INFEASIBLE;
    return PQP_OK;
}

int pqp_path_solve(pqp_handle* h, int batch, int n, const double* ref, const double* lin, const double* bounds,
                   const double* scal, int passes, int warm, double* out, int32_t* status, int32_t* iters, double* info) {
    return path_solve_host(h, batch, n, nullptr, ref, lin, bounds, scal, passes, warm, out, status, iters, info);
}

int pqp_path_solve_var(pqp_handle* h, int batch, int n_max, const int32_t* n_of, const double* ref, const double* lin, const double* bounds,
                       const double* scal, int passes, int warm, double* out, int32_t* status, int32_t* iters, double* info) {
    if (!n_of) return fail(PQP_ERR_INVALID, "pqp_path_solve_var: n_of is null");
    return path_solve_host(h, batch, n_max, n_of, ref, lin, bounds, scal, passes, warm, out, status, iters, info);
}

int pqp_path_get_solution(pqp_handle* h, int batch, int n, int precise, double* x, double* y) {
    if (!h || batch < 1 || n < 2 || precise < 0 || precise > n) return fail(PQP_ERR_INVALID, "pqp_path_get_solution: bad argument");
    if (h->warm_batch != batch || h->warm_n != n || !h->warm_stored)
        return fail(PQP_ERR_INVALID, "pqp_path_get_solution: no solve of that shape on this handle (or PQP_OPT_STORE_WARM is off)");
    PQP_HIP(hipSetDevice(h->device));
    pqp::RefIndex R{n, precise};
    int rc;
    if ((rc = h->s_a.ensure((size_t)batch * R.vars() * 8)) || (rc = h->s_l.ensure((size_t)batch * R.cons() * 8))) return rc;
    PQP_HIP(hipMemsetAsync(h->s_a.p, 0, (size_t)batch * R.vars() * 8, h->stream));
    PQP_HIP(hipMemsetAsync(h->s_l.p, 0, (size_t)batch * R.cons() * 8, h->stream));
    const int total = batch * n;
    hipLaunchKernelGGL(pqp::path_gather_solution, dim3((total + 255) / 256), dim3(256), 0, h->stream, R, batch, h->wx.as<double>(),
                       h->wy.as<double>(), h->wye.as<double>(), x ? h->s_a.as<double>() : nullptr, y ? h->s_l.as<double>() : nullptr);
    PQP_HIP(hipGetLastError());
    if (x) PQP_HIP(hipMemcpyAsync(x, h->s_a.p, (size_t)batch * R.vars() * 8, hipMemcpyDeviceToHost, h->stream));
    if (y) PQP_HIP(hipMemcpyAsync(y, h->s_l.p, (size_t)batch * R.cons() * 8, hipMemcpyDeviceToHost, h->stream));
    PQP_HIP(hipStreamSynchronize(h->stream));
    return PQP_OK;
}

int pqp_constrain_angle_device(pqp_handle* h, int count, const double* in, double* out) {
    if (!h || !in || !out || count < 1) return fail(PQP_ERR_INVALID, "pqp_constrain_angle: bad argument");
    PQP_HIP(hipSetDevice(h->device));
    hipLaunchKernelGGL(pqp::constrain_angle_kernel, dim3((count + 255) / 256), dim3(256), 0, h->stream, count, in, out);
    PQP_HIP(hipGetLastError());
    return PQP_OK;
}

int pqp_last_path_kernel(pqp_handle* h) {
    if (!h) return fail(PQP_ERR_INVALID, "pqp_last_path_kernel: null handle");
    return h->last_path_kernel;
}

int pqp_last_kernel_ms(pqp_handle* h, float* ms) {
    if (!h || !ms) return fail(PQP_ERR_INVALID, "pqp_last_kernel_ms: null argument");
    if (!h->timed) return fail(PQP_ERR_INVALID, "pqp_last_kernel_ms: nothing was launched yet");
    PQP_HIP(hipSetDevice(h->device));
    PQP_HIP(hipEventSynchronize(h->ev1));
    PQP_HIP(hipEventElapsedTime(ms, h->ev0, h->ev1));
    return PQP_OK;
}

int pqp_kernel_ms_history(pqp_handle* h, float* ms, int count) {
    if (!h || !ms || count < 1) return fail(PQP_ERR_INVALID, "pqp_kernel_ms_history: bad argument");
    if (count > pqp_handle::kEvRing || (long long)count > h->ev_count)
        return fail(PQP_ERR_INVALID, "pqp_kernel_ms_history: more launches asked for than the ring holds (256) or were made");
    PQP_HIP(hipSetDevice(h->device));
    PQP_HIP(hipStreamSynchronize(h->stream));
    for (int k = 0; k < count; ++k) {          // oldest of the requested launches first
        const long long idx = (h->ev_count - count + k) % pqp_handle::kEvRing;
        PQP_HIP(hipEventElapsedTime(ms + k, h->evs0[idx], h->evs1[idx]));
    }
    return PQP_OK;
}

// ---------------------------------------------------------------------------------------------------------
// smoother QPs
// ---------------------------------------------------------------------------------------------------------
namespace {
enum { SM_TENSION2 = 0, SM_TENSION = 1, SM_POST = 2 };

struct SmShape { int nv, nc, bw, pbw, stride; };
SmShape sm_shape(int type, int n) {
    if (type == SM_TENSION2) return {4 * n - 1, 3 * (n - 1) + 2, 4, 4, 4};
    if (type == SM_TENSION) return {3 * n, 3 * n, 9, 9, 3};
    return {3 * n, 3 * n - 2, 3, 0, 3};
}

// shared sparsity of a smoother QP in the interleaved variable order: integer host logic (like pqp_path_sizes)
int sm_upload_structure(pqp_handle* h, int type, int n) {
    if (h->b_struct_type == type && h->b_struct_n == n) return PQP_OK;
    // a host -> device copy from vectors that go out of scope + a synchronise: not something a capturing stream may do.  The capture of
    // pqp_optimize_path_device is abandoned cleanly (the body fails, the call falls back to plain launches and never captures these arguments again:
    // a chain whose smoothers alternate between two structure types on the generic core uploads on every call)
    if (h->capturing) return fail(PQP_ERR_INVALID, "smoother structure upload inside a graph capture");
    const SmShape sh = sm_shape(type, n);
    std::vector<int> acol((size_t)sh.nc * pqp::kRMax, -1), trow((size_t)sh.nv * pqp::kCMax, -1), tslot((size_t)sh.nv * pqp::kCMax, 0);
    auto row = [&](int r, int c0, int c1, int c2) { int* a = &acol[(size_t)r * pqp::kRMax]; a[0] = c0; a[1] = c1; a[2] = c2; };
    if (type == SM_TENSION2) {
        for (int i = 0; i < n - 1; ++i) {
            row(i, 4 * (i + 1), 4 * i, 4 * i + 2);
            row(n - 1 + i, 4 * (i + 1) + 1, 4 * i + 1, 4 * i + 2);
            row(2 * (n - 1) + i, 4 * (i + 1) + 2, 4 * i + 2, 4 * i + 3);
        }
        row(3 * (n - 1), 0, -1, -1);
        row(3 * (n - 1) + 1, 1, -1, -1);
    } else if (type == SM_TENSION) {
        for (int i = 0; i < n; ++i) { row(i, 3 * i, 3 * i + 2, -1); row(n + i, 3 * i + 1, 3 * i + 2, -1); row(2 * n + i, 3 * i + 2, -1, -1); }
    } else {
        for (int i = 0; i < n; ++i) row(i, 3 * i, -1, -1);
        for (int i = 0; i < n - 1; ++i) { row(n + i, 3 * (i + 1), 3 * i, 3 * i + 1); row(2 * n - 1 + i, 3 * (i + 1) + 1, 3 * i + 1, 3 * i + 2); }
    }
    std::vector<int> fill(sh.nv, 0);
    for (int r = 0; r < sh.nc; ++r)
        for (int s = 0; s < pqp::kRMax; ++s) {
            const int c = acol[(size_t)r * pqp::kRMax + s];
            if (c < 0) continue;
            if (fill[c] >= pqp::kCMax) return fail(PQP_ERR_INVALID, "smoother structure: column overflow");
            trow[(size_t)c * pqp::kCMax + fill[c]] = r; tslot[(size_t)c * pqp::kCMax + fill[c]] = s; ++fill[c];
        }
    int rc;
    if ((rc = h->b_acol.ensure(acol.size() * 4)) || (rc = h->b_trow.ensure(trow.size() * 4)) || (rc = h->b_tslot.ensure(tslot.size() * 4))) return rc;
    PQP_HIP(hipMemcpyAsync(h->b_acol.p, acol.data(), acol.size() * 4, hipMemcpyHostToDevice, h->stream));
    PQP_HIP(hipMemcpyAsync(h->b_trow.p, trow.data(), trow.size() * 4, hipMemcpyHostToDevice, h->stream));
    PQP_HIP(hipMemcpyAsync(h->b_tslot.p, tslot.data(), tslot.size() * 4, hipMemcpyHostToDevice, h->stream));
    PQP_HIP(hipStreamSynchronize(h->stream));      // the vectors go out of scope
    h->b_struct_type = type; h->b_struct_n = n;
    return PQP_OK;
}

// assemble (already enqueued by the caller into b_pband ...) -> banded ADMM solve -> finish.  All device pointers.
int sm_solve(pqp_handle* h, int type, int batch, int n, int32_t* status, int32_t* iters, double* info) {
    const SmShape sh = sm_shape(type, n);
    pqp::BandedQpArgs a;
    std::memset(&a, 0, sizeof(a));
    a.batch = batch; a.nv = sh.nv; a.nc = sh.nc; a.bw = sh.bw; a.pbw = sh.pbw;
    a.pband = h->b_pband.as<double>(); a.q = h->b_q.as<double>(); a.acol = h->b_acol.as<int>(); a.aval = h->b_aval.as<double>();
    a.trow = h->b_trow.as<int>(); a.tslot = h->b_tslot.as<int>(); a.lo = h->b_lo.as<double>(); a.up = h->b_up.as<double>();
    a.x = h->b_x.as<double>(); a.y = h->b_y.as<double>(); a.status = status; a.iters = iters; a.info = info; a.prm = h->prm;
    pqp::resolve_banded_params(&a.prm);
    // the row data of A, the index lists and q staged in LDS once per QP (256-lane kernels: always - two of them still share a CU's LDS up
    // to 80 KB each; 512-lane kernels: when it fits; 1024-lane kernels: never)
    const size_t lds0 = (size_t)pqp::BqLayout{sh.nv, sh.nc, sh.bw}.total(false) * 8, lds1 = (size_t)pqp::BqLayout{sh.nv, sh.nc, sh.bw}.total(true) * 8;
    if (lds0 > 160 * 1024) return fail(PQP_ERR_CAPACITY, "smoother QP too large for one CU's LDS");
    const int nbb = pqp::BqLayout{sh.nv, sh.nc, sh.bw}.nbb();
    const int threads = 64 * ((nbb + 63) / 64);        // one lane per (padded) variable
    if (threads > 1024) return fail(PQP_ERR_CAPACITY, "smoother QP has more than 1024 variables");
    const bool stage = threads <= 512 && lds1 <= 160 * 1024;
    const size_t lds = stage ? lds1 : lds0;
    const void* fn = nullptr;
#define PQP_BQ_PICK(BB) fn = (threads <= 256 && stage) ? (const void*)pqp::banded_solve_kernel<BB, 256, true> : threads <= 512 ? (stage ? (const void*)pqp::banded_solve_kernel<BB, 512, true> : (const void*)pqp::banded_solve_kernel<BB, 512, false>) : (const void*)pqp::banded_solve_kernel<BB, 1024, false>
    switch (sh.bw) {
        case 3: PQP_BQ_PICK(3); break;
        case 4: PQP_BQ_PICK(4); break;
        case 9: PQP_BQ_PICK(9); break;
        default: return fail(PQP_ERR_INVALID, "unsupported smoother block size");
    }
#undef PQP_BQ_PICK
    if (lds > 64 * 1024) PQP_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    h->next_event_pair();
    if (!h->capturing) PQP_HIP(hipEventRecord(h->ev0, h->stream));
    void* kargs[] = {(void*)&a};
    PQP_HIP(hipLaunchKernel(fn, dim3(batch), dim3(threads), kargs, lds, h->stream));
    PQP_HIP(hipGetLastError());
    if (!h->capturing) PQP_HIP(hipEventRecord(h->ev1, h->stream));
    h->timed = true;
    return PQP_OK;
}

// does the generic banded core hold a smoother QP of this size (its vectors, factor rows and row data in one CU's LDS, one lane per padded variable)?
bool sm_generic_fits(int type, int n) {
    const SmShape sh = sm_shape(type, n);
    const pqp::BqLayout lay{sh.nv, sh.nc, sh.bw};
    return (size_t)lay.total(false) * 8 <= 160 * 1024 && 64 * ((lay.nbb() + 63) / 64) <= 1024;
}

int sm_alloc(pqp_handle* h, int type, int batch, int n) {
    const SmShape sh = sm_shape(type, n);
    int rc;
    if ((rc = h->b_pband.ensure((size_t)batch * (sh.pbw + 1) * sh.nv * 8)) || (rc = h->b_q.ensure((size_t)batch * sh.nv * 8)) ||
        (rc = h->b_aval.ensure((size_t)batch * sh.nc * pqp::kRMax * 8)) || (rc = h->b_lo.ensure((size_t)batch * sh.nc * 8)) ||
        (rc = h->b_up.ensure((size_t)batch * sh.nc * 8)) || (rc = h->b_x.ensure((size_t)batch * sh.nv * 8)) || (rc = h->b_y.ensure((size_t)batch * sh.nc * 8)))
        return rc;
    return sm_upload_structure(h, type, n);
}
}  // namespace

// TensionSmoother2::osqpSmooth (tension_smoother_2.cpp:20-72), device pointers, all lists [batch][n]; n_of [batch] (device) or nullptr
static int smooth_tension2_impl(pqp_handle* h, int batch, int n, const int32_t* n_of, const double* x_list, const double* y_list,
                                const double* angle_list, const double* k_list, const double* s_list, double* out_x, double* out_y, double* out_s,
                                int32_t* status, int32_t* iters, double* info) {
    if (!h || !x_list || !y_list || !angle_list || !k_list || !s_list || !out_x || !out_y || !out_s || batch < 1 || n < 3)
        return fail(PQP_ERR_INVALID, "pqp_smooth_tension2: bad argument");
    PQP_HIP(hipSetDevice(h->device));
    int rc;
    if (h->prm.polish != 0 || !sm_generic_fits(SM_TENSION2, n)) {
        // exact optima asked for (or more points than the generic core holds: 4 n variables on at most 1024 lanes, tension_smoother_2.cpp:20-72 has
        // no cap): the QP has equality rows only - its optimum by one Riccati sweep per scenario (tension2_exact_kernel)
        if (!status) return fail(PQP_ERR_INVALID, "pqp_smooth_tension2: status is null");
        if ((rc = h->b_pband.ensure((size_t)batch * n * 5 * 8)) || (rc = h->b_aval.ensure((size_t)batch * n * 6 * 8))) return rc;
        hipLaunchKernelGGL(pqp::tension2_stage_kernel, dim3((batch * n + 255) / 256), dim3(256), 0, h->stream, batch, n, n_of, x_list, y_list, angle_list, k_list,
                           s_list, h->b_aval.as<double>());
        PQP_HIP(hipGetLastError());
        h->next_event_pair();
        if (!h->capturing) PQP_HIP(hipEventRecord(h->ev0, h->stream));
        hipLaunchKernelGGL(pqp::tension2_exact_kernel, dim3((batch + 63) / 64), dim3(64), 0, h->stream, batch, n, n_of, x_list, y_list,
                           h->prm.tension2_deviation_weight, h->prm.tension2_curvature_weight, h->prm.tension2_curvature_rate_weight, h->b_aval.as<double>(),
                           h->b_pband.as<double>(), out_x, out_y, out_s, status, iters, info);
        PQP_HIP(hipGetLastError());
        if (!h->capturing) PQP_HIP(hipEventRecord(h->ev1, h->stream));
        h->timed = true;
        return PQP_OK;
    }
    if ((rc = sm_alloc(h, SM_TENSION2, batch, n))) return rc;
    const int total = batch * n;
    hipLaunchKernelGGL(pqp::tension2_assemble_kernel, dim3((total + 255) / 256), dim3(256), 0, h->stream, batch, n, n_of, x_list, y_list, angle_list,
                       k_list, s_list, h->prm.tension2_deviation_weight, h->prm.tension2_curvature_weight, h->prm.tension2_curvature_rate_weight,
                       h->b_pband.as<double>(), h->b_q.as<double>(), h->b_aval.as<double>(), h->b_lo.as<double>(), h->b_up.as<double>());
    PQP_HIP(hipGetLastError());
    if ((rc = sm_solve(h, SM_TENSION2, batch, n, status, iters, info))) return rc;
    hipLaunchKernelGGL(pqp::tension_finish_kernel, dim3(batch), dim3(64), (size_t)n * 8, h->stream, batch, n, n_of, 4 * n - 1, 4, h->b_x.as<double>(), out_x, out_y, out_s);
    PQP_HIP(hipGetLastError());
    return PQP_OK;
}

int pqp_smooth_tension2_device(pqp_handle* h, int batch, int n, const double* x_list, const double* y_list, const double* angle_list,
                               const double* k_list, const double* s_list, double* out_x, double* out_y, double* out_s, int32_t* status,
                               int32_t* iters, double* info) {
    return smooth_tension2_impl(h, batch, n, nullptr, x_list, y_list, angle_list, k_list, s_list, out_x, out_y, out_s, status, iters, info);
}

int pqp_smooth_tension2_var_device(pqp_handle* h, int batch, int n_max, const int32_t* n_of, const double* x_list, const double* y_list,
                                   const double* angle_list, const double* k_list, const double* s_list, double* out_x, double* out_y,
                                   double* out_s, int32_t* status, int32_t* iters, double* info) {
    if (!n_of) return fail(PQP_ERR_INVALID, "pqp_smooth_tension2_var: n_of is null");
    return smooth_tension2_impl(h, batch, n_max, n_of, x_list, y_list, angle_list, k_list, s_list, out_x, out_y, out_s, status, iters, info);
}

// TensionSmoother::osqpSmooth (tension_smoother.cpp:49-100); clearance[batch][n] = Map::getObstacleDistance at each point; n_of [batch]
// (device) or nullptr
static int smooth_tension_impl(pqp_handle* h, int batch, int n, const int32_t* n_of, const double* x_list, const double* y_list, const double* angle_list,
                               const double* clearance, double* out_x, double* out_y, double* out_s, int32_t* status, int32_t* iters, double* info) {
    if (!h || !x_list || !y_list || !angle_list || !clearance || !out_x || !out_y || !out_s || batch < 1 || n < 4)
        return fail(PQP_ERR_INVALID, "pqp_smooth_tension: bad argument");
    PQP_HIP(hipSetDevice(h->device));
    int rc;
    // The generic block-cyclic-reduction core keeps a QP's vectors, factor rows and row data in one compute unit's LDS: in TensionSmoother's
    // 9 x 9 blocks that ends near 166 points.  The reference has no such limit (tension_smoother.cpp:49-100; segmentRawReference gives a
    // point per metre of line).  Beyond it, also a handle in the reference's ADMM setting gets the exact kernel's optimum: a point with
    // zero residuals meets OSQP's termination test at any eps, so it IS a valid result of that setting (iters = 0; OSQP itself would
    // stop at a less accurate one).
    const bool generic_fits = sm_generic_fits(SM_TENSION, n);
    if (h->prm.polish == 1 || !generic_fits) {
        // exact optima asked for (or the only kernel that holds the QP): the box QP in the lateral shifts alone, one wavefront per scenario (tension_exact_kernel)
        if (!status) return fail(PQP_ERR_INVALID, "pqp_smooth_tension: status is null");
        h->next_event_pair();
        if (!h->capturing) PQP_HIP(hipEventRecord(h->ev0, h->stream));
        const double wk = h->prm.cartesian_curvature_weight, wdk = h->prm.cartesian_curvature_rate_weight, wdev = h->prm.cartesian_deviation_weight, tol = h->prm.polish_tol;
        // PQP_OPT_CARRY_CYCLES: the active set every line ended with is kept on the handle; a solve of the shape of the previous one starts from it
        signed char* act_io = nullptr;
        int carry = 0;
        if (h->opt_carry) {
            const void* before = h->sm_act[0].p;
            if ((rc = h->sm_act[0].ensure((size_t)batch * n))) return rc;
            act_io = h->sm_act[0].as<signed char>();
            carry = (h->sm_act_batch[0] == batch && h->sm_act_n[0] == n && before == h->sm_act[0].p) ? 1 : 0;
            h->sm_act_batch[0] = batch; h->sm_act_n[0] = n;
        }
        if (n <= 64) hipLaunchKernelGGL(pqp::tension_exact_kernel<1>, dim3(batch), dim3(64), 0, h->stream, batch, n, n_of, x_list, y_list, angle_list, clearance, wk, wdk, wdev, tol, out_x, out_y, out_s, status, iters, info, act_io, carry, nullptr);
        else if (n <= 128) hipLaunchKernelGGL(pqp::tension_exact_kernel<2>, dim3(batch), dim3(64), 0, h->stream, batch, n, n_of, x_list, y_list, angle_list, clearance, wk, wdk, wdev, tol, out_x, out_y, out_s, status, iters, info, act_io, carry, nullptr);
        else if (n <= 256) hipLaunchKernelGGL(pqp::tension_exact_kernel<4>, dim3(batch), dim3(64), 0, h->stream, batch, n, n_of, x_list, y_list, angle_list, clearance, wk, wdk, wdev, tol, out_x, out_y, out_s, status, iters, info, act_io, carry, nullptr);
        else if (n <= 384) hipLaunchKernelGGL(pqp::tension_exact_kernel<6>, dim3(batch), dim3(64), 0, h->stream, batch, n, n_of, x_list, y_list, angle_list, clearance, wk, wdk, wdev, tol, out_x, out_y, out_s, status, iters, info, act_io, carry, nullptr);
        else if (n <= 512) hipLaunchKernelGGL(pqp::tension_exact_kernel<8>, dim3(batch), dim3(64), 0, h->stream, batch, n, n_of, x_list, y_list, angle_list, clearance, wk, wdk, wdev, tol, out_x, out_y, out_s, status, iters, info, act_io, carry, nullptr);
        // (twelve / sixteen points per lane: the lane state no longer fits the registers - 1.8 / 3.3 KB of scratch per lane - but lines that long are
        //  rare, a point per metre of reference line, and the recursion down the lanes, not the spills, is what their time goes to)
        else if (n <= 768) hipLaunchKernelGGL(pqp::tension_exact_kernel<12>, dim3(batch), dim3(64), 0, h->stream, batch, n, n_of, x_list, y_list, angle_list, clearance, wk, wdk, wdev, tol, out_x, out_y, out_s, status, iters, info, act_io, carry, nullptr);
        else if (n <= 1024) hipLaunchKernelGGL(pqp::tension_exact_kernel<16>, dim3(batch), dim3(64), 0, h->stream, batch, n, n_of, x_list, y_list, angle_list, clearance, wk, wdk, wdev, tol, out_x, out_y, out_s, status, iters, info, act_io, carry, nullptr);
        else {
            // any longer line (the reference has no cap: tension_smoother.cpp:49-100): the same kernel with its arrays in HBM (SmHbm)
            if ((rc = h->b_pband.ensure((size_t)batch * pqp::kTensionExactArrays * (64 * (((size_t)n + 63) / 64)) * 8))) return rc;
            hipLaunchKernelGGL(pqp::tension_exact_kernel<0>, dim3(batch), dim3(64), 0, h->stream, batch, n, n_of, x_list, y_list, angle_list, clearance, wk, wdk, wdev, tol, out_x, out_y, out_s, status, iters, info, act_io, carry, h->b_pband.as<double>());
        }
        PQP_HIP(hipGetLastError());
        if (!h->capturing) PQP_HIP(hipEventRecord(h->ev1, h->stream));
        h->timed = true;
        return PQP_OK;
    }
    if ((rc = sm_alloc(h, SM_TENSION, batch, n))) return rc;
    const int total = batch * n;
    hipLaunchKernelGGL(pqp::tension_assemble_kernel, dim3((total + 255) / 256), dim3(256), 0, h->stream, batch, n, n_of, x_list, y_list, angle_list, clearance,
                       h->prm.cartesian_curvature_weight, h->prm.cartesian_curvature_rate_weight, h->prm.cartesian_deviation_weight,
                       h->b_pband.as<double>(), h->b_q.as<double>(), h->b_aval.as<double>(), h->b_lo.as<double>(), h->b_up.as<double>());
    PQP_HIP(hipGetLastError());
    if ((rc = sm_solve(h, SM_TENSION, batch, n, status, iters, info))) return rc;
    hipLaunchKernelGGL(pqp::tension_finish_kernel, dim3(batch), dim3(64), (size_t)n * 8, h->stream, batch, n, n_of, 3 * n, 3, h->b_x.as<double>(), out_x, out_y, out_s);
    PQP_HIP(hipGetLastError());
    return PQP_OK;
}

int pqp_smooth_tension_device(pqp_handle* h, int batch, int n, const double* x_list, const double* y_list, const double* angle_list,
                              const double* clearance, double* out_x, double* out_y, double* out_s, int32_t* status, int32_t* iters, double* info) {
    return smooth_tension_impl(h, batch, n, nullptr, x_list, y_list, angle_list, clearance, out_x, out_y, out_s, status, iters, info);
}

int pqp_smooth_tension_var_device(pqp_handle* h, int batch, int n_max, const int32_t* n_of, const double* x_list, const double* y_list,
                                  const double* angle_list, const double* clearance, double* out_x, double* out_y, double* out_s, int32_t* status,
                                  int32_t* iters, double* info) {
    if (!n_of) return fail(PQP_ERR_INVALID, "pqp_smooth_tension_var: n_of is null");
    return smooth_tension_impl(h, batch, n_max, n_of, x_list, y_list, angle_list, clearance, out_x, out_y, out_s, status, iters, info);
}

// ReferencePathSmoother::postSmooth QP (reference_path_smoother.cpp:526-558): out_l[batch][m] = the lateral offsets l_i
static int post_smooth_impl(pqp_handle* h, int batch, int m, const int32_t* m_of, const double* layers_s, const double* lb, const double* ub,
                            const double* vehicle_l, double* out_l, int32_t* status, int32_t* iters, double* info) {
    if (!h || !layers_s || !lb || !ub || !vehicle_l || !out_l || batch < 1 || m < 4) return fail(PQP_ERR_INVALID, "pqp_post_smooth: bad argument (m >= 4, reference_path_smoother.cpp:528)");
    PQP_HIP(hipSetDevice(h->device));
    // (beyond what the generic core holds in a CU's LDS also a handle in the reference's ADMM setting gets the exact kernel's optimum, as in smooth_tension_impl)
    if (h->prm.polish == 1 || !sm_generic_fits(SM_POST, m)) {
        // exact optima asked for: the box QP in the offsets alone, one wavefront per scenario (post_exact_kernel)
        if (!status) return fail(PQP_ERR_INVALID, "pqp_post_smooth: status is null");
        h->next_event_pair();
        if (!h->capturing) PQP_HIP(hipEventRecord(h->ev0, h->stream));
        const double tol = h->prm.polish_tol;
        signed char* act_io = nullptr;          // PQP_OPT_CARRY_CYCLES, as in smooth_tension_impl
        int carry = 0;
        if (h->opt_carry) {
            const void* before = h->sm_act[1].p;
            int rc_a;
            if ((rc_a = h->sm_act[1].ensure((size_t)batch * m))) return rc_a;
            act_io = h->sm_act[1].as<signed char>();
            carry = (h->sm_act_batch[1] == batch && h->sm_act_n[1] == m && before == h->sm_act[1].p) ? 1 : 0;
            h->sm_act_batch[1] = batch; h->sm_act_n[1] = m;
        }
        if (m <= 64) hipLaunchKernelGGL(pqp::post_exact_kernel<1>, dim3(batch), dim3(64), 0, h->stream, batch, m, m_of, layers_s, lb, ub, vehicle_l, tol, out_l, status, iters, info, act_io, carry, nullptr);
        else if (m <= 128) hipLaunchKernelGGL(pqp::post_exact_kernel<2>, dim3(batch), dim3(64), 0, h->stream, batch, m, m_of, layers_s, lb, ub, vehicle_l, tol, out_l, status, iters, info, act_io, carry, nullptr);
        else if (m <= 256) hipLaunchKernelGGL(pqp::post_exact_kernel<4>, dim3(batch), dim3(64), 0, h->stream, batch, m, m_of, layers_s, lb, ub, vehicle_l, tol, out_l, status, iters, info, act_io, carry, nullptr);
        else if (m <= 384) hipLaunchKernelGGL(pqp::post_exact_kernel<6>, dim3(batch), dim3(64), 0, h->stream, batch, m, m_of, layers_s, lb, ub, vehicle_l, tol, out_l, status, iters, info, act_io, carry, nullptr);
        else if (m <= 512) hipLaunchKernelGGL(pqp::post_exact_kernel<8>, dim3(batch), dim3(64), 0, h->stream, batch, m, m_of, layers_s, lb, ub, vehicle_l, tol, out_l, status, iters, info, act_io, carry, nullptr);
        else if (m <= 768) hipLaunchKernelGGL(pqp::post_exact_kernel<12>, dim3(batch), dim3(64), 0, h->stream, batch, m, m_of, layers_s, lb, ub, vehicle_l, tol, out_l, status, iters, info, act_io, carry, nullptr);
        else if (m <= 1024) hipLaunchKernelGGL(pqp::post_exact_kernel<16>, dim3(batch), dim3(64), 0, h->stream, batch, m, m_of, layers_s, lb, ub, vehicle_l, tol, out_l, status, iters, info, act_io, carry, nullptr);
        else {
            // any longer corridor (reference_path_smoother.cpp:526-580 has no cap): the same kernel with its arrays in HBM (SmHbm)
            int rc_w;
            if ((rc_w = h->b_pband.ensure((size_t)batch * pqp::kPostExactArrays * (64 * (((size_t)m + 63) / 64)) * 8))) return rc_w;
            hipLaunchKernelGGL(pqp::post_exact_kernel<0>, dim3(batch), dim3(64), 0, h->stream, batch, m, m_of, layers_s, lb, ub, vehicle_l, tol, out_l, status, iters, info, act_io, carry, h->b_pband.as<double>());
        }
        PQP_HIP(hipGetLastError());
        if (!h->capturing) PQP_HIP(hipEventRecord(h->ev1, h->stream));
        h->timed = true;
        return PQP_OK;
    }
    int rc;
    if ((rc = sm_alloc(h, SM_POST, batch, m))) return rc;
    const int total = batch * m;
    hipLaunchKernelGGL(pqp::post_assemble_kernel, dim3((total + 255) / 256), dim3(256), 0, h->stream, batch, m, m_of, layers_s, lb, ub, vehicle_l,
                       h->b_pband.as<double>(), h->b_q.as<double>(), h->b_aval.as<double>(), h->b_lo.as<double>(), h->b_up.as<double>());
    PQP_HIP(hipGetLastError());
    if ((rc = sm_solve(h, SM_POST, batch, m, status, iters, info))) return rc;
    hipLaunchKernelGGL(pqp::post_finish_kernel, dim3((total + 255) / 256), dim3(256), 0, h->stream, batch, m, h->b_x.as<double>(), out_l);
    PQP_HIP(hipGetLastError());
    return PQP_OK;
}

int pqp_post_smooth_device(pqp_handle* h, int batch, int m, const double* layers_s, const double* lb, const double* ub, const double* vehicle_l,
                           double* out_l, int32_t* status, int32_t* iters, double* info) {
    return post_smooth_impl(h, batch, m, nullptr, layers_s, lb, ub, vehicle_l, out_l, status, iters, info);
}

int pqp_post_smooth_var_device(pqp_handle* h, int batch, int m_max, const int32_t* m_of, const double* layers_s, const double* lb, const double* ub,
                               const double* vehicle_l, double* out_l, int32_t* status, int32_t* iters, double* info) {
    if (!m_of) return fail(PQP_ERR_INVALID, "pqp_post_smooth_var: m_of is null");
    return post_smooth_impl(h, batch, m_max, m_of, layers_s, lb, ub, vehicle_l, out_l, status, iters, info);
}

// host-pointer conveniences: nin input lists of [batch][n] (+ optional [batch] scalar list), nout output lists
static int sm_host_call(pqp_handle* h, int which, int batch, int n, const double* const* in, int nin, const double* scalar_in, double* const* out,
                        int nout, int32_t* status, int32_t* iters) {
    PQP_HIP(hipSetDevice(h->device));
    const size_t bytes = (size_t)batch * n * 8;
    int rc;
    for (int k = 0; k < nin; ++k) {
        if ((rc = h->b_in[k].ensure(bytes))) return rc;
        PQP_HIP(hipMemcpyAsync(h->b_in[k].p, in[k], bytes, hipMemcpyHostToDevice, h->stream));
    }
    if (scalar_in) {
        if ((rc = h->b_in[4].ensure((size_t)batch * 8))) return rc;
        PQP_HIP(hipMemcpyAsync(h->b_in[4].p, scalar_in, (size_t)batch * 8, hipMemcpyHostToDevice, h->stream));
    }
    for (int k = 0; k < nout; ++k) if ((rc = h->b_out[k].ensure(bytes))) return rc;
    if ((rc = h->s_status.ensure((size_t)batch * 4)) || (rc = h->s_iters.ensure((size_t)batch * 4))) return rc;
    double* i0 = h->b_in[0].as<double>(); double* i1 = h->b_in[1].as<double>(); double* i2 = h->b_in[2].as<double>(); double* i3 = h->b_in[3].as<double>();
    double* o0 = h->b_out[0].as<double>(); double* o1 = h->b_out[1].as<double>(); double* o2 = h->b_out[2].as<double>();
    if (which == SM_TENSION2) {
        if ((rc = h->b_in[4].ensure(bytes))) return rc;
        PQP_HIP(hipMemcpyAsync(h->b_in[4].p, in[4], bytes, hipMemcpyHostToDevice, h->stream));
        rc = pqp_smooth_tension2_device(h, batch, n, i0, i1, i2, i3, h->b_in[4].as<double>(), o0, o1, o2, h->s_status.as<int32_t>(), h->s_iters.as<int32_t>(), nullptr);
    } else if (which == SM_TENSION) {
        rc = pqp_smooth_tension_device(h, batch, n, i0, i1, i2, i3, o0, o1, o2, h->s_status.as<int32_t>(), h->s_iters.as<int32_t>(), nullptr);
    } else {
        rc = pqp_post_smooth_device(h, batch, n, i0, i1, i2, h->b_in[4].as<double>(), o0, h->s_status.as<int32_t>(), h->s_iters.as<int32_t>(), nullptr);
    }
    if (rc) return rc;
    for (int k = 0; k < nout; ++k) PQP_HIP(hipMemcpyAsync(out[k], h->b_out[k].p, bytes, hipMemcpyDeviceToHost, h->stream));
    if (status) PQP_HIP(hipMemcpyAsync(status, h->s_status.p, (size_t)batch * 4, hipMemcpyDeviceToHost, h->stream));
    if (iters) PQP_HIP(hipMemcpyAsync(iters, h->s_iters.p, (size_t)batch * 4, hipMemcpyDeviceToHost, h->stream));
    PQP_HIP(hipStreamSynchronize(h->stream));
    return PQP_OK;
}

int pqp_smooth_tension2(pqp_handle* h, int batch, int n, const double* x_list, const double* y_list, const double* angle_list, const double* k_list,
                        const double* s_list, double* out_x, double* out_y, double* out_s, int32_t* status, int32_t* iters) {
    if (!h || !x_list || !y_list || !angle_list || !k_list || !s_list || !out_x || !out_y || !out_s || batch < 1 || n < 3)
        return fail(PQP_ERR_INVALID, "pqp_smooth_tension2: bad argument");
    const double* in[5] = {x_list, y_list, angle_list, k_list, s_list};
    double* out[3] = {out_x, out_y, out_s};
    return sm_host_call(h, SM_TENSION2, batch, n, in, 4, nullptr, out, 3, status, iters);
}
int pqp_smooth_tension(pqp_handle* h, int batch, int n, const double* x_list, const double* y_list, const double* angle_list, const double* clearance,
                       double* out_x, double* out_y, double* out_s, int32_t* status, int32_t* iters) {
    if (!h || !x_list || !y_list || !angle_list || !clearance || !out_x || !out_y || !out_s || batch < 1 || n < 4)
        return fail(PQP_ERR_INVALID, "pqp_smooth_tension: bad argument");
    const double* in[4] = {x_list, y_list, angle_list, clearance};
    double* out[3] = {out_x, out_y, out_s};
    return sm_host_call(h, SM_TENSION, batch, n, in, 4, nullptr, out, 3, status, iters);
}
int pqp_post_smooth(pqp_handle* h, int batch, int m, const double* layers_s, const double* lb, const double* ub, const double* vehicle_l, double* out_l,
                    int32_t* status, int32_t* iters) {
    if (!h || !layers_s || !lb || !ub || !vehicle_l || !out_l || batch < 1 || m < 4) return fail(PQP_ERR_INVALID, "pqp_post_smooth: bad argument");
    const double* in[3] = {layers_s, lb, ub};
    double* out[1] = {out_l};
    return sm_host_call(h, SM_POST, batch, m, in, 3, vehicle_l, out, 1, status, iters);
}

// ---- corridor bounds from the distance map -----------------------------------------------------------
void pqp_corridor_default_params(pqp_corridor_params* p) {
    if (!p) return;
    p->front_length = 3.9; p->rear_length = -1.0;       // planning_flags.cpp:20,18
    p->car_width = 2.0; p->safety_margin = 0.3;         // planning_flags.cpp:10,14
    p->epsilon = 1e-6;                                  // planning_flags.cpp:108
    p->search_radius = 0.5; p->delta_s = 0.3; p->smaller_ds = 0.05; p->search_range = 6.0; p->min_space = 0.2;   // reference_path_impl.cpp:238-304
    p->projection_window = 5.0;                         // reference_path_impl.cpp:194
}

// a distance-map layer the kernels can index with 32 bits (pqp_corridor_kernels.inc: obstacle_distance)
static bool geometry_ok(const pqp_grid_geometry* g) {
    return g && g->rows >= 2 && g->cols >= 2 && g->resolution > 0.0 && (long long)g->rows * g->cols < (1ll << 30);
}

int pqp_corridor_bounds_device(pqp_handle* h, int batch, int n, int m, const double* ref, const int32_t* n_of, const double* spline,
                               const double* spline_ext, const float* dist, const int32_t* map_of, const pqp_grid_geometry* geom,
                               const pqp_corridor_params* prm, double* bounds, int32_t* n_valid) {
    if (!h || !ref || !spline || !spline_ext || !dist || !geom || !prm || !bounds || !n_valid || batch < 1 || n < 1 || m < 3 ||
        !geometry_ok(geom) || !(prm->delta_s > 0.0) || !(prm->smaller_ds > 0.0))
        return fail(PQP_ERR_INVALID, "pqp_corridor_bounds: bad argument (m >= 3 knots: spline.cpp:164; a map layer of 2 x 2 to 2^30 cells)");
    PQP_HIP(hipSetDevice(h->device));
    pqp::CorridorArgs a;
    a.batch = batch; a.n = n; a.m = m; a.ref = ref; a.spl = spline; a.spl_ext = spline_ext; a.dist = dist; a.map_of = map_of; a.n_of = n_of;
    a.g = *geom; a.p = *prm; a.bounds = bounds; a.n_valid = n_valid;
    int threads = 64 * ((3 * n + 63) / 64);
    if (threads > 1024) threads = 1024;
    h->next_event_pair();
    if (!h->capturing) PQP_HIP(hipEventRecord(h->ev0, h->stream));
    // a whole scenario's probes in LDS when they fit (9 m + 33 n doubles), tiles of waypoints otherwise: any path length
    a.tile = n;
    if (pqp::CorridorLds{m, n}.total_bytes() > 160 * 1024) {
        const long long room = 160 * 1024 - (long long)pqp::CorridorLds{m, 0}.total_bytes(), per_waypoint = (long long)(pqp::CorridorLds{m, 1}.total_bytes() - pqp::CorridorLds{m, 0}.total_bytes());
        if (room < 16 * per_waypoint) return fail(PQP_ERR_CAPACITY, "pqp_corridor_bounds: the line's spline table (9 m doubles) does not leave room for the probes in one CU's LDS");
        a.tile = (int)(room / per_waypoint);
    }
    const size_t lds = pqp::CorridorLds{m, a.tile}.total_bytes();
    if (lds > 160 * 1024) return fail(PQP_ERR_CAPACITY, "pqp_corridor_bounds: scenario too large for one CU's LDS (about 9 m + 31 n doubles)");
    if (lds > 48 * 1024) PQP_HIP(hipFuncSetAttribute((const void*)pqp::corridor_bounds_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    threads = 512;                   // the sample loops are strided; 512 lanes per scenario keep the most gathers in flight per CU (measured at batch
                                     // 1024 x n = 80: 1024 lanes 142 us - two scenarios per CU -, 512: 121, 256: 120, 128: 146)
    hipLaunchKernelGGL(pqp::corridor_bounds_kernel, dim3(batch), dim3(threads), lds, h->stream, a);
    PQP_HIP(hipGetLastError());
    if (!h->capturing) PQP_HIP(hipEventRecord(h->ev1, h->stream));
    h->timed = true;
    return PQP_OK;
}

int pqp_corridor_bounds(pqp_handle* h, int batch, int n, int m, const double* ref, const int32_t* n_of, const double* spline,
                        const double* spline_ext, const float* dist, int n_maps, const int32_t* map_of, const pqp_grid_geometry* geom,
                        const pqp_corridor_params* prm, double* bounds, int32_t* n_valid) {
    if (!h || !ref || !spline || !spline_ext || !dist || !geom || !prm || !bounds || !n_valid || batch < 1 || n < 1 || m < 3 || n_maps < 1)
        return fail(PQP_ERR_INVALID, "pqp_corridor_bounds: bad argument");
    PQP_HIP(hipSetDevice(h->device));
    const size_t b_ref = (size_t)batch * n * PQP_REF_STRIDE * 8, b_spl = (size_t)batch * 9 * m * 8, b_ext = (size_t)batch * 4 * 8;
    const size_t b_map = (size_t)n_maps * geom->rows * geom->cols * 4, b_of = (size_t)batch * 4;
    const size_t b_bnd = (size_t)batch * n * PQP_BOUNDS_STRIDE * 8, b_nv = (size_t)batch * 4;
    const size_t sizes[7] = {b_ref, b_spl, b_ext, b_map, b_of, b_bnd, b_nv};
    int rc;
    for (int k = 0; k < 7; ++k) if ((rc = h->c_buf[k].ensure(sizes[k]))) return rc;
    PQP_HIP(hipMemcpyAsync(h->c_buf[0].p, ref, b_ref, hipMemcpyHostToDevice, h->stream));
    PQP_HIP(hipMemcpyAsync(h->c_buf[1].p, spline, b_spl, hipMemcpyHostToDevice, h->stream));
    PQP_HIP(hipMemcpyAsync(h->c_buf[2].p, spline_ext, b_ext, hipMemcpyHostToDevice, h->stream));
    PQP_HIP(hipMemcpyAsync(h->c_buf[3].p, dist, b_map, hipMemcpyHostToDevice, h->stream));
    if (map_of) PQP_HIP(hipMemcpyAsync(h->c_buf[4].p, map_of, b_of, hipMemcpyHostToDevice, h->stream));
    if (n_of) PQP_HIP(hipMemcpyAsync(h->c_buf[6].p, n_of, b_nv, hipMemcpyHostToDevice, h->stream));     // n_valid is written after n_of is read
    if ((rc = pqp_corridor_bounds_device(h, batch, n, m, h->c_buf[0].as<double>(), n_of ? h->c_buf[6].as<int32_t>() : nullptr,
                                         h->c_buf[1].as<double>(), h->c_buf[2].as<double>(),
                                         h->c_buf[3].as<float>(), map_of ? h->c_buf[4].as<int32_t>() : nullptr, geom, prm,
                                         h->c_buf[5].as<double>(), h->c_buf[6].as<int32_t>())))
        return rc;
    PQP_HIP(hipMemcpyAsync(bounds, h->c_buf[5].p, b_bnd, hipMemcpyDeviceToHost, h->stream));
    PQP_HIP(hipMemcpyAsync(n_valid, h->c_buf[6].p, b_nv, hipMemcpyDeviceToHost, h->stream));
    PQP_HIP(hipStreamSynchronize(h->stream));
    return PQP_OK;
}

// ---- reference states + initial error ----------------------------------------------------------------
int pqp_reference_states_device(pqp_handle* h, int batch, int n_max, int m, const double* spline, const double* spline_ext,
                                const double* max_s, const double* start, double ds_small, double ds_large, int dynamic, double* ref,
                                int32_t* count, double* init_err) {
    if (!h || !spline || !spline_ext || !max_s || !ref || !count || batch < 1 || n_max < 1 || m < 3 || !(ds_small > 0.0) ||
        !(ds_large >= ds_small) || (init_err && !start))
        return fail(PQP_ERR_INVALID, "pqp_reference_states: bad argument (0 < ds_small <= ds_large: reference_path_impl.cpp:315)");
    PQP_HIP(hipSetDevice(h->device));
    pqp::RefStatesArgs a;
    a.batch = batch; a.n_max = n_max; a.m = m; a.spl = spline; a.spl_ext = spline_ext; a.max_s = max_s; a.start = start;
    a.ds_small = ds_small; a.ds_large = ds_large; a.dynamic = dynamic ? 1 : 0; a.ref = ref; a.count = count; a.init_err = init_err;
    a.lx = a.ly = a.ls = a.langle = a.lk = nullptr;
    h->next_event_pair();
    if (!h->capturing) PQP_HIP(hipEventRecord(h->ev0, h->stream));
    const size_t lds = ((size_t)9 * m + n_max) * 8;
    if (lds > 160 * 1024) return fail(PQP_ERR_CAPACITY, "pqp_reference_states: 9 m + n_max doubles exceed one CU's LDS");
    if (lds > 48 * 1024) PQP_HIP(hipFuncSetAttribute((const void*)pqp::reference_states_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(pqp::reference_states_kernel, dim3(batch), dim3(64), lds, h->stream, a);
    PQP_HIP(hipGetLastError());
    if (!h->capturing) PQP_HIP(hipEventRecord(h->ev1, h->stream));
    h->timed = true;
    return PQP_OK;
}

// ---- raw reference line -> the smoother QPs' input lists (ReferencePathSmoother::segmentRawReference) ------------------------------
int pqp_segment_raw_reference_device(pqp_handle* h, int batch, int n_max, int m, const double* spline, const double* spline_ext,
                                     const double* max_s, double delta_s, double* x, double* y, double* s, double* angle, double* k,
                                     int32_t* count) {
    if (!h || !spline || !spline_ext || !max_s || !x || !y || !s || !angle || !k || !count || batch < 1 || n_max < 1 || m < 3 || !(delta_s > 0.0))
        return fail(PQP_ERR_INVALID, "pqp_segment_raw_reference: bad argument");
    PQP_HIP(hipSetDevice(h->device));
    pqp::RefStatesArgs a;
    a.batch = batch; a.n_max = n_max; a.m = m; a.spl = spline; a.spl_ext = spline_ext; a.max_s = max_s; a.start = nullptr;
    a.ds_small = delta_s; a.ds_large = delta_s; a.dynamic = 2; a.ref = nullptr; a.count = count; a.init_err = nullptr;
    a.lx = x; a.ly = y; a.ls = s; a.langle = angle; a.lk = k;
    h->next_event_pair();
    if (!h->capturing) PQP_HIP(hipEventRecord(h->ev0, h->stream));
    const size_t lds = ((size_t)9 * m + n_max) * 8;
    if (lds > 160 * 1024) return fail(PQP_ERR_CAPACITY, "pqp_segment_raw_reference: 9 m + n_max doubles exceed one CU's LDS");
    if (lds > 48 * 1024) PQP_HIP(hipFuncSetAttribute((const void*)pqp::reference_states_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(pqp::reference_states_kernel, dim3(batch), dim3(64), lds, h->stream, a);
    PQP_HIP(hipGetLastError());
    if (!h->capturing) PQP_HIP(hipEventRecord(h->ev1, h->stream));
    h->timed = true;
    return PQP_OK;
}

int pqp_segment_raw_reference(pqp_handle* h, int batch, int n_max, int m, const double* spline, const double* spline_ext, const double* max_s,
                              double delta_s, double* x, double* y, double* s, double* angle, double* k, int32_t* count) {
    if (!h || !spline || !spline_ext || !max_s || !x || !y || !s || !angle || !k || !count || batch < 1 || n_max < 1 || m < 3)
        return fail(PQP_ERR_INVALID, "pqp_segment_raw_reference: bad argument");
    PQP_HIP(hipSetDevice(h->device));
    const size_t b_spl = (size_t)batch * 9 * m * 8, b_ext = (size_t)batch * 4 * 8, b_s = (size_t)batch * 8;
    const size_t b_list = (size_t)batch * n_max * 8, b_cnt = (size_t)batch * 4;
    const size_t sizes[5] = {5 * b_list, b_spl, b_ext, b_s, b_cnt};
    int rc;
    for (int j = 0; j < 5; ++j) if ((rc = h->c_buf[j].ensure(sizes[j]))) return rc;
    PQP_HIP(hipMemcpyAsync(h->c_buf[1].p, spline, b_spl, hipMemcpyHostToDevice, h->stream));
    PQP_HIP(hipMemcpyAsync(h->c_buf[2].p, spline_ext, b_ext, hipMemcpyHostToDevice, h->stream));
    PQP_HIP(hipMemcpyAsync(h->c_buf[3].p, max_s, b_s, hipMemcpyHostToDevice, h->stream));
    PQP_HIP(hipMemsetAsync(h->c_buf[0].p, 0, 5 * b_list, h->stream));
    double* l = h->c_buf[0].as<double>();
    const size_t bn = (size_t)batch * n_max;
    if ((rc = pqp_segment_raw_reference_device(h, batch, n_max, m, h->c_buf[1].as<double>(), h->c_buf[2].as<double>(), h->c_buf[3].as<double>(),
                                               delta_s, l, l + bn, l + 2 * bn, l + 3 * bn, l + 4 * bn, h->c_buf[4].as<int32_t>())))
        return rc;
    double* outs[5] = {x, y, s, angle, k};
    for (int j = 0; j < 5; ++j) PQP_HIP(hipMemcpyAsync(outs[j], l + j * bn, b_list, hipMemcpyDeviceToHost, h->stream));
    PQP_HIP(hipMemcpyAsync(count, h->c_buf[4].p, b_cnt, hipMemcpyDeviceToHost, h->stream));
    PQP_HIP(hipStreamSynchronize(h->stream));
    return PQP_OK;
}

int pqp_reference_states(pqp_handle* h, int batch, int n_max, int m, const double* spline, const double* spline_ext, const double* max_s,
                         const double* start, double ds_small, double ds_large, int dynamic, double* ref, int32_t* count,
                         double* init_err) {
    if (!h || !spline || !spline_ext || !max_s || !ref || !count || batch < 1 || n_max < 1 || m < 3)
        return fail(PQP_ERR_INVALID, "pqp_reference_states: bad argument");
    PQP_HIP(hipSetDevice(h->device));
    const size_t b_spl = (size_t)batch * 9 * m * 8, b_ext = (size_t)batch * 4 * 8, b_s = (size_t)batch * 8, b_st = (size_t)batch * 3 * 8;
    const size_t b_ref = (size_t)batch * n_max * PQP_REF_STRIDE * 8, b_cnt = (size_t)batch * 4, b_err = (size_t)batch * 2 * 8;
    const size_t sizes[7] = {b_ref, b_spl, b_ext, b_s, b_st, b_err, b_cnt};
    int rc;
    for (int k = 0; k < 7; ++k) if ((rc = h->c_buf[k].ensure(sizes[k]))) return rc;
    PQP_HIP(hipMemcpyAsync(h->c_buf[1].p, spline, b_spl, hipMemcpyHostToDevice, h->stream));
    PQP_HIP(hipMemcpyAsync(h->c_buf[2].p, spline_ext, b_ext, hipMemcpyHostToDevice, h->stream));
    PQP_HIP(hipMemcpyAsync(h->c_buf[3].p, max_s, b_s, hipMemcpyHostToDevice, h->stream));
    if (start) PQP_HIP(hipMemcpyAsync(h->c_buf[4].p, start, b_st, hipMemcpyHostToDevice, h->stream));
    PQP_HIP(hipMemsetAsync(h->c_buf[0].p, 0, b_ref, h->stream));
    if ((rc = pqp_reference_states_device(h, batch, n_max, m, h->c_buf[1].as<double>(), h->c_buf[2].as<double>(), h->c_buf[3].as<double>(),
                                          start ? h->c_buf[4].as<double>() : nullptr, ds_small, ds_large, dynamic, h->c_buf[0].as<double>(),
                                          h->c_buf[6].as<int32_t>(), (init_err && start) ? h->c_buf[5].as<double>() : nullptr)))
        return rc;
    PQP_HIP(hipMemcpyAsync(ref, h->c_buf[0].p, b_ref, hipMemcpyDeviceToHost, h->stream));
    PQP_HIP(hipMemcpyAsync(count, h->c_buf[6].p, b_cnt, hipMemcpyDeviceToHost, h->stream));
    if (init_err && start) PQP_HIP(hipMemcpyAsync(init_err, h->c_buf[5].p, b_err, hipMemcpyDeviceToHost, h->stream));
    PQP_HIP(hipStreamSynchronize(h->stream));
    return PQP_OK;
}

// ---- lateral offsets on a line -> points with chord-length abscissae (tail of ReferencePathSmoother::postSmooth) ---------------------
int pqp_offsets_to_points_device(pqp_handle* h, int batch, int m_spline, int m, const double* spline, const double* spline_ext, const double* at_s,
                                 const double* l, const int32_t* m_of, double* x, double* y, double* s) {
    if (!h || !spline || !spline_ext || !at_s || !l || !x || !y || !s || batch < 1 || m_spline < 3 || m < 1)
        return fail(PQP_ERR_INVALID, "pqp_offsets_to_points: bad argument");
    PQP_HIP(hipSetDevice(h->device));
    pqp::OffsetsArgs a;
    a.batch = batch; a.m_spl = m_spline; a.m = m; a.spl = spline; a.spl_ext = spline_ext; a.at_s = at_s; a.l = l; a.m_of = m_of;
    a.x = x; a.y = y; a.s = s;
    const size_t lds = ((size_t)9 * m_spline + 2 * (size_t)m) * 8;
    if (lds > 160 * 1024) return fail(PQP_ERR_CAPACITY, "pqp_offsets_to_points: 9 m_spline + 2 m doubles exceed one CU's LDS");
    if (lds > 48 * 1024) PQP_HIP(hipFuncSetAttribute((const void*)pqp::offsets_to_points_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    h->next_event_pair();
    if (!h->capturing) PQP_HIP(hipEventRecord(h->ev0, h->stream));
    hipLaunchKernelGGL(pqp::offsets_to_points_kernel, dim3(batch), dim3(64), lds, h->stream, a);
    PQP_HIP(hipGetLastError());
    if (!h->capturing) PQP_HIP(hipEventRecord(h->ev1, h->stream));
    h->timed = true;
    return PQP_OK;
}

int pqp_offsets_to_points(pqp_handle* h, int batch, int m_spline, int m, const double* spline, const double* spline_ext, const double* at_s,
                          const double* l, const int32_t* m_of, double* x, double* y, double* s) {
    if (!h || !spline || !spline_ext || !at_s || !l || !x || !y || !s || batch < 1 || m_spline < 3 || m < 1)
        return fail(PQP_ERR_INVALID, "pqp_offsets_to_points: bad argument");
    PQP_HIP(hipSetDevice(h->device));
    const size_t b_spl = (size_t)batch * 9 * m_spline * 8, b_ext = (size_t)batch * 4 * 8, b_list = (size_t)batch * m * 8, b_n = (size_t)batch * 4;
    const size_t sizes[6] = {3 * b_list, b_spl, b_ext, b_list, b_list, b_n};
    int rc;
    for (int j = 0; j < 6; ++j) if ((rc = h->c_buf[j].ensure(sizes[j]))) return rc;
    PQP_HIP(hipMemcpyAsync(h->c_buf[1].p, spline, b_spl, hipMemcpyHostToDevice, h->stream));
    PQP_HIP(hipMemcpyAsync(h->c_buf[2].p, spline_ext, b_ext, hipMemcpyHostToDevice, h->stream));
    PQP_HIP(hipMemcpyAsync(h->c_buf[3].p, at_s, b_list, hipMemcpyHostToDevice, h->stream));
    PQP_HIP(hipMemcpyAsync(h->c_buf[4].p, l, b_list, hipMemcpyHostToDevice, h->stream));
    if (m_of) PQP_HIP(hipMemcpyAsync(h->c_buf[5].p, m_of, b_n, hipMemcpyHostToDevice, h->stream));
    PQP_HIP(hipMemsetAsync(h->c_buf[0].p, 0, 3 * b_list, h->stream));
    double* o = h->c_buf[0].as<double>();
    const size_t bn = (size_t)batch * m;
    if ((rc = pqp_offsets_to_points_device(h, batch, m_spline, m, h->c_buf[1].as<double>(), h->c_buf[2].as<double>(), h->c_buf[3].as<double>(),
                                           h->c_buf[4].as<double>(), m_of ? h->c_buf[5].as<int32_t>() : nullptr, o, o + bn, o + 2 * bn)))
        return rc;
    PQP_HIP(hipMemcpyAsync(x, o, b_list, hipMemcpyDeviceToHost, h->stream));
    PQP_HIP(hipMemcpyAsync(y, o + bn, b_list, hipMemcpyDeviceToHost, h->stream));
    PQP_HIP(hipMemcpyAsync(s, o + 2 * bn, b_list, hipMemcpyDeviceToHost, h->stream));
    PQP_HIP(hipStreamSynchronize(h->stream));
    return PQP_OK;
}

// ---- length of the reference line up to the target state (PathOptimizer::setReferencePathLength) ---------------------------------
int pqp_reference_length_device(pqp_handle* h, int batch, int m, const double* spline, const double* spline_ext, const double* length,
                                const double* target, double* length_out) {
    if (!h || !spline || !spline_ext || !length || !target || !length_out || batch < 1 || m < 3)
        return fail(PQP_ERR_INVALID, "pqp_reference_length: bad argument");
    PQP_HIP(hipSetDevice(h->device));
    pqp::RefLengthArgs a;
    a.batch = batch; a.m = m; a.spl = spline; a.spl_ext = spline_ext; a.length = length; a.target = target; a.length_out = length_out;
    const size_t lds = (size_t)9 * m * 8;
    if (lds > 160 * 1024) return fail(PQP_ERR_CAPACITY, "pqp_reference_length: 9 m doubles exceed one CU's LDS");
    if (lds > 48 * 1024) PQP_HIP(hipFuncSetAttribute((const void*)pqp::reference_length_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    h->next_event_pair();
    if (!h->capturing) PQP_HIP(hipEventRecord(h->ev0, h->stream));
    hipLaunchKernelGGL(pqp::reference_length_kernel, dim3(batch), dim3(64), lds, h->stream, a);
    PQP_HIP(hipGetLastError());
    if (!h->capturing) PQP_HIP(hipEventRecord(h->ev1, h->stream));
    h->timed = true;
    return PQP_OK;
}

int pqp_reference_length(pqp_handle* h, int batch, int m, const double* spline, const double* spline_ext, const double* length,
                         const double* target, double* length_out) {
    if (!h || !spline || !spline_ext || !length || !target || !length_out || batch < 1 || m < 3)
        return fail(PQP_ERR_INVALID, "pqp_reference_length: bad argument");
    PQP_HIP(hipSetDevice(h->device));
    const size_t b_spl = (size_t)batch * 9 * m * 8, b_ext = (size_t)batch * 4 * 8, b_s = (size_t)batch * 8, b_t = (size_t)batch * 3 * 8;
    const size_t sizes[5] = {b_s, b_spl, b_ext, b_s, b_t};
    int rc;
    for (int j = 0; j < 5; ++j) if ((rc = h->c_buf[j].ensure(sizes[j]))) return rc;
    PQP_HIP(hipMemcpyAsync(h->c_buf[1].p, spline, b_spl, hipMemcpyHostToDevice, h->stream));
    PQP_HIP(hipMemcpyAsync(h->c_buf[2].p, spline_ext, b_ext, hipMemcpyHostToDevice, h->stream));
    PQP_HIP(hipMemcpyAsync(h->c_buf[3].p, length, b_s, hipMemcpyHostToDevice, h->stream));
    PQP_HIP(hipMemcpyAsync(h->c_buf[4].p, target, b_t, hipMemcpyHostToDevice, h->stream));
    if ((rc = pqp_reference_length_device(h, batch, m, h->c_buf[1].as<double>(), h->c_buf[2].as<double>(), h->c_buf[3].as<double>(),
                                          h->c_buf[4].as<double>(), h->c_buf[0].as<double>())))
        return rc;
    PQP_HIP(hipMemcpyAsync(length_out, h->c_buf[0].p, b_s, hipMemcpyDeviceToHost, h->stream));
    PQP_HIP(hipStreamSynchronize(h->stream));
    return PQP_OK;
}

// ---- input points -> dense raw reference line (ReferencePathSmoother::bSpline) --------------------------------------------------
int pqp_bspline_resample_device(pqp_handle* h, int batch, int p_max, int n_max, const double* points, const int32_t* n_points, double* x,
                                double* y, double* s, int32_t* count) {
    if (!h || !points || !n_points || !x || !y || !s || !count || batch < 1 || p_max < 4 || n_max < 2)
        return fail(PQP_ERR_INVALID, "pqp_bspline_resample: bad argument (at least 4 input points: reference_path_smoother.cpp:33)");
    PQP_HIP(hipSetDevice(h->device));
    pqp::BsplineArgs a;
    a.batch = batch; a.p_max = p_max; a.n_max = n_max; a.pts = points; a.n_pts = n_points; a.x = x; a.y = y; a.s = s; a.count = count;
    const size_t lds = ((size_t)3 * p_max + 6 + (size_t)3 * n_max) * 8;
    if (lds > 160 * 1024) return fail(PQP_ERR_CAPACITY, "pqp_bspline_resample: 3 p_max + 3 n_max doubles exceed one CU's LDS");
    if (lds > 48 * 1024) PQP_HIP(hipFuncSetAttribute((const void*)pqp::bspline_resample_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    h->next_event_pair();
    if (!h->capturing) PQP_HIP(hipEventRecord(h->ev0, h->stream));
    hipLaunchKernelGGL(pqp::bspline_resample_kernel, dim3(batch), dim3(64), lds, h->stream, a);
    PQP_HIP(hipGetLastError());
    if (!h->capturing) PQP_HIP(hipEventRecord(h->ev1, h->stream));
    h->timed = true;
    return PQP_OK;
}

int pqp_bspline_resample(pqp_handle* h, int batch, int p_max, int n_max, const double* points, const int32_t* n_points, double* x, double* y,
                         double* s, int32_t* count) {
    if (!h || !points || !n_points || !x || !y || !s || !count || batch < 1 || p_max < 4 || n_max < 2)
        return fail(PQP_ERR_INVALID, "pqp_bspline_resample: bad argument");
    PQP_HIP(hipSetDevice(h->device));
    const size_t b_pts = (size_t)batch * p_max * 2 * 8, b_n = (size_t)batch * 4, b_list = (size_t)batch * n_max * 8;
    const size_t sizes[4] = {3 * b_list, b_pts, b_n, b_n};
    int rc;
    for (int j = 0; j < 4; ++j) if ((rc = h->c_buf[j].ensure(sizes[j]))) return rc;
    PQP_HIP(hipMemcpyAsync(h->c_buf[1].p, points, b_pts, hipMemcpyHostToDevice, h->stream));
    PQP_HIP(hipMemcpyAsync(h->c_buf[2].p, n_points, b_n, hipMemcpyHostToDevice, h->stream));
    PQP_HIP(hipMemsetAsync(h->c_buf[0].p, 0, 3 * b_list, h->stream));
    double* l = h->c_buf[0].as<double>();
    const size_t bn = (size_t)batch * n_max;
    if ((rc = pqp_bspline_resample_device(h, batch, p_max, n_max, h->c_buf[1].as<double>(), h->c_buf[2].as<int32_t>(), l, l + bn, l + 2 * bn,
                                          h->c_buf[3].as<int32_t>())))
        return rc;
    PQP_HIP(hipMemcpyAsync(x, l, b_list, hipMemcpyDeviceToHost, h->stream));
    PQP_HIP(hipMemcpyAsync(y, l + bn, b_list, hipMemcpyDeviceToHost, h->stream));
    PQP_HIP(hipMemcpyAsync(s, l + 2 * bn, b_list, hipMemcpyDeviceToHost, h->stream));
    PQP_HIP(hipMemcpyAsync(count, h->c_buf[3].p, b_n, hipMemcpyDeviceToHost, h->stream));
    PQP_HIP(hipStreamSynchronize(h->stream));
    return PQP_OK;
}

// ---- spline fit ---------------------------------------------------------------------------------------
static int spline_fit_impl(pqp_handle* h, int batch, int m, const int32_t* m_of, const double* s, const double* x, const double* y, double* spline,
                           double* spline_ext) {
    if (!h || !s || !x || !y || !spline || !spline_ext || batch < 1 || m < 3)
        return fail(PQP_ERR_INVALID, "pqp_spline_fit: bad argument (m >= 3: spline.cpp:164)");
    PQP_HIP(hipSetDevice(h->device));
    pqp::SplineFitArgs a;
    a.m_of = m_of;
    a.batch = batch; a.m = m; a.s = s; a.vx = x; a.vy = y; a.spl = spline; a.spl_ext = spline_ext;
    const size_t lds = (size_t)7 * m * 8;
    if (lds > 160 * 1024) return fail(PQP_ERR_CAPACITY, "pqp_spline_fit: 7 m doubles exceed one CU's LDS");
    if (lds > 48 * 1024) PQP_HIP(hipFuncSetAttribute((const void*)pqp::spline_fit_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    h->next_event_pair();
    if (!h->capturing) PQP_HIP(hipEventRecord(h->ev0, h->stream));
    hipLaunchKernelGGL(pqp::spline_fit_kernel, dim3(2 * batch), dim3(64), lds, h->stream, a);
    PQP_HIP(hipGetLastError());
    if (!h->capturing) PQP_HIP(hipEventRecord(h->ev1, h->stream));
    h->timed = true;
    return PQP_OK;
}

int pqp_spline_fit_device(pqp_handle* h, int batch, int m, const double* s, const double* x, const double* y, double* spline,
                          double* spline_ext) {
    return spline_fit_impl(h, batch, m, nullptr, s, x, y, spline, spline_ext);
}

int pqp_spline_fit_var_device(pqp_handle* h, int batch, int m_max, const int32_t* m_of, const double* s, const double* x, const double* y,
                              double* spline, double* spline_ext) {
    if (!m_of) return fail(PQP_ERR_INVALID, "pqp_spline_fit_var: m_of is null");
    return spline_fit_impl(h, batch, m_max, m_of, s, x, y, spline, spline_ext);
}

int pqp_spline_fit(pqp_handle* h, int batch, int m, const double* s, const double* x, const double* y, double* spline, double* spline_ext) {
    if (!h || !s || !x || !y || !spline || !spline_ext || batch < 1 || m < 3) return fail(PQP_ERR_INVALID, "pqp_spline_fit: bad argument");
    PQP_HIP(hipSetDevice(h->device));
    const size_t b_in = (size_t)batch * m * 8, b_spl = (size_t)batch * 9 * m * 8, b_ext = (size_t)batch * 4 * 8;
    const size_t sizes[5] = {b_in, b_in, b_in, b_spl, b_ext};
    int rc;
    for (int k = 0; k < 5; ++k) if ((rc = h->c_buf[k].ensure(sizes[k]))) return rc;
    PQP_HIP(hipMemcpyAsync(h->c_buf[0].p, s, b_in, hipMemcpyHostToDevice, h->stream));
    PQP_HIP(hipMemcpyAsync(h->c_buf[1].p, x, b_in, hipMemcpyHostToDevice, h->stream));
    PQP_HIP(hipMemcpyAsync(h->c_buf[2].p, y, b_in, hipMemcpyHostToDevice, h->stream));
    if ((rc = pqp_spline_fit_device(h, batch, m, h->c_buf[0].as<double>(), h->c_buf[1].as<double>(), h->c_buf[2].as<double>(),
                                    h->c_buf[3].as<double>(), h->c_buf[4].as<double>())))
        return rc;
    PQP_HIP(hipMemcpyAsync(spline, h->c_buf[3].p, b_spl, hipMemcpyDeviceToHost, h->stream));
    PQP_HIP(hipMemcpyAsync(spline_ext, h->c_buf[4].p, b_ext, hipMemcpyDeviceToHost, h->stream));
    PQP_HIP(hipStreamSynchronize(h->stream));
    return PQP_OK;
}

// ---- layered DP corridor search ------------------------------------------------------------------------
void pqp_dp_default_params(pqp_dp_params* p) {
    if (!p) return;
    p->lateral_range = 10.0; p->longitudinal_spacing = 1.5; p->lateral_spacing = 0.6; p->car_width = 2.0;      // planning_flags.cpp:38-42,10
}

int pqp_dp_corridor_device(pqp_handle* h, int batch, int m, int max_layers, const double* spline, const double* spline_ext,
                           const double* length, const double* start, const float* dist, const int32_t* map_of,
                           const pqp_grid_geometry* geom, const pqp_dp_params* prm, double* layers_s, double* lb, double* ub,
                           int32_t* count, double* vehicle_l) {
    if (!h || !spline || !spline_ext || !length || !start || !dist || !geom || !prm || !layers_s || !lb || !ub || !count || !vehicle_l ||
        batch < 1 || m < 3 || max_layers < 2 || !geometry_ok(geom) || !(prm->lateral_spacing > 0.0) || !(prm->longitudinal_spacing > 0.0) ||
        2.0 * prm->lateral_range / prm->lateral_spacing + 1.0 > 64.0)
        return fail(PQP_ERR_INVALID, "pqp_dp_corridor: bad argument (at most 64 lateral samples per layer)");
    PQP_HIP(hipSetDevice(h->device));
    pqp::DpArgs a;
    a.batch = batch; a.m = m; a.max_layers = max_layers; a.spl = spline; a.spl_ext = spline_ext; a.length = length; a.start = start;
    a.dist = dist; a.map_of = map_of; a.g = *geom; a.p = *prm; a.layers_s = layers_s; a.lb = lb; a.ub = ub; a.count = count; a.vehicle_l = vehicle_l;
    const size_t lds = pqp::DpLds{m, max_layers, pqp::dp_lateral_samples(prm->lateral_range, prm->lateral_spacing)}.total_bytes();
    if (lds > 160 * 1024) return fail(PQP_ERR_CAPACITY, "pqp_dp_corridor: 9 m + 17 max_layers doubles (+ the edge table) exceed one CU's LDS");
    if (lds > 48 * 1024) PQP_HIP(hipFuncSetAttribute((const void*)pqp::dp_corridor_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    h->next_event_pair();
    if (!h->capturing) PQP_HIP(hipEventRecord(h->ev0, h->stream));
    hipLaunchKernelGGL(pqp::dp_corridor_kernel, dim3(batch), dim3(pqp::kDpThreads), lds, h->stream, a);
    PQP_HIP(hipGetLastError());
    if (!h->capturing) PQP_HIP(hipEventRecord(h->ev1, h->stream));
    h->timed = true;
    return PQP_OK;
}

int pqp_dp_corridor(pqp_handle* h, int batch, int m, int max_layers, const double* spline, const double* spline_ext, const double* length,
                    const double* start, const float* dist, int n_maps, const int32_t* map_of, const pqp_grid_geometry* geom,
                    const pqp_dp_params* prm, double* layers_s, double* lb, double* ub, int32_t* count, double* vehicle_l) {
    if (!h || !spline || !spline_ext || !length || !start || !dist || !geom || !prm || !layers_s || !lb || !ub || !count || !vehicle_l ||
        batch < 1 || m < 3 || max_layers < 2 || n_maps < 1)
        return fail(PQP_ERR_INVALID, "pqp_dp_corridor: bad argument");
    PQP_HIP(hipSetDevice(h->device));
    const size_t b_spl = (size_t)batch * 9 * m * 8, b_ext = (size_t)batch * 4 * 8, b_len = (size_t)batch * 8, b_st = (size_t)batch * 3 * 8;
    const size_t b_map = (size_t)n_maps * geom->rows * geom->cols * 4, b_of = (size_t)batch * 4, b_out = (size_t)batch * max_layers * 8;
    const size_t sizes[11] = {b_spl, b_ext, b_len, b_st, b_map, b_of, b_out, b_out, b_out, b_of, b_len};
    int rc;
    for (int k = 0; k < 11; ++k) if ((rc = h->c_buf[k].ensure(sizes[k]))) return rc;
    PQP_HIP(hipMemcpyAsync(h->c_buf[0].p, spline, b_spl, hipMemcpyHostToDevice, h->stream));
    PQP_HIP(hipMemcpyAsync(h->c_buf[1].p, spline_ext, b_ext, hipMemcpyHostToDevice, h->stream));
    PQP_HIP(hipMemcpyAsync(h->c_buf[2].p, length, b_len, hipMemcpyHostToDevice, h->stream));
    PQP_HIP(hipMemcpyAsync(h->c_buf[3].p, start, b_st, hipMemcpyHostToDevice, h->stream));
    PQP_HIP(hipMemcpyAsync(h->c_buf[4].p, dist, b_map, hipMemcpyHostToDevice, h->stream));
    if (map_of) PQP_HIP(hipMemcpyAsync(h->c_buf[5].p, map_of, b_of, hipMemcpyHostToDevice, h->stream));
    for (int k = 6; k < 9; ++k) PQP_HIP(hipMemsetAsync(h->c_buf[k].p, 0, b_out, h->stream));
    if ((rc = pqp_dp_corridor_device(h, batch, m, max_layers, h->c_buf[0].as<double>(), h->c_buf[1].as<double>(), h->c_buf[2].as<double>(),
                                     h->c_buf[3].as<double>(), h->c_buf[4].as<float>(), map_of ? h->c_buf[5].as<int32_t>() : nullptr, geom, prm,
                                     h->c_buf[6].as<double>(), h->c_buf[7].as<double>(), h->c_buf[8].as<double>(), h->c_buf[9].as<int32_t>(),
                                     h->c_buf[10].as<double>())))
        return rc;
    PQP_HIP(hipMemcpyAsync(layers_s, h->c_buf[6].p, b_out, hipMemcpyDeviceToHost, h->stream));
    PQP_HIP(hipMemcpyAsync(lb, h->c_buf[7].p, b_out, hipMemcpyDeviceToHost, h->stream));
    PQP_HIP(hipMemcpyAsync(ub, h->c_buf[8].p, b_out, hipMemcpyDeviceToHost, h->stream));
    PQP_HIP(hipMemcpyAsync(count, h->c_buf[9].p, b_of, hipMemcpyDeviceToHost, h->stream));
    PQP_HIP(hipMemcpyAsync(vehicle_l, h->c_buf[10].p, b_len, hipMemcpyDeviceToHost, h->stream));
    PQP_HIP(hipStreamSynchronize(h->stream));
    return PQP_OK;
}

}  // extern "C"

#include "pqp_chain.inc"
