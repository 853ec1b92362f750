// pqp_path_solve.hip - the lane-per-waypoint path-QP kernel (gfx950), one translation unit per workgroup width: compiled four times with
// -DPQP_NW=1 / 2 / 4 / 8 (wavefronts per QP; __graft_entry__.py), so that the four register allocations of this 12 000-instruction
// kernel are made side by side instead of one after the other and an experiment on one width rebuilds in a fifth of the time.
//
//   path_solve_kernel<NW, CERT>   persistent workgroups (64*NW lanes, one waypoint per lane) draw QPs from a ticket counter, most expensive
//                         first when the handle knows the QPs' previous cost: assemble -> Ruiz metrics -> block-cyclic-reduction factor ->
//                         ADMM loop + KKT-verified polish -> unpack -> re-linearise -> warm re-solve, everything in VGPRs + 79 KB of LDS
//                         (T = 128: exchange buffers, polish save area, parked Ruiz vectors; two QPs per CU); lanes of one row of 16
//                         exchange through DPP operands, the solver's control state is wave-uniform (scalar branches); HBM is read once
//                         (scenario) and written once (path).  Algorithm: pqp_path_lane.hpp; launcher: pqp_kernels.hip (path_solve_impl).
//                         The last workgroup to finish a launch writes the ticket -> QP map of the next one (order_next_launch).
#include <hip/hip_runtime.h>

#include "pqp_defaults.hpp"
#include "pqp_path_lane.hpp"
#include "pqp_wave.hpp"

#ifndef PQP_NW
#error "compile with -DPQP_NW=1, 2, 4 or 8 (wavefronts per QP)"
#endif

namespace pqp {

// -------------------------------------------------------------------------------------------------------
// device execution context for PathQp: a phase is the code between two workgroup barriers
// -------------------------------------------------------------------------------------------------------
template <int NW, int K, bool MAX>
__device__ __forceinline__ void wg_reduce(double (&v)[K], double* shp) {
#pragma unroll
    for (int k = 0; k < K; ++k) {
        double x = v[k];
        v[k] = MAX ? wave_max(x) : wave_sum(x);
    }
    if (NW > 1) {
        double* red = shp + ShLayout{64 * NW}.red();
        const int w = threadIdx.x >> 6;
        if ((threadIdx.x & 63) == 0)
            for (int k = 0; k < K; ++k) red[k * 16 + w] = v[k];
        __syncthreads();
        // (every load before the first readfirstlane: with the uniform() inside the loop over k each value was an LDS round trip of its own -
        //  six in a row per residual evaluation, round 6)
        double x[K][NW];
#pragma unroll
        for (int k = 0; k < K; ++k) {
#pragma unroll
            for (int j = 0; j < NW; ++j) x[k][j] = red[k * 16 + j];
        }
#pragma unroll
        for (int k = 0; k < K; ++k) {
            double m = x[k][0];
#pragma unroll
            for (int j = 1; j < NW; ++j) m = MAX ? fmax(m, x[k][j]) : m + x[k][j];
            v[k] = uniform(m);
        }
    }
    __syncthreads();
}

// Lane-less context + out-of-line evaluation of the infeasibility certificate (all data in LDS): nothing of it lives in the
// registers of the ADMM loop.
template <int NW>
struct LaneLessCtx {
    double* shp;
    __device__ __forceinline__ int T() const { return 64 * NW; }
    template <class F>
    __device__ __forceinline__ void phase(F f) {
        f((int)threadIdx.x);
        __syncthreads();
    }
    template <int K, class F>
    __device__ __forceinline__ void reduce_max(double (&out)[K], F f) {
        f((int)threadIdx.x, out);
        wg_reduce<NW, K, true>(out, shp);
    }
    template <int K, class F>
    __device__ __forceinline__ void reduce_sum(double (&out)[K], F f) {
        f((int)threadIdx.x, out);
        wg_reduce<NW, K, false>(out, shp);
    }
};
template <int NW>
__device__ __noinline__ bool dev_certificate(double* sh, double fl, double rl, double kap, double eps, double cscale) {
    LaneLessCtx<NW> c{sh};
    return primal_certificate(c, sh, 64 * NW, fl, rl, kap, eps, cscale);
}

template <int NW>
__device__ __noinline__ bool dev_late_certificate(double* sh, int t, double* snap, bool have, LateCertIn in, double fl, double rl, double kap, double eps,
                                                  double cscale) {
    LaneLessCtx<NW> c{sh};
    return late_certificate(c, sh, 64 * NW, t, snap, have, in, fl, rl, kap, eps, cscale);
}

// Hot context: the lane state is a local struct that SROA turns into registers; a phase is the code between two
// workgroup barriers (for a one-wave workgroup the barrier is only a wait on outstanding LDS traffic).
template <int NW>
struct DevCtx {
    // kSaveLds: up to 256 lanes the polish save area fits beside the exchange buffers (72 KB per QP at T = 128, two QPs per CU)
    // kCstLds: pass constants in LDS instead of in registers (no gain at one wavefront per SIMD); kParkScale: the Ruiz vectors are parked
    // between the passes (+2 %, profiles/r02b_variants.txt)
    static constexpr bool kCstLds = false, kParkScale = true, kSaveLds = NW <= kSaveLdsMaxNw;
    static constexpr bool kFinalRefine = NW >= 4;      // pqp_params::polish_final_refine is honoured: the contexts of paths beyond 128 waypoints
    // DPP moves: the value of the lane H below / above in the same row of 16 lanes, of the previous row's last lane; 0 where
    // there is no such lane.  Must run with every lane enabled (a disabled source lane reads as "no lane").  (profiles/r02j_dpp_exchanges.txt)
    static constexpr bool kDpp = true;
    // kCstAcc: twelve pass constants of a waypoint that a solve reads once live in the accumulator registers, written and read by hand
#ifdef PQP_CST_ACC
    static constexpr bool kCstAcc = true;
#else
    static constexpr bool kCstAcc = false;
#endif
    __device__ __forceinline__ static void acc_write(double v, int& lo, int& hi) {
        asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(lo) : "v"(__double2loint(v)));
        asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(hi) : "v"(__double2hiint(v)));
    }
    __device__ __forceinline__ static double acc_read(int lo, int hi) {
        int l, h;
        asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(l) : "a"(lo));
        asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(h) : "a"(hi));
        return __hiloint2double(h, l);
    }
    template <int CTRL, int ROW_MASK>
    __device__ __forceinline__ static double dpp0(double v) {
        const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, 0xf, true);
        const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROW_MASK, 0xf, true);
        return __hiloint2double(hi, lo);
    }
    template <int H> __device__ __forceinline__ static double lane_below(double v) { return dpp0<0x110 + H, 0xf>(v); }     // row_shr:H
    template <int H> __device__ __forceinline__ static double lane_above(double v) { return dpp0<0x100 + H, 0xf>(v); }     // row_shl:H
    __device__ __forceinline__ static double prev_row_last(double v) { return dpp0<0x142, 0xe>(v); }                      // row_bcast:15
    // "every one of these loaded values is needed HERE": one empty asm statement that takes them all.  The loads are then issued together and waited
    // for once; left alone the scheduler - this kernel has no register to spare - loads a few, sums them, reuses their registers for the next
    // few: one LDS round trip after the other on the solve's critical path (round 6: profiles/r06*_lds_round_trips.txt).  Only for batches of six
    // and more loads: behind two loads the fence costs more than the order it enforces (profiles/r06g_*)
    typedef double v3[3];
    __device__ __forceinline__ static void join(v3& a, v3& b, v3& c, v3& d, v3& e, v3& f) {
        asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(c[0]), "+v"(c[1]), "+v"(c[2]),
                          "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(e[0]), "+v"(e[1]), "+v"(e[2]), "+v"(f[0]), "+v"(f[1]), "+v"(f[2]));
    }
    __device__ __forceinline__ static void join(v3& a, v3& b, v3& c, v3& d, v3& e, v3& f, v3& g, v3& h) {
        asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(c[0]), "+v"(c[1]), "+v"(c[2]),
                          "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(e[0]), "+v"(e[1]), "+v"(e[2]), "+v"(f[0]), "+v"(f[1]), "+v"(f[2]),
                          "+v"(g[0]), "+v"(g[1]), "+v"(g[2]), "+v"(h[0]), "+v"(h[1]), "+v"(h[2]));
    }
    __device__ __forceinline__ static double uni(double x) { return uniform(x); }      // a wave-uniform value that reaches control flow
    __device__ __forceinline__ static int uni_int(int x) { return __builtin_amdgcn_readfirstlane(x); }
    Lane lane;
    double* shp;
    __device__ __forceinline__ long long clock() const { return (long long)wall_clock64(); }     // 100 MHz
    __device__ __forceinline__ bool certificate(double* sh, int, double fl, double rl, double kap, double eps, double cscale) {
        return uniform(dev_certificate<NW>(sh, fl, rl, kap, eps, cscale));
    }
    __device__ __forceinline__ bool late_certificate(double* sh, int t, double* snap, bool have, const LateCertIn& in, double fl, double rl, double kap,
                                                     double eps, double cscale) {
        return uniform(dev_late_certificate<NW>(sh, t, snap, have, in, fl, rl, kap, eps, cscale));
    }
    __device__ __forceinline__ int T() const { return 64 * NW; }
    __device__ __forceinline__ double* sh() { return shp; }
    // (round 6: an opaque lane index - asm volatile("" : "+v"(t)) per phase, so that the lane predicates "eliminated at level h" are recomputed where they
    //  are used instead of living as spilled SGPR pairs - has fewer spills and 3 % less throughput: profiles/r06b_*)
    __device__ __forceinline__ static int lane_index() { return (int)threadIdx.x; }
    template <class F>
    __device__ __forceinline__ void phase(F f) {
        f(lane_index(), lane);
        __syncthreads();
    }
    // wave-local phase: LDS operations of one wavefront execute in program order, so lanes of the same wavefront see each
    // other's writes without a workgroup barrier; the fence only stops the compiler from moving LDS accesses across it
    template <class F>
    __device__ __forceinline__ void phase_w(F f) {
        f(lane_index(), lane);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    template <int K, class F>
    __device__ __forceinline__ void reduce_max(double (&out)[K], F f) {
        f(lane_index(), lane, out);
        wg_reduce<NW, K, true>(out, shp);
    }
    template <int K, class F>
    __device__ __forceinline__ void reduce_sum(double (&out)[K], F f) {
        f(lane_index(), lane, out);
        wg_reduce<NW, K, false>(out, shp);
    }
    // The cold operations (assemble, Ruiz, factorisation, polish bookkeeping, unpack) run inline on the same lane state: out of line, on a
    // memory-resident copy, the lane state's round trips cost more than the spills they avoid (0.36x, profiles/r04h_cold_ops_out_of_line_ab.txt)
    template <class PQ>
    __device__ __forceinline__ void cold(PQ& pq, int op, int i0, int i1, double d0) { pq.do_cold(op, i0, i1, d0); }
};

// ticket -> QP of the NEXT launch, most expensive first: cost bins in descending order, within a bin in whatever order this workgroup's
// lanes draw their ranks (results do not depend on the order).  Run by the last workgroup of a launch to leave its ticket loop (every workgroup counts itself out on hist[kCostBins]):
// all keys and bin counts of the launch are complete then.  No separate kernel: with two launches in flight a tiny ordering kernel
// waits for a free CU slot behind the other launch's persistent workgroups (measured: 266 us instead of 3).
// `start`: 256 ints of the workgroup's dynamic LDS (free once its last QP is done).  Not a static array: 1 KB more per workgroup is what kept a FOURTH
// one-wavefront workgroup (paths of up to 64 waypoints: 40.4 KB each) off a compute unit's 160 KB - three QPs per CU, one SIMD idle (round 5).
__device__ void order_next_launch(const PathSolveArgs& args, int* start) {
    __shared__ int s_last;
    // release / acquire around the count-out: this workgroup's keys and bin counts (relaxed agent-scope atomics of record_cost) are
    // visible before its count is, and the last workgroup reads the other XCDs' keys only after it has seen every count
    if (threadIdx.x == 0) {
        __threadfence();
        s_last = atomicAdd(args.cost_hist + kCostBins, 1) == (int)gridDim.x - 1;
        __threadfence();
    }
    __syncthreads();
    if (!s_last) return;
    const int nt = blockDim.x;
    for (int b = threadIdx.x; b < kCostBins; b += nt) start[b] = __hip_atomic_load(args.cost_hist + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (threadIdx.x == 0) {                 // exclusive suffix sum over 256 bins: QPs in more expensive bins
        // (thr: the cheapest bin b such that the QPs in bins >= b are still no more than 1 / k of the batch - a bin that would overshoot that share stays
        //  out whole: 999 QPs in one bin and one above it carry the one, not all thousand.  k = the handle's PQP_OPT_CARRY_CYCLES on EVERY launch,
        //  the cold first one included: the second launch's threshold is then the user's k too, not a fallback)
        int acc = 0, thr = kCostBins;
        const long long k = args.carry_k > 1 ? args.carry_k : 8;
        for (int b = kCostBins - 1; b >= 0; --b) {
            const int c = start[b]; start[b] = acc;
            if (thr == b + 1 && k * (acc + c) <= args.batch) thr = b;
            acc += c;
        }
        // PQP_OPT_CARRY_CYCLES = k >= 2: the bin from which on a QP counts as one of the launch's expensive ones (read by the next launch's QPs)
        __hip_atomic_store(args.cost_hist + kCostBins + 1, thr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    for (int q = threadIdx.x; q < args.batch; q += nt) {
        const int k = __hip_atomic_load(args.cost_key + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int pos = atomicAdd(&start[(k >> 24) & 0xff], 1);          // (LDS atomic: the rank of q within its bin)
        if (pos < args.batch) args.order_next[pos] = q;
    }
    __syncthreads();
    for (int b = threadIdx.x; b <= kCostBins; b += nt) __hip_atomic_store(args.cost_hist + b, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // for the next launch
}

// One wavefront per SIMD: the lane state + the factorisation take the whole 512-register budget (occupancy 2 on 256: 0.41-0.53x,
// profiles/r04f_occupancy2_ab.txt)
template <int NW, bool CERT>
__global__ void __launch_bounds__(64 * NW, 1) path_solve_kernel(const PathSolveArgs args) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    __shared__ int s_ticket;
    // Persistent workgroups: every workgroup draws tickets until the batch is used up (each workgroup ends on one ticket beyond
    // it, so a launch consumes exactly batch + gridDim.x tickets and the host knows the next launch's base without a reset).
    // (Drawing the next ticket while the current QP is solved - to hide the ~2 us of the returning atomic - was measured and dropped: at
    // batch 1024 on 512 slots every workgroup then reserves its second QP the moment it starts its first, the most expensive QPs of the
    // first round pair up with the most expensive of the rest, and the launch takes 0.69 instead of 0.53 ms.)
    for (;;) {
#ifdef PQP_TIMING
        const long long t_ticket0 = (long long)wall_clock64();
#endif
        if (threadIdx.x == 0) s_ticket = (int)(atomicAdd(args.ticket, 1ull) - args.ticket_base);
        __syncthreads();
        const int ticket = __builtin_amdgcn_readfirstlane(s_ticket);
        __syncthreads();
        if ((unsigned)ticket >= (unsigned)args.batch) break;      // (unsigned: a ticket below the base - a host/device counter mismatch - ends the workgroup too)
        const int qp = args.order ? args.order[ticket] : ticket;
        // (written out rather than through PathQp::count_of: with the call here the register allocator spills 170 VGPRs of the loop)
        if ((args.n_of ? args.n_of[qp] : args.n) < 2) {
            // nothing to optimise: defined outputs for everything a later call may read (status, counters, warm state)
            if (threadIdx.x == 0) {
                if (args.status) args.status[qp] = PQP_STATUS_UNSOLVED;
                if (args.iters) args.iters[qp] = 0;
                if (args.info) for (int k = 0; k < PQP_INFO_STRIDE; ++k) args.info[(size_t)qp * PQP_INFO_STRIDE + k] = 0.0;
                args.wrho[qp] = args.prm.rho;
                args.wye[2 * (size_t)qp] = 0.0; args.wye[2 * (size_t)qp + 1] = 0.0;
                if (args.cost_key) record_cost(args, qp, 0);
            }
            if (args.store_warm)
                for (int k = threadIdx.x; k < args.n * 6; k += blockDim.x) {
                    args.wx[(size_t)qp * args.n * 6 + k] = 0.0;
                    args.wy[(size_t)qp * args.n * 6 + k] = 0.0;
                }
            continue;
        }
        DevCtx<NW> ctx;
        ctx.shp = smem;
        PathQp<DevCtx<NW>, CERT> solver(ctx, args, qp, (int)blockIdx.x);
#ifdef PQP_TIMING
        const long long t_ticket1 = (long long)wall_clock64();
#endif
        solver.run();
        __syncthreads();
#ifdef PQP_TIMING
        if (threadIdx.x == 0) {                                                                                          // debug build only
            double* dbg = args.out + (size_t)qp * args.n * PQP_OUT_STRIDE;
            dbg[8] = (double)(t_ticket1 - t_ticket0);
            dbg[9] = (double)t_ticket0; dbg[10] = (double)(long long)wall_clock64(); dbg[11] = (double)blockIdx.x;       // the schedule: start, end, slot
        }
#endif
    }
    if (args.cost_key) order_next_launch(args, reinterpret_cast<int*>(smem));
}

}  // namespace pqp

// the two kernels of this width for the launcher (a kernel's host stub lives in the translation unit that defines it)
#define PQP_CAT2(a, b) a##b
#define PQP_CAT(a, b) PQP_CAT2(a, b)
extern "C" __attribute__((visibility("hidden"))) const void* PQP_CAT(pqp_path_solve_fn_nw, PQP_NW)(int cert) {
    return cert ? (const void*)pqp::path_solve_kernel<PQP_NW, true> : (const void*)pqp::path_solve_kernel<PQP_NW, false>;
}
