// pqp_path_stream.hip - the lane-per-QP path-QP kernel (algorithm: pqp_path_lq.hpp).  A translation unit of its own: the kernel is one
// large function and compiles in seconds instead of the 90 s of pqp_kernels.hip.
//
// One wavefront = 64 QPs in lock-step, every per-waypoint quantity streamed through the batch-interleaved workspace
// [wavefront][waypoint][field][lane]: each load / store instruction of a wavefront is one contiguous 512-byte line.  No cross-lane
// traffic: HBM (or the Infinity Cache, while the workspace of 240 n bytes per QP fits its 256 MiB) bandwidth bounds it - once its latency
// is hidden: one wavefront per SIMD, nothing else to switch to, and a sweep is a dependency chain that wants a waypoint's record every
// ~0.8 us of arithmetic while a load takes ~2 us to come back.  Round 6: the sweeps' records are copied workspace -> LDS by LDS-direct
// loads (global_load_lds_dwordx4: no destination registers, so kStageDepth waypoints ahead cost none - in registers every depth beyond 1 spilled and lost).
// Replaces: the OSQP solves called at src/solver/base_solver.cpp:88,110 for batches that fill the chip's 65 536 lanes.
#include <hip/hip_runtime.h>

#include <type_traits>

#include "pqp_path_lq.hpp"

namespace pqp {

// ---- the workspace view of the device: StridedWs + LDS staging of the sweeps' records --------------------------------------------------
// A slot = one workspace block ([chunk][lane][16 bytes], lq::kBlockChunks KB); one global_load_lds_dwordx4 copies chunk c of every lane's OWN
// waypoint into it (16 bytes per lane from the lane's address, LDS address = M0 + 16 * lane) - lanes of a wavefront may be at different
// waypoints (ragged batches) or masked off (their QP's rounds are over): each lane's copy only feeds that lane.
// The copies are inline assembly: through the builtin the compiler knows that an LDS-direct load writes LDS and waits for ALL outstanding ones
// (s_waitcnt vmcnt(0)) before every LDS read - the pipeline's depth would be gone.  Here the wait is staged_wait(): at most `later` records'
// copies may still be in flight; whatever else the wavefront issued in between (its stores) only makes the wait longer, never too short.
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"      // (M0 is reserved, and named as clobbered: the compiler then restores its own uses)
struct StagedWs : lq::ChunkWs {
    static constexpr int kStageDepth = 2;
    const double* lds;        // the wavefront's slots [kStageDepth][kBlockDoubles][64]
    unsigned lds_addr;        // their LDS byte address
    __device__ __forceinline__ void stage_chunk(int slot, int chunk, int i) const {
        const double* g = block + chunk_at(chunk, i);
        const unsigned to = lds_addr + (unsigned)((slot * lq::kBlockChunks + chunk) * 1024);
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(g), "s"(to) : "memory", "m0");
    }
    // Several chunks of ONE waypoint in one statement: the instruction's 13-bit signed offset moves the source address AND the LDS destination (measured:
    // tools/probes/glds_offset_probe.hip), and a slot is laid out like a workspace block - so the chunks within -4 ... +3 of a centre chunk share one address
    // register pair and one M0: two VALU and two SALU instructions less per chunk after a group's first.
    template <int CENTER, int... CS> __device__ __forceinline__ void stage_group(int slot, int i) const {
        constexpr int n = sizeof...(CS);
        constexpr int o[] = {(CS - CENTER) * 1024 ...};
        static_assert(n >= 1 && n <= 5, "one asm string per count");
        static_assert(((CS - CENTER >= -4 && CS - CENTER <= 3) && ...), "the offset field holds -4096 ... 4095");
        const double* g = block + chunk_at(CENTER, i);
        const unsigned to = lds_addr + (unsigned)((slot * lq::kBlockChunks + CENTER) * 1024);
#define PQP_GLDS "\n\tglobal_load_lds_dwordx4 %0, off offset:"
        if constexpr (n == 1) asm volatile("s_mov_b32 m0, %1\n\ts_nop 0" PQP_GLDS "%2" : : "v"(g), "s"(to), "n"(o[0]) : "memory", "m0");
        if constexpr (n == 2) asm volatile("s_mov_b32 m0, %1\n\ts_nop 0" PQP_GLDS "%2" PQP_GLDS "%3" : : "v"(g), "s"(to), "n"(o[0]), "n"(o[n > 1 ? 1 : 0]) : "memory", "m0");
        if constexpr (n == 3) asm volatile("s_mov_b32 m0, %1\n\ts_nop 0" PQP_GLDS "%2" PQP_GLDS "%3" PQP_GLDS "%4" : : "v"(g), "s"(to), "n"(o[0]), "n"(o[n > 1 ? 1 : 0]), "n"(o[n > 2 ? 2 : 0]) : "memory", "m0");
        if constexpr (n == 4) asm volatile("s_mov_b32 m0, %1\n\ts_nop 0" PQP_GLDS "%2" PQP_GLDS "%3" PQP_GLDS "%4" PQP_GLDS "%5" : : "v"(g), "s"(to), "n"(o[0]), "n"(o[n > 1 ? 1 : 0]), "n"(o[n > 2 ? 2 : 0]), "n"(o[n > 3 ? 3 : 0]) : "memory", "m0");
        if constexpr (n == 5) asm volatile("s_mov_b32 m0, %1\n\ts_nop 0" PQP_GLDS "%2" PQP_GLDS "%3" PQP_GLDS "%4" PQP_GLDS "%5" PQP_GLDS "%6" : : "v"(g), "s"(to), "n"(o[0]), "n"(o[n > 1 ? 1 : 0]), "n"(o[n > 2 ? 2 : 0]), "n"(o[n > 3 ? 3 : 0]), "n"(o[n > 4 ? 4 : 0]) : "memory", "m0");
#undef PQP_GLDS
    }
    __device__ __forceinline__ const double* slot_chunk(int slot, int chunk) const { return lds + ((slot * lq::kBlockChunks + chunk) * 64 + lane) * 2; }
    __device__ __forceinline__ double slot_ld(int slot, int f) const { return slot_chunk(slot, f >> 1)[f & 1]; }
    __device__ __forceinline__ float slot_ldf(int slot, int f) const { return reinterpret_cast<const float*>(slot_chunk(slot, lq::kFieldsD / 2 + (f >> 2)))[f & 3]; }
    // A slot is read (ds_read) and then refilled by the copies of the record two waypoints on.  The reads are issued first, and a copy's data comes back a memory
    // round trip later - but nothing in the ISA orders an LDS read against a later LDS-direct write: the reads are waited for before the slot is handed over.
    __device__ __forceinline__ void reads_done() const { asm volatile("s_waitcnt lgkmcnt(0)" : : : "memory"); }
    template <int N> __device__ __forceinline__ static void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" : : "n"(N < 63 ? N : 63) : "memory"); }
    // before a record is read: at most the copies of `later` (0 .. kStageDepth - 1) records issued after it may be outstanding, C copies each
    template <int C> __device__ __forceinline__ void staged_wait(int later) const {
        if (later >= 1) wait_vm<C>(); else wait_vm<0>();
        static_assert(kStageDepth == 2, "one case per depth");
    }
};
#pragma clang diagnostic pop

// slot -> QP of the NEXT launch, heaviest phase keys first: a counting sort over the 32 768 key bins, run by the last wavefront of the launch to finish (all
// keys and bin counts are complete then; no separate kernel: with a second launch in flight a tiny ordering kernel waits for a free SIMD behind its 1024
// resident wavefronts).  64 lanes: an exclusive scan of the bins in descending key order, 64 bins per step, then every QP draws its rank from its bin.
__device__ void stream_order_next(const lq::Args& a) {
    const int lane = (int)threadIdx.x;
    // (every loop below keeps several independent device-scope accesses in flight per lane: one returning atomic at a time is ~1.2 us, and the first
    //  form of this function - one per iteration - took 1.2 ms of a 10.6 ms launch: exactly what the order had saved)
    constexpr int US = 8;                                                       // bins per lane and scan step
    int carry = 0;
    for (int base = lq::kOrderBins - 64 * US; base >= 0; base -= 64 * US) {
        // group u = bins [base + 64 (US - 1 - u), + 64): lane 0 of group 0 takes the heaviest bin
        int c[US];
#pragma unroll
        for (int u = 0; u < US; ++u) c[u] = __hip_atomic_load(a.hist + base + 64 * (US - 1 - u) + 63 - lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
        for (int u = 0; u < US; ++u) {
            int incl = c[u];
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) { const int t = __shfl_up(incl, d); if (lane >= d) incl += t; }
            __hip_atomic_store(a.hist + base + 64 * (US - 1 - u) + 63 - lane, carry + incl - c[u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // QPs in heavier bins
            carry += __shfl(incl, 63);
        }
    }
    __threadfence();
    constexpr int UQ = 16;                                                      // QPs per lane and scatter step
    for (int q0 = 0; q0 < a.batch; q0 += 64 * UQ) {
        int k[UQ], pos[UQ];
#pragma unroll
        for (int u = 0; u < UQ; ++u) { const int q = q0 + 64 * u + lane; k[u] = q < a.batch ? __hip_atomic_load(a.key_out + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : -1; }
#pragma unroll
        for (int u = 0; u < UQ; ++u) pos[u] = k[u] >= 0 ? __hip_atomic_fetch_add(a.hist + (k[u] & (lq::kOrderBins - 1)), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : a.batch;
#pragma unroll
        for (int u = 0; u < UQ; ++u) if (pos[u] < a.batch) a.order_next[pos[u]] = q0 + 64 * u + lane;
    }
    __threadfence();
    for (int b = lane; b <= lq::kOrderBins; b += 64) __hip_atomic_store(a.hist + b, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // for the next launch
}

// (one wavefront per SIMD: the whole 512-register budget; two / four per SIMD spill and lose, half-filled wavefronts two per SIMD lose 1.5x -
//  profiles/r03a_stream_first.txt)
// STAGED: the [chunk][lane] workspace with the sweeps' records staged in LDS (launches that leave SIMDs idle); else the [field][lane] workspace with the
// register prefetch (launches that fill the chip) - Args::staged, chosen by the launcher
template <bool STAGED>
__global__ void __launch_bounds__(64, 1) path_stream_kernel(const lq::Args a_by_value) {
    using Ws = typename std::conditional<STAGED, StagedWs, lq::StridedWs>::type;
    const int lanes = 64;
    // Wavefronts of similar work (a.order, PQP_OPT_ORDER_BY_COST): the 64 lanes of a wavefront run every phase - interior-point iterations and active-set
    // rounds of each pass - as often as their slowest lane.  Sorting by the TOTAL sweeps of the previous solve changes nothing (round 3: 11 % less
    // traffic, the same 10.7 ms - profiles/r03d_stream_ordered.txt); sorting by the four phase counts, first pass first, does: 25.4 -> 16.8 lock-step
    // phases per wavefront for 16.5 per lane, 11.1 -> 9.4 ms at 65 536 QPs with the counts of the identical batch, 10.3 ms with those of the previous
    // planning cycle (profiles/r05g_stream_sorted_probe.txt).
    // (STAGED: the argument block is read where the launch put it - a reference to the by-value parameter becomes a private copy as soon as the function holds a
    //  statement that may write memory - the LDS copies are such statements - and the solver would read pqp_params from scratch)
    const lq::Args* ap = &a_by_value;
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (STAGED) ap = (const lq::Args*)__builtin_amdgcn_kernarg_segment_ptr();
#endif
    const lq::Args& a = *ap;
    const int slot = blockIdx.x * lanes + threadIdx.x;
    const bool live = slot < a.batch;
    if (live) {
        const int qp = a.order ? a.order[slot] : slot;
        Ws ws;
        ws.block = a.ws + (size_t)blockIdx.x * a.n * lq::kBlockDoubles * lanes; ws.lane = (int)threadIdx.x; ws.lanes = lanes;
        if constexpr (STAGED) {
            __shared__ __attribute__((aligned(16))) double stage_lds[StagedWs::kStageDepth * lq::kBlockDoubles * 64];      // 30 KB: four wavefronts per CU keep 120 of its 160 KB
            ws.lds = stage_lds; ws.lds_addr = (unsigned)(size_t)(__attribute__((address_space(3))) double*)stage_lds;
        }
        lq::Solver<Ws> s(a, qp, ws);
        s.run();
        if (a.key_out) {
            const int it1 = s.ipm_iters_first < 31 ? s.ipm_iters_first : 31, s1 = s.set_rounds_first < 7 ? s.set_rounds_first : 7;
            const int it2r = s.ipm_iters - s.ipm_iters_first, s2r = s.set_rounds - s.set_rounds_first;
            const int it2 = it2r < 15 ? it2r : 15, s2 = s2r < 7 ? s2r : 7;
            const int key = (it1 << 10) | (s1 << 7) | (it2 << 3) | s2;
            // (device-scope: the last wavefront of the launch, on whatever XCD, reads them with device-scope loads)
            __hip_atomic_store(a.key_out + qp, key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            (void)__hip_atomic_fetch_add(a.hist + key, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (a.key_out) {
        int last = 0;
        if (threadIdx.x == 0) {
            __threadfence();
            last = atomicAdd(a.hist + lq::kOrderBins, 1) == (int)gridDim.x - 1;
            __threadfence();
        }
        if (__shfl(last, 0)) stream_order_next(a);
    }
}

}  // namespace pqp

extern "C" hipError_t pqp_stream_launch(const pqp::lq::Args* a, int waves, void* stream) {
    (void)waves;
    const int lanes = 64;
    if (a->staged) hipLaunchKernelGGL(pqp::path_stream_kernel<true>, dim3((a->batch + lanes - 1) / lanes), dim3(lanes), 0, (hipStream_t)stream, *a);
    else hipLaunchKernelGGL(pqp::path_stream_kernel<false>, dim3((a->batch + lanes - 1) / lanes), dim3(lanes), 0, (hipStream_t)stream, *a);
    return hipGetLastError();
}
