// pqp_path_stream.hip - the lane-per-QP path-QP kernel (algorithm: pqp_path_lq.hpp).  A translation unit of its own: the kernel is one
// large function and compiles in seconds instead of the 90 s of pqp_kernels.hip.
//
// One wavefront = 64 QPs in lock-step, every per-waypoint quantity streamed through the batch-interleaved workspace
// [wavefront][waypoint][field][lane]: each load / store instruction of a wavefront is one contiguous 512-byte line.  No LDS, no
// cross-lane traffic: HBM (or the Infinity Cache, while the workspace of 256 n bytes per QP fits its 256 MiB) bandwidth bounds it.
// Replaces: the OSQP solves called at src/solver/base_solver.cpp:88,110 for batches that fill the chip's 65 536 lanes.
#include <hip/hip_runtime.h>

#include "pqp_path_lq.hpp"

namespace pqp {
#ifndef PQP_STREAM_OCC
#define PQP_STREAM_OCC 1
#endif
__global__ void __launch_bounds__(64, PQP_STREAM_OCC) path_stream_kernel(const lq::Args a) {
    // (blockDim.x = lanes per wavefront in use: 64, or 32 - half-filled wavefronts, two per SIMD: PQP_STREAM_LANES)
    const int lanes = (int)blockDim.x;
    const int slot = blockIdx.x * lanes + threadIdx.x;
    if (slot >= a.batch) return;
    const int qp = a.order ? a.order[slot] : slot;
    lq::StridedWs ws{a.ws + (size_t)blockIdx.x * a.n * lq::kBlockDoubles * lanes, (int)threadIdx.x, lanes};
    lq::Solver<lq::StridedWs> s(a, qp, ws);
    s.run();
}

// lane slot -> QP of the next launch: QPs sorted by the Riccati sweeps they took in the previous solve of the same batch (a planner
// re-solves nearly the same scenarios cycle after cycle), so that the 64 lanes of a wavefront finish together.  Counting sort in one
// workgroup: 128 bins in LDS; the order inside a bin is whatever the atomics give (results do not depend on the order).
__global__ void __launch_bounds__(1024) stream_order_kernel(int batch, const int32_t* __restrict__ cost, int32_t* __restrict__ order) {
    __shared__ int start[128];
    for (int b = threadIdx.x; b < 128; b += blockDim.x) start[b] = 0;
    __syncthreads();
    for (int q = threadIdx.x; q < batch; q += blockDim.x) atomicAdd(&start[min(max(cost[q], 0), 127)], 1);
    __syncthreads();
    if (threadIdx.x == 0) {
        int acc = 0;
        for (int b = 127; b >= 0; --b) { const int c = start[b]; start[b] = acc; acc += c; }       // most expensive first
    }
    __syncthreads();
    for (int q = threadIdx.x; q < batch; q += blockDim.x) order[atomicAdd(&start[min(max(cost[q], 0), 127)], 1)] = q;
}
}  // namespace pqp

extern "C" hipError_t pqp_stream_order_launch(int batch, const int32_t* cost, int32_t* order, void* stream) {
    hipLaunchKernelGGL(pqp::stream_order_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, batch, cost, order);
    return hipGetLastError();
}

#ifndef PQP_STREAM_LANES
#define PQP_STREAM_LANES 64
#endif
extern "C" hipError_t pqp_stream_launch(const pqp::lq::Args* a, int waves, void* stream) {
    (void)waves;
    const int lanes = PQP_STREAM_LANES;
    hipLaunchKernelGGL(pqp::path_stream_kernel, dim3((a->batch + lanes - 1) / lanes), dim3(lanes), 0, (hipStream_t)stream, *a);
    return hipGetLastError();
}
