// pqp_path_stream.hip - the lane-per-QP path-QP kernel (algorithm: pqp_path_lq.hpp).  A translation unit of its own: the kernel is one
// large function and compiles in seconds instead of the 90 s of pqp_kernels.hip.
//
// One wavefront = 64 QPs in lock-step, every per-waypoint quantity streamed through the batch-interleaved workspace
// [wavefront][waypoint][field][lane]: each load / store instruction of a wavefront is one contiguous 512-byte line.  No LDS, no
// cross-lane traffic: HBM (or the Infinity Cache, while the workspace of 240 n bytes per QP fits its 256 MiB) bandwidth bounds it.
// Replaces: the OSQP solves called at src/solver/base_solver.cpp:88,110 for batches that fill the chip's 65 536 lanes.
#include <hip/hip_runtime.h>

#include "pqp_path_lq.hpp"

namespace pqp {
// (one wavefront per SIMD: the whole 512-register budget; two / four per SIMD spill and lose, half-filled wavefronts two per SIMD lose 1.5x -
//  profiles/r03a_stream_first.txt)
__global__ void __launch_bounds__(64, 1) path_stream_kernel(const lq::Args a) {
    const int lanes = 64;
    // (Sorting the QPs by the sweeps they took in the previous solve, so that a wavefront's 64 lanes finish together, was built and
    // measured: 11 % less traffic and 14 % fewer instructions, the same 10.7 ms at 65 536 QPs - every wavefront is resident at once and the
    // launch lasts as long as its slowest one - and its ordering kernel waited milliseconds for a free slot behind a second launch in
    // flight.  Dropped: profiles/r03d_stream_ordered.txt.)
    const int qp = blockIdx.x * lanes + threadIdx.x;
    if (qp >= a.batch) return;
    lq::StridedWs ws{a.ws + (size_t)blockIdx.x * a.n * lq::kBlockDoubles * lanes, (int)threadIdx.x, lanes};
    lq::Solver<lq::StridedWs> s(a, qp, ws);
    s.run();
}

}  // namespace pqp

extern "C" hipError_t pqp_stream_launch(const pqp::lq::Args* a, int waves, void* stream) {
    (void)waves;
    const int lanes = 64;
    hipLaunchKernelGGL(pqp::path_stream_kernel, dim3((a->batch + lanes - 1) / lanes), dim3(lanes), 0, (hipStream_t)stream, *a);
    return hipGetLastError();
}
