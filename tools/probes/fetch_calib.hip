// fetch_calib.hip - what rocprofv3's FETCH_SIZE / WRITE_SIZE report for a KNOWN byte count moved with path_stream_kernel's access
// pattern (csrc/pqp_path_lq_abi.hpp: a wavefront walks its own block [waypoint][field][lane], every load / store instruction one
// contiguous line of 64 lanes x 8 B or 64 x 4 B), next to the 16 B/lane streaming read the MI355X guide calibrated (x2).
// The footprint (4 GiB) is far beyond the 256 MiB Infinity Cache.  Driver: tools/fetch_calib.py (runs this under rocprofv3 --pmc).
//   hipcc --offload-arch=gfx950 -O3 -o fetch_calib fetch_calib.hip && ./fetch_calib
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

constexpr size_t kWaves = 4096, kBlockBytes = 1u << 20;        // 4096 wavefronts x 1 MiB each
constexpr int kFields = 22;                                      // consecutive lines a "waypoint" reads (the fp64 fields of the workspace)

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

// 8 B per lane: lines of 512 B, kFields of them per step (the loads of one waypoint are issued together, as the kernel's are)
__global__ void __launch_bounds__(64) read_b64(const double* buf, double* sink) {
    const double* blk = buf + (size_t)blockIdx.x * (kBlockBytes / 8);
    const int steps = (int)(kBlockBytes / 8 / 64 / kFields);
    double acc = 0.0;
    for (int i = 0; i < steps; ++i) {
#pragma unroll
        for (int f = 0; f < kFields; ++f) acc += blk[((size_t)i * kFields + f) * 64 + threadIdx.x];
    }
    if (acc == 12345.678) sink[blockIdx.x * 64 + threadIdx.x] = acc;
}
// 4 B per lane: lines of 256 B (the fp32 interior-point fields)
__global__ void __launch_bounds__(64) read_b32(const float* buf, float* sink) {
    const float* blk = buf + (size_t)blockIdx.x * (kBlockBytes / 4);
    const int steps = (int)(kBlockBytes / 4 / 64 / kFields);
    float acc = 0.0f;
    for (int i = 0; i < steps; ++i) {
#pragma unroll
        for (int f = 0; f < kFields; ++f) acc += blk[((size_t)i * kFields + f) * 64 + threadIdx.x];
    }
    if (acc == 12345.678f) sink[blockIdx.x * 64 + threadIdx.x] = acc;
}
// 16 B per lane: the guide's reference pattern (FETCH_SIZE = 1/2 of the bytes)
__global__ void __launch_bounds__(64) read_b128(const double2* buf, double* sink) {
    const double2* blk = buf + (size_t)blockIdx.x * (kBlockBytes / 16);
    const int steps = (int)(kBlockBytes / 16 / 64 / kFields);
    double acc = 0.0;
    for (int i = 0; i < steps; ++i) {
#pragma unroll
        for (int f = 0; f < kFields; ++f) { const double2 v = blk[((size_t)i * kFields + f) * 64 + threadIdx.x]; acc += v.x + v.y; }
    }
    if (acc == 12345.678) sink[blockIdx.x * 64 + threadIdx.x] = acc;
}
__global__ void __launch_bounds__(64) write_b64(double* buf) {
    double* blk = buf + (size_t)blockIdx.x * (kBlockBytes / 8);
    const int steps = (int)(kBlockBytes / 8 / 64 / kFields);
    for (int i = 0; i < steps; ++i) {
#pragma unroll
        for (int f = 0; f < kFields; ++f) blk[((size_t)i * kFields + f) * 64 + threadIdx.x] = (double)(i + f);
    }
}
__global__ void __launch_bounds__(64) write_b32(float* buf) {
    float* blk = buf + (size_t)blockIdx.x * (kBlockBytes / 4);
    const int steps = (int)(kBlockBytes / 4 / 64 / kFields);
    for (int i = 0; i < steps; ++i) {
#pragma unroll
        for (int f = 0; f < kFields; ++f) blk[((size_t)i * kFields + f) * 64 + threadIdx.x] = (float)(i + f);
    }
}
// one waypoint record of the workspace as an interior-point sweep touches it: 22 fp64 + 16 fp32 fields read, 5 fp64 + 15 fp32 written
__global__ void __launch_bounds__(64) sweep_like(double* buf) {
    constexpr int kRec = 30;                                     // doubles per waypoint and lane (kBlockDoubles)
    double* blk = buf + (size_t)blockIdx.x * (kBlockBytes / 8);
    const int steps = (int)(kBlockBytes / 8 / 64 / kRec);
    double acc = 0.0;
    for (int i = 0; i < steps; ++i) {
        double* rec = blk + (size_t)i * kRec * 64;
        float* recf = reinterpret_cast<float*>(rec + 22 * 64);
#pragma unroll
        for (int f = 0; f < 22; ++f) acc += rec[f * 64 + threadIdx.x];
#pragma unroll
        for (int f = 0; f < 16; ++f) acc += recf[f * 64 + threadIdx.x];
#pragma unroll
        for (int f = 12; f < 17; ++f) rec[f * 64 + threadIdx.x] = acc + f;
#pragma unroll
        for (int f = 0; f < 15; ++f) recf[f * 64 + threadIdx.x] = (float)(acc - f);
    }
}

int main() {
    const size_t bytes = kWaves * kBlockBytes;
    void* buf = nullptr;
    double* sink = nullptr;
    CHECK(hipMalloc(&buf, bytes));
    CHECK(hipMalloc(&sink, kWaves * 64 * 8));
    CHECK(hipMemset(buf, 0, bytes));
    CHECK(hipDeviceSynchronize());
    const int steps64 = (int)(kBlockBytes / 8 / 64 / kFields), steps32 = (int)(kBlockBytes / 4 / 64 / kFields), steps128 = (int)(kBlockBytes / 16 / 64 / kFields);
    const int steps_rec = (int)(kBlockBytes / 8 / 64 / 30);
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(read_b64, dim3(kWaves), dim3(64), 0, 0, (const double*)buf, sink);
        hipLaunchKernelGGL(read_b32, dim3(kWaves), dim3(64), 0, 0, (const float*)buf, (float*)sink);
        hipLaunchKernelGGL(read_b128, dim3(kWaves), dim3(64), 0, 0, (const double2*)buf, sink);
        hipLaunchKernelGGL(write_b64, dim3(kWaves), dim3(64), 0, 0, (double*)buf);
        hipLaunchKernelGGL(write_b32, dim3(kWaves), dim3(64), 0, 0, (float*)buf);
        hipLaunchKernelGGL(sweep_like, dim3(kWaves), dim3(64), 0, 0, (double*)buf);
        CHECK(hipDeviceSynchronize());
    }
    // the byte counts the counters are compared with (per launch)
    std::printf("{\"read_b64\": {\"read\": %zu, \"written\": 0}, \"read_b32\": {\"read\": %zu, \"written\": 0}, \"read_b128\": {\"read\": %zu, \"written\": 0}, "
                "\"write_b64\": {\"read\": 0, \"written\": %zu}, \"write_b32\": {\"read\": 0, \"written\": %zu}, "
                "\"sweep_like\": {\"read\": %zu, \"written\": %zu, \"read_fp64\": %zu, \"read_fp32\": %zu, \"written_fp64\": %zu, \"written_fp32\": %zu}}\n",
                kWaves * (size_t)steps64 * kFields * 64 * 8, kWaves * (size_t)steps32 * kFields * 64 * 4, kWaves * (size_t)steps128 * kFields * 64 * 16,
                kWaves * (size_t)steps64 * kFields * 64 * 8, kWaves * (size_t)steps32 * kFields * 64 * 4,
                kWaves * (size_t)steps_rec * 64 * (22 * 8 + 16 * 4), kWaves * (size_t)steps_rec * 64 * (5 * 8 + 15 * 4),
                kWaves * (size_t)steps_rec * 64 * 22 * 8, kWaves * (size_t)steps_rec * 64 * 16 * 4, kWaves * (size_t)steps_rec * 64 * 5 * 8, kWaves * (size_t)steps_rec * 64 * 15 * 4);
    CHECK(hipFree(buf));
    CHECK(hipFree(sink));
    return 0;
}
