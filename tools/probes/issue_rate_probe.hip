// What does ONE wavefront on a SIMD get per instruction?  (Both path-QP kernels run one wavefront per SIMD.)  8 independent chains per wavefront, so no instruction waits
// for its operands; clock64() ticks (= shader cycles) and wall-clock per instruction and wavefront, for 1 / 2 / 4 one-wavefront workgroups per SIMD.
//   hipcc --offload-arch=gfx950 -O2 -o issue_rate_probe issue_rate_probe.hip && ./issue_rate_probe
#include <hip/hip_runtime.h>
#include <cstdio>
template <int WHICH>
__global__ void __launch_bounds__(64, 1) k(double* out, long long* cyc, int iters) {
    double a[8];
    unsigned s[8];
    for (int j = 0; j < 8; ++j) { a[j] = 1.0 + 0.001 * (threadIdx.x + 64 * j); s[j] = __builtin_amdgcn_readfirstlane(j + blockIdx.x); }
    const double c = 1.0000001, d = 1e-9;
    const long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u)          // 64 statements per trip: the loop's own branch is amortised
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (WHICH == 0) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[j]) : "v"(c), "v"(d));
            if (WHICH == 1) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(*(float*)&a[j]) : "v"((float)c), "v"((float)d));
            if (WHICH == 2) asm volatile("v_mov_b32 %0, %1" : "=v"(*(unsigned*)&a[j]) : "v"(*(unsigned*)&a[(j + 4) & 7]));
            if (WHICH == 3) asm volatile("s_add_u32 %0, %0, 1" : "+s"(s[j]) : : "scc");
            if (WHICH == 4) asm volatile("v_fma_f64 %0, %0, %2, %3\n\ts_add_u32 %1, %1, 1" : "+v"(a[j]), "+s"(s[j]) : "v"(c), "v"(d) : "scc");
            if (WHICH == 5) asm volatile("v_accvgpr_write_b32 a0, %0\n\tv_accvgpr_read_b32 %0, a0" : "+v"(*(unsigned*)&a[j]) : : "a0");
            if (WHICH == 6) asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "=v"(*(unsigned*)&a[j]) : "v"(*(unsigned*)&a[(j + 4) & 7]));
            if (WHICH == 7) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[0]) : "v"(c), "v"(d));        // ONE dependent chain
        }
    }
    const long long t1 = clock64();
    double r = 0; for (int j = 0; j < 8; ++j) r += a[j] + s[j];
    out[blockIdx.x * 64 + threadIdx.x] = r;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int W> void run(int blocks, double* out, long long* cyc, int iters) { hipLaunchKernelGGL(k<W>, dim3(blocks), dim3(64), 0, 0, out, cyc, iters); }
int main() {
    double* out; long long* cyc; (void)hipMalloc(&out, 4096 * 64 * 8); (void)hipMalloc(&cyc, 4096 * 8);
    const char* names[] = {"v_fma_f64", "v_fma_f32", "v_mov_b32", "s_add_u32", "v_fma_f64 + s_add_u32 (pair)", "v_accvgpr_write + read (pair)", "v_mov_b32_dpp", "v_fma_f64, one dependent chain"};
    const int iters = 4000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int blocks : {1024, 2048, 4096}) {
        printf("%d one-wavefront workgroups (%d per SIMD)\n", blocks, blocks / 1024);
        for (int w = 0; w < 8; ++w) {
            float ms = 0;
            for (int rep = 0; rep < 2; ++rep) {
                (void)hipEventRecord(e0, 0);
                switch (w) { case 0: run<0>(blocks, out, cyc, iters); break; case 1: run<1>(blocks, out, cyc, iters); break; case 2: run<2>(blocks, out, cyc, iters); break; case 3: run<3>(blocks, out, cyc, iters); break;
                             case 4: run<4>(blocks, out, cyc, iters); break; case 5: run<5>(blocks, out, cyc, iters); break; case 6: run<6>(blocks, out, cyc, iters); break; default: run<7>(blocks, out, cyc, iters); }
                (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1); (void)hipEventElapsedTime(&ms, e0, e1);
            }
            static long long h[4096]; (void)hipMemcpy(h, cyc, blocks * sizeof(long long), hipMemcpyDeviceToHost);
            double m = 0; for (int b = 0; b < blocks; ++b) m += h[b];
            printf("  %-32s %7.3f cycles per statement and wavefront  (kernel %.3f ms)\n", names[w], m / blocks / (64.0 * iters), ms);
        }
    }
    return 0;
}
