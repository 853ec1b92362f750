// Issue cost of v_rcp_f64 against v_fma_f64 and friends for lone wavefronts (the lane-per-QP kernel's situation): clock64() ticks per instruction - read the RATIOS.
//   hipcc --offload-arch=gfx950 -O2 -o rcp_rate_probe rcp_rate_probe.hip && ./rcp_rate_probe
#include <hip/hip_runtime.h>
#include <cstdio>
template <int WHICH>
__global__ void __launch_bounds__(64, 1) k(double* out, long long* cyc, int iters) {
    double a[8];
    for (int j = 0; j < 8; ++j) a[j] = 1.0 + 0.001 * (threadIdx.x + 64 * j);
    const double c = 1.0000001, d = 1e-9;
    const long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (WHICH == 0) asm volatile("v_rcp_f64 %0, %0" : "+v"(a[j]));
            if (WHICH == 1) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[j]) : "v"(c), "v"(d));
            if (WHICH == 2) asm volatile("v_rsq_f64 %0, %0" : "+v"(a[j]));
            if (WHICH == 3) asm volatile("v_rcp_f32 %0, %0" : "+v"(*(float*)&a[j]));
            if (WHICH == 4) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a[j]) : "v"(c));
            if (WHICH == 5) asm volatile("v_max_f64 %0, %0, %1" : "+v"(a[j]) : "v"(c));
            if (WHICH == 6) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(*(float*)&a[j]) : "v"(a[(j + 1) & 7]));
        }
    }
    const long long t1 = clock64();
    double s = 0; for (int j = 0; j < 8; ++j) s += a[j];
    out[blockIdx.x * 64 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
int main() {
    double* out; long long* cyc; hipMalloc(&out, 1024 * 64 * 8); hipMalloc(&cyc, 1024 * 8);
    const char* names[] = {"v_rcp_f64", "v_fma_f64", "v_rsq_f64", "v_rcp_f32", "v_mul_f64", "v_max_f64", "v_cvt_f32_f64"};
    const int iters = 20000;
    for (int w = 0; w < 7; ++w) {
        for (int rep = 0; rep < 2; ++rep) {
            if (w == 0) hipLaunchKernelGGL(k<0>, dim3(1024), dim3(64), 0, 0, out, cyc, iters);
            if (w == 1) hipLaunchKernelGGL(k<1>, dim3(1024), dim3(64), 0, 0, out, cyc, iters);
            if (w == 2) hipLaunchKernelGGL(k<2>, dim3(1024), dim3(64), 0, 0, out, cyc, iters);
            if (w == 3) hipLaunchKernelGGL(k<3>, dim3(1024), dim3(64), 0, 0, out, cyc, iters);
            if (w == 4) hipLaunchKernelGGL(k<4>, dim3(1024), dim3(64), 0, 0, out, cyc, iters);
            if (w == 5) hipLaunchKernelGGL(k<5>, dim3(1024), dim3(64), 0, 0, out, cyc, iters);
            if (w == 6) hipLaunchKernelGGL(k<6>, dim3(1024), dim3(64), 0, 0, out, cyc, iters);
            hipDeviceSynchronize();
        }
        long long h[1024]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
        double m = 0; for (int b = 0; b < 1024; ++b) m += h[b];
        printf("%-14s %.3f clock64() ticks per instruction  (8 independent chains per wavefront, 1024 one-wavefront workgroups)\n", names[w], m / 1024 / (8.0 * iters));
    }
    return 0;
}
