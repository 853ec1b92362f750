// accuracy of v_rcp_f64 (the hardware seed) and of one / two Newton steps on it: what pqp::rcp builds on
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
__global__ void k(const double* x, double* o, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double v = x[i];
    double r0 = __builtin_amdgcn_rcp(v);
    double r1 = fma(fma(-v, r0, 1.0), r0, r0);
    double r2 = fma(fma(-v, r1, 1.0), r1, r1);
    o[3 * i] = r0; o[3 * i + 1] = r1; o[3 * i + 2] = r2;
}
int main() {
    const int n = 1 << 20;
    std::vector<double> x(n), o(3 * n);
    unsigned long long s = 88172645463325252ull;
    for (int i = 0; i < n; ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; x[i] = std::ldexp(1.0 + (double)(s >> 11) / 9007199254740992.0, (int)(s % 61) - 30); }
    double *dx, *dout;
    hipMalloc(&dx, n * 8); hipMalloc(&dout, 3 * n * 8);
    hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, dout, n);
    hipMemcpy(o.data(), dout, 3 * n * 8, hipMemcpyDeviceToHost);
    double e[3] = {0, 0, 0};
    for (int i = 0; i < n; ++i) for (int j = 0; j < 3; ++j) e[j] = std::fmax(e[j], std::fabs(o[3 * i + j] * x[i] - 1.0));
    std::printf("max relative error: seed %.3e, one Newton step %.3e, two %.3e\n", e[0], e[1], e[2]);
    return 0;
}
