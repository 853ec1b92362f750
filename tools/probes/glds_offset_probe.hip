// Where does global_load_lds_dwordx4 with an instruction offset read from and write to?  One wavefront, lane j points at g + 4 j (uints: 16 bytes per lane),
// M0 = the LDS base; the instruction carries offset:1024.  Prints the global index range that arrived and the LDS offset it arrived at.
//   hipcc --offload-arch=gfx950 -O2 -o glds_offset_probe glds_offset_probe.hip && ./glds_offset_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void probe(const unsigned* g, unsigned* out) {
    __shared__ __attribute__((aligned(16))) unsigned lds[2048];
    for (int k = threadIdx.x; k < 2048; k += 64) lds[k] = 0xffffffffu;
    __syncthreads();
    const unsigned* p = g + 4 * threadIdx.x;
    const unsigned base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned*)lds;
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off offset:1024\n\ts_waitcnt vmcnt(0)" : : "v"(p), "s"(base) : "memory", "m0");
    __syncthreads();
    for (int k = threadIdx.x; k < 2048; k += 64) out[k] = lds[k];
}
int main() {
    std::vector<unsigned> h(4096);
    for (int i = 0; i < 4096; ++i) h[i] = i;
    unsigned *g, *o;
    hipMalloc(&g, 4096 * 4); hipMalloc(&o, 2048 * 4);
    hipMemcpy(g, h.data(), 4096 * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, g, o);
    std::vector<unsigned> r(2048);
    hipMemcpy(r.data(), o, 2048 * 4, hipMemcpyDeviceToHost);
    int first = -1, last = -1;
    for (int k = 0; k < 2048; ++k) if (r[k] != 0xffffffffu) { if (first < 0) first = k; last = k; }
    if (first < 0) { printf("nothing arrived\n"); return 0; }
    printf("LDS dwords [%d, %d] written; first holds global dword %u, last %u  (offset:1024 = 256 dwords: source %s, destination %s)\n", first, last, r[first], r[last],
           r[first] == 256 ? "+offset" : (r[first] == 0 ? "without offset" : "?"), first == 256 ? "+offset" : (first == 0 ? "without offset" : "?"));
    return 0;
}
