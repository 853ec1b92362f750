// A torch-free caller of the multi-GPU driver + its RCCL gather (tests/test_multi_and_batched.py): reads one batch (binary: int32 batch, n; then
// ref, bounds, scal as doubles), solves it on `shards` shards (devices 0 .. shards - 1), gathers the paths on every shard's GPU with
// pqp_multi_gather_paths and compares each GPU's copy with the host copy of pqp_multi_path_solve.  Prints "gather ok <shards> <batch>".
#include <hip/hip_runtime_api.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/pqp.h"

int main(int argc, char** argv) {
    if (argc < 3) { std::fprintf(stderr, "usage: gather_demo <file> <shards>\n"); return 2; }
    const int shards = std::atoi(argv[2]);
    FILE* f = std::fopen(argv[1], "rb");
    if (!f) { std::perror("open"); return 2; }
    int32_t hdr[2];
    if (std::fread(hdr, 4, 2, f) != 2) return 2;
    const int batch = hdr[0], n = hdr[1];
    std::vector<double> ref((size_t)batch * n * PQP_REF_STRIDE), bounds((size_t)batch * n * PQP_BOUNDS_STRIDE), scal((size_t)batch * PQP_SCAL_STRIDE);
    if (std::fread(ref.data(), 8, ref.size(), f) != ref.size() || std::fread(bounds.data(), 8, bounds.size(), f) != bounds.size() ||
        std::fread(scal.data(), 8, scal.size(), f) != scal.size()) return 2;
    std::fclose(f);
    pqp_params prm;
    pqp_production_params(&prm);
    pqp_multi* m = nullptr;
    if (pqp_multi_create(&m, &prm, shards, nullptr, (batch + shards - 1) / shards, n) != PQP_OK) { std::fprintf(stderr, "create: %s\n", pqp_last_error()); return 1; }
    const size_t total = (size_t)batch * n * PQP_OUT_STRIDE;
    std::vector<double> out(total), back(total);
    std::vector<int32_t> status(batch);
    if (pqp_multi_path_solve(m, batch, n, nullptr, ref.data(), nullptr, bounds.data(), scal.data(), 1, out.data(), status.data(), nullptr, nullptr) != PQP_OK) {
        std::fprintf(stderr, "solve: %s\n", pqp_last_error()); return 1;
    }
    std::vector<double*> full(shards, nullptr);
    for (int g = 0; g < shards; ++g) {
        if (hipSetDevice(g) != hipSuccess || hipMalloc((void**)&full[g], total * 8) != hipSuccess || hipMemset(full[g], 0xff, total * 8) != hipSuccess) { std::fprintf(stderr, "hipMalloc on device %d\n", g); return 1; }
        (void)hipDeviceSynchronize();
    }
    if (pqp_multi_gather_paths(m, batch, n, full.data()) != PQP_OK) { std::fprintf(stderr, "gather: %s\n", pqp_last_error()); return 1; }
    for (int g = 0; g < shards; ++g) {
        if (hipSetDevice(g) != hipSuccess || hipMemcpy(back.data(), full[g], total * 8, hipMemcpyDeviceToHost) != hipSuccess) return 1;
        if (std::memcmp(back.data(), out.data(), total * 8) != 0) { std::fprintf(stderr, "device %d holds other paths than the host copy\n", g); return 1; }
        (void)hipFree(full[g]);
    }
    int solved = 0;
    for (int q = 0; q < batch; ++q) solved += status[q] == PQP_STATUS_SOLVED;
    (void)pqp_multi_destroy(m);
    std::printf("gather ok %d %d solved %d\n", shards, batch, solved);
    return 0;
}
