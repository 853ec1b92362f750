// pqp_multi.cpp — the multi-GPU driver of the C ABI (include/pqp.h: pqp_multi_*): the batch of independent QPs is cut
// into contiguous shards, every shard has its own handle (= its own GPU, stream and workspaces) and its own host thread; a call moves
// each shard's slice of the caller's host arrays to its GPU, solves it there and brings the paths back - per-GPU copies, no
// collective: with the consumer on the host that is strictly better than a device-side gather; the device-resident
// RCCL all-gather of the result slabs is path_optimizer_2_amd/shard.py (torch.distributed, one process per GPU).
// Plain host C++ over the entry points of pqp_kernels.hip; part of libpqp_hip.so.
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../include/pqp.h"

extern "C" void pqp_set_last_error(const char* msg);      // pqp_kernels.hip (thread-local message of pqp_last_error)

struct pqp_multi {
    std::vector<pqp_handle*> shards;
    std::vector<int> devices;
};

namespace {
int mfail(int code, const std::string& msg) {
    pqp_set_last_error(msg.c_str());
    return code;
}
}  // namespace

extern "C" {

// Contiguous split (the same rule as path_optimizer_2_amd/shard.py::shard_range): rank g gets QPs [first, first + count),
// the first (total % world) ranks one more.
void pqp_shard_range(int total, int world, int rank, int* first, int* count) {
    const int base = total / world, extra = total % world;
    if (count) *count = base + (rank < extra ? 1 : 0);
    if (first) *first = rank * base + (rank < extra ? rank : extra);
}

int pqp_multi_create(pqp_multi** out, const pqp_params* params, int n_shards, const int* devices, int max_batch_per_shard, int max_n) {
    if (!out || n_shards < 1) return mfail(PQP_ERR_INVALID, "pqp_multi_create: need an output pointer and n_shards >= 1");
    *out = nullptr;
    pqp_multi* m = new (std::nothrow) pqp_multi();
    if (!m) return mfail(PQP_ERR_INVALID, "pqp_multi_create: out of host memory");
    for (int g = 0; g < n_shards; ++g) {
        const int dev = devices ? devices[g] : g;
        pqp_handle* h = nullptr;
        const int rc = pqp_create(&h, params, dev, max_batch_per_shard, max_n);
        if (rc != PQP_OK) {
            const std::string why = pqp_last_error();
            (void)pqp_multi_destroy(m);
            return mfail(rc, "pqp_multi_create: shard " + std::to_string(g) + " on device " + std::to_string(dev) + ": " + why);
        }
        m->shards.push_back(h);
        m->devices.push_back(dev);
    }
    *out = m;
    return PQP_OK;
}

int pqp_multi_destroy(pqp_multi* m) {
    if (!m) return PQP_OK;
    for (pqp_handle* h : m->shards) (void)pqp_destroy(h);
    delete m;
    return PQP_OK;
}

int pqp_multi_shards(const pqp_multi* m) { return m ? (int)m->shards.size() : 0; }

pqp_handle* pqp_multi_handle(pqp_multi* m, int shard) {
    return (m && shard >= 0 && shard < (int)m->shards.size()) ? m->shards[shard] : nullptr;
}

int pqp_multi_set_option(pqp_multi* m, int option, int value) {
    if (!m) return mfail(PQP_ERR_INVALID, "pqp_multi_set_option: null handle");
    for (pqp_handle* h : m->shards) {
        const int rc = pqp_set_option(h, option, value);
        if (rc != PQP_OK) return rc;
    }
    return PQP_OK;
}

int pqp_multi_path_solve(pqp_multi* m, int batch, int n, const int32_t* n_of, const double* ref, const double* lin, const double* bounds,
                         const double* scal, int passes, double* out, int32_t* status, int32_t* iters, double* info) {
    if (!m || !ref || !bounds || !scal || !out || batch < 1 || n < 2 || passes < 0)
        return mfail(PQP_ERR_INVALID, "pqp_multi_path_solve: bad argument");
    const int world = (int)m->shards.size();
    std::vector<int> rcs(world, PQP_OK);
    std::vector<std::string> errs(world);
    auto work = [&](int g) {
        int first = 0, count = 0;
        pqp_shard_range(batch, world, g, &first, &count);
        if (count == 0) return;
        const size_t o = (size_t)first;
        const size_t on = o * (size_t)n;
        const double* lin_g = lin ? lin + on * PQP_LIN_STRIDE : nullptr;
        int rc;
        if (n_of)
            rc = pqp_path_solve_var(m->shards[g], count, n, n_of + o, ref + on * PQP_REF_STRIDE, lin_g, bounds + on * PQP_BOUNDS_STRIDE,
                                    scal + o * PQP_SCAL_STRIDE, passes, 0, out + on * PQP_OUT_STRIDE, status ? status + o : nullptr,
                                    iters ? iters + o : nullptr, info ? info + o * PQP_INFO_STRIDE : nullptr);
        else
            rc = pqp_path_solve(m->shards[g], count, n, ref + on * PQP_REF_STRIDE, lin_g, bounds + on * PQP_BOUNDS_STRIDE,
                                scal + o * PQP_SCAL_STRIDE, passes, 0, out + on * PQP_OUT_STRIDE, status ? status + o : nullptr,
                                iters ? iters + o : nullptr, info ? info + o * PQP_INFO_STRIDE : nullptr);
        rcs[g] = rc;
        if (rc != PQP_OK) errs[g] = pqp_last_error();      // (thread-local in the worker: carried back by hand)
    };
    std::vector<std::thread> threads;          // one host thread per shard (a handle is single-owner; different handles may run concurrently)
    for (int g = 1; g < world; ++g) threads.emplace_back(work, g);
    work(0);
    for (auto& t : threads) t.join();
    for (int g = 0; g < world; ++g)
        if (rcs[g] != PQP_OK) return mfail(rcs[g], "pqp_multi_path_solve: shard " + std::to_string(g) + ": " + errs[g]);
    return PQP_OK;
}

}  // extern "C"
