// pqp_multi.cpp — the multi-GPU host loop of the C ABI (include/pqp.h: pqp_multi_*): the batch of independent QPs is cut
// into contiguous shards; every shard has its own handle (= its own GPU, stream and workspaces), its own PERSISTENT host thread
// (created in pqp_multi_create, parked on a condition variable between calls), pinned staging buffers and device buffers.  A call
// hands every worker its slice of the caller's host arrays; the worker stages it into pinned memory, enqueues H2D copies, the
// device-resident solve and the D2H copies on its handle's stream - true asynchronous DMA, so the copies of one shard run beside the
// solves of the others also when two shards share a GPU - waits for its stream and writes the caller's output slice.  No collective in
// that call: with the consumer on the host per-GPU copies beat a device-side gather.  For a consumer ON the GPUs there is
// pqp_multi_gather_paths below: the result slabs the shards keep in device memory, gathered over RCCL (xGMI) so that every GPU holds every
// path - the one exchange step north_star names, and the only place this library touches RCCL (dlopen'ed on first use: a caller that never
// gathers needs no librccl).  The one-process-per-GPU counterpart is path_optimizer_2_amd/shard.py (torch.distributed).
// Plain host C++ over the entry points of pqp_kernels.hip + the HIP runtime's memory API; part of libpqp_hip.so.
// The reference has no counterpart (one path per call, single-threaded: base_solver.cpp:56-95).
#include <dlfcn.h>
#include <hip/hip_runtime_api.h>

#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "../../include/pqp.h"

extern "C" void pqp_set_last_error(const char* msg);      // pqp_kernels.hip (thread-local message of pqp_last_error)

namespace {
int mfail(int code, const std::string& msg) {
    pqp_set_last_error(msg.c_str());
    return code;
}

struct Job {
    int count = 0, n = 0, passes = 0;
    const int32_t* n_of = nullptr;
    const double *ref = nullptr, *lin = nullptr, *bounds = nullptr, *scal = nullptr;
    double* out = nullptr;
    int32_t *status = nullptr, *iters = nullptr;
    double* info = nullptr;
};

// a buffer pair: pinned host memory + device memory of the same size, grown on demand
struct Staged {
    void *pin = nullptr, *dev = nullptr;
    size_t bytes = 0;
    hipError_t ensure(size_t need) {
        if (need <= bytes) return hipSuccess;
        release();
        hipError_t e = hipHostMalloc(&pin, need, hipHostMallocDefault);
        if (e == hipSuccess) e = hipMalloc(&dev, need);
        if (e != hipSuccess) { release(); return e; }
        bytes = need;
        return hipSuccess;
    }
    void release() {
        if (pin) (void)hipHostFree(pin);
        if (dev) (void)hipFree(dev);
        pin = dev = nullptr; bytes = 0;
    }
};

struct Shard {
    pqp_handle* h = nullptr;
    int device = 0;
    std::thread worker;
    std::mutex mu;
    std::condition_variable cv;
    bool has_job = false, done = false, quit = false;
    Job job;
    int rc = PQP_OK;
    std::string err;
    Staged ref, lin, bounds, scal, counts, out, status, iters, info;
    int last_count = -1, last_n = 0;        // shape of the slab out.dev holds (pqp_multi_gather_paths)

    int hip(hipError_t e, const char* what) {
        if (e == hipSuccess) return PQP_OK;
        err = std::string(what) + ": " + hipGetErrorString(e);
        return PQP_ERR_HIP;
    }

    // one shard's slice: host -> pinned -> device, solve, device -> pinned -> host
    // (every error return goes through here: the stream is drained before the pinned staging buffers can be reused - a copy still in
    //  flight would race with the next call's memcpy into them - and the slab shape is invalidated, so that a later
    //  pqp_multi_gather_paths cannot broadcast a slab this failed call left half written)
    int run(const Job& j) {
        last_count = -1;
        const int r = run_body(j);
        if (r != PQP_OK) {
            void* sv = nullptr;
            if (pqp_get_stream(h, &sv) == PQP_OK && sv) (void)hipStreamSynchronize((hipStream_t)sv);
            last_count = -1;
        }
        return r;
    }
    int run_body(const Job& j) {
        int rc_;
        static const bool trace = std::getenv("PQP_MULTI_TRACE") != nullptr;         // phase times of every call on stderr (debugging aid)
        const auto t0 = std::chrono::steady_clock::now();
        auto us = [&] { return (long)std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count(); };
        long t_stage = 0, t_enq = 0, t_sync = 0;
        if ((rc_ = hip(hipSetDevice(device), "hipSetDevice"))) return rc_;
        void* sv = nullptr;
        if (pqp_get_stream(h, &sv) != PQP_OK) { err = pqp_last_error(); return PQP_ERR_INVALID; }
        hipStream_t stream = (hipStream_t)sv;
        const size_t bn = (size_t)j.count * j.n, b = (size_t)j.count;
        struct In { Staged* s; const void* src; size_t bytes; };
        const In ins[] = {{&ref, j.ref, bn * PQP_REF_STRIDE * 8}, {&lin, j.lin, j.lin ? bn * PQP_LIN_STRIDE * 8 : 0}, {&bounds, j.bounds, bn * PQP_BOUNDS_STRIDE * 8},
                          {&scal, j.scal, b * PQP_SCAL_STRIDE * 8}, {&counts, j.n_of, j.n_of ? b * 4 : 0}};
        for (const In& in : ins) {
            if (!in.bytes) continue;
            if ((rc_ = hip(in.s->ensure(in.bytes), "staging buffer"))) return rc_;
            std::memcpy(in.s->pin, in.src, in.bytes);
            if ((rc_ = hip(hipMemcpyAsync(in.s->dev, in.s->pin, in.bytes, hipMemcpyHostToDevice, stream), "hipMemcpyAsync H2D"))) return rc_;
        }
        t_stage = us();
        if ((rc_ = hip(out.ensure(bn * PQP_OUT_STRIDE * 8), "staging buffer")) || (rc_ = hip(status.ensure(b * 4), "staging buffer")) ||
            (rc_ = hip(iters.ensure(b * 4), "staging buffer")) || (rc_ = hip(info.ensure(b * PQP_INFO_STRIDE * 8), "staging buffer")))
            return rc_;
        if (j.n_of && (rc_ = hip(hipMemsetAsync(out.dev, 0, bn * PQP_OUT_STRIDE * 8, stream), "hipMemsetAsync"))) return rc_;      // rows beyond a QP's own count
        const double* d_lin = j.lin ? (const double*)lin.dev : nullptr;
        int rc_solve;
        if (j.n_of)
            rc_solve = pqp_path_solve_var_device(h, j.count, j.n, (const int32_t*)counts.dev, (const double*)ref.dev, d_lin, (const double*)bounds.dev,
                                                 (const double*)scal.dev, j.passes, 0, (double*)out.dev, (int32_t*)status.dev, (int32_t*)iters.dev, (double*)info.dev);
        else
            rc_solve = pqp_path_solve_device(h, j.count, j.n, (const double*)ref.dev, d_lin, (const double*)bounds.dev, (const double*)scal.dev, j.passes, 0,
                                             (double*)out.dev, (int32_t*)status.dev, (int32_t*)iters.dev, (double*)info.dev);
        if (rc_solve != PQP_OK) { err = pqp_last_error(); return rc_solve; }
        if ((rc_ = hip(hipMemcpyAsync(out.pin, out.dev, bn * PQP_OUT_STRIDE * 8, hipMemcpyDeviceToHost, stream), "hipMemcpyAsync D2H"))) return rc_;
        if (j.status && (rc_ = hip(hipMemcpyAsync(status.pin, status.dev, b * 4, hipMemcpyDeviceToHost, stream), "hipMemcpyAsync D2H"))) return rc_;
        if (j.iters && (rc_ = hip(hipMemcpyAsync(iters.pin, iters.dev, b * 4, hipMemcpyDeviceToHost, stream), "hipMemcpyAsync D2H"))) return rc_;
        if (j.info && (rc_ = hip(hipMemcpyAsync(info.pin, info.dev, b * PQP_INFO_STRIDE * 8, hipMemcpyDeviceToHost, stream), "hipMemcpyAsync D2H"))) return rc_;
        t_enq = us();
        if ((rc_ = hip(hipStreamSynchronize(stream), "hipStreamSynchronize"))) return rc_;
        t_sync = us();
        std::memcpy(j.out, out.pin, bn * PQP_OUT_STRIDE * 8);
        if (j.status) std::memcpy(j.status, status.pin, b * 4);
        if (j.iters) std::memcpy(j.iters, iters.pin, b * 4);
        if (j.info) std::memcpy(j.info, info.pin, b * PQP_INFO_STRIDE * 8);
        last_count = j.count; last_n = j.n;
        if (trace) std::fprintf(stderr, "[pqp_multi] device %d, %d QPs: staged + H2D enqueued %ld us, solve + D2H enqueued %ld us, stream done %ld us, copied out %ld us\n",
                                device, j.count, t_stage, t_enq, t_sync, us());
        return PQP_OK;
    }

    void loop() {
        for (;;) {
            Job j;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return has_job || quit; });
                if (quit) return;
                j = job;
                has_job = false;
            }
            err.clear();
            if (j.count <= 0) { last_count = j.count == 0 ? 0 : -1; last_n = j.n; }
            const int r = j.count > 0 ? run(j) : PQP_OK;
            {
                std::lock_guard<std::mutex> lk(mu);
                rc = r;
                done = true;
            }
            cv.notify_all();
        }
    }
    void submit(const Job& j) {
        {
            std::lock_guard<std::mutex> lk(mu);
            job = j; has_job = true; done = false;
        }
        cv.notify_all();
    }
    void wait() {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return done; });
    }
    void stop() {
        if (worker.joinable()) {
            { std::lock_guard<std::mutex> lk(mu); quit = true; }
            cv.notify_all();
            worker.join();
        }
        (void)hipSetDevice(device);
        for (Staged* s : {&ref, &lin, &bounds, &scal, &counts, &out, &status, &iters, &info}) s->release();
    }
};

// RCCL, resolved at run time.  One communicator per shard (single process, one rank per device), created by the first gather.
// The handful of types and prototypes the gather uses are declared here (NCCL's stable C ABI: rccl.h's ncclComm_t, ncclResult_t with
// ncclSuccess = 0, ncclDataType_t with ncclDouble = 8), so that the library builds on a ROCm install without the RCCL development headers.
struct ncclComm;
typedef ncclComm* ncclComm_t;
typedef int ncclResult_t;
constexpr ncclResult_t ncclSuccess = 0;
constexpr int ncclDouble = 8;
struct Rccl {
    void* lib = nullptr;
    ncclResult_t (*comm_init_all)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*comm_destroy)(ncclComm_t) = nullptr;
    ncclResult_t (*comm_count)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*group_start)() = nullptr;
    ncclResult_t (*group_end)() = nullptr;
    ncclResult_t (*broadcast)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    const char* (*error_string)(ncclResult_t) = nullptr;
    std::string why;
    std::mutex mu;
    // (callers of two drivers may gather at once: the load is serialised; a library that lacks a symbol is closed again)
    bool load() {
        std::lock_guard<std::mutex> lk(mu);
        if (lib) return true;
        for (const char* name : {"librccl.so.1", "librccl.so"}) {
            lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (lib) break;
        }
        if (!lib) { const char* e = dlerror(); why = std::string("librccl.so not found (") + (e ? e : "?") + ")"; return false; }
        auto sym = [&](const char* n) { void* f = dlsym(lib, n); if (!f) why = std::string("librccl.so lacks ") + n; return f; };
        comm_count = (decltype(comm_count))sym("ncclCommCount");
        comm_init_all = (decltype(comm_init_all))sym("ncclCommInitAll");
        comm_destroy = (decltype(comm_destroy))sym("ncclCommDestroy");
        group_start = (decltype(group_start))sym("ncclGroupStart");
        group_end = (decltype(group_end))sym("ncclGroupEnd");
        broadcast = (decltype(broadcast))sym("ncclBroadcast");
        error_string = (decltype(error_string))sym("ncclGetErrorString");
        if (!comm_init_all || !comm_destroy || !comm_count || !group_start || !group_end || !broadcast || !error_string) { (void)dlclose(lib); lib = nullptr; return false; }
        return true;
    }
};
Rccl& rccl() { static Rccl r; return r; }
}  // namespace

struct pqp_multi {
    std::vector<Shard*> shards;
    std::vector<ncclComm_t> comms;          // empty until the first pqp_multi_gather_paths
    std::mutex gather_mu;
};

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
        Shard* s = new (std::nothrow) Shard();
        if (!s) { (void)pqp_destroy(h); (void)pqp_multi_destroy(m); return mfail(PQP_ERR_INVALID, "pqp_multi_create: out of host memory"); }
        s->h = h; s->device = dev;
        // a caller of the multi driver takes its paths home after every call: nobody reads the handles' warm state
        (void)pqp_set_option(h, PQP_OPT_STORE_WARM, 0);
        m->shards.push_back(s);
        s->worker = std::thread([s] { s->loop(); });            // the shard's host thread, for the life of the driver
    }
    *out = m;
    return PQP_OK;
}

int pqp_multi_destroy(pqp_multi* m) {
    if (!m) return PQP_OK;
    for (ncclComm_t c : m->comms) (void)rccl().comm_destroy(c);
    for (Shard* s : m->shards) {
        s->stop();
        (void)pqp_destroy(s->h);
        delete s;
    }
    delete m;
    return PQP_OK;
}

int pqp_multi_shards(const pqp_multi* m) { return m ? (int)m->shards.size() : 0; }

pqp_handle* pqp_multi_handle(pqp_multi* m, int shard) {
    return (m && shard >= 0 && shard < (int)m->shards.size()) ? m->shards[shard]->h : nullptr;
}

int pqp_multi_set_option(pqp_multi* m, int option, int value) {
    if (!m) return mfail(PQP_ERR_INVALID, "pqp_multi_set_option: null handle");
    for (Shard* s : m->shards) {
        const int rc = pqp_set_option(s->h, option, value);      // (PQP_OPT_CARRY_CYCLES: every shard's handle carries its own slice)
        if (rc != PQP_OK) return rc;
    }
    return PQP_OK;
}

int pqp_multi_path_solve(pqp_multi* m, int batch, int n, const int32_t* n_of, const double* ref, const double* lin, const double* bounds,
                         const double* scal, int passes, double* out, int32_t* status, int32_t* iters, double* info) {
    if (!m || !ref || !bounds || !scal || !out || batch < 1 || n < 2 || passes < 0)
        return mfail(PQP_ERR_INVALID, "pqp_multi_path_solve: bad argument");
    // (an inverted collision box is refused as OSQP refuses it at setup - the host-pointer entry points' rule, pqp.h)
    std::vector<int32_t> counts;
    for (int q = 0; q < batch; ++q) {
        const int cnt = n_of ? n_of[q] : n;
        bool bad = false;
        for (int i = 0; i < cnt && i < n && !bad; ++i) {
            const double* b = bounds + ((size_t)q * n + i) * PQP_BOUNDS_STRIDE;
            bad = b[0] > b[1] || b[2] > b[3] || b[4] > b[5];
        }
        if (bad) {
            if (counts.empty()) { if (n_of) counts.assign(n_of, n_of + batch); else counts.assign(batch, n); }
            counts[q] = -1;
        }
    }
    const int32_t* cnt_ptr = counts.empty() ? n_of : counts.data();
    const int world = (int)m->shards.size();
    for (int g = 0; g < world; ++g) {
        int first = 0, count = 0;
        pqp_shard_range(batch, world, g, &first, &count);
        const size_t o = (size_t)first, on = o * (size_t)n;
        Job j;
        j.count = count; j.n = n; j.passes = passes;
        j.n_of = cnt_ptr ? cnt_ptr + o : nullptr;
        j.ref = ref + on * PQP_REF_STRIDE; j.lin = lin ? lin + on * PQP_LIN_STRIDE : nullptr; j.bounds = bounds + on * PQP_BOUNDS_STRIDE;
        j.scal = scal + o * PQP_SCAL_STRIDE; j.out = out + on * PQP_OUT_STRIDE;
        j.status = status ? status + o : nullptr; j.iters = iters ? iters + o : nullptr; j.info = info ? info + o * PQP_INFO_STRIDE : nullptr;
        m->shards[g]->submit(j);
    }
    int rc = PQP_OK;
    std::string why;
    for (int g = 0; g < world; ++g) {
        m->shards[g]->wait();
        if (m->shards[g]->rc != PQP_OK && rc == PQP_OK) { rc = m->shards[g]->rc; why = "pqp_multi_path_solve: shard " + std::to_string(g) + ": " + m->shards[g]->err; }
    }
    if (rc != PQP_OK) return mfail(rc, why);
    if (!counts.empty() && status)
        for (int q = 0; q < batch; ++q)
            if (counts[q] < 0) status[q] = PQP_STATUS_PRIMAL_INFEASIBLE;
    return PQP_OK;
}

// After pqp_multi_path_solve: every shard's result slab is still in its GPU's memory.  Gathers the slabs so that full_out[g] - device memory
// of shard g's GPU, [batch][n][PQP_OUT_STRIDE] doubles - holds the paths of the whole batch in the caller's order, over RCCL: shard r broadcasts
// its contiguous slab (pqp_shard_range) to everyone, all broadcasts of all shards in one group on the shards' own streams, which the call waits
// for.  (Broadcasts rather than one all-gather: the slabs of a batch that does not divide by the shard count differ by one QP.)
// Needs one device per shard (RCCL refuses two ranks on one device) and librccl.so at run time.
int pqp_multi_gather_paths(pqp_multi* m, int batch, int n, double* const* full_out) {
    if (!m || !full_out || batch < 1 || n < 2) return mfail(PQP_ERR_INVALID, "pqp_multi_gather_paths: bad argument");
    const int world = (int)m->shards.size();
    std::lock_guard<std::mutex> lk(m->gather_mu);
    std::vector<int> first(world), count(world);
    for (int g = 0; g < world; ++g) {
        pqp_shard_range(batch, world, g, &first[g], &count[g]);
        if (!full_out[g]) return mfail(PQP_ERR_INVALID, "pqp_multi_gather_paths: null output buffer");
        if (m->shards[g]->last_count != count[g] || m->shards[g]->last_n != n)
            return mfail(PQP_ERR_INVALID, "pqp_multi_gather_paths: batch and n are not those of the preceding pqp_multi_path_solve");
        for (int r = 0; r < g; ++r)
            if (m->shards[r]->device == m->shards[g]->device)
                return mfail(PQP_ERR_INVALID, "pqp_multi_gather_paths: two shards share device " + std::to_string(m->shards[g]->device) + " (RCCL takes one rank per device)");
    }
    Rccl& R = rccl();
    if (!R.load()) return mfail(PQP_ERR_HIP, "pqp_multi_gather_paths: " + R.why);
    auto nfail = [&](ncclResult_t e, const char* what) { return mfail(PQP_ERR_HIP, std::string("pqp_multi_gather_paths: ") + what + ": " + R.error_string(e)); };
    if (m->comms.empty()) {
        std::vector<int> devs(world);
        for (int g = 0; g < world; ++g) devs[g] = m->shards[g]->device;
        std::vector<ncclComm_t> comms(world);
        const ncclResult_t e = R.comm_init_all(comms.data(), world, devs.data());
        if (e != ncclSuccess) return nfail(e, "ncclCommInitAll");
        m->comms = comms;
    }
    std::vector<hipStream_t> streams(world);
    for (int g = 0; g < world; ++g) {
        void* sv = nullptr;
        if (pqp_get_stream(m->shards[g]->h, &sv) != PQP_OK) return PQP_ERR_INVALID;
        streams[g] = (hipStream_t)sv;
    }
    ncclResult_t e = R.group_start();
    if (e != ncclSuccess) return nfail(e, "ncclGroupStart");
    ncclResult_t bad = ncclSuccess;
    for (int g = 0; g < world && bad == ncclSuccess; ++g) {
        if (hipSetDevice(m->shards[g]->device) != hipSuccess) { (void)R.group_end(); return mfail(PQP_ERR_HIP, "pqp_multi_gather_paths: hipSetDevice"); }
        for (int r = 0; r < world && bad == ncclSuccess; ++r) {
            if (count[r] == 0) continue;
            double* dst = full_out[g] + (size_t)first[r] * n * PQP_OUT_STRIDE;
            const void* src = r == g ? m->shards[g]->out.dev : (const void*)dst;        // (read on the root only)
            bad = R.broadcast(src, dst, (size_t)count[r] * n * PQP_OUT_STRIDE, ncclDouble, r, m->comms[g], streams[g]);
        }
    }
    e = R.group_end();
    if (bad != ncclSuccess) return nfail(bad, "ncclBroadcast");
    if (e != ncclSuccess) return nfail(e, "ncclGroupEnd");
    for (int g = 0; g < world; ++g) {
        if (hipSetDevice(m->shards[g]->device) != hipSuccess || hipStreamSynchronize(streams[g]) != hipSuccess)
            return mfail(PQP_ERR_HIP, "pqp_multi_gather_paths: hipStreamSynchronize after the gather");
    }
    return PQP_OK;
}

// Ranks of the gather's communicator as RCCL itself counts them (ncclCommCount of shard 0's communicator); 0 before the first
// pqp_multi_gather_paths, < 0 on error.  What a scaling report should quote beside its own shard count.
int pqp_multi_gather_ranks(pqp_multi* m) {
    if (!m) return mfail(PQP_ERR_INVALID, "pqp_multi_gather_ranks: null driver");
    std::lock_guard<std::mutex> lk(m->gather_mu);
    if (m->comms.empty()) return 0;
    int ranks = 0;
    const ncclResult_t e = rccl().comm_count(m->comms[0], &ranks);
    if (e != ncclSuccess) return mfail(PQP_ERR_HIP, std::string("pqp_multi_gather_ranks: ncclCommCount: ") + rccl().error_string(e));
    return ranks;
}

}  // extern "C"
