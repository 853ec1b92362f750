// Host build of the lane-per-QP solver (path_optimizer_2_amd/csrc/pqp_path_lq.hpp): every QP of the batch runs the device algorithm
// source on the CPU, one after the other.  TEST INFRASTRUCTURE: it lets tests/test_lq_emulation.py check the algorithm against the
// oracle in a container without a GPU; nothing in the product links it.
#include <cstdint>
#include <cstring>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "../../path_optimizer_2_amd/csrc/pqp_defaults.hpp"
#include "../../path_optimizer_2_amd/csrc/pqp_path_lq.hpp"

extern "C" {
void pqp_emu_lq_production_params(pqp_params* p) { pqp::production_params(p); }

int pqp_emu_lq_fields(void) { return pqp::lq::kBlockDoubles; }

// carry != 0: `persist` [batch][n * kBlockDoubles] keeps every QP's workspace between calls (Args::carry: the next call's first pass starts from it)
void pqp_emu_lq_solve_carry(const pqp_params* prm, int batch, int n, const int32_t* n_of, const double* ref, const double* lin, const double* bounds,
                            const double* scal, int passes, double* out, int32_t* status, int32_t* iters, double* info, double* persist, int carry) {
    pqp::lq::Args a;
    std::memset(&a, 0, sizeof(a));
    a.batch = batch; a.n = n; a.passes = passes; a.n_of = n_of; a.ref = ref; a.lin = lin; a.bounds = bounds; a.scal = scal; a.out = out;
    a.status = status; a.iters = iters; a.info = info; a.prm = *prm; a.carry = carry; a.ws = persist;
#pragma omp parallel for schedule(dynamic, 16)
    for (int q = 0; q < batch; ++q) {
        pqp::lq::Solver<pqp::lq::StridedWs> s(a, q, pqp::lq::StridedWs{persist + (size_t)q * n * pqp::lq::kBlockDoubles, 0, 1});
        s.run();
    }
}

void pqp_emu_lq_solve(const pqp_params* prm, int batch, int n, const int32_t* n_of, const double* ref, const double* lin, const double* bounds,
                      const double* scal, int passes, double* out, int32_t* status, int32_t* iters, double* info) {
    pqp::lq::Args a;
    std::memset(&a, 0, sizeof(a));
    a.batch = batch; a.n = n; a.passes = passes; a.n_of = n_of; a.ref = ref; a.lin = lin; a.bounds = bounds; a.scal = scal; a.out = out;
    a.status = status; a.iters = iters; a.info = info; a.prm = *prm;
    // (one QP per OpenMP task when built with -fopenmp: bench.py's "same algorithm on the host cores" line; serial otherwise)
#pragma omp parallel
    {
        std::vector<double> ws((size_t)n * pqp::lq::kBlockDoubles);
#pragma omp for schedule(dynamic, 16)
        for (int q = 0; q < batch; ++q) {
            std::fill(ws.begin(), ws.end(), 0.0);
            pqp::lq::Solver<pqp::lq::StridedWs> s(a, q, pqp::lq::StridedWs{ws.data(), 0, 1});
            s.run();
        }
    }
}

// the same with Args::order set, as a launch has it whose wavefronts are sorted by their phase counts (PQP_OPT_ORDER_BY_COST from the second launch on): the solver
// itself only asks WHETHER the launch is sorted (the re-linearised pass then starts with active-set rounds on the previous pass's set)
void pqp_emu_lq_solve_sorted(const pqp_params* prm, int batch, int n, const int32_t* n_of, const double* ref, const double* lin, const double* bounds,
                             const double* scal, int passes, double* out, int32_t* status, int32_t* iters, double* info) {
    pqp::lq::Args a;
    std::memset(&a, 0, sizeof(a));
    static const int32_t sorted_marker = 0;
    a.batch = batch; a.n = n; a.passes = passes; a.n_of = n_of; a.ref = ref; a.lin = lin; a.bounds = bounds; a.scal = scal; a.out = out;
    a.status = status; a.iters = iters; a.info = info; a.prm = *prm; a.order = &sorted_marker;
#pragma omp parallel
    {
        std::vector<double> ws((size_t)n * pqp::lq::kBlockDoubles);
#pragma omp for schedule(dynamic, 16)
        for (int q = 0; q < batch; ++q) {
            std::fill(ws.begin(), ws.end(), 0.0);
            pqp::lq::Solver<pqp::lq::StridedWs> s(a, q, pqp::lq::StridedWs{ws.data(), 0, 1});
            s.run();
        }
    }
}

// the same solver over the OTHER workspace layout ([chunk][lane][16 bytes], lq::ChunkWs: what the device's staged form addresses): the same arithmetic on the
// same values, so the same bits as pqp_emu_lq_solve
void pqp_emu_lq_solve_chunk_layout(const pqp_params* prm, int batch, int n, const int32_t* n_of, const double* ref, const double* lin, const double* bounds,
                                   const double* scal, int passes, double* out, int32_t* status, int32_t* iters, double* info, int sorted) {
    pqp::lq::Args a;
    std::memset(&a, 0, sizeof(a));
    static const int32_t sorted_marker = 0;
    a.batch = batch; a.n = n; a.passes = passes; a.n_of = n_of; a.ref = ref; a.lin = lin; a.bounds = bounds; a.scal = scal; a.out = out;
    a.status = status; a.iters = iters; a.info = info; a.prm = *prm; a.order = sorted ? &sorted_marker : nullptr; a.staged = 1;
#pragma omp parallel
    {
        std::vector<double> ws((size_t)n * pqp::lq::kBlockDoubles + 2);
        double* base = ws.data() + (((uintptr_t)ws.data() & 15) ? 1 : 0);          // 16-byte aligned: the chunk accesses
#pragma omp for schedule(dynamic, 16)
        for (int q = 0; q < batch; ++q) {
            std::fill(ws.begin(), ws.end(), 0.0);
            pqp::lq::ChunkWs w; w.block = base; w.lane = 0; w.lanes = 1;
            pqp::lq::Solver<pqp::lq::ChunkWs> s(a, q, w);
            s.run();
        }
    }
}

int pqp_emu_lq_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
}
