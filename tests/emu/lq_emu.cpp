// Host build of the lane-per-QP solver (path_optimizer_2_amd/csrc/pqp_path_lq.hpp): every QP of the batch runs the device algorithm
// source on the CPU, one after the other.  TEST INFRASTRUCTURE: it lets tests/test_lq_emulation.py check the algorithm against the
// oracle in a container without a GPU; nothing in the product links it.
#include <cstring>
#include <vector>

#include "../../path_optimizer_2_amd/csrc/pqp_defaults.hpp"
#include "../../path_optimizer_2_amd/csrc/pqp_path_lq.hpp"

extern "C" {
void pqp_emu_lq_production_params(pqp_params* p) { pqp::production_params(p); }

int pqp_emu_lq_fields(void) { return pqp::lq::kBlockDoubles; }

void pqp_emu_lq_solve(const pqp_params* prm, int batch, int n, const int32_t* n_of, const double* ref, const double* lin, const double* bounds,
                      const double* scal, int passes, double* out, int32_t* status, int32_t* iters, double* info) {
    pqp::lq::Args a;
    std::memset(&a, 0, sizeof(a));
    a.batch = batch; a.n = n; a.passes = passes; a.n_of = n_of; a.ref = ref; a.lin = lin; a.bounds = bounds; a.scal = scal; a.out = out;
    a.status = status; a.iters = iters; a.info = info; a.prm = *prm;
    std::vector<double> ws((size_t)n * pqp::lq::kBlockDoubles);
    a.ws = ws.data();
    for (int q = 0; q < batch; ++q) {
        std::fill(ws.begin(), ws.end(), 0.0);
        pqp::lq::Solver<pqp::lq::StridedWs> s(a, q, pqp::lq::StridedWs{ws.data(), 0, 1});
        s.run();
    }
}
}
