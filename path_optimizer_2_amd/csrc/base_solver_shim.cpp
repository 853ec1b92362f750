// base_solver_shim.cpp — host C++ side of the drop-in: PathOptimizationNS::BaseSolver over the C ABI (include/pqp.h).
// Uses only public getters of ReferencePath / VehicleState, so it compiles unchanged against the reference's own
// headers (define PQP_USE_REFERENCE_TYPES and include them first) or against include/pqp_types.hpp.
#include "../../include/pqp_base_solver.hpp"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <vector>

namespace PathOptimizationNS {

namespace {
// The reference constructs one BaseSolver per planning cycle (src/path_optimizer.cpp:138) and lets it die at the end of optimizePath.  A
// pqp_handle owns a HIP stream, 2 x 256 events and its device workspaces: creating one costs a few hundred microseconds to milliseconds
// (bench.py: secondary.base_solver_shim_batch1), several times the batch-1 solve itself.  So handles outlive the instances: a destructor
// parks its handle here, the next instance on the same device takes it (workspaces grow on demand; parameters are set at every take).
// Parked handles live until the process ends (BaseSolver::releaseCachedHandles() for a caller that wants them gone earlier: they are
// never destroyed from a static destructor, where the HIP runtime may already be down).
struct HandlePool {
    std::mutex mu;
    std::vector<std::pair<int, pqp_handle*>> idle;      // (device, handle)
    pqp_handle* take(int device) {
        std::lock_guard<std::mutex> lk(mu);
        for (size_t k = 0; k < idle.size(); ++k)
            if (idle[k].first == device) { pqp_handle* h = idle[k].second; idle.erase(idle.begin() + (long)k); return h; }
        return nullptr;
    }
    void park(int device, pqp_handle* h) {
        std::lock_guard<std::mutex> lk(mu);
        idle.emplace_back(device, h);
    }
    void clear() {
        std::lock_guard<std::mutex> lk(mu);
        for (auto& e : idle) (void)pqp_destroy(e.second);
        idle.clear();
    }
};
HandlePool& pool() { static HandlePool* p = new HandlePool(); return *p; }       // (leaked on purpose: see above)

// FLAGS-like process default of the device ordinal: PQP_DEVICE (the reference has no notion of a device; one per process is the
// common case: one planner process per GPU)
int default_device() {
    const char* e = std::getenv("PQP_DEVICE");
    if (!e || !*e) return 0;
    char* end = nullptr;
    const long v = std::strtol(e, &end, 10);
    return (end && *end == 0 && v >= 0 && v < 1024) ? (int)v : 0;
}
}  // namespace

void BaseSolver::releaseCachedHandles() { pool().clear(); }

BaseSolver::BaseSolver(const ReferencePath& reference_path, const VehicleState& vehicle_state, const std::vector<SlState>& input_path)
    : n_(input_path.size()), reference_path_(reference_path), vehicle_state_(vehicle_state), input_path_(input_path), device_(default_device()) {
    pqp_default_params(&params_);
    updateSizes();
}

// base_solver.cpp:22-37 through the library's own size function.  The reference reads FLAGS_rough_constraints_far_away /
// FLAGS_precise_planning_length at construction; here they are fields of params(), so the sizes follow setParams() too
void BaseSolver::updateSizes() {
    std::vector<double> s(n_);
    for (size_t i = 0; i < n_; ++i) s[i] = input_path_[i].s;
    pqp_sizes sz;
    if (n_ >= 2 && pqp_path_sizes(&params_, (int)n_, s.data(), &sz) == PQP_OK) {
        state_size_ = sz.state; control_size_ = sz.control; slack_size_ = sz.slack;
        vars_size_ = sz.vars; cons_size_ = sz.cons; precise_planning_size_ = sz.precise;
    }
}

BaseSolver::~BaseSolver() {
    if (handle_) {
        if (cache_handles_) pool().park(device_, handle_);
        else pqp_destroy(handle_);
    }
}

void BaseSolver::setDevice(int device) {
    if (device == device_ || device < 0) return;
    if (handle_) { if (cache_handles_) pool().park(device_, handle_); else pqp_destroy(handle_); handle_ = nullptr; }
    solved_once_ = false;          // (the next handle - possibly a pooled one with another instance's warm state - has not seen this instance's cold solve)
    device_ = device;
}

void BaseSolver::setParams(const pqp_params& p) {
    params_ = p;
    updateSizes();          // vars() / cons() / the precise planning size depend on rough_constraints_far_away and precise_planning_length
    if (handle_) pqp_set_params(handle_, &params_);
}

bool BaseSolver::run(const std::vector<SlState>& lin, bool warm, std::vector<SlState>* out) {
    if (!out || n_ < 2) return false;
    // (a pooled handle may still hold another instance's warm state: a warm solve needs THIS instance's own cold solve first - the reference's
    //  updateBounds() on a solver that was never initialised fails the same way, base_solver.cpp:106)
    if (warm && !solved_once_) return false;
    const auto& ref_states = reference_path_.getReferenceStates();
    const auto& bounds = reference_path_.getBounds();
    if (ref_states.size() < n_ || bounds.size() < n_ || lin.size() != n_) return false;
    if (!handle_) {
        if (cache_handles_ && (handle_ = pool().take(device_)) != nullptr) {
            pqp_set_params(handle_, &params_);
        } else if (pqp_create(&handle_, &params_, device_, 1, (int)n_) != PQP_OK) {
            handle_ = nullptr;
            std::fprintf(stderr, "BaseSolver: %s\n", pqp_last_error());
            return false;      // no CPU fallback: without the GPU engine solve() fails, as a failed initSolver() does
        }
    }
    std::vector<double> ref(5 * n_), lin3(3 * n_), bnd(6 * n_), result(7 * n_);
    for (size_t i = 0; i < n_; ++i) {
        ref[5 * i] = ref_states[i].s; ref[5 * i + 1] = ref_states[i].k; ref[5 * i + 2] = ref_states[i].heading;
        ref[5 * i + 3] = ref_states[i].x; ref[5 * i + 4] = ref_states[i].y;
        lin3[3 * i] = lin[i].l; lin3[3 * i + 1] = lin[i].d_heading; lin3[3 * i + 2] = lin[i].k;
        bnd[6 * i] = bounds[i].front.lb; bnd[6 * i + 1] = bounds[i].front.ub;
        bnd[6 * i + 2] = bounds[i].rear.lb; bnd[6 * i + 3] = bounds[i].rear.ub;
        bnd[6 * i + 4] = bounds[i].center.lb; bnd[6 * i + 5] = bounds[i].center.ub;
    }
    const auto init_error = vehicle_state_.getInitError();                 // base_solver.cpp:217-218
    double scal[PQP_SCAL_STRIDE] = {init_error[0], init_error[1], vehicle_state_.getStartState().k,
                                    vehicle_state_.getTargetState().heading, reference_path_.isBlocked() == nullptr ? 0.0 : 1.0,
                                    max_steering_angle_};
    int32_t status = 0, iters = 0;
    const int rc = pqp_path_solve(handle_, 1, (int)n_, ref.data(), lin3.data(), bnd.data(), scal, 0, warm ? 1 : 0, result.data(),
                                  &status, &iters, nullptr);
    status_ = status; iters_ = iters;
    if (rc != PQP_OK) {
        std::fprintf(stderr, "BaseSolver: %s\n", pqp_last_error());
        return false;
    }
    // the cold solve was launched on this handle: its warm state is this instance's own, whatever the status - the reference's solver stays
    // initialised after a solve() that ends at max_iter or infeasible, and updateProblemFormulationAndSolve still runs (base_solver.cpp:97-110)
    // (not a QP the host entry refused - an inverted box, PRIMAL_INFEASIBLE without an iteration: nothing ran on its data)
    solved_once_ = !(status == PQP_STATUS_PRIMAL_INFEASIBLE && iters == 0);
    if (status != PQP_STATUS_SOLVED) return false;                           // osqp-eigen: solve() true only for "solved"
    out->clear();                                                            // base_solver.cpp:266
    out->reserve(n_);
    for (size_t i = 0; i < n_; ++i) {                                        // base_solver.cpp:269-287: s, v, a stay 0
        SlState pt;
        pt.x = result[7 * i]; pt.y = result[7 * i + 1]; pt.heading = result[7 * i + 2];
        pt.l = result[7 * i + 3]; pt.d_heading = result[7 * i + 4]; pt.k = result[7 * i + 5]; pt.d_k = result[7 * i + 6];
        out->push_back(pt);
    }
    return true;
}

bool BaseSolver::solve(std::vector<SlState>* optimized_path) {
    return run(input_path_, false, optimized_path);
}

bool BaseSolver::updateProblemFormulationAndSolve(const std::vector<SlState>& input_path, std::vector<SlState>* optimized_path) {
    input_path_ = input_path;               // copy first: in and out may alias (base_solver.cpp:100, path_optimizer.cpp:153)
    if (input_path_.size() != n_) return false;
    return run(input_path_, true, optimized_path);
}

}  // namespace PathOptimizationNS
