// pqp_batched_solver.hpp — the batched C++ surface of the engine: what PathOptimizer::optimizePath does for ONE scenario
// (reference src/path_optimizer.cpp:124-161: input path (0, 0, k_ref) -> BaseSolver::solve -> updateProblemFormulationAndSolve),
// done for MANY scenarios in one launch per GPU.  Header-only over the C ABI (include/pqp.h: pqp_multi_*); the scenario types are the
// reference's (ReferencePath / VehicleState / SlState; define PQP_USE_REFERENCE_TYPES and include the reference's headers first) or
// the mirror types of pqp_types.hpp.
//
//   PathOptimizationNS::BatchedPathSolver   add(reference_path, vehicle_state) ... optimizePaths(&paths, &ok)
//   PathOptimizationNS::MapFreePathOptimizer  the reference's PathOptimizer without the grid map (path_optimizer.hpp:24-57 needs
//                                           grid_map::GridMap to build reference states and bounds; here they are handed in):
//                                           solve(reference_path, vehicle_state, &final_path) == optimizePath for one scenario
// Scenarios may have different numbers of waypoints (a road cut short by an obstacle): the batch is solved with a count per QP.
// Not copyable, not thread-safe (one object per host thread), no exceptions; without a usable GPU every solve returns false.
#pragma once
#include <cstdio>
#include <vector>

#include "pqp.h"
#ifndef PQP_USE_REFERENCE_TYPES
#include "pqp_types.hpp"
#endif

namespace PathOptimizationNS {

class BatchedPathSolver {
 public:
    // n_devices GPUs of this node (device ordinals 0 .. n_devices - 1), one handle + host thread each; params == nullptr: the engine's
    // production setting (pqp_production_params); pass pqp_default_params for the reference's own OSQP setting (eps 2e-3, no polish)
    explicit BatchedPathSolver(int n_devices = 1, const pqp_params* params = nullptr) : n_devices_(n_devices < 1 ? 1 : n_devices) {
        if (params) params_ = *params; else pqp_production_params(&params_);
    }
    BatchedPathSolver(const BatchedPathSolver&) = delete;
    BatchedPathSolver& operator=(const BatchedPathSolver&) = delete;
    ~BatchedPathSolver() { if (multi_) pqp_multi_destroy(multi_); }

    // One scenario: what BaseSolver's constructor takes (base_solver.cpp:15-39) minus the input path, which optimizePath derives from
    // the reference states (path_optimizer.cpp:128-137).  Returns the scenario's index in the batch.
    int add(const ReferencePath& reference_path, const VehicleState& vehicle_state, double max_steering_angle = 35.0 * 3.14159265358979323846 / 180.0) {
        const auto& states = reference_path.getReferenceStates();
        const auto& bounds = reference_path.getBounds();
        Scenario s;
        s.n = (int)(states.size() < bounds.size() ? states.size() : bounds.size());
        s.ref.resize((size_t)s.n * PQP_REF_STRIDE);
        s.bounds.resize((size_t)s.n * PQP_BOUNDS_STRIDE);
        for (int i = 0; i < s.n; ++i) {
            double* r = &s.ref[(size_t)i * PQP_REF_STRIDE];
            r[0] = states[i].s; r[1] = states[i].k; r[2] = states[i].heading; r[3] = states[i].x; r[4] = states[i].y;
            double* b = &s.bounds[(size_t)i * PQP_BOUNDS_STRIDE];
            b[0] = bounds[i].front.lb; b[1] = bounds[i].front.ub; b[2] = bounds[i].rear.lb; b[3] = bounds[i].rear.ub;
            b[4] = bounds[i].center.lb; b[5] = bounds[i].center.ub;
        }
        const auto init_error = vehicle_state.getInitError();                 // base_solver.cpp:217-218
        s.scal[0] = init_error[0]; s.scal[1] = init_error[1]; s.scal[2] = vehicle_state.getStartState().k;
        s.scal[3] = vehicle_state.getTargetState().heading; s.scal[4] = reference_path.isBlocked() == nullptr ? 0.0 : 1.0;
        s.scal[5] = max_steering_angle;
        scenarios_.push_back(std::move(s));
        return (int)scenarios_.size() - 1;
    }
    void clear() { scenarios_.clear(); }
    // A planner that re-solves the same scenarios (same count, same order, equal waypoint counts) every cycle: the first solve of scenario k starts
    // from the optimum scenario k had in the previous optimizePaths call instead of cold.  Same paths (the optimum is unique), about a quarter less
    // time.  The reference builds a fresh BaseSolver per cycle (path_optimizer.cpp:138): off by default.
    void setCarryCycles(bool on) { setCarryCycles(on ? 1 : 0); }
    // ... k >= 2: only the scenarios that were among the most expensive 1 / k of the previous call start from their previous optimum, all others
    // start cold - the ones a call waits for (PQP_OPT_CARRY_CYCLES = k; switches PQP_OPT_ORDER_BY_COST on, whose cost keys it needs)
    void setCarryCycles(int k) {
        carry_cycles_ = k < 0 ? 0 : k;
        if (multi_) {
            if (carry_cycles_ >= 2) pqp_multi_set_option(multi_, PQP_OPT_ORDER_BY_COST, 1);
            pqp_multi_set_option(multi_, PQP_OPT_CARRY_CYCLES, carry_cycles_);
        }
    }
    size_t size() const { return scenarios_.size(); }

    // optimizePath for every scenario added so far: cold solve + one re-linearised warm re-solve, fused in one launch per GPU.
    // paths[q] = the SlState fields getOptimizedPath fills (base_solver.cpp:263-288); ok[q] = what the reference's bool pair says
    // (both solves "solved").  Returns false when the engine itself failed (no GPU, bad sizes): nothing was solved then.
    bool optimizePaths(std::vector<std::vector<SlState>>* paths, std::vector<bool>* ok) {
        if (!paths || scenarios_.empty()) return false;
        const int batch = (int)scenarios_.size();
        int n_max = 0;
        for (const auto& s : scenarios_) n_max = s.n > n_max ? s.n : n_max;
        if (n_max < 2) return false;
        if (!multi_) {
            const int per_shard = (batch + n_devices_ - 1) / n_devices_;
            if (pqp_multi_create(&multi_, &params_, n_devices_, nullptr, per_shard, n_max) != PQP_OK) {
                std::fprintf(stderr, "BatchedPathSolver: %s\n", pqp_last_error());
                multi_ = nullptr;
                return false;                      // no CPU fallback
            }
            pqp_multi_set_option(multi_, PQP_OPT_STORE_WARM, 0);       // every call is a complete optimizePath
            if (carry_cycles_ >= 2) pqp_multi_set_option(multi_, PQP_OPT_ORDER_BY_COST, 1);
            if (carry_cycles_) pqp_multi_set_option(multi_, PQP_OPT_CARRY_CYCLES, carry_cycles_);
        }
        const size_t bn = (size_t)batch * n_max;
        ref_.assign(bn * PQP_REF_STRIDE, 0.0); bounds_.assign(bn * PQP_BOUNDS_STRIDE, 0.0); scal_.assign((size_t)batch * PQP_SCAL_STRIDE, 0.0);
        out_.assign(bn * PQP_OUT_STRIDE, 0.0); n_of_.assign(batch, 0); status_.assign(batch, 0); iters_.assign(batch, 0);
        for (int q = 0; q < batch; ++q) {
            const Scenario& s = scenarios_[q];
            n_of_[q] = s.n;
            for (size_t k = 0; k < s.ref.size(); ++k) ref_[(size_t)q * n_max * PQP_REF_STRIDE + k] = s.ref[k];
            for (size_t k = 0; k < s.bounds.size(); ++k) bounds_[(size_t)q * n_max * PQP_BOUNDS_STRIDE + k] = s.bounds[k];
            for (int k = 0; k < PQP_SCAL_STRIDE; ++k) scal_[(size_t)q * PQP_SCAL_STRIDE + k] = s.scal[k];
        }
        bool ragged = false;
        for (int q = 0; q < batch; ++q) ragged = ragged || n_of_[q] != n_max;
        // (equal sizes: no count array - the shape the driver can carry from one planning cycle to the next, setCarryCycles)
        const int rc = pqp_multi_path_solve(multi_, batch, n_max, ragged ? n_of_.data() : nullptr, ref_.data(), nullptr, bounds_.data(), scal_.data(), /*passes=*/1,
                                            out_.data(), status_.data(), iters_.data(), nullptr);
        if (rc != PQP_OK) {
            std::fprintf(stderr, "BatchedPathSolver: %s\n", pqp_last_error());
            return false;
        }
        paths->assign(batch, {});
        if (ok) ok->assign(batch, false);
        for (int q = 0; q < batch; ++q) {
            const bool solved = status_[q] == PQP_STATUS_SOLVED;                     // osqp-eigen: true only for "solved"
            if (ok) (*ok)[q] = solved;
            if (!solved) continue;                                                   // the reference leaves its output vector untouched
            auto& path = (*paths)[q];
            path.reserve(n_of_[q]);
            for (int i = 0; i < n_of_[q]; ++i) {                                    // base_solver.cpp:269-287: s, v, a stay 0
                const double* o = &out_[((size_t)q * n_max + i) * PQP_OUT_STRIDE];
                SlState pt;
                pt.x = o[0]; pt.y = o[1]; pt.heading = o[2]; pt.l = o[3]; pt.d_heading = o[4]; pt.k = o[5]; pt.d_k = o[6];
                path.push_back(pt);
            }
        }
        return true;
    }
    int status(int q) const { return status_[q]; }          // pqp_status of scenario q in the last optimizePaths
    int iterations(int q) const { return iters_[q]; }       // ADMM iterations of both solves
    const pqp_params& params() const { return params_; }

 private:
    struct Scenario {
        int n = 0;
        std::vector<double> ref, bounds;
        double scal[PQP_SCAL_STRIDE] = {0, 0, 0, 0, 0, 0};
    };
    int n_devices_;
    pqp_params params_;
    pqp_multi* multi_{nullptr};
    std::vector<Scenario> scenarios_;
    std::vector<double> ref_, bounds_, scal_, out_;
    std::vector<int32_t> n_of_, status_, iters_;
    int carry_cycles_ = 0;      // PQP_OPT_CARRY_CYCLES value
};

// PathOptimizer::optimizePath (path_optimizer.cpp:124-161) for one scenario whose reference states and bounds already exist.
class MapFreePathOptimizer {
 public:
    explicit MapFreePathOptimizer(const pqp_params* params = nullptr) : batch_(1, params) {}
    bool solve(const ReferencePath& reference_path, const VehicleState& vehicle_state, std::vector<SlState>* final_path) {
        if (!final_path) return false;
        batch_.clear();
        batch_.add(reference_path, vehicle_state);
        std::vector<std::vector<SlState>> paths;
        std::vector<bool> ok;
        if (!batch_.optimizePaths(&paths, &ok) || !ok[0]) return false;           // "Solving failed!" (path_optimizer.cpp:143-156)
        *final_path = std::move(paths[0]);
        return true;
    }

 private:
    BatchedPathSolver batch_;
};

}  // namespace PathOptimizationNS
