// pqp_base_solver.hpp — drop-in for the reference's solver class (include/solver/base_solver.hpp:21-69):
// same namespace, same constructor and method signatures, same bool semantics; the OsqpEigen::Solver member is
// replaced by a handle of the MI355X engine (include/pqp.h).  Batch = 1 goes through exactly the same kernels as
// batch = 65 536.
//
//   reference                                                   here
//   BaseSolver(const ReferencePath&, const VehicleState&,       same; copies what the kernels read into flat arrays
//              const std::vector<SlState>&)  base_solver.cpp:15   (sizes: pqp_path_sizes)
//   bool solve(std::vector<SlState>*)        base_solver.cpp:56   pqp_path_solve(warm = 0, passes = 0)
//   bool updateProblemFormulationAndSolve(const std::vector<SlState>&, std::vector<SlState>*)   base_solver.cpp:97
//                                                                 pqp_path_solve(warm = 1, lin = input, passes = 0)
// In/out may alias in updateProblemFormulationAndSolve (path_optimizer.cpp:153): the input is copied first, as
// base_solver.cpp:100 does.  Not copyable, not thread-safe (like the reference); one GPU handle per instance, taken from / parked in a
// process-wide pool (setHandleCaching) so that the reference's one-solver-per-cycle pattern does not pay pqp_create every cycle.
#pragma once
#include <vector>

#include "pqp.h"
#ifndef PQP_USE_REFERENCE_TYPES
#include "pqp_types.hpp"
#endif

namespace PathOptimizationNS {

class BaseSolver {
 public:
    BaseSolver() = delete;
    BaseSolver(const ReferencePath& reference_path, const VehicleState& vehicle_state, const std::vector<SlState>& input_path);
    BaseSolver(const BaseSolver&) = delete;
    BaseSolver& operator=(const BaseSolver&) = delete;
    virtual ~BaseSolver();

    virtual bool solve(std::vector<SlState>* optimized_path);
    virtual bool updateProblemFormulationAndSolve(const std::vector<SlState>& input_path, std::vector<SlState>* optimized_path);

    // extras the reference does not have
    void setParams(const pqp_params& p);          // the gflags the path reads + solver settings (defaults = reference)
    void setMaxSteeringAngle(double rad) { max_steering_angle_ = rad; }   // FLAGS_max_steering_angle (planning_flags.cpp:22)
    void setDevice(int device);                   // GPU of this instance (default: environment PQP_DEVICE, else 0); before the first solve
    int device() const { return device_; }
    // Handles (stream + device workspaces) are parked when an instance dies and taken over by the next instance on the same device: the
    // reference builds a BaseSolver per planning cycle (path_optimizer.cpp:138), and creating a handle costs more than a batch-1 solve.
    void setHandleCaching(bool on) { cache_handles_ = on; }
    static void releaseCachedHandles();           // destroys the parked handles (e.g. before hipDeviceReset)
    const pqp_params& params() const { return params_; }
    int lastStatus() const { return status_; }    // pqp_status of the last solve
    int lastIterations() const { return iters_; }
    size_t vars() const { return vars_size_; }
    size_t cons() const { return cons_size_; }
    size_t precisePlanningSize() const { return precise_planning_size_; }     // base_solver.cpp:25-34

 protected:
    void updateSizes();
    bool run(const std::vector<SlState>& lin, bool warm, std::vector<SlState>* out);

    const size_t n_{};
    size_t state_size_{}, control_size_{}, slack_size_{}, vars_size_{}, cons_size_{}, precise_planning_size_{};
    const ReferencePath& reference_path_;
    const VehicleState& vehicle_state_;
    std::vector<SlState> input_path_;
    pqp_params params_;
    pqp_handle* handle_{nullptr};
    int device_{0};
    bool cache_handles_{true};
    bool solved_once_{false};
    double max_steering_angle_{35.0 * 3.14159265358979323846 / 180.0};
    int status_{0}, iters_{0};
};

}  // namespace PathOptimizationNS
