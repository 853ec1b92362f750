// pqp_types.hpp — the minimal host types the BaseSolver shim reads, with the accessor names of the reference
// (LiJiangnanBit/path_optimizer_2: include/data_struct/data_struct.hpp:14-32,74-93, include/data_struct/reference_path.hpp,
// include/data_struct/vehicle_state_frenet.hpp).  Inside the reference tree these are NOT needed: the shim
// (path_optimizer_2_amd/csrc/base_solver_shim.cpp) only uses the public getters listed in INTEGRATION.md and
// compiles against the reference's own headers.  Stand-alone users (tests, the demo in INTEGRATION.md) use these.
#pragma once
#include <memory>
#include <vector>

namespace PathOptimizationNS {

struct State {
    State() = default;
    State(double x_, double y_, double heading_ = 0.0, double k_ = 0.0, double s_ = 0.0, double v_ = 0.0, double a_ = 0.0)
        : x(x_), y(y_), heading(heading_), k(k_), s(s_), v(v_), a(a_) {}          // data_struct.hpp:16-17 (d_k has no constructor argument)
    double x{}, y{}, heading{}, k{}, d_k{}, s{}, v{}, a{};
};

struct SlState : public State {
    double l{}, d_heading{};
};

struct VehicleStateBound {
    struct SingleBound {
        double ub{}, lb{};       // left / right
        double x{}, y{}, heading{};
    } front, rear, center;
};

// What BaseSolver reads from ReferencePath (reference_path.cpp:49-55,73)
class ReferencePath {
 public:
    const std::vector<State>& getReferenceStates() const { return states_; }
    const std::vector<VehicleStateBound>& getBounds() const { return bounds_; }
    const State* isBlocked() const { return blocked_.get(); }          // nullptr when the corridor is open
    double getLength() const { return states_.empty() ? 0.0 : states_.back().s; }
    void setReferenceStates(std::vector<State> s) { states_ = std::move(s); }
    void setBounds(std::vector<VehicleStateBound> b) { bounds_ = std::move(b); }
    void setBlocked(const State& at) { blocked_.reset(new State(at)); }

 private:
    std::vector<State> states_;
    std::vector<VehicleStateBound> bounds_;
    std::shared_ptr<State> blocked_;
};

// What BaseSolver reads from VehicleState (vehicle_state_frenet.cpp:29-47)
class VehicleState {
 public:
    VehicleState() = default;
    VehicleState(const State& start, const State& target, double init_l, double init_psi)
        : start_(start), target_(target), init_error_{init_l, init_psi} {}
    const State& getStartState() const { return start_; }
    const State& getTargetState() const { return target_; }
    std::vector<double> getInitError() const { return init_error_; }
    void setInitError(double l, double psi) { init_error_ = {l, psi}; }

 private:
    State start_, target_;
    std::vector<double> init_error_{0.0, 0.0};
};

}  // namespace PathOptimizationNS
