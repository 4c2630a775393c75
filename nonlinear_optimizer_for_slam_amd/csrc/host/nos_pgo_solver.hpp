// nos_pgo_solver.hpp — drop-in pose-graph optimiser with the reference's public surface
// (nonlinear_optimizer/pose_graph_optimizer/pose_graph_optimizer.h:21-108: SetPose / SetConstraint /
// SetPoseConstant / Solve(options)), backed by the GPU entry points nos_pgo_* of include/nos.h.
//
// It fills the role of PoseGraphOptimizerAnalytic, whose Solve() is an empty TODO loop in the reference
// (pose_graph_optimizer_analytic.cc:12-51): linearise → solve the damped normal equations → update poses →
// check convergence, with the LM bookkeeping of the reference's other analytic solvers (always step,
// lambda x2 / x0.6 on the cost, clamp [1e-6, 1e-2], convergence tested after the update).  Odometry
// constraints keep switch 1; loop constraints carry a free switch variable exactly as the Ceres path sets
// them up (pose_graph_optimizer_ceres.cc:29-42).
#ifndef NOS_PGO_SOLVER_HPP_
#define NOS_PGO_SOLVER_HPP_

#include <iostream>
#include <memory>
#include <set>
#include <unordered_map>
#include <vector>

#include "nos_hip_solvers.hpp"

namespace nonlinear_optimizer {
namespace pose_graph_optimizer {

#ifndef NOS_IN_REFERENCE_TREE
// Stand-alone mirror of the abstract base (in the reference tree the real header is used).
class PoseGraphOptimizer {
 public:
  PoseGraphOptimizer() {}
  virtual ~PoseGraphOptimizer() {}

  void SetLossFunction(const std::shared_ptr<LossFunction>& loss_function) { loss_function_ = loss_function; }

  void SetConstraint(const Constraint& constraint) {
    if (index_to_pose_ptr_.count(constraint.query_pose_index) == 0 ||
        index_to_pose_ptr_.count(constraint.reference_pose_index) == 0) {
      std::cerr << "Constraint is invalid.\n";
      return;
    }
    constraints_.push_back(constraint);
  }

  void SetPose(const int pose_index, Pose* pose_ptr) { index_to_pose_ptr_[pose_index] = pose_ptr; }

  void SetPoseConstant(const int pose_index) {
    if (index_to_pose_ptr_.count(pose_index) == 0) {
      std::cerr << "Queried pose index is never registered into the solver.\n";
      return;
    }
    fixed_pose_index_set_.insert(pose_index);
  }

  virtual bool Solve(const Options& options) = 0;

 protected:
  std::shared_ptr<LossFunction> loss_function_{nullptr};
  std::unordered_map<int, Pose*> index_to_pose_ptr_;
  std::set<int> fixed_pose_index_set_;
  std::vector<Constraint> constraints_;
};
#endif

struct PgoHipOptions {
  int device_id{0};
  int pcg_max_iterations{2000};
  double pcg_relative_tolerance{1e-10};
  bool print_summary{true};
};

struct PgoSolveReport {
  int iterations{0};
  double initial_cost{0.0};
  double final_cost{0.0};
  double final_gradient_norm{0.0};
  double final_step_norm{0.0};
  long total_pcg_iterations{0};
  int status{0};
};

class PoseGraphOptimizerHip : public PoseGraphOptimizer {
 public:
  PoseGraphOptimizerHip() {}
  explicit PoseGraphOptimizerHip(const PgoHipOptions& hip_options) : hip_options_(hip_options) {}

  /// Registered poses are overwritten with the optimised ones on success (same contract as the reference,
  /// pose_graph_optimizer.h:62-67).
  bool Solve(const Options& options) final;

  const PgoSolveReport& report() const { return report_; }
  /// Optimised switch values of the constraints, in SetConstraint order (1 for odometry constraints).
  const std::vector<double>& switch_parameters() const { return switches_; }

 private:
  PgoHipOptions hip_options_;
  PgoSolveReport report_;
  std::vector<double> switches_;
};

}  // namespace pose_graph_optimizer
}  // namespace nonlinear_optimizer

#endif  // NOS_PGO_SOLVER_HPP_
