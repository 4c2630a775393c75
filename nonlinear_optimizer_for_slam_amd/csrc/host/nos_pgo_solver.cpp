// nos_pgo_solver.cpp — see nos_pgo_solver.hpp.
#include "nos_pgo_solver.hpp"

#include <algorithm>
#include <cmath>
#include <limits>
#include <map>

#include "nos_lm.hpp"

namespace nonlinear_optimizer {
namespace pose_graph_optimizer {

namespace {

void PoseToParameter(const Pose& pose, double out[7]) {
  double R[9];
  for (int i = 0; i < 3; ++i) {
    out[i] = pose.translation()(i);
    for (int j = 0; j < 3; ++j) R[3 * i + j] = pose.linear()(i, j);
  }
  const nos_host::Quat q = nos_host::QuatFromMatrix(R);  // Orientation quaternion(pose_ptr->rotation())
  out[3] = q.w;
  out[4] = q.x;
  out[5] = q.y;
  out[6] = q.z;
}

}  // namespace

bool PoseGraphOptimizerHip::Solve(const Options& options) {
#ifdef NOS_IN_REFERENCE_TREE
  std::map<int, Pose*> ordered;
  for (const auto& [index, parameter] : optimized_pose_map_) {
    (void)parameter;
    ordered[index] = index_to_pose_ptr_bimap_.GetValue(index);
  }
#else
  const std::map<int, Pose*> ordered(index_to_pose_ptr_.begin(), index_to_pose_ptr_.end());
#endif
  report_ = PgoSolveReport();
  if (ordered.empty()) return false;
  // compact pose indices (the reference keys poses by arbitrary ints)
  std::map<int, int> slot_of;
  std::vector<double> poses(7 * ordered.size());
  std::vector<unsigned char> fixed(ordered.size(), 0);
  {
    int k = 0;
    for (const auto& [index, pose_ptr] : ordered) {
      slot_of[index] = k;
      PoseToParameter(*pose_ptr, &poses[7 * static_cast<size_t>(k)]);
      if (fixed_pose_index_set_.count(index) != 0) fixed[k] = 1;
      ++k;
    }
  }
  const size_t m = constraints_.size();
  std::vector<int32_t> ref(m), qry(m);
  std::vector<double> meas(7 * std::max<size_t>(m, 1)), sw(std::max<size_t>(m, 1), 1.0);
  std::vector<unsigned char> sw_free(std::max<size_t>(m, 1), 0);
  for (size_t e = 0; e < m; ++e) {
    const Constraint& c = constraints_[e];
    ref[e] = slot_of.at(c.reference_pose_index);
    qry[e] = slot_of.at(c.query_pose_index);
    PoseToParameter(c.relative_pose_from_reference_to_query, &meas[7 * e]);
    sw[e] = c.switch_parameter;
    sw_free[e] = (c.type == ConstraintType::kLoop) ? 1 : 0;  // pose_graph_optimizer_ceres.cc:29-36
  }

  nos_ctx* ctx = nullptr;
  int status = nos_ctx_create(&hip_options_.device_id, 1, &ctx);
  nos_pose_graph* pg = nullptr;
  if (status == NOS_OK)
    status = nos_pgo_create(ctx, ordered.size(), poses.data(), m, ref.data(), qry.data(), meas.data(), sw.data(),
                            sw_free.data(), fixed.data(), &pg);
  if (status != NOS_OK) {
    std::cerr << "[nos-hip] pose-graph setup failed: " << nos_status_string(status) << " — " << nos_last_error()
              << std::endl;
    report_.status = status;
    if (ctx != nullptr) nos_ctx_destroy(ctx);
    return false;
  }

  double lambda = 1e-3;
  double previous_cost = std::numeric_limits<double>::max();
  int iteration = 0;
  for (; iteration < options.max_iterations; ++iteration) {
    double cost = 0.0, gradient_norm = 0.0, rel_residual = 0.0, step_norm = 0.0;
    int pcg_iterations = 0;
    status = nos_pgo_linearize(pg, &cost, &gradient_norm);
    if (status == NOS_OK)
      status = nos_pgo_solve(pg, lambda, hip_options_.pcg_max_iterations, hip_options_.pcg_relative_tolerance,
                             &pcg_iterations, &rel_residual, &step_norm);
    if (status == NOS_OK) status = nos_pgo_retract(pg);
    if (status != NOS_OK) break;
    if (iteration == 0) report_.initial_cost = cost;
    report_.final_cost = cost;
    report_.final_gradient_norm = gradient_norm;
    report_.final_step_norm = step_norm;
    report_.total_pcg_iterations += pcg_iterations;
    if (step_norm < options.convergence_handle.parameter_tolerance) break;
    if (gradient_norm < options.convergence_handle.gradient_tolerance) break;
    lambda = std::clamp(lambda * (cost > previous_cost ? 2.0 : 0.6), nos_host::kMinLambda, nos_host::kMaxLambda);
    previous_cost = cost;
  }
  report_.iterations = iteration;
  report_.status = status;
  switches_.assign(m, 1.0);
  if (status == NOS_OK) status = nos_pgo_get_state(pg, poses.data(), m > 0 ? switches_.data() : nullptr);
  nos_pgo_destroy(pg);
  nos_ctx_destroy(ctx);
  if (status != NOS_OK) {
    std::cerr << "[nos-hip] pose-graph optimisation failed: " << nos_status_string(status) << " — "
              << nos_last_error() << std::endl;
    report_.status = status;
    return false;  // "If not, the poses are not changed."
  }
  if (hip_options_.print_summary)
    std::cerr << "COST: " << report_.initial_cost << " -> " << report_.final_cost << ", iter: " << report_.iterations
              << ", pcg iterations: " << report_.total_pcg_iterations << std::endl;
  // UpdateOptimizedPose (pose_graph_optimizer.h:89-101)
  int k = 0;
  for (const auto& [index, pose_ptr] : ordered) {
    (void)index;
    const double* p = &poses[7 * static_cast<size_t>(k)];
    nos_host::Quat q;
    const double n = std::sqrt(p[3] * p[3] + p[4] * p[4] + p[5] * p[5] + p[6] * p[6]);
    q.w = p[3] / n;
    q.x = p[4] / n;
    q.y = p[5] / n;
    q.z = p[6] / n;
    double R[9];
    nos_host::QuatToMatrix(q, R);
    for (int i = 0; i < 3; ++i) {
      pose_ptr->translation()(i) = p[i];
      for (int j = 0; j < 3; ++j) pose_ptr->linear()(i, j) = R[3 * i + j];
    }
    ++k;
  }
  return true;
}

}  // namespace pose_graph_optimizer
}  // namespace nonlinear_optimizer

// ---- C entry point for the Python test-suite: builds Pose / Constraint objects and calls the class ----
extern "C" int nos_host_pgo_solve(size_t n_poses, const int* pose_indices, double* poses /*[n][7] in/out*/,
                                  size_t n_constraints, const int* ref_index, const int* qry_index,
                                  const double* meas /*[m][7]*/, const unsigned char* is_loop, size_t n_fixed,
                                  const int* fixed_indices, int max_iterations, double gradient_tolerance,
                                  double parameter_tolerance, int pcg_max_iterations, double pcg_tolerance,
                                  double* switches_out, double report[6]) {
  using namespace nonlinear_optimizer;
  using namespace nonlinear_optimizer::pose_graph_optimizer;
  try {
    auto to_pose = [](const double* p) {
      Pose pose = Pose::Identity();
      nos_host::Quat q;
      q.w = p[3];
      q.x = p[4];
      q.y = p[5];
      q.z = p[6];
      double R[9];
      nos_host::QuatToMatrix(q, R);
      for (int i = 0; i < 3; ++i) {
        pose.translation()(i) = p[i];
        for (int j = 0; j < 3; ++j) pose.linear()(i, j) = R[3 * i + j];
      }
      return pose;
    };
    std::vector<Pose> pose_objects(n_poses);
    for (size_t i = 0; i < n_poses; ++i) pose_objects[i] = to_pose(poses + 7 * i);
    PgoHipOptions hip;
    hip.pcg_max_iterations = pcg_max_iterations;
    hip.pcg_relative_tolerance = pcg_tolerance;
    hip.print_summary = false;
    PoseGraphOptimizerHip optimizer(hip);
    for (size_t i = 0; i < n_poses; ++i) optimizer.SetPose(pose_indices[i], &pose_objects[i]);
    for (size_t i = 0; i < n_fixed; ++i) optimizer.SetPoseConstant(fixed_indices[i]);
    for (size_t e = 0; e < n_constraints; ++e) {
      Constraint c;
      c.reference_pose_index = ref_index[e];
      c.query_pose_index = qry_index[e];
      c.relative_pose_from_reference_to_query = to_pose(meas + 7 * e);
      c.type = is_loop[e] ? ConstraintType::kLoop : ConstraintType::kOdometry;
      optimizer.SetConstraint(c);
    }
    Options options;
    options.max_iterations = max_iterations;
    options.convergence_handle.gradient_tolerance = gradient_tolerance;
    options.convergence_handle.parameter_tolerance = parameter_tolerance;
    const bool ok = optimizer.Solve(options);
    const PgoSolveReport& r = optimizer.report();
    if (report != nullptr) {
      report[0] = r.iterations;
      report[1] = r.initial_cost;
      report[2] = r.final_cost;
      report[3] = r.final_gradient_norm;
      report[4] = double(r.total_pcg_iterations);
      report[5] = r.status;
    }
    if (!ok) return 0;
    for (size_t i = 0; i < n_poses; ++i) {
      double R[9];
      for (int a = 0; a < 3; ++a) {
        poses[7 * i + a] = pose_objects[i].translation()(a);
        for (int b = 0; b < 3; ++b) R[3 * a + b] = pose_objects[i].linear()(a, b);
      }
      const nos_host::Quat q = nos_host::QuatFromMatrix(R);
      poses[7 * i + 3] = q.w;
      poses[7 * i + 4] = q.x;
      poses[7 * i + 5] = q.y;
      poses[7 * i + 6] = q.z;
    }
    if (switches_out != nullptr)
      for (size_t e = 0; e < optimizer.switch_parameters().size(); ++e) switches_out[e] = optimizer.switch_parameters()[e];
    return 1;
  } catch (...) {
    return 0;
  }
}
