// nos_reference_api.hpp — stand-alone mirror of the reference's public C++ surface for the
// two solvers on the hot path, so that this repository builds and tests WITHOUT the
// reference tree and without Eigen (neither exists on the build / GPU boxes).
//
// When the HIP solver classes are dropped into the real reference tree this header is not
// used at all: the classes include the reference's own
//   nonlinear_optimizer/options.h, loss_function.h, types.h,
//   mahalanobis_distance_minimizer/{types.h, mahalanobis_distance_minimizer.h},
//   reprojection_error_minimizer/{types.h, reprojection_error_minimizer.h}
// (define NOS_IN_REFERENCE_TREE).  Names, members, defaults and error
// behaviour below follow those headers:
//   Options                      NO/options.h:15-28
//   LossFunction & subclasses    NO/loss_function.h:11-77 (scalar overloads only: the
//                                simd::Scalar overloads need the external simd_helper)
//   Vec*/Mat*/Pose/Orientation   NO/types.h:8-58 (Eigen typedefs → minimal value types)
//   NDT, Correspondence          MDM/types.h:11-26
//   CameraIntrinsics, Correspondence   REM/types.h:14-28
//   MahalanobisDistanceMinimizer MDM/mahalanobis_distance_minimizer.h:20-42
//   ReprojectionErrorMinimizer   REM/reprojection_error_minimizer.h:14-55
//   MultiThreadExecutor          NO/multi_thread_executor.h (accepted, never used: the GPU
//                                grid replaces the thread fan-out)
//   Constraint, PoseParameter    NO/pose_graph_optimizer/types.h:11-19, pose_graph_optimizer.h:16-19
#ifndef NOS_REFERENCE_API_HPP_
#define NOS_REFERENCE_API_HPP_

#include <cmath>
#include <memory>
#include <optional>
#include <stdexcept>
#include <vector>

namespace nonlinear_optimizer {

// ---- options ---------------------------------------------------------------------
enum class MinimizerType { kGaussNewton = 0, kGradientDescent, kQuasiNewton, kLevenbergMarquardt };
enum class LinearSolverType { kDenseQR = 0, kDenseCholesky, kSparseCholesky };

struct Options {
  int max_iterations{40};
  MinimizerType minimizer_type{MinimizerType::kGaussNewton};        // ignored by analytic solvers
  LinearSolverType linear_solver_type{LinearSolverType::kDenseQR};  // ignored by analytic solvers
  struct {
    double function_tolerance{1e-6};  // ignored by analytic solvers
    double gradient_tolerance{1e-6};
    double parameter_tolerance{1e-6};
  } convergence_handle;
  struct {
    double min_lambda{1e-6};  // ignored: λ bounds are constexpr in the solvers
    double max_lambda{1e-2};
  } optimization_handle;
};

// ---- robust losses ---------------------------------------------------------------
class LossFunction {
 public:
  LossFunction() {}
  virtual ~LossFunction() {}
  // output[0] = rho(s), output[1] = rho'(s) (the weight), output[2] = rho''(s) (exponential only)
  virtual void Evaluate(const double squared_residual, double* output) = 0;
};

class ExponentialLossFunction : public LossFunction {
 public:
  ExponentialLossFunction(const double c1, const double c2) : c1_{c1}, c2_{c2}, two_c1c2_{2.0 * c1 * c2} {
    if (c1_ < 0.0) throw std::out_of_range("`c1_` should be positive number.");
    if (c2_ < 0.0) throw std::out_of_range("`c2_` should be positive number.");
  }
  void Evaluate(const double squared_residual, double output[3]) final {
    const double exp_term = std::exp(-c2_ * squared_residual);
    output[0] = c1_ - c1_ * exp_term;
    output[1] = two_c1c2_ * exp_term;
    output[2] = -2.0 * c2_ * output[1];
  }

 private:
  double c1_{0.0};
  double c2_{0.0};
  double two_c1c2_{0.0};
};

class HuberLossFunction : public LossFunction {
 public:
  explicit HuberLossFunction(const double threshold)
      : threshold_{threshold}, squared_threshold_{threshold * threshold} {
    if (threshold_ <= 0.0) throw std::out_of_range("threshold value should be larger than zero.");
  }
  void Evaluate(const double squared_residual, double output[2]) final {
    if (squared_residual > squared_threshold_) {
      const double residual = std::sqrt(squared_residual);
      output[0] = 2.0 * threshold_ * residual - squared_threshold_;
      output[1] = threshold_ / residual;
    } else {
      output[0] = squared_residual;
      output[1] = 1.0;
    }
  }

 private:
  double threshold_{0.0};
  const double squared_threshold_;
};

// ---- minimal value types standing in for the Eigen typedefs ------------------------
struct Vec2 {
  double v[2]{0.0, 0.0};
  Vec2() {}
  Vec2(double a, double b) : v{a, b} {}
  static Vec2 Zero() { return Vec2(); }
  double& operator()(int i) { return v[i]; }
  const double& operator()(int i) const { return v[i]; }
  double& x() { return v[0]; }
  double& y() { return v[1]; }
  double x() const { return v[0]; }
  double y() const { return v[1]; }
};

struct Vec3 {
  double v[3]{0.0, 0.0, 0.0};
  Vec3() {}
  Vec3(double a, double b, double c) : v{a, b, c} {}
  static Vec3 Zero() { return Vec3(); }
  double& operator()(int i) { return v[i]; }
  const double& operator()(int i) const { return v[i]; }
  double& x() { return v[0]; }
  double& y() { return v[1]; }
  double& z() { return v[2]; }
  double x() const { return v[0]; }
  double y() const { return v[1]; }
  double z() const { return v[2]; }
  Vec3 operator+(const Vec3& o) const { return Vec3(v[0] + o.v[0], v[1] + o.v[1], v[2] + o.v[2]); }
  Vec3 operator-(const Vec3& o) const { return Vec3(v[0] - o.v[0], v[1] - o.v[1], v[2] - o.v[2]); }
  Vec3 operator*(double s) const { return Vec3(v[0] * s, v[1] * s, v[2] * s); }
  double norm() const { return std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); }
};

struct Mat3x3 {
  double m[9]{0, 0, 0, 0, 0, 0, 0, 0, 0};  // row-major
  static Mat3x3 Zero() { return Mat3x3(); }
  static Mat3x3 Identity() {
    Mat3x3 r;
    r.m[0] = r.m[4] = r.m[8] = 1.0;
    return r;
  }
  double& operator()(int i, int j) { return m[3 * i + j]; }
  const double& operator()(int i, int j) const { return m[3 * i + j]; }
  Vec3 operator*(const Vec3& p) const {
    return Vec3(m[0] * p.v[0] + m[1] * p.v[1] + m[2] * p.v[2], m[3] * p.v[0] + m[4] * p.v[1] + m[5] * p.v[2],
                m[6] * p.v[0] + m[7] * p.v[1] + m[8] * p.v[2]);
  }
  Mat3x3 operator*(const Mat3x3& o) const {
    Mat3x3 r;
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) r.m[3 * i + j] = m[3 * i] * o.m[j] + m[3 * i + 1] * o.m[3 + j] + m[3 * i + 2] * o.m[6 + j];
    return r;
  }
  Mat3x3 transpose() const {
    Mat3x3 r;
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) r.m[3 * i + j] = m[3 * j + i];
    return r;
  }
};

// Stand-in for Eigen::Isometry3d with the members the solvers and tests touch.
class Pose {
 public:
  static Pose Identity() { return Pose(); }
  Vec3& translation() { return t_; }
  const Vec3& translation() const { return t_; }
  Mat3x3& linear() { return R_; }
  const Mat3x3& linear() const { return R_; }
  const Mat3x3& rotation() const { return R_; }
  Pose inverse() const {
    Pose r;
    r.R_ = R_.transpose();
    const Vec3 nt = r.R_ * t_;
    r.t_ = Vec3(-nt.v[0], -nt.v[1], -nt.v[2]);
    return r;
  }
  Vec3 operator*(const Vec3& p) const { return R_ * p + t_; }
  Pose operator*(const Pose& o) const {
    Pose r;
    r.R_ = R_ * o.R_;
    r.t_ = R_ * o.t_ + t_;
    return r;
  }

 private:
  Mat3x3 R_{Mat3x3::Identity()};
  Vec3 t_{};
};

// Stand-in for MultiThreadExecutor: the HIP solvers accept one (API compatibility) and never
// use it — the kernel grid and, across GPUs, the context's shards replace the thread pool.
class MultiThreadExecutor {
 public:
  explicit MultiThreadExecutor(int num_threads) : num_threads_{num_threads} {}
  int GetNumOfTotalThreads() const { return num_threads_; }

 private:
  int num_threads_{0};
};

namespace mahalanobis_distance_minimizer {

struct NDT {
  int count{0};
  Vec3 sum{Vec3::Zero()};
  Mat3x3 moment{Mat3x3::Identity()};

  Vec3 mean{Vec3::Zero()};
  Mat3x3 information{Mat3x3::Identity()};
  Mat3x3 sqrt_information{Mat3x3::Identity()};
  bool is_valid{false};
  bool is_planar{false};
};

struct Correspondence {
  Vec3 point{Vec3::Zero()};
  NDT ndt;
};

class MahalanobisDistanceMinimizer {
 public:
  MahalanobisDistanceMinimizer() {}
  virtual ~MahalanobisDistanceMinimizer() {}

  void SetMultiThreadExecutor(const std::shared_ptr<MultiThreadExecutor>& multi_thread_executor) {
    multi_thread_executor_ = multi_thread_executor;
  }
  void SetLossFunction(const std::shared_ptr<LossFunction>& loss_function) { loss_function_ = loss_function; }

  virtual bool Solve(const Options& options, const std::vector<Correspondence>& correspondences, Pose* pose) = 0;

 protected:
  std::shared_ptr<LossFunction> loss_function_{nullptr};
  std::shared_ptr<MultiThreadExecutor> multi_thread_executor_{nullptr};
};

}  // namespace mahalanobis_distance_minimizer

namespace reprojection_error_minimizer {

struct CameraIntrinsics {
  double fx{0.0};
  double fy{0.0};
  double cx{0.0};
  double cy{0.0};
  double inv_fx{0.0};
  double inv_fy{0.0};
  int width{0};
  int height{0};
};

struct Correspondence {
  Vec3 local_point{Vec3::Zero()};    // represented in reference frame
  Vec2 matched_pixel{Vec2::Zero()};  // represented in query frame
};

class ReprojectionErrorMinimizer {
 public:
  ReprojectionErrorMinimizer() {}
  virtual ~ReprojectionErrorMinimizer() {}

  void SetLossFunction(const std::shared_ptr<LossFunction>& loss_function) { loss_function_ = loss_function; }

  virtual bool Solve(const Options& options, const std::vector<Correspondence>& correspondences,
                     const CameraIntrinsics& camera_intrinsics, Pose* pose) = 0;

 protected:
  std::shared_ptr<LossFunction> loss_function_{nullptr};
};

}  // namespace reprojection_error_minimizer

namespace pose_graph_optimizer {

// NO/pose_graph_optimizer/types.h:11-19
enum class ConstraintType { kOdometry = 0, kLoop = 1 };

struct Constraint {
  int reference_pose_index{-1};
  int query_pose_index{-1};
  Pose relative_pose_from_reference_to_query{Pose::Identity()};
  double switch_parameter{1.0};
  ConstraintType type{ConstraintType::kOdometry};
};

// NO/pose_graph_optimizer/pose_graph_optimizer.h:16-19
struct PoseParameter {
  double position[3];
  double orientation[4];  // w x y z
};

}  // namespace pose_graph_optimizer

}  // namespace nonlinear_optimizer

#endif  // NOS_REFERENCE_API_HPP_
