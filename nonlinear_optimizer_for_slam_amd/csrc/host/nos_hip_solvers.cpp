// nos_hip_solvers.cpp — see nos_hip_solvers.hpp.
#include "nos_hip_solvers.hpp"

#include <cmath>
#include <iostream>
#include <map>
#include <mutex>

#include "nos_lm.hpp"

namespace nonlinear_optimizer {

namespace {

// Contexts (stream, pinned blocks, workspaces, buffer pool) are cached per device list for the life of the process:
// the reference's callers construct a solver object per optimisation call (tests/simple_optimization_test.cc builds
// one inside every OptimizePose* function), and creating a context costs ≈ 1.5 ms — more than a whole small Solve().
// The cache is deliberately never destroyed at exit (HIP may already be gone during static destruction);
// ReleaseHipRuntimes() drops it explicitly.
std::mutex g_runtime_mutex;
using RuntimeCache = std::map<std::vector<int>, std::shared_ptr<HipRuntime>>;
RuntimeCache& Runtimes() {
  static RuntimeCache* cache = new RuntimeCache();
  return *cache;
}

std::shared_ptr<HipRuntime> AcquireRuntime(const std::vector<int>& device_ids) {
  std::lock_guard<std::mutex> lock(g_runtime_mutex);
  RuntimeCache& cache = Runtimes();
  auto it = cache.find(device_ids);
  if (it != cache.end() && it->second->status() == NOS_OK) return it->second;
  auto fresh = std::make_shared<HipRuntime>(device_ids);
  if (fresh->status() == NOS_OK) cache[device_ids] = fresh;  // a failed context (no device) is not cached: retried next time
  return fresh;
}

}  // namespace

void ReleaseHipRuntimes() {
  std::lock_guard<std::mutex> lock(g_runtime_mutex);
  Runtimes().clear();
}

namespace {

void ReportFailure(const char* where, int status) {
  std::cerr << "[nos-hip] " << where << " failed: " << nos_status_string(status) << " — " << nos_last_error()
            << std::endl;
}

nos_host::LmSettings SettingsFrom(const Options& options) {
  nos_host::LmSettings s;
  s.max_iterations = options.max_iterations;
  s.gradient_tolerance = options.convergence_handle.gradient_tolerance;
  s.parameter_tolerance = options.convergence_handle.parameter_tolerance;
  return s;
}

nos_host::LmSettings SettingsFrom(const Options& options, const HipOptions& hip, bool ndt) {
  nos_host::LmSettings settings = SettingsFrom(options);
  settings.float_schedule = (hip.simd_class && ndt) ? 1 : 0;  // host loop; the device loop reads it off the dataset
  return settings;
}

void FillReport(const nos_host::LmReport& lm, int status, HipSolveReport* rep) {
  rep->iterations = lm.iterations;
  rep->printed_cost = lm.printed_cost;
  rep->last_cost = lm.last_cost;
  rep->final_lambda = lm.final_lambda;
  rep->status = status;
}

// Device-resident loop when the options allow it and the context supports it.  Returns true if the loop ran on
// the device (lm / status filled in); false = the caller runs the host loop.
template <typename SolveFn>
bool TryDeviceLoop(const HipOptions& hip, const Options& options, SolveFn&& solve, nos_host::LmReport* lm, int* status) {
  if (!hip.device_loop) return false;
  nos_lm_options o{};
  o.max_iterations = options.max_iterations;
  o.launches_in_flight = 0;
  o.gradient_tolerance = options.convergence_handle.gradient_tolerance;
  o.parameter_tolerance = options.convergence_handle.parameter_tolerance;
  o.cost_history = nullptr;
  nos_lm_report r{};
  const int rc = solve(&o, &r);
  if (rc == NOS_ERR_UNSUPPORTED) return false;  // multi-device context: host loop sums the shards
  *status = rc;
  lm->iterations = r.iterations;
  lm->printed_cost = r.printed_cost;
  lm->last_cost = r.last_cost;
  lm->final_lambda = r.final_lambda;
  lm->ok = rc == NOS_OK && r.ok != 0;
  return true;
}

template <typename PoseT>
void ReadPose(const PoseT& pose, double t[3], double R[9]) {
  for (int i = 0; i < 3; ++i) {
    t[i] = pose.translation()(i);
    for (int j = 0; j < 3; ++j) R[3 * i + j] = pose.linear()(i, j);
  }
}

template <typename PoseT>
void WritePose(const double t[3], const double R[9], PoseT* pose) {
  for (int i = 0; i < 3; ++i) {
    pose->translation()(i) = t[i];
    for (int j = 0; j < 3; ++j) pose->linear()(i, j) = R[3 * i + j];
  }
}

}  // namespace

HipRuntime::HipRuntime(const std::vector<int>& device_ids) {
  status_ = nos_ctx_create(device_ids.data(), static_cast<int>(device_ids.size()), &ctx_);
  if (status_ != NOS_OK) ReportFailure("nos_ctx_create", status_);
}

HipRuntime::~HipRuntime() {
  if (ctx_ != nullptr) nos_ctx_destroy(ctx_);
}

bool DescribeLossFunction(LossFunction* loss_function, nos_loss* out) {
  out->kind = NOS_LOSS_NONE;
  out->reserved = 0;
  out->a = 0.0;
  out->b = 0.0;
  if (loss_function == nullptr) return true;
  if (dynamic_cast<ExponentialLossFunction*>(loss_function) != nullptr) {
    // rho'(0) = 2 c1 c2 and rho''(0) = -2 c2 rho'(0)  (loss_function.h:28-33)
    double at0[3] = {0.0, 0.0, 0.0};
    loss_function->Evaluate(0.0, at0);
    out->kind = NOS_LOSS_EXPONENTIAL;
    if (at0[1] == 0.0) return true;  // c1*c2 == 0: rho == 0, w == 0 everywhere
    const double c2 = -at0[2] / (2.0 * at0[1]);
    out->b = c2;
    out->a = at0[1] / (2.0 * c2);
    // rho(s) → c1 exactly once exp(-c2 s) underflows to 0; prefer that over the quotient
    double far[3] = {0.0, 0.0, 0.0};
    loss_function->Evaluate(1e300, far);
    if (c2 * 1e300 > 800.0 && std::isfinite(far[0])) out->a = far[0];
    return true;
  }
  if (dynamic_cast<HuberLossFunction*>(loss_function) != nullptr) {
    // far in the linear zone rho'(s) = th / sqrt(s)  (loss_function.h:57-66); s = 2^200
    // has an exact square root, so th is recovered without rounding for th < 2^100.
    const double s = std::ldexp(1.0, 200);
    double far[3] = {0.0, 0.0, 0.0};
    loss_function->Evaluate(s, far);
    const double th = far[1] * std::ldexp(1.0, 100);
    if (!(th > 0.0) || !std::isfinite(th)) return false;
    out->kind = NOS_LOSS_HUBER;
    out->a = th;
    return true;
  }
  return false;  // unknown subclass: a host virtual cannot run on the GPU and there is no CPU path
}

namespace mahalanobis_distance_minimizer {

namespace {

// Byte offsets of the 15 doubles the solvers read, taken from a live object so the same code
// is right for Eigen's column-major 3x3 and for the stand-in's row-major one.
void NdtFieldOffsets(size_t offsets[NOS_NDT_PLANES]) {
  static const Correspondence probe{};
  const char* base = reinterpret_cast<const char*>(&probe);
  for (int i = 0; i < 3; ++i) {
    offsets[i] = static_cast<size_t>(reinterpret_cast<const char*>(&probe.point(i)) - base);
    offsets[3 + i] = static_cast<size_t>(reinterpret_cast<const char*>(&probe.ndt.mean(i)) - base);
    for (int j = 0; j < 3; ++j)
      offsets[6 + 3 * i + j] =
          static_cast<size_t>(reinterpret_cast<const char*>(&probe.ndt.sqrt_information(i, j)) - base);
  }
}

}  // namespace

MahalanobisDistanceMinimizerHip::MahalanobisDistanceMinimizerHip() {}

MahalanobisDistanceMinimizerHip::MahalanobisDistanceMinimizerHip(const HipOptions& hip_options)
    : hip_options_(hip_options) {}

MahalanobisDistanceMinimizerHip::~MahalanobisDistanceMinimizerHip() { ReleasePrepared(); }

void MahalanobisDistanceMinimizerHip::ReleasePrepared() {
  if (prepared_ != nullptr) {
    nos_dataset_destroy(prepared_);
    prepared_ = nullptr;
  }
}

bool MahalanobisDistanceMinimizerHip::Prepare(const std::vector<Correspondence>& correspondences) {
  ReleasePrepared();
  if (!runtime_) runtime_ = AcquireRuntime(hip_options_.device_ids);
  if (runtime_->status() != NOS_OK) {
    report_.status = runtime_->status();
    return false;
  }
  size_t offsets[NOS_NDT_PLANES];
  NdtFieldOffsets(offsets);
  const size_t count = hip_options_.simd_class ? SimdClassCount(correspondences.size(), SimdClassThreads())
                                               : correspondences.size();
  int rc = nos_ndt_dataset_create_from_records(runtime_->ctx(), count, correspondences.data(), sizeof(Correspondence),
                                               offsets, hip_options_.dtype, &prepared_);
  if (rc == NOS_OK && hip_options_.simd_class) rc = nos_dataset_set_simd_class(prepared_, 1);
  if (rc != NOS_OK) {
    ReportFailure("nos_ndt_dataset_create_from_records", rc);
    report_.status = rc;
    if (prepared_ != nullptr) nos_dataset_destroy(prepared_);
    prepared_ = nullptr;
    return false;
  }
  return true;
}

bool MahalanobisDistanceMinimizerHip::SolvePrepared(const Options& options, Pose* pose) {
  if (prepared_ == nullptr || pose == nullptr) return false;
  nos_loss loss;
  if (!DescribeLossFunction(loss_function_.get(), &loss)) {
    std::cerr << "[nos-hip] unsupported LossFunction subclass: only Exponential and Huber have a device "
                 "restatement and there is no CPU fallback"
              << std::endl;
    report_.status = NOS_ERR_UNSUPPORTED;
    return false;
  }
  return RunLoop(options, prepared_, loss, pose);
}

bool MahalanobisDistanceMinimizerHip::SolveDataset(const Options& options, nos_dataset* dataset, Pose* pose) {
  if (dataset == nullptr || pose == nullptr) return false;
  nos_loss loss;
  if (!DescribeLossFunction(loss_function_.get(), &loss)) {
    std::cerr << "[nos-hip] unsupported LossFunction subclass: only Exponential and Huber have a device "
                 "restatement and there is no CPU fallback"
              << std::endl;
    report_.status = NOS_ERR_UNSUPPORTED;
    return false;
  }
  return RunLoop(options, dataset, loss, pose);
}

bool MahalanobisDistanceMinimizerHip::Solve(const Options& options,
                                            const std::vector<Correspondence>& correspondences, Pose* pose) {
  if (!Prepare(correspondences)) return false;
  const bool ok = SolvePrepared(options, pose);
  ReleasePrepared();
  return ok;
}

bool MahalanobisDistanceMinimizerHip::RunLoop(const Options& options, nos_dataset* dataset, const nos_loss& loss,
                                              Pose* pose) {
  double t[3], R[9];
  ReadPose(*pose, t, R);
  int status = NOS_OK;
  nos_host::LmReport lm;
  if (!TryDeviceLoop(hip_options_, options,
                     [&](const nos_lm_options* o, nos_lm_report* r) { return nos_ndt6_solve(dataset, R, t, &loss, o, r); },
                     &lm, &status))
    lm = nos_host::RunLm6(
        SettingsFrom(options, hip_options_, true),
        [&](const double* Rc, const double* tc, double* out28) {
          status = nos_ndt6_accumulate(dataset, Rc, tc, &loss, out28);
          return status == NOS_OK;
        },
        t, R);
  FillReport(lm, status, &report_);
  if (!lm.ok) {
    if (status != NOS_OK) ReportFailure("nos_ndt6_accumulate / nos_ndt6_solve", status);
    return false;
  }
  if (hip_options_.print_cost_line)
    std::cerr << "COST: " << lm.printed_cost << ", iter: " << lm.iterations << std::endl;
  WritePose(t, R, pose);
  return true;
}

bool MahalanobisDistanceMinimizerHip3DOF::RunLoop(const Options& options, nos_dataset* dataset, const nos_loss& loss,
                                                  Pose* pose) {
  // planar state = top-left 2x2 of the rotation and (x, y); z / roll / pitch pass through
  // untouched (…_analytic_3dof.cc:23-25,104-105)
  double R2[4] = {pose->linear()(0, 0), pose->linear()(0, 1), pose->linear()(1, 0), pose->linear()(1, 1)};
  double t2[2] = {pose->translation()(0), pose->translation()(1)};
  int status = NOS_OK;
  nos_host::LmReport lm;
  if (!TryDeviceLoop(hip_options_, options,
                     [&](const nos_lm_options* o, nos_lm_report* r) { return nos_ndt3_solve(dataset, R2, t2, &loss, o, r); },
                     &lm, &status))
    lm = nos_host::RunLm3(
        SettingsFrom(options, hip_options_, true),
        [&](const double* Rc, const double* tc, double* out10) {
          status = nos_ndt3_accumulate(dataset, Rc, tc, &loss, out10);
          return status == NOS_OK;
        },
        t2, R2);
  FillReport(lm, status, &report_);
  if (!lm.ok) {
    if (status != NOS_OK) ReportFailure("nos_ndt3_accumulate / nos_ndt3_solve", status);
    return false;
  }
  if (hip_options_.print_cost_line)
    std::cerr << "COST: " << lm.printed_cost << ", iter: " << lm.iterations << std::endl;
  pose->translation()(0) = t2[0];
  pose->translation()(1) = t2[1];
  pose->linear()(0, 0) = R2[0];
  pose->linear()(0, 1) = R2[1];
  pose->linear()(1, 0) = R2[2];
  pose->linear()(1, 1) = R2[3];
  return true;
}

}  // namespace mahalanobis_distance_minimizer

namespace reprojection_error_minimizer {

namespace {

void ReprojFieldOffsets(size_t offsets[NOS_REPROJ_PLANES]) {
  static const Correspondence probe{};
  const char* base = reinterpret_cast<const char*>(&probe);
  for (int i = 0; i < 3; ++i)
    offsets[i] = static_cast<size_t>(reinterpret_cast<const char*>(&probe.local_point(i)) - base);
  for (int i = 0; i < 2; ++i)
    offsets[3 + i] = static_cast<size_t>(reinterpret_cast<const char*>(&probe.matched_pixel(i)) - base);
}

constexpr double kMinDepth = 0.03;  // REM/reprojection_error_minimizer_analytic.cc:111

}  // namespace

ReprojectionErrorMinimizerHip::ReprojectionErrorMinimizerHip() {}

ReprojectionErrorMinimizerHip::ReprojectionErrorMinimizerHip(const HipOptions& hip_options)
    : hip_options_(hip_options) {}

ReprojectionErrorMinimizerHip::~ReprojectionErrorMinimizerHip() {}

bool ReprojectionErrorMinimizerHip::Solve(const Options& options, const std::vector<Correspondence>& correspondences,
                                          const CameraIntrinsics& camera_intrinsics, Pose* pose) {
  if (pose == nullptr) return false;
  if (!runtime_) runtime_ = AcquireRuntime(hip_options_.device_ids);
  if (runtime_->status() != NOS_OK) {
    report_.status = runtime_->status();
    return false;
  }
  nos_loss loss;
  if (!DescribeLossFunction(loss_function_.get(), &loss)) {
    std::cerr << "[nos-hip] unsupported LossFunction subclass: only Exponential and Huber have a device "
                 "restatement and there is no CPU fallback"
              << std::endl;
    report_.status = NOS_ERR_UNSUPPORTED;
    return false;
  }
  size_t offsets[NOS_REPROJ_PLANES];
  ReprojFieldOffsets(offsets);
  nos_dataset* dataset = nullptr;
  const size_t count = hip_options_.simd_class ? SimdClassCount(correspondences.size(), 1) : correspondences.size();
  int status = nos_reproj_dataset_create_from_records(runtime_->ctx(), count, correspondences.data(), sizeof(Correspondence),
                                                      offsets, hip_options_.dtype, &dataset);
  if (status == NOS_OK && hip_options_.simd_class) status = nos_dataset_set_simd_class(dataset, 1);
  if (status != NOS_OK) {
    ReportFailure("nos_reproj_dataset_create_from_records", status);
    report_.status = status;
    if (dataset != nullptr) nos_dataset_destroy(dataset);
    return false;
  }
  double intr[4] = {camera_intrinsics.inv_fx, camera_intrinsics.inv_fy, camera_intrinsics.cx, camera_intrinsics.cy};
  if (hip_options_.simd_class) {  // REM/..._analytic_simd.cc:29-30: 1.0f / fx, not the struct's double inv_fx
    intr[0] = double(1.0f / float(camera_intrinsics.fx));
    intr[1] = double(1.0f / float(camera_intrinsics.fy));
  }
  double t[3], R[9];
  ReadPose(*pose, t, R);
  nos_host::LmReport lm;
  if (!TryDeviceLoop(hip_options_, options,
                     [&](const nos_lm_options* o, nos_lm_report* r) {
                       return nos_reproj_solve(dataset, R, t, intr, &loss, kMinDepth, o, r);
                     },
                     &lm, &status))
    lm = nos_host::RunLm6(
        SettingsFrom(options),
        [&](const double* Rc, const double* tc, double* out28) {
          status = nos_reproj_accumulate(dataset, Rc, tc, intr, &loss, kMinDepth, out28);
          return status == NOS_OK;
        },
        t, R);
  nos_dataset_destroy(dataset);
  FillReport(lm, status, &report_);
  if (!lm.ok) {
    if (status != NOS_OK) ReportFailure("nos_reproj_accumulate / nos_reproj_solve", status);
    return false;
  }
  if (hip_options_.print_cost_line)
    std::cerr << "COST: " << lm.printed_cost << ", iter: " << lm.iterations << std::endl;
  WritePose(t, R, pose);
  return true;
}

}  // namespace reprojection_error_minimizer

}  // namespace nonlinear_optimizer
