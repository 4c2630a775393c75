// nos_hip_solvers.hpp — the drop-in solver classes: same abstract bases, same Solve()
// signatures, same stderr line and the same LM loop as the reference's analytic classes,
// with the per-iteration normal-equation assembly running on MI355X through include/nos.h.
//
//   MahalanobisDistanceMinimizerHip      ↔ MahalanobisDistanceMinimizerAnalytic[SIMD]
//        (MDM/mahalanobis_distance_minimizer_analytic.cc:54-157, ..._analytic_simd.cc:16-111)
//   MahalanobisDistanceMinimizerHip3DOF  ↔ MahalanobisDistanceMinimizerAnalytic3DOF[SIMD]
//        (MDM/mahalanobis_distance_minimizer_analytic_3dof.cc:14-108)
//   ReprojectionErrorMinimizerHip        ↔ ReprojectionErrorMinimizerAnalytic[SIMD]
//        (REM/reprojection_error_minimizer_analytic.cc:12-105)
//
// Inside the reference tree compile with -DNOS_IN_REFERENCE_TREE (uses the reference's own
// headers and Eigen types); stand-alone it uses nos_reference_api.hpp.
#ifndef NOS_HIP_SOLVERS_HPP_
#define NOS_HIP_SOLVERS_HPP_

#include <memory>
#include <vector>

#ifdef NOS_IN_REFERENCE_TREE
#include "nonlinear_optimizer/mahalanobis_distance_minimizer/mahalanobis_distance_minimizer.h"
#include "nonlinear_optimizer/reprojection_error_minimizer/reprojection_error_minimizer.h"
#else
#include "nos_reference_api.hpp"
#endif

#include "../../../include/nos.h"

namespace nonlinear_optimizer {

// Additive GPU knobs; options.h stays untouched.
struct HipOptions {
  std::vector<int> device_ids{0};  // one shard per entry (contiguous ranges of correspondences)
  int dtype{NOS_F64};              // NOS_F64 (scalar-class arithmetic) or NOS_F32 (SIMD-class arithmetic)
  bool print_cost_line{true};      // the reference's "COST: <previous_cost>, iter: <n>" stderr line
  bool device_loop{true};          // run the whole LM loop device-resident (nos_*_solve); false = host loop around
                                   // nos_*_accumulate.  Multi-device contexts always use the host loop.
  // Semantics of the reference's fp32 ("SIMD") classes instead of the scalar classes' (nos_dataset_set_simd_class in
  // include/nos.h), to be combined with dtype = NOS_F32 for the classes' arithmetic (the reference's own "SIMD … Double"
  // variants are this with NOS_F64, results/maha_amd64.txt:16-21): only the first T * floor(floor(N/8)/T) * 8
  // correspondences are used (T = simd_class_threads = the thread count of the executor the reference class would be
  // given, 1 = none: MDM/..._analytic_simd.cc:46-69), float lambda / previous_cost (NDT), depth > 0 mask on the weight
  // only and a float 1/fx (reprojection).  The lane arithmetic of the un-vendored simd_helper is not reproduced bit for bit.
  bool simd_class{false};
  int simd_class_threads{1};
};

// Correspondences the reference's fp32 classes actually use out of n (see HipOptions::simd_class).
inline size_t SimdClassCount(size_t n, int threads) {
  const size_t t = threads > 1 ? size_t(threads) : 1;
  return t * ((n / 8) / t) * 8;
}

// What the last Solve() did (additive; the reference exposes only the stderr line).
struct HipSolveReport {
  int iterations{0};
  double printed_cost{0.0};
  double last_cost{0.0};
  double final_lambda{0.0};
  int status{0};  // nos_status of the failing call, 0 if none
};

// Shared context (stream + workspaces per device), created on first use.
class HipRuntime {
 public:
  explicit HipRuntime(const std::vector<int>& device_ids);
  ~HipRuntime();
  HipRuntime(const HipRuntime&) = delete;
  HipRuntime& operator=(const HipRuntime&) = delete;
  nos_ctx* ctx() const { return ctx_; }
  int status() const { return status_; }

 private:
  nos_ctx* ctx_{nullptr};
  int status_{0};
};

// Solver objects share one cached context per device list for the life of the process (creating one costs more than
// a small Solve()); this drops the cache — e.g. before unloading the library.  Solvers alive at that time keep theirs.
void ReleaseHipRuntimes();

// Recovers the POD loss descriptor the kernels need from a host LossFunction object whose
// parameters are private (NO/loss_function.h:43-46,74-76): dynamic_cast to the two known
// classes, then probe Evaluate().  nullptr → NOS_LOSS_NONE.  Unknown subclass → false.
bool DescribeLossFunction(LossFunction* loss_function, nos_loss* out);

namespace mahalanobis_distance_minimizer {

class MahalanobisDistanceMinimizerHip : public MahalanobisDistanceMinimizer {
 public:
  MahalanobisDistanceMinimizerHip();
  explicit MahalanobisDistanceMinimizerHip(const HipOptions& hip_options);
  ~MahalanobisDistanceMinimizerHip();

  bool Solve(const Options& options, const std::vector<Correspondence>& correspondences, Pose* pose) final;

  // Additive API: upload once, Solve() many times (cold Solve is ingestion-bound).
  bool Prepare(const std::vector<Correspondence>& correspondences);
  bool SolvePrepared(const Options& options, Pose* pose);
  void ReleasePrepared();
  // Additive API for GPU-resident pipelines: solve on a dataset that already lives on the device
  // (e.g. the output of nos_ndt_match); the dataset stays owned by the caller.
  bool SolveDataset(const Options& options, nos_dataset* dataset, Pose* pose);

  const HipSolveReport& report() const { return report_; }

 protected:
  virtual bool RunLoop(const Options& options, nos_dataset* dataset, const nos_loss& loss, Pose* pose);
  // T of the SIMD class's tail drop T * floor(floor(N/8)/T) * 8: the 6-DoF class splits over its executor's threads
  // (MDM/..._analytic_simd.cc:46-69); the 3-DoF SIMD class has no executor and always uses floor(N/8)*8
  // (MDM/..._analytic_3dof_simd.cc:83-86).
  virtual int SimdClassThreads() const { return hip_options_.simd_class_threads; }
  HipOptions hip_options_;
  std::shared_ptr<HipRuntime> runtime_;
  nos_dataset* prepared_{nullptr};
  HipSolveReport report_;
};

class MahalanobisDistanceMinimizerHip3DOF : public MahalanobisDistanceMinimizerHip {
 public:
  MahalanobisDistanceMinimizerHip3DOF() {}
  explicit MahalanobisDistanceMinimizerHip3DOF(const HipOptions& hip_options)
      : MahalanobisDistanceMinimizerHip(hip_options) {}

 protected:
  bool RunLoop(const Options& options, nos_dataset* dataset, const nos_loss& loss, Pose* pose) final;
  int SimdClassThreads() const final { return 1; }
};

}  // namespace mahalanobis_distance_minimizer

namespace reprojection_error_minimizer {

class ReprojectionErrorMinimizerHip : public ReprojectionErrorMinimizer {
 public:
  ReprojectionErrorMinimizerHip();
  explicit ReprojectionErrorMinimizerHip(const HipOptions& hip_options);
  ~ReprojectionErrorMinimizerHip();

  bool Solve(const Options& options, const std::vector<Correspondence>& correspondences,
             const CameraIntrinsics& camera_intrinsics, Pose* pose) final;

  const HipSolveReport& report() const { return report_; }

 private:
  HipOptions hip_options_;
  std::shared_ptr<HipRuntime> runtime_;
  HipSolveReport report_;
};

}  // namespace reprojection_error_minimizer

}  // namespace nonlinear_optimizer

#endif  // NOS_HIP_SOLVERS_HPP_
