// nos_host_capi.cpp — C entry points that drive the C++ drop-in solver classes
// (nos_hip_solvers.hpp) exactly the way a C++ caller of the reference would: build a
// std::vector<Correspondence> (array of structures), set a LossFunction object, call Solve().
// They exist so the Python test-suite and bench can exercise the C++ host layer; they add no
// arithmetic of their own.
#include <cstddef>
#include <chrono>
#include <algorithm>
#include <memory>
#include <vector>

#include "nos_hip_solvers.hpp"

using nonlinear_optimizer::ExponentialLossFunction;
using nonlinear_optimizer::HipOptions;
using nonlinear_optimizer::HipSolveReport;
using nonlinear_optimizer::HuberLossFunction;
using nonlinear_optimizer::LossFunction;
using nonlinear_optimizer::Options;
using nonlinear_optimizer::Pose;

namespace {

std::shared_ptr<LossFunction> MakeLoss(int kind, double a, double b) {
  if (kind == NOS_LOSS_EXPONENTIAL) return std::make_shared<ExponentialLossFunction>(a, b);
  if (kind == NOS_LOSS_HUBER) return std::make_shared<HuberLossFunction>(a);
  return nullptr;
}

HipOptions MakeHipOptions(int dtype, const int* device_ids, int n_devices, int print_cost_line) {
  HipOptions h;
  h.dtype = dtype;
  // flag word: bit 0 = print the COST line, bit 1 = host loop, bit 2 = the fp32 classes' semantics (HipOptions::simd_class),
  // bits 8-15 = simd_class_threads (0 = 1)
  h.print_cost_line = (print_cost_line & 1) != 0;
  h.device_loop = (print_cost_line & 2) == 0;
  h.simd_class = (print_cost_line & 4) != 0;
  h.simd_class_threads = std::max(1, (print_cost_line >> 8) & 0xff);
  if (device_ids != nullptr && n_devices > 0) h.device_ids.assign(device_ids, device_ids + n_devices);
  return h;
}

Options MakeOptions(int max_iterations, double gradient_tolerance, double parameter_tolerance) {
  Options o;
  o.max_iterations = max_iterations;
  o.convergence_handle.gradient_tolerance = gradient_tolerance;
  o.convergence_handle.parameter_tolerance = parameter_tolerance;
  return o;
}

void LoadPose(const double t[3], const double R[9], Pose* pose) {
  for (int i = 0; i < 3; ++i) {
    pose->translation()(i) = t[i];
    for (int j = 0; j < 3; ++j) pose->linear()(i, j) = R[3 * i + j];
  }
}

void StorePose(const Pose& pose, double t[3], double R[9]) {
  for (int i = 0; i < 3; ++i) {
    t[i] = pose.translation()(i);
    for (int j = 0; j < 3; ++j) R[3 * i + j] = pose.linear()(i, j);
  }
}

void StoreReport(const HipSolveReport& r, double report[5]) {
  if (report == nullptr) return;
  report[0] = r.iterations;
  report[1] = r.printed_cost;
  report[2] = r.last_cost;
  report[3] = r.final_lambda;
  report[4] = r.status;
}

}  // namespace

extern "C" {

// dof: 6 → MahalanobisDistanceMinimizerHip, 3 → MahalanobisDistanceMinimizerHip3DOF.
// repeat_solves > 1 re-solves from the same initial pose on the prepared dataset (the additive
// Prepare/SolvePrepared API) and returns the last result.  Returns 1 on success, 0 on failure
// (the bool of Solve()).
int nos_host_ndt_solve(int dof, size_t n, const double* const planes[15], int loss_kind, double loss_a,
                       double loss_b, int max_iterations, double gradient_tolerance, double parameter_tolerance,
                       int dtype, const int* device_ids, int n_devices, int print_cost_line, int repeat_solves,
                       double t[3], double R[9], double report[5]) {
  namespace mdm = nonlinear_optimizer::mahalanobis_distance_minimizer;
  try {
    std::vector<mdm::Correspondence> correspondences(n);
    for (size_t i = 0; i < n; ++i) {
      mdm::Correspondence& c = correspondences[i];
      for (int k = 0; k < 3; ++k) {
        c.point(k) = planes[k][i];
        c.ndt.mean(k) = planes[3 + k][i];
        for (int j = 0; j < 3; ++j) c.ndt.sqrt_information(k, j) = planes[6 + 3 * k + j][i];
      }
      c.ndt.is_valid = true;
    }
    const HipOptions hip = MakeHipOptions(dtype, device_ids, n_devices, print_cost_line);
    std::unique_ptr<mdm::MahalanobisDistanceMinimizerHip> solver;
    if (dof == 3)
      solver = std::make_unique<mdm::MahalanobisDistanceMinimizerHip3DOF>(hip);
    else
      solver = std::make_unique<mdm::MahalanobisDistanceMinimizerHip>(hip);
    solver->SetLossFunction(MakeLoss(loss_kind, loss_a, loss_b));
    const Options options = MakeOptions(max_iterations, gradient_tolerance, parameter_tolerance);
    Pose pose = Pose::Identity();
    bool ok = false;
    if (repeat_solves <= 1) {
      LoadPose(t, R, &pose);
      ok = solver->Solve(options, correspondences, &pose);
    } else {
      ok = solver->Prepare(correspondences);
      for (int r = 0; ok && r < repeat_solves; ++r) {
        LoadPose(t, R, &pose);
        ok = solver->SolvePrepared(options, &pose);
      }
      solver->ReleasePrepared();
    }
    StoreReport(solver->report(), report);
    if (ok) StorePose(pose, t, R);
    return ok ? 1 : 0;
  } catch (...) {
    return 0;
  }
}

// Wall time of the drop-in call as a user makes it: one solver object, `repeats` cold Solve() calls on the same
// std::vector<Correspondence> (each: records → device, LM loop, free), and the split Prepare / SolvePrepared.
// ms_out = {min Solve, mean Solve, min Prepare, min SolvePrepared, iterations of the last solve}.
int nos_host_ndt_cold_solve_timing(size_t n, const double* const planes[15], int loss_kind, double loss_a, double loss_b,
                                   int max_iterations, int dtype, int repeats, double ms_out[5]) {
  namespace mdm = nonlinear_optimizer::mahalanobis_distance_minimizer;
  try {
    std::vector<mdm::Correspondence> correspondences(n);
    for (size_t i = 0; i < n; ++i) {
      mdm::Correspondence& c = correspondences[i];
      for (int k = 0; k < 3; ++k) {
        c.point(k) = planes[k][i];
        c.ndt.mean(k) = planes[3 + k][i];
        for (int j = 0; j < 3; ++j) c.ndt.sqrt_information(k, j) = planes[6 + 3 * k + j][i];
      }
      c.ndt.is_valid = true;
    }
    const int dev = 0;
    HipOptions hip = MakeHipOptions(dtype, &dev, 1, 0);
    mdm::MahalanobisDistanceMinimizerHip solver(hip);
    solver.SetLossFunction(MakeLoss(loss_kind, loss_a, loss_b));
    const Options options = MakeOptions(max_iterations, 1e-6, 1e-6);
    using clock = std::chrono::steady_clock;
    auto ms = [](clock::time_point a, clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    double best = 1e300, sum = 0.0, best_prep = 1e300, best_loop = 1e300;
    Pose pose = Pose::Identity();
    if (!solver.Solve(options, correspondences, &pose)) return 0;  // context creation, module load
    for (int r = 0; r < repeats; ++r) {
      pose = Pose::Identity();
      const auto t0 = clock::now();
      if (!solver.Solve(options, correspondences, &pose)) return 0;
      const double d = ms(t0, clock::now());
      best = std::min(best, d);
      sum += d;
    }
    for (int r = 0; r < repeats; ++r) {
      pose = Pose::Identity();
      const auto t0 = clock::now();
      if (!solver.Prepare(correspondences)) return 0;
      const auto t1 = clock::now();
      if (!solver.SolvePrepared(options, &pose)) return 0;
      const auto t2 = clock::now();
      solver.ReleasePrepared();
      best_prep = std::min(best_prep, ms(t0, t1));
      best_loop = std::min(best_loop, ms(t1, t2));
    }
    ms_out[0] = best;
    ms_out[1] = sum / repeats;
    ms_out[2] = best_prep;
    ms_out[3] = best_loop;
    ms_out[4] = solver.report().iterations;
    return 1;
  } catch (...) {
    return 0;
  }
}

// intr = {fx, fy, cx, cy}; inv_fx / inv_fy are derived as the reference's test does
// (REM/tests/simple_optimization_test.cc:43-51).
int nos_host_reproj_solve(size_t n, const double* const planes[5], const double intr[4], int loss_kind, double loss_a,
                          double loss_b, int max_iterations, double gradient_tolerance, double parameter_tolerance,
                          int dtype, const int* device_ids, int n_devices, int print_cost_line, double t[3],
                          double R[9], double report[5]) {
  namespace rem = nonlinear_optimizer::reprojection_error_minimizer;
  try {
    std::vector<rem::Correspondence> correspondences(n);
    for (size_t i = 0; i < n; ++i) {
      for (int k = 0; k < 3; ++k) correspondences[i].local_point(k) = planes[k][i];
      correspondences[i].matched_pixel(0) = planes[3][i];
      correspondences[i].matched_pixel(1) = planes[4][i];
    }
    rem::CameraIntrinsics cam;
    cam.fx = intr[0];
    cam.fy = intr[1];
    cam.cx = intr[2];
    cam.cy = intr[3];
    cam.inv_fx = 1.0 / cam.fx;
    cam.inv_fy = 1.0 / cam.fy;
    cam.width = 640;
    cam.height = 480;
    rem::ReprojectionErrorMinimizerHip solver(MakeHipOptions(dtype, device_ids, n_devices, print_cost_line));
    solver.SetLossFunction(MakeLoss(loss_kind, loss_a, loss_b));
    const Options options = MakeOptions(max_iterations, gradient_tolerance, parameter_tolerance);
    Pose pose = Pose::Identity();
    LoadPose(t, R, &pose);
    const bool ok = solver.Solve(options, correspondences, cam, &pose);
    StoreReport(solver.report(), report);
    if (ok) StorePose(pose, t, R);
    return ok ? 1 : 0;
  } catch (...) {
    return 0;
  }
}

// Loss-descriptor recovery exposed for CPU-only unit tests (no GPU involved).
int nos_host_describe_loss(int kind, double a, double b, int* out_kind, double* out_a, double* out_b) {
  try {
    std::shared_ptr<LossFunction> loss = MakeLoss(kind, a, b);
    nos_loss d;
    const bool ok = nonlinear_optimizer::DescribeLossFunction(loss.get(), &d);
    *out_kind = d.kind;
    *out_a = d.a;
    *out_b = d.b;
    return ok ? 1 : 0;
  } catch (...) {
    return 0;
  }
}

size_t nos_host_sizeof_ndt_correspondence(void) {
  return sizeof(nonlinear_optimizer::mahalanobis_distance_minimizer::Correspondence);
}

}  // extern "C"

// ---- host-logic hooks (no GPU): the LM loop and the damped step with a caller-supplied
// accumulate callback, so the C++ host layer can be unit-tested on a CPU-only box and so a
// one-process-per-GPU caller can put an all-reduce between the kernel and the host step. ----
#include "nos_lm.hpp"

extern "C" {

typedef int (*nos_host_accumulate6_fn)(void* user, const double R[9], const double t[3], double out28[28]);
typedef int (*nos_host_accumulate3_fn)(void* user, const double R2[4], const double t2[2], double out10[10]);

// Returns 1 if the loop ran to a normal end, 0 if the callback or the 6x6 solve failed.
int nos_host_lm6_run(nos_host_accumulate6_fn accumulate, void* user, int max_iterations, double gradient_tolerance,
                     double parameter_tolerance, double t[3], double R[9], double report[5]) {
  nos_host::LmSettings s;
  s.max_iterations = max_iterations;
  s.gradient_tolerance = gradient_tolerance;
  s.parameter_tolerance = parameter_tolerance;
  const nos_host::LmReport lm = nos_host::RunLm6(
      s, [&](const double* Rc, const double* tc, double* out) { return accumulate(user, Rc, tc, out) == 0; }, t, R);
  if (report != nullptr) {
    report[0] = lm.iterations;
    report[1] = lm.printed_cost;
    report[2] = lm.last_cost;
    report[3] = lm.final_lambda;
    report[4] = lm.ok ? 0 : 1;
  }
  return lm.ok ? 1 : 0;
}

int nos_host_lm3_run(nos_host_accumulate3_fn accumulate, void* user, int max_iterations, double gradient_tolerance,
                     double parameter_tolerance, double t2[2], double R2[4], double report[5]) {
  nos_host::LmSettings s;
  s.max_iterations = max_iterations;
  s.gradient_tolerance = gradient_tolerance;
  s.parameter_tolerance = parameter_tolerance;
  const nos_host::LmReport lm = nos_host::RunLm3(
      s, [&](const double* Rc, const double* tc, double* out) { return accumulate(user, Rc, tc, out) == 0; }, t2, R2);
  if (report != nullptr) {
    report[0] = lm.iterations;
    report[1] = lm.printed_cost;
    report[2] = lm.last_cost;
    report[3] = lm.final_lambda;
    report[4] = lm.ok ? 0 : 1;
  }
  return lm.ok ? 1 : 0;
}

// One step of the HOST loop (nos_host::LmAdvance6 / LmAdvance3) on given sums and a given state: the twin of
// nos_debug_lm_step (include/nos.h), same argument layout.
int nos_host_lm_advance(int dof, const double* sums, const double settings[4], double state[22]) {
  if (!sums || !settings || !state || (dof != 6 && dof != 3)) return 1;
  nos_host::LmState st;
  for (int k = 0; k < 9; ++k) st.R[k] = state[k];
  for (int k = 0; k < 3; ++k) st.t[k] = state[9 + k];
  st.q.w = state[12], st.q.x = state[13], st.q.y = state[14], st.q.z = state[15];
  st.lambda = state[16], st.previous_cost = state[17], st.cost = state[18];
  st.iteration = int(state[19]), st.done = int(state[20]), st.ok = int(state[21]);
  nos_host::LmSettings s;
  s.max_iterations = int(settings[0]);
  s.gradient_tolerance = settings[1];
  s.parameter_tolerance = settings[2];
  s.float_schedule = int(settings[3]);
  if (dof == 6)
    nos_host::LmAdvance6(s, sums, &st);
  else
    nos_host::LmAdvance3(s, sums, &st);
  for (int k = 0; k < 9; ++k) state[k] = st.R[k];
  for (int k = 0; k < 3; ++k) state[9 + k] = st.t[k];
  state[12] = st.q.w, state[13] = st.q.x, state[14] = st.q.y, state[15] = st.q.z;
  state[16] = st.lambda, state[17] = st.previous_cost, state[18] = st.cost;
  state[19] = st.iteration, state[20] = st.done, state[21] = st.ok;
  return 0;
}

int nos_host_damped_step6(const double out28[28], double lambda, double step[6]) {
  return nos_host::DampedStep<6>(out28, lambda, step) ? 1 : 0;
}

int nos_host_damped_step3(const double out10[10], double lambda, double step[3]) {
  return nos_host::DampedStep<3>(out10, lambda, step) ? 1 : 0;
}

}  // extern "C"

// ---- fixed-iteration LM loops on an existing device dataset (bench / steady-state use): the
// whole loop — kernel launch, 224-byte readback, damping, 6x6 LDLT, pose update, lambda
// schedule — runs in C++ with no Python between iterations.  Tolerances are 0 so exactly
// `iterations` GN/LM iterations execute. ----
extern "C" {

int nos_host_ndt6_iterate(nos_dataset* dataset, const nos_loss* loss, int iterations, double t[3], double R[9],
                          double report[5]) {
  nos_host::LmSettings s;
  s.max_iterations = iterations;
  s.gradient_tolerance = 0.0;
  s.parameter_tolerance = 0.0;
  int status = NOS_OK;
  const nos_host::LmReport lm = nos_host::RunLm6(
      s,
      [&](const double* Rc, const double* tc, double* out) {
        status = nos_ndt6_accumulate(dataset, Rc, tc, loss, out);
        return status == NOS_OK;
      },
      t, R);
  if (report != nullptr) {
    report[0] = lm.iterations;
    report[1] = lm.printed_cost;
    report[2] = lm.last_cost;
    report[3] = lm.final_lambda;
    report[4] = status;
  }
  return lm.ok ? 1 : 0;
}

// planar loop: t = (t_x, t_y, ·), R row-major 3x3 of which the top-left 2x2 is used and written back
// (MDM/…_analytic_3dof.cc:23-25,104-105)
int nos_host_ndt3_iterate(nos_dataset* dataset, const nos_loss* loss, int iterations, double t[3], double R[9],
                          double report[5]) {
  nos_host::LmSettings s;
  s.max_iterations = iterations;
  s.gradient_tolerance = 0.0;
  s.parameter_tolerance = 0.0;
  int status = NOS_OK;
  double R2[4] = {R[0], R[1], R[3], R[4]}, t2[2] = {t[0], t[1]};
  const nos_host::LmReport lm = nos_host::RunLm3(
      s,
      [&](const double* Rc, const double* tc, double* out) {
        status = nos_ndt3_accumulate(dataset, Rc, tc, loss, out);
        return status == NOS_OK;
      },
      t2, R2);
  R[0] = R2[0];
  R[1] = R2[1];
  R[3] = R2[2];
  R[4] = R2[3];
  t[0] = t2[0];
  t[1] = t2[1];
  if (report != nullptr) {
    report[0] = lm.iterations;
    report[1] = lm.printed_cost;
    report[2] = lm.last_cost;
    report[3] = lm.final_lambda;
    report[4] = status;
  }
  return lm.ok ? 1 : 0;
}

int nos_host_reproj_iterate(nos_dataset* dataset, const double intr[4], const nos_loss* loss, double min_depth,
                            int iterations, double t[3], double R[9], double report[5]) {
  nos_host::LmSettings s;
  s.max_iterations = iterations;
  s.gradient_tolerance = 0.0;
  s.parameter_tolerance = 0.0;
  int status = NOS_OK;
  const nos_host::LmReport lm = nos_host::RunLm6(
      s,
      [&](const double* Rc, const double* tc, double* out) {
        status = nos_reproj_accumulate(dataset, Rc, tc, intr, loss, min_depth, out);
        return status == NOS_OK;
      },
      t, R);
  if (report != nullptr) {
    report[0] = lm.iterations;
    report[1] = lm.printed_cost;
    report[2] = lm.last_cost;
    report[3] = lm.final_lambda;
    report[4] = status;
  }
  return lm.ok ? 1 : 0;
}

}  // extern "C"

// Solve() of the drop-in classes on a dataset that already lives on the device (matcher output).
extern "C" int nos_host_ndt_solve_dataset(int dof, nos_dataset* dataset, int loss_kind, double loss_a, double loss_b,
                                          int max_iterations, double gradient_tolerance, double parameter_tolerance,
                                          int print_cost_line, double t[3], double R[9], double report[5]) {
  namespace mdm = nonlinear_optimizer::mahalanobis_distance_minimizer;
  try {
    HipOptions hip;
    hip.print_cost_line = (print_cost_line & 1) != 0;
    hip.device_loop = (print_cost_line & 2) == 0;
    hip.simd_class = (print_cost_line & 4) != 0;  // host loop only: the device loop reads the flag off the dataset
    std::unique_ptr<mdm::MahalanobisDistanceMinimizerHip> solver;
    if (dof == 3)
      solver = std::make_unique<mdm::MahalanobisDistanceMinimizerHip3DOF>(hip);
    else
      solver = std::make_unique<mdm::MahalanobisDistanceMinimizerHip>(hip);
    solver->SetLossFunction(MakeLoss(loss_kind, loss_a, loss_b));
    const Options options = MakeOptions(max_iterations, gradient_tolerance, parameter_tolerance);
    Pose pose = Pose::Identity();
    LoadPose(t, R, &pose);
    const bool ok = solver->SolveDataset(options, dataset, &pose);
    StoreReport(solver->report(), report);
    if (ok) StorePose(pose, t, R);
    return ok ? 1 : 0;
  } catch (...) {
    return 0;
  }
}
