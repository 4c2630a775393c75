// nos_lm.hpp — host side of one Levenberg-Marquardt solve around the GPU assembly.
//
// The outer loop of the reference is kept as it is (north star: "loss_function.h, options.h
// and the outer LM loop are unchanged"): multiplicative damping H_kk *= 1 + λ, a tiny dense
// solve on the host, right-multiplicative pose update, convergence tested after the update,
// λ schedule ×2 / ×0.6 clamped to the hard-coded [1e-6, 1e-2].  Restated from
//   NO/mahalanobis_distance_minimizer/mahalanobis_distance_minimizer_analytic_simd.cc:30-108
//   NO/mahalanobis_distance_minimizer/mahalanobis_distance_minimizer_analytic_3dof.cc:17-108
//   NO/reprojection_error_minimizer/reprojection_error_minimizer_analytic.cc:15-105
// Only plain doubles appear here so the same code serves an Eigen build of the reference
// tree and the Eigen-free build of this repository.
#ifndef NOS_LM_HPP_
#define NOS_LM_HPP_

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <limits>

// The step of the loop (LmAdvance6 / LmAdvance3 and what they call) also compiles as HIP device code: the
// device-resident loop of libnos_hip.so (nos_*_solve) runs the very same function in the last workgroup of each
// launch, so host loop and device loop cannot drift apart.
#if defined(__HIPCC__)
#define NOS_HD __host__ __device__
#else
#define NOS_HD
#endif

#if defined(NOS_USE_EIGEN) && __has_include(<Eigen/Dense>)
#include <Eigen/Dense>
#define NOS_HAVE_EIGEN 1
#endif

namespace nos_host {

struct Quat {  // w, x, y, z
  double w = 1.0, x = 0.0, y = 0.0, z = 0.0;
};

// Rotation matrix (row-major) → unit quaternion, the branch structure Eigen's
// Quaternion(Matrix3) uses, so the starting orientation equals the reference's
// `Orientation optimized_orientation(initial_pose.rotation())`.
NOS_HD inline Quat QuatFromMatrix(const double R[9]) {
  Quat q;
  const double tr = R[0] + R[4] + R[8];
  if (tr > 0.0) {
    const double s = sqrt(tr + 1.0);
    const double f = 0.5 / s;
    q.w = 0.5 * s;
    q.x = (R[7] - R[5]) * f;
    q.y = (R[2] - R[6]) * f;
    q.z = (R[3] - R[1]) * f;
    return q;
  }
  int i = (R[4] > R[0]) ? 1 : 0;
  if (R[8] > R[4 * i]) i = 2;
  const int j = (i + 1) % 3, k = (j + 1) % 3;
  const double s = sqrt(R[4 * i] - R[4 * j] - R[4 * k] + 1.0);
  const double f = 0.5 / s;
  double v[3];
  v[i] = 0.5 * s;
  v[j] = (R[3 * j + i] + R[3 * i + j]) * f;
  v[k] = (R[3 * k + i] + R[3 * i + k]) * f;
  q.w = (R[3 * k + j] - R[3 * j + k]) * f;
  q.x = v[0];
  q.y = v[1];
  q.z = v[2];
  return q;
}

NOS_HD inline void QuatToMatrix(const Quat& q, double R[9]) {
  const double tx = 2.0 * q.x, ty = 2.0 * q.y, tz = 2.0 * q.z;
  const double twx = tx * q.w, twy = ty * q.w, twz = tz * q.w;
  const double txx = tx * q.x, txy = ty * q.x, txz = tz * q.x;
  const double tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
  R[0] = 1.0 - (tyy + tzz);
  R[1] = txy - twz;
  R[2] = txz + twy;
  R[3] = txy + twz;
  R[4] = 1.0 - (txx + tzz);
  R[5] = tyz - twx;
  R[6] = txz - twy;
  R[7] = tyz + twx;
  R[8] = 1.0 - (txx + tyy);
}

// so(3) → quaternion, MahalanobisDistanceMinimizer::ComputeQuaternion
// (NO/mahalanobis_distance_minimizer/mahalanobis_distance_minimizer.cc:20-33).
NOS_HD inline Quat ExpQuat(const double w[3]) {
  Quat q;
  const double theta = sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
  if (theta < 1e-6) {
    q.w = 1.0;
    q.x = 0.5 * w[0];
    q.y = 0.5 * w[1];
    q.z = 0.5 * w[2];
  } else {
    const double half = 0.5 * theta;
    double sn, cs;
#if defined(__HIP_DEVICE_COMPILE__)
    sincos(half, &sn, &cs);  // one argument reduction for both
#else
    sn = sin(half);
    cs = cos(half);
#endif
    const double k = sn / theta;
    q.w = cs;
    q.x = k * w[0];
    q.y = k * w[1];
    q.z = k * w[2];
  }
  return q;
}

// q ← normalize(q ⊗ dq)
NOS_HD inline void RightMultiplyNormalize(Quat* q, const Quat& d) {
  const Quat a = *q;
  Quat r;
  r.w = a.w * d.w - a.x * d.x - a.y * d.y - a.z * d.z;
  r.x = a.w * d.x + a.x * d.w + a.y * d.z - a.z * d.y;
  r.y = a.w * d.y + a.y * d.w + a.z * d.x - a.x * d.z;
  r.z = a.w * d.z + a.z * d.w + a.x * d.y - a.y * d.x;
  const double inv_n = 1.0 / sqrt(r.x * r.x + r.y * r.y + r.z * r.z + r.w * r.w);  // one division (see SolveLdlt)
  q->w = r.w * inv_n;
  q->x = r.x * inv_n;
  q->y = r.y * inv_n;
  q->z = r.z * inv_n;
}

// 1 / d for a positive finite pivot.  Host: the IEEE divide.  Device: the hardware seed and two Newton steps (full fp64
// accuracy, within an ulp of the divide) — the LM step runs on ONE GPU lane, where the divide's twelve dependent
// instructions per pivot are a measurable part of an LM iteration of the small problem sizes.
NOS_HD inline double Reciprocal(double d) {
#if defined(__HIP_DEVICE_COMPILE__)
  double y = __builtin_amdgcn_rcp(d);
  double e = __builtin_fma(-d, y, 1.0);
  y = __builtin_fma(y, e, y);
  e = __builtin_fma(-d, y, 1.0);
  return __builtin_fma(y, e, y);
#else
  return 1.0 / d;
#endif
}

// Solve (A) x = b for a symmetric positive definite A (N ≤ 6) by LDLᵀ.  After the
// multiplicative damping the normal matrix is SPD, so no pivoting is required; a
// non-positive pivot reports failure instead of producing garbage.
template <int N>
NOS_HD inline bool SolveLdlt(const double* A, const double* b, double* x) {
#if defined(NOS_HAVE_EIGEN) && !defined(__HIP_DEVICE_COMPILE__)
  using Mat = Eigen::Matrix<double, N, N, Eigen::RowMajor>;
  using Vec = Eigen::Matrix<double, N, 1>;
  Eigen::Map<const Mat> Am(A);
  Eigen::Map<const Vec> bm(b);
  Eigen::Map<Vec> xm(x);
  xm = Am.ldlt().solve(bm);
  return xm.allFinite();
#else
  // Right-looking LDLᵀ on the lower triangle with reciprocal pivots: the trailing updates of one column are independent
  // of each other (the step runs in ONE GPU lane inside the device-resident loop, where a chain of dependent fp64
  // operations — and above all of divisions, ≈ 45 ns each — is what it costs: 6 divisions here instead of 21).
  double a[N][N];
  for (int i = 0; i < N; ++i)
    for (int k = 0; k <= i; ++k) a[i][k] = A[N * i + k];
  double inv[N];
  for (int j = 0; j < N; ++j) {
    const double d = a[j][j];
    if (!(d > 0.0) || !(d <= DBL_MAX)) return false;
    inv[j] = Reciprocal(d);
    double l[N];
    for (int i = j + 1; i < N; ++i) l[i] = a[i][j] * inv[j];
    for (int i = j + 1; i < N; ++i)
      for (int k = j + 1; k <= i; ++k) a[i][k] -= l[i] * a[k][j];
    for (int i = j + 1; i < N; ++i) a[i][j] = l[i];  // column j of L
  }
  double y[N];
  for (int i = 0; i < N; ++i) y[i] = b[i];
  for (int k = 0; k < N; ++k)
    for (int i = k + 1; i < N; ++i) y[i] -= a[i][k] * y[k];
  for (int i = 0; i < N; ++i) x[i] = y[i] * inv[i];
  for (int k = N - 1; k > 0; --k)
    for (int i = 0; i < k; ++i) x[i] -= a[k][i] * x[k];
  return true;
#endif
}

// out = {upper triangle row-major | gradient | cost}  →  damped step  δ = -(H∘(1+λ on diag))⁻¹ g
template <int N>
NOS_HD inline bool DampedStep(const double* out, double lambda, double* step) {
  double H[N * N], mg[N];
  int k = 0;
  for (int r = 0; r < N; ++r)
    for (int c = r; c < N; ++c) {
      H[N * r + c] = out[k];
      H[N * c + r] = out[k];
      ++k;
    }
  for (int r = 0; r < N; ++r) H[N * r + r] *= 1.0 + lambda;
  for (int r = 0; r < N; ++r) mg[r] = -out[k + r];
  return SolveLdlt<N>(H, mg, step);
}

template <int N>
NOS_HD inline double Norm(const double* v) {
  double s = 0.0;
  for (int i = 0; i < N; ++i) s += v[i] * v[i];
  return sqrt(s);
}

struct LmSettings {
  int max_iterations = 40;
  double gradient_tolerance = 1e-6;
  double parameter_tolerance = 1e-6;
  // 1: λ and previous_cost live in `float`, as in the reference's fp32 NDT classes (MDM/…_analytic_simd.cc:38-39,
  // …_3dof_simd.cc:73-74: `float lambda = 0.001f; float previous_cost = numeric_limits<float>::max()`); the cost itself
  // stays double there too.  Selected by the "SIMD class" semantics of a dataset (nos_dataset_set_simd_class).
  int float_schedule = 0;
};

struct LmReport {
  int iterations = 0;            // loop index at exit: the "iter:" of the reference's stderr line
  double printed_cost = 0.0;     // previous_cost at exit: the "COST:" of that line
  double last_cost = 0.0;
  double final_lambda = 0.0;
  bool ok = true;                // false if an accumulate call or the 6x6 solve failed
};

constexpr double kMinLambda = 1e-6;  // constexpr in the reference too (…_analytic_simd.cc:30-31),
constexpr double kMaxLambda = 1e-2;  // not Options::optimization_handle

// Everything the loop carries from one iteration to the next.  6-DoF: orientation q (R is its matrix, refreshed
// after every update) and translation t.  Planar: R[0..3] is the 2x2 rotation, t[0..1] the translation.
struct LmState {
  double R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  double t[3] = {0, 0, 0};
  Quat q;
  double lambda = 1e-3;
  double previous_cost = DBL_MAX;  // std::numeric_limits<double>::max() in the reference
  double cost = 0.0;
  int iteration = 0;  // loop index; at exit the "iter:" of the reference's stderr line
  int done = 0;       // 1 once a convergence test fired, the iteration budget ran out or a solve failed
  int ok = 1;         // 0 if the damped solve failed
};

NOS_HD inline void LmInitSchedule(LmState* st, int float_schedule) {
  if (float_schedule) {
    st->lambda = double(0.001f);
    st->previous_cost = double(FLT_MAX);
  }
}

NOS_HD inline void LmInit6(LmState* st, const double R[9], const double t[3], int max_iterations, int float_schedule = 0) {
  *st = LmState();
  LmInitSchedule(st, float_schedule);
  st->q = QuatFromMatrix(R);
  QuatToMatrix(st->q, st->R);
  for (int k = 0; k < 3; ++k) st->t[k] = t[k];
  st->done = max_iterations <= 0 ? 1 : 0;
}

NOS_HD inline void LmInit3(LmState* st, const double R2[4], const double t2[2], int max_iterations, int float_schedule = 0) {
  *st = LmState();
  LmInitSchedule(st, float_schedule);
  for (int k = 0; k < 4; ++k) st->R[k] = R2[k];
  st->t[0] = t2[0];
  st->t[1] = t2[1];
  st->done = max_iterations <= 0 ? 1 : 0;
}

NOS_HD inline double NextLambda(double lambda, double cost, double previous_cost) {
  const double l = lambda * (cost > previous_cost ? 2.0 : 0.6);
  return l < kMinLambda ? kMinLambda : (l > kMaxLambda ? kMaxLambda : l);
}
// The same with λ in float: `lambda *= (cost > previous_cost ? 2.0 : 0.6)` (product in double, stored to float), then
// std::clamp against float bounds (MDM/…_analytic_simd.cc:30-31,99-100).  `lambda` / `previous_cost` hold float values.
NOS_HD inline double NextLambdaFloat(double lambda, double cost, double previous_cost) {
  const float l = float(double(float(lambda)) * (cost > previous_cost ? 2.0 : 0.6));
  const float lo = 1e-6f, hi = 1e-2f;
  return double(l < lo ? lo : (l > hi ? hi : l));
}
// λ schedule + cost bookkeeping at the end of an iteration that did not converge
NOS_HD inline void LmSchedule(const LmSettings& s, LmState* st) {
  if (s.float_schedule) {
    st->lambda = NextLambdaFloat(st->lambda, st->cost, st->previous_cost);
    st->previous_cost = double(float(st->cost));
  } else {
    st->lambda = NextLambda(st->lambda, st->cost, st->previous_cost);
    st->previous_cost = st->cost;
  }
}

// One pass of the loop body after ComputeCostAndDerivatives: damped solve, pose update, the two convergence
// tests (after the update, as in the reference), λ schedule.  `out` = {21 H upper | 6 g | cost} evaluated at
// st->R, st->t.
NOS_HD inline void LmAdvance6(const LmSettings& s, const double out[28], LmState* st) {
  double step[6];
  st->cost = out[27];
  if (!DampedStep<6>(out, st->lambda, step)) {
    st->ok = 0;
    st->done = 1;
    return;
  }
  st->t[0] += step[0];
  st->t[1] += step[1];
  st->t[2] += step[2];
  RightMultiplyNormalize(&st->q, ExpQuat(step + 3));
  QuatToMatrix(st->q, st->R);
  if (Norm<6>(step) < s.parameter_tolerance || Norm<6>(out + 21) < s.gradient_tolerance) {
    st->done = 1;
    return;
  }
  LmSchedule(s, st);
  if (++st->iteration >= s.max_iterations) st->done = 1;
}

// Planar form; `out` = {6 H upper | 3 g | cost}.
NOS_HD inline void LmAdvance3(const LmSettings& s, const double out[10], LmState* st) {
  double step[3];
  st->cost = out[9];
  if (!DampedStep<3>(out, st->lambda, step)) {
    st->ok = 0;
    st->done = 1;
    return;
  }
  st->t[0] += step[0];
  st->t[1] += step[1];
  double c, sn;
#if defined(__HIP_DEVICE_COMPILE__)
  sincos(step[2], &sn, &c);
#else
  c = cos(step[2]);
  sn = sin(step[2]);
#endif
  const double a = st->R[0], b = st->R[1], d = st->R[2], e = st->R[3];
  st->R[0] = a * c + b * sn;  // linear ← linear · Rot2(δθ)   (Isometry2d::rotate)
  st->R[1] = b * c - a * sn;
  st->R[2] = d * c + e * sn;
  st->R[3] = e * c - d * sn;
  if (Norm<3>(step) < s.parameter_tolerance || Norm<3>(out + 6) < s.gradient_tolerance) {
    st->done = 1;
    return;
  }
  LmSchedule(s, st);
  if (++st->iteration >= s.max_iterations) st->done = 1;
}

inline LmReport ReportOf(const LmState& st) {
  LmReport rep;
  rep.iterations = st.iteration;
  rep.printed_cost = st.previous_cost;
  rep.last_cost = st.cost;
  rep.final_lambda = st.lambda;
  rep.ok = st.ok != 0;
  return rep;
}

// 6-DoF loop.  `accumulate(R, t, out28)` returns false on failure.
template <typename Accumulate>
inline LmReport RunLm6(const LmSettings& s, Accumulate&& accumulate, double t[3], double R[9]) {
  LmState st;
  LmInit6(&st, R, t, s.max_iterations, s.float_schedule);
  while (!st.done) {
    double out[28];
    if (!accumulate(st.R, st.t, out)) {
      st.ok = 0;
      break;
    }
    LmAdvance6(s, out, &st);
  }
  for (int k = 0; k < 9; ++k) R[k] = st.R[k];
  for (int k = 0; k < 3; ++k) t[k] = st.t[k];
  return ReportOf(st);
}

// Planar loop: state is the 2x2 rotation (row-major) and (x, y).  `accumulate(R2, t2, out10)`.
template <typename Accumulate>
inline LmReport RunLm3(const LmSettings& s, Accumulate&& accumulate, double t2[2], double R2[4]) {
  LmState st;
  LmInit3(&st, R2, t2, s.max_iterations, s.float_schedule);
  while (!st.done) {
    double out[10];
    if (!accumulate(st.R, st.t, out)) {
      st.ok = 0;
      break;
    }
    LmAdvance3(s, out, &st);
  }
  for (int k = 0; k < 4; ++k) R2[k] = st.R[k];
  t2[0] = st.t[0];
  t2[1] = st.t[1];
  return ReportOf(st);
}

}  // namespace nos_host

#endif  // NOS_LM_HPP_
