// mapexact_kernels.hpp — NDT map construction that reproduces the reference's harness BIT FOR BIT
// (the NOS_MAP_REFERENCE_EXACT mode of nos_ndt_map_build).
//
// Why it exists: the reference's captured NDT runs (results/maha_*.txt) depend on the rounding noise of
// UpdateNdtMap (nonlinear_optimizer/mahalanobis_distance_minimizer/tests/simple_optimization_test.cc:236-281) —
// a full 1 m x 1 m floor / wall patch has two equal in-plane variances, Eigen's eigenvectors of it sit at a
// noise-determined angle and `sqrt_information = D^-1/2 * V` (:275-276) turns that angle into the direction of
// the voxel's strong constraint.  "Results identical to the reference's" for the map → match →
// Solve pipeline therefore needs
//   1. count / sum / moment accumulated per voxel SEQUENTIALLY IN POINT ORDER (:240-248; moment starts at
//      identity, MDM/types.h:14) with the multiply-adds fused exactly where the reference's -O2 -march=native
//      binary fuses them,
//   2. mean, covariance (:256-259) with the same contraction,
//   3. Eigen::SelfAdjointEigenSolver<Matrix3d>::compute (:262): scale by the largest coefficient, the 3x3
//      tridiagonalization_inplace special case, computeFromTridiagonal_impl (deflation test, implicit symmetric QR
//      step with Wilkinson shift, JacobiRotation::makeGivens, numext::hypot), ascending selection sort —
//      Eigen is an un-vendored, un-pinned dependency of the reference (find_package(Eigen3 REQUIRED),
//      MDM/CMakeLists.txt:6); restated from its published algorithm, 3.3.x and 3.4.0 semantics,
//   4. eigenvalue floor and sqrt_information = D^-1/2 * V (:268-276).
//
// GPU form: the stable radix sort by voxel cell keeps the points of a voxel in point order; one wave per voxel
// walks them through LDS, and the 12 running sums of a voxel (3 of `sum`, 9 of `moment`) are 12 independent
// sequential chains — one lane each — so the additions happen in exactly the reference's order, twelve at a time.
// A second kernel (one lane per voxel) does steps 2-4.
//
// THIS FILE MUST BE COMPILED WITH -ffp-contract=off (csrc/Makefile does): every multiply-add that is fused is an
// explicit fma(), everything else must stay a separately rounded multiply and add.  The pragma below says the same
// to the compiler for the functions of this file.
#pragma once

#include <hip/hip_runtime.h>
#include <float.h>
#include <stdint.h>

#pragma clang fp contract(off)

namespace nos {
namespace mapexact {

// multiply-add sites, one bit each (the same encoding as the CPU restatement the tests compare against):
// bits 2..5 the solver's statements, bits 8..16 `moment(e) += p p^T` per row-major element e,
// bits 17..25 `moment(e) / count - mean mean^T` per element.
constexpr int kSiteTridiag = 4, kSiteSqrt1p = 8, kSiteQr = 16, kSiteQ = 32, kMomentShift = 8, kCovShift = 17;
// The setting that reproduces the reference's captured x86-64 runs (tests/golden/make_ndt_scene_golden.py): solver
// statements contracted; `moment += p p^T` contracted only in the elements where Eigen's packet-of-two evaluation
// through a temporary lets GCC forward the product into the add (row-major 0, 2, 3, 5, 8); the lazy outer product
// of the covariance contracted everywhere.
constexpr int kReferenceFmaMask = (4 | 8 | 16 | 32) | ((1 | 4 | 8 | 32 | 256) << kMomentShift) | (0x1ff << kCovShift);
constexpr int kReferenceEigenVersion = 34;

__device__ __forceinline__ double madd(int fused, double a, double b, double c) {
  const double f = __builtin_fma(a, b, c);
  const double p = a * b;
  const double u = p + c;
  return fused ? f : u;
}

struct Givens {
  double c, s;
};

// Eigen/src/Jacobi/Jacobi.h, JacobiRotation<double>::makeGivens(p, q), real case.
__device__ inline Givens make_givens(double p, double q, int fm) {
  Givens g;
  if (q == 0.0) {
    g.c = p < 0.0 ? -1.0 : 1.0;
    g.s = 0.0;
  } else if (p == 0.0) {
    g.c = 0.0;
    g.s = q < 0.0 ? 1.0 : -1.0;
  } else if (fabs(p) > fabs(q)) {
    const double t = q / p;
    double u = sqrt(madd(fm & kSiteSqrt1p, t, t, 1.0));
    if (p < 0.0) u = -u;
    g.c = 1.0 / u;
    g.s = -t * g.c;
  } else {
    const double t = p / q;
    double u = sqrt(madd(fm & kSiteSqrt1p, t, t, 1.0));
    if (q < 0.0) u = -u;
    g.s = -1.0 / u;
    g.c = -t * g.s;
  }
  return g;
}

// Eigen/src/Core/MathFunctions.h, numext::hypot (Eigen's own scaled form).
__device__ inline double eigen_hypot(double x, double y, int fm) {
  const double ax = fabs(x), ay = fabs(y);
  double p, qp;
  if (ax > ay) {
    p = ax;
    qp = ay / p;
  } else {
    p = ay;
    qp = ax / p;
  }
  if (p == 0.0) return 0.0;
  return p * sqrt(madd(fm & kSiteSqrt1p, qp, qp, 1.0));
}

// internal::tridiagonal_qr_step (SelfAdjointEigenSolver.h); Q column-major 3x3.
__device__ inline void qr_step(double* diag, double* sub, int start, int end, double* Q, int version, int fm) {
  const double td = (diag[end - 1] - diag[end]) * 0.5;
  const double e = sub[end - 1];
  double mu = diag[end];
  if (version == 34) {
    if (td == 0.0) {
      mu -= fabs(e);
    } else if (e != 0.0) {
      const double e2 = e * e;
      const double h = eigen_hypot(td, e, fm);
      if (e2 == 0.0)
        mu -= e / ((td + (td > 0.0 ? h : -h)) / e);
      else
        mu -= e2 / (td + (td > 0.0 ? h : -h));
    }
  } else {
    if (td == 0.0) {
      mu -= fabs(e);
    } else {
      const double e2 = e * e;
      const double h = eigen_hypot(td, e, fm);
      if (e2 == 0.0)
        mu -= (e / (td + (td > 0.0 ? 1.0 : -1.0))) * (e / h);
      else
        mu -= e2 / (td + (td > 0.0 ? h : -h));
    }
  }
  double x = diag[start] - mu;
  double z = sub[start];
  const int fq = fm & kSiteQr;
  const int fqq = fm & kSiteQ;
  for (int k = start; k < end && (version != 34 || z != 0.0); ++k) {
    const Givens r = make_givens(x, z, fm);
    const double c = r.c, s = r.s;
    // T = G' T G; contraction shapes as g++ 11 -O2 -mfma chooses them for Eigen's statements: the products
    // with c are shared plain multiplies, the products with s are the fused ones
    const double c_sub = c * sub[k];
    const double sdk = madd(fq, s, diag[k], c_sub);
    const double dkp1 = madd(fq, s, sub[k], c * diag[k + 1]);
    const double in1 = madd(fq, -s, sub[k], c * diag[k]);
    const double in2 = madd(fq, -s, diag[k + 1], c_sub);
    diag[k] = madd(fq, c, in1, -(s * in2));
    diag[k + 1] = madd(fq, s, sdk, c * dkp1);
    sub[k] = madd(fq, c, sdk, -(s * dkp1));
    if (k > start) sub[k - 1] = madd(fq, c, sub[k - 1], -(s * z));
    x = sub[k];
    if (k < end - 1) {
      z = -s * sub[k + 1];
      sub[k + 1] = c * sub[k + 1];
    }
    // Q = Q * G
    for (int i = 0; i < 3; ++i) {
      const double xi = Q[k * 3 + i], yi = Q[(k + 1) * 3 + i];
      Q[k * 3 + i] = madd(fqq, c, xi, -(s * yi));
      Q[(k + 1) * 3 + i] = madd(fqq, s, xi, c * yi);
    }
  }
}

// SelfAdjointEigenSolver<Matrix3d>::compute(A, ComputeEigenvectors).  A row-major symmetric (the lower triangle is
// read, as in Eigen).  evals ascending, evecs COLUMN-major (evecs[3 k + i] = component i of eigenvector k).
// Returns 0 = Success, 1 = NoConvergence.
__device__ inline int eigen_selfadjoint3(const double* A, int version, int fm, double* evals, double* evecs) {
  double m[3][3];
  double diag[3], sub[2];
  double Q[9];
  double scale = 0.0;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      m[i][j] = j <= i ? A[3 * i + j] : 0.0;
      if (fabs(m[i][j]) > scale) scale = fabs(m[i][j]);
    }
  if (scale == 0.0) scale = 1.0;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j <= i; ++j) m[i][j] /= scale;
  {  // tridiagonalization_inplace_selector<MatrixType, 3, false>::run
    const int ft = fm & kSiteTridiag;
    diag[0] = m[0][0];
    const double v1norm2 = m[2][0] * m[2][0];
    for (int i = 0; i < 9; ++i) Q[i] = 0.0;
    if (v1norm2 <= DBL_MIN) {
      diag[1] = m[1][1];
      diag[2] = m[2][2];
      sub[0] = m[1][0];
      sub[1] = m[2][1];
      Q[0] = Q[4] = Q[8] = 1.0;
    } else {
      const double beta = sqrt(madd(ft, m[1][0], m[1][0], v1norm2));
      const double inv_beta = 1.0 / beta;
      const double m01 = m[1][0] * inv_beta;
      const double m02 = m[2][0] * inv_beta;
      const double q = madd(ft, 2.0 * m01, m[2][1], m02 * (m[2][2] - m[1][1]));
      diag[1] = madd(ft, m02, q, m[1][1]);
      diag[2] = madd(ft, -m02, q, m[2][2]);
      sub[0] = beta;
      sub[1] = madd(ft, -m01, q, m[2][1]);
      Q[0] = 1.0;
      Q[4] = m01;
      Q[5] = m02;
      Q[7] = m02;
      Q[8] = -m01;
    }
  }
  int iter = 0;
  {  // computeFromTridiagonal_impl
    const int n = 3, max_iterations = 30;
    int end = n - 1, start = 0;
    const double consider_as_zero = DBL_MIN;
    const double precision_inv = 1.0 / DBL_EPSILON;
    const double precision = 2.0 * DBL_EPSILON;
    while (end > 0) {
      for (int i = start; i < end; ++i) {
        if (version == 34) {
          if (fabs(sub[i]) < consider_as_zero) {
            sub[i] = 0.0;
          } else {
            const double scaled = precision_inv * sub[i];
            if (scaled * scaled <= (fabs(diag[i]) + fabs(diag[i + 1]))) sub[i] = 0.0;
          }
        } else {
          if (fabs(sub[i]) <= (fabs(diag[i]) + fabs(diag[i + 1])) * precision || fabs(sub[i]) <= consider_as_zero)
            sub[i] = 0.0;
        }
      }
      while (end > 0 && sub[end - 1] == 0.0) end--;
      if (end <= 0) break;
      iter++;
      if (iter > max_iterations * n) break;
      start = end - 1;
      while (start > 0 && sub[start - 1] != 0.0) start--;
      qr_step(diag, sub, start, end, Q, version, fm);
    }
    if (iter > max_iterations * n) return 1;
    for (int i = 0; i < n - 1; ++i) {  // selection sort, ascending, vectors follow
      int k = 0;
      for (int j = 1; j < n - i; ++j)
        if (diag[i + j] < diag[i + k]) k = j;
      if (k > 0) {
        double tmp = diag[i];
        diag[i] = diag[k + i];
        diag[k + i] = tmp;
        for (int j = 0; j < n; ++j) {
          tmp = Q[i * n + j];
          Q[i * n + j] = Q[(k + i) * n + j];
          Q[(k + i) * n + j] = tmp;
        }
      }
    }
  }
  for (int i = 0; i < 3; ++i) evals[i] = diag[i] * scale;
  for (int i = 0; i < 9; ++i) evecs[i] = Q[i];
  return 0;
}

constexpr int kAccWavesPerBlock = 4;

// One wave per voxel; lane c < 12 owns running sum c (0..2 = sum x y z, 3 + e = moment element e, row-major).
// sorted_idx[seg_offset[v] .. + seg_count[v]) are the voxel's points in ascending point order (stable sort).
// acc_out[12 v + c].
__global__ __launch_bounds__(64 * kAccWavesPerBlock) void voxel_accumulate_exact_kernel(
    const double* __restrict__ px, const double* __restrict__ py, const double* __restrict__ pz,
    const uint32_t* __restrict__ sorted_idx, const uint32_t* __restrict__ seg_offset,
    const uint32_t* __restrict__ seg_count, uint32_t n_voxels, int fma_mask, double* __restrict__ acc_out) {
  __shared__ double stage[kAccWavesPerBlock][4][64];  // [wave][x y z one][point of the chunk]
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const uint32_t v = blockIdx.x * kAccWavesPerBlock + wave;
  if (v >= n_voxels) return;  // wave-uniform; no block-wide barrier below
  const uint32_t begin = seg_offset[v], count = seg_count[v];
  const int chain = lane < 12 ? lane : 0;
  const int e = chain - 3;
  const int ia = chain < 3 ? chain : e / 3;  // first factor
  const int ib = chain < 3 ? 3 : e % 3;      // second factor (plane 3 holds 1.0: sum += p is p * 1 + sum, exactly)
  const int fused = chain >= 3 ? ((fma_mask >> (kMomentShift + e)) & 1) : 0;
  double m = (chain >= 3 && (e % 4) == 0) ? 1.0 : 0.0;  // NDT::moment starts at Identity (MDM/types.h:14)
  stage[wave][3][lane] = 1.0;
  for (uint32_t base = 0; base < count; base += 64) {
    const uint32_t k = base + uint32_t(lane);
    if (k < count) {
      const uint32_t i = sorted_idx[begin + k];
      stage[wave][0][lane] = px[i];
      stage[wave][1][lane] = py[i];
      stage[wave][2][lane] = pz[i];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const int todo = int(count - base < 64u ? count - base : 64u);
    const double* pa = stage[wave][ia];
    const double* pb = stage[wave][ib];
    for (int j = 0; j < todo; ++j) m = madd(fused, pa[j], pb[j], m);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
  if (lane < 12) acc_out[12 * size_t(v) + lane] = m;
}

struct FinalizeParams {
  int min_points;         // 5    (…test.cc:258)
  double min_eigenvalue;  // 0.01 on the largest eigenvalue (:264)
  double eig_floor;       // 0.01 ratio (:268)
  int fma_mask;
  int eigen_version;      // 33 or 34
};

// One lane per voxel: mean, covariance, Eigen's solver, floor, sqrt_information = D^-1/2 V.  Outputs follow the CPU
// restatement's conventions: an invalid voxel has mean 0 and sqrt_information = I; evals are the un-floored
// eigenvalues (0 when the voxel has too few points), evecs row-major V (evecs[9 v + 3 i + k] = component i of
// eigenvector k).  first_idx[v] = lowest point index of the voxel (first-seen order of the reference's map).
__global__ __launch_bounds__(64) void voxel_finalize_exact_kernel(const double* __restrict__ acc,
                                                                  const uint32_t* __restrict__ sorted_idx,
                                                                  const uint32_t* __restrict__ seg_offset,
                                                                  const uint32_t* __restrict__ seg_count,
                                                                  uint32_t n_voxels, FinalizeParams prm,
                                                                  double* __restrict__ mean_out,
                                                                  double* __restrict__ sqrt_info_out,
                                                                  unsigned char* __restrict__ valid_out,
                                                                  double* __restrict__ evals_out,
                                                                  double* __restrict__ evecs_out,
                                                                  uint32_t* __restrict__ first_idx) {
  const uint32_t v = blockIdx.x * 64 + threadIdx.x;
  if (v >= n_voxels) return;
  const uint32_t count = seg_count[v];
  first_idx[v] = sorted_idx[seg_offset[v]];
  double mean[3] = {0.0, 0.0, 0.0};
  double S[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  double ev[3] = {0.0, 0.0, 0.0};
  double Vrm[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  unsigned char ok = 0;
  if (count >= uint32_t(prm.min_points)) {
    const double cnt = double(count);
    const double* a = acc + 12 * size_t(v);
    double mu[3], cov[9], U[9];
    for (int k = 0; k < 3; ++k) mu[k] = a[k] / cnt;
    const int fcov = (prm.fma_mask >> kCovShift) & 0x1ff;
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j)
        cov[3 * i + j] = madd((fcov >> (3 * i + j)) & 1, -mu[i], mu[j], a[3 + 3 * i + j] / cnt);
    const int info = eigen_selfadjoint3(cov, prm.eigen_version, prm.fma_mask, ev, U);
    for (int i = 0; i < 3; ++i)
      for (int k = 0; k < 3; ++k) Vrm[3 * i + k] = U[3 * k + i];
    if (info == 0 && !(ev[2] < prm.min_eigenvalue)) {
      double d[3] = {ev[0], ev[1], ev[2]};
      const double fl = d[2] * prm.eig_floor;
      d[0] = d[0] > fl ? d[0] : fl;  // std::max(eigvals(0), eigvals(2) * ratio)
      d[1] = d[1] > fl ? d[1] : fl;
      for (int k = 0; k < 3; ++k) mean[k] = mu[k];
      for (int i = 0; i < 3; ++i) {
        const double w = sqrt(1.0 / d[i]);  // eigvals.cwiseInverse().cwiseSqrt()
        for (int k = 0; k < 3; ++k) S[3 * i + k] = w * Vrm[3 * i + k];  // (D^-1/2 V)(i, k)
      }
      ok = 1;
    }
  }
  for (int k = 0; k < 3; ++k) mean_out[3 * size_t(v) + k] = mean[k];
  for (int k = 0; k < 9; ++k) sqrt_info_out[9 * size_t(v) + k] = S[k];
  for (int k = 0; k < 3; ++k) evals_out[3 * size_t(v) + k] = ev[k];
  for (int k = 0; k < 9; ++k) evecs_out[9 * size_t(v) + k] = Vrm[k];
  valid_out[v] = ok;
}

}  // namespace mapexact
}  // namespace nos
