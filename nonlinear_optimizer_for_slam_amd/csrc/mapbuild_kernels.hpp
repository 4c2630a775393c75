// mapbuild_kernels.hpp — NDT map construction on the GPU.
//
// Restates UpdateNdtMap of the reference's test harness
// (nonlinear_optimizer/mahalanobis_distance_minimizer/tests/simple_optimization_test.cc:236-281):
//   per point:  voxel key, ++count, sum += p, moment += p pᵀ   (moment starts at IDENTITY, MDM/types.h:14)
//   per voxel:  count < 5 → invalid;  mean = sum / count;  cov = moment / count − mean meanᵀ;
//               eigen-decomposition (ascending);  largest eigenvalue < 0.01 → invalid;
//               the two smaller eigenvalues are floored at 0.01 × largest (:268-273);
//               sqrt_information = diag(eigvals^-1/2) · eigenvectors (:275-276)
// GPU form: voxel keys → stable radix sort of (key, point id) → run-length encode → one wave per
// voxel sums its points in a fixed order (lane-strided, then butterfly), so the statistics are
// deterministic, and finishes the 3×3 symmetric eigenproblem with cyclic Jacobi rotations.
// Eigenvector sign convention (the reference inherits Eigen's, which is not reproducible here):
// the first component of each eigenvector whose magnitude is within 1e-6 of its largest is positive.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "match_kernels.hpp"

namespace nos {

__global__ __launch_bounds__(256) void voxel_key_kernel(const double* __restrict__ px, const double* __restrict__ py,
                                                        const double* __restrict__ pz, uint64_t n, double inv_res,
                                                        uint64_t* __restrict__ keys, uint32_t* __restrict__ idx) {
  const uint64_t i = uint64_t(blockIdx.x) * 256 + threadIdx.x;
  if (i >= n) return;
  keys[i] = pack_cell(int64_t(floor(px[i] * inv_res)), int64_t(floor(py[i] * inv_res)), int64_t(floor(pz[i] * inv_res)));
  idx[i] = uint32_t(i);
}

// Compact keys (round 4).  The packed key above spends 63 bits whatever the scene's extent, and a radix sort pays for every
// one of them: 8 passes over 10 M (key, index) pairs = 0.75 of the map build's 6.8 ms.  With the cells' bounding box known,
// key = ((x - x0) NY + (y - y0)) NZ + (z - z0) orders the cells exactly as the packed key does — lexicographically in
// (x, y, z) — in ceil(log2(NX NY NZ)) bits: 18 for a 100 x 100 x 10 m scene at 1 m, i.e. 3 passes.
//   box[0..2] = min cell, box[3..5] = max cell, initialised to INT64_MAX / INT64_MIN; grid-stride so that the whole grid
//   sends a few thousand atomics, not one per wave.
__global__ __launch_bounds__(256) void voxel_box_kernel(const double* __restrict__ px, const double* __restrict__ py,
                                                        const double* __restrict__ pz, uint64_t n, double inv_res,
                                                        long long* __restrict__ box) {
  long long lo[3] = {0x7FFFFFFFFFFFFFFFll, 0x7FFFFFFFFFFFFFFFll, 0x7FFFFFFFFFFFFFFFll};
  long long hi[3] = {-0x7FFFFFFFFFFFFFFFll - 1, -0x7FFFFFFFFFFFFFFFll - 1, -0x7FFFFFFFFFFFFFFFll - 1};
  for (uint64_t i = uint64_t(blockIdx.x) * 256 + threadIdx.x; i < n; i += uint64_t(gridDim.x) * 256) {
    const double c[3] = {floor(px[i] * inv_res), floor(py[i] * inv_res), floor(pz[i] * inv_res)};
#pragma unroll
    for (int k = 0; k < 3; ++k)
      if (c[k] >= -9.0e18 && c[k] <= 9.0e18) {  // finite and representable (a NaN fails both tests)
        const long long v = (long long)c[k];
        lo[k] = v < lo[k] ? v : lo[k];
        hi[k] = v > hi[k] ? v : hi[k];
      }
  }
  __shared__ long long s_lo[3][256], s_hi[3][256];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    s_lo[k][threadIdx.x] = lo[k];
    s_hi[k][threadIdx.x] = hi[k];
  }
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (int(threadIdx.x) < o) {
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const long long a = s_lo[k][threadIdx.x + o], b = s_hi[k][threadIdx.x + o];
        if (a < s_lo[k][threadIdx.x]) s_lo[k][threadIdx.x] = a;
        if (b > s_hi[k][threadIdx.x]) s_hi[k][threadIdx.x] = b;
      }
    }
    __syncthreads();
  }
  if (threadIdx.x < 3) {
    atomicMin(&box[threadIdx.x], s_lo[threadIdx.x][0]);
    atomicMax(&box[3 + threadIdx.x], s_hi[threadIdx.x][0]);
  }
}

// origin = min cell, dims = NX, NY, NZ (their product < 2^62); coordinates outside the box (non-finite points) are clamped in
__global__ __launch_bounds__(256) void voxel_compact_key_kernel(const double* __restrict__ px, const double* __restrict__ py,
                                                                const double* __restrict__ pz, uint64_t n, double inv_res,
                                                                long long x0, long long y0, long long z0, long long nx,
                                                                long long ny, long long nz, uint64_t* __restrict__ keys,
                                                                uint32_t* __restrict__ idx) {
  const uint64_t i = uint64_t(blockIdx.x) * 256 + threadIdx.x;
  if (i >= n) return;
  auto rel = [](double c, long long origin, long long dim) -> long long {
    if (!(c >= -9.0e18 && c <= 9.0e18)) return 0;
    const long long v = (long long)c - origin;
    return v < 0 ? 0 : (v >= dim ? dim - 1 : v);
  };
  const long long x = rel(floor(px[i] * inv_res), x0, nx), y = rel(floor(py[i] * inv_res), y0, ny),
                  z = rel(floor(pz[i] * inv_res), z0, nz);
  keys[i] = uint64_t((x * ny + y) * nz + z);
  idx[i] = uint32_t(i);
}

// Cyclic Jacobi for a symmetric 3x3 (row-major a[9]); eigenvalues ascending in w, eigenvectors in
// the COLUMNS of V (row-major), signs fixed as described in the file header.
__host__ __device__ inline void symmetric_eigen3(const double* A, double* w, double* V) {
  double a[9];
  for (int i = 0; i < 9; ++i) a[i] = A[i];
  for (int i = 0; i < 9; ++i) V[i] = (i % 4 == 0) ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 60; ++sweep) {
    const double off = a[1] * a[1] + a[2] * a[2] + a[5] * a[5];
    const double diag = a[0] * a[0] + a[4] * a[4] + a[8] * a[8];
    if (off <= 1e-26 * diag) break;
    for (int p = 0; p < 2; ++p)
      for (int q = p + 1; q < 3; ++q) {
        const double apq = a[3 * p + q];
        // off-diagonals at rounding-noise level are treated as zero, so numerically diagonal matrices
        // (axis-aligned patches) keep axis-aligned eigenvectors instead of a noise-driven rotation
        if (fabs(apq) <= 1e-13 * (fabs(a[3 * p + p]) + fabs(a[3 * q + q]))) continue;
        const double theta = (a[3 * q + q] - a[3 * p + p]) / (2.0 * apq);
        const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
        for (int k = 0; k < 3; ++k) {
          const double akp = a[3 * k + p], akq = a[3 * k + q];
          a[3 * k + p] = c * akp - s * akq;
          a[3 * k + q] = s * akp + c * akq;
        }
        for (int k = 0; k < 3; ++k) {
          const double apk = a[3 * p + k], aqk = a[3 * q + k];
          a[3 * p + k] = c * apk - s * aqk;
          a[3 * q + k] = s * apk + c * aqk;
        }
        for (int k = 0; k < 3; ++k) {
          const double vkp = V[3 * k + p], vkq = V[3 * k + q];
          V[3 * k + p] = c * vkp - s * vkq;
          V[3 * k + q] = s * vkp + c * vkq;
        }
      }
  }
  int o0 = 0, o1 = 1, o2 = 2;  // sort the three diagonal entries ascending (stable)
  if (a[4 * o1] < a[4 * o0]) { const int t = o0; o0 = o1; o1 = t; }
  if (a[4 * o2] < a[4 * o1]) { const int t = o1; o1 = o2; o2 = t; }
  if (a[4 * o1] < a[4 * o0]) { const int t = o0; o0 = o1; o1 = t; }
  const int order[3] = {o0, o1, o2};
  double Vs[9];
  for (int c = 0; c < 3; ++c) {
    w[c] = a[4 * order[c]];
    // sign convention: the first component whose magnitude is within 1e-6 of the largest is positive
    // (tolerant form of "largest component positive", so exact ties such as (1, -1, 0)/sqrt(2) are not
    // decided by rounding noise).  On the reference's room scene this convention reproduces the captured
    // run's per-solve costs to 0.06 % (17448.5 vs 17438.4), i.e. it is close to what Eigen returns there.
    double vmax = 0.0;
    for (int r = 0; r < 3; ++r) vmax = fmax(vmax, fabs(V[3 * r + order[c]]));
    int big = 0;
    while (big < 2 && fabs(V[3 * big + order[c]]) < vmax * (1.0 - 1e-6)) ++big;
    const double sign = V[3 * big + order[c]] < 0 ? -1.0 : 1.0;
    for (int r = 0; r < 3; ++r) Vs[3 * r + c] = sign * V[3 * r + order[c]];
  }
  // Repeated eigenvalues (planar patches: the two in-plane variances tie) leave the eigenbasis of the
  // degenerate plane undetermined, and rounding noise would pick it.  Fix it instead: take the
  // Householder reflection that maps e_0 onto the eigenvector n of the distinct eigenvalue (or e_2 onto
  // it when the two SMALL eigenvalues tie).  Its columns are an orthonormal eigenbasis, it is symmetric,
  // and therefore the harness formula D^-1/2 V coincides with the true square root D^-1/2 V^T there.
  {
    const double tol = 1e-9 * fabs(w[2]);
    const bool tie_hi = fabs(w[2] - w[1]) <= tol, tie_lo = fabs(w[1] - w[0]) <= tol;
    if (tie_hi && tie_lo) {
      for (int i = 0; i < 9; ++i) Vs[i] = (i % 4 == 0) ? 1.0 : 0.0;
    } else if (tie_hi || tie_lo) {
      const int col = tie_hi ? 0 : 2;  // the column that holds the distinct eigenvector
      double nvec[3] = {Vs[col], Vs[3 + col], Vs[6 + col]};
      if (nvec[col] > 0) {             // reflect e_col onto -n when that is the better conditioned choice
        nvec[0] = -nvec[0];
        nvec[1] = -nvec[1];
        nvec[2] = -nvec[2];
      }
      double hv[3] = {-nvec[0], -nvec[1], -nvec[2]};
      hv[col] += 1.0;                   // hv = e_col - n
      const double hh = hv[0] * hv[0] + hv[1] * hv[1] + hv[2] * hv[2];
      for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) Vs[3 * r + c] = (r == c ? 1.0 : 0.0) - 2.0 * hv[r] * hv[c] / hh;
      // no per-column sign flips here: they would break the symmetry that makes the formula well posed
    }
  }
  for (int i = 0; i < 9; ++i) V[i] = Vs[i];
}

struct MapBuildParams {
  int min_points;        // 5   (:258)
  double min_eigenvalue; // 0.01 on the largest eigenvalue (:264)
  double eig_floor;      // 0.01 ratio (:268)
  int proper_transpose;  // 0: sqrt_information = D^-1/2 V (the harness formula, :275-276);
                         // 1: D^-1/2 V^T (the actual square root of the inverse covariance — invariant to
                         //    eigenvector signs and to rotations inside degenerate eigenspaces)
};

// Two kernels since round 4.  In the one-kernel form lane 0 of every wave ran the 3x3 eigen-decomposition while 63 lanes
// idled: at 796 k voxels of ≈ 12 points that was 3.0 of the build's 12 ms (profiles/r04_mapbuild_summary.json: issue stalls
// 46 %, one launch 3 041 µs).  Same additions in the same order, same eigen routine.
//
// The points once more as 32-byte records {x, y, z, 0}: the sums kernel GATHERS points by sorted index, and a gather of three
// 8-byte values from three planes touches three 64-byte sectors per point (1.9 GB of HBM traffic for 10 M points, 0.9 ms:
// profiles/r04_mapbuild_summary.json), one aligned 32-byte record one.
__global__ __launch_bounds__(256) void points_to_records_kernel(const double* __restrict__ px, const double* __restrict__ py,
                                                                const double* __restrict__ pz, uint64_t n,
                                                                double* __restrict__ rec /* [n][4] */) {
  const uint64_t i = uint64_t(blockIdx.x) * 256 + threadIdx.x;
  if (i >= n) return;
  using V2 = double __attribute__((ext_vector_type(2)));
  V2* out = reinterpret_cast<V2*>(rec) + 2 * i;
  out[0] = V2{px[i], py[i]};
  out[1] = V2{pz[i], 0.0};
}

// (1) One wave per voxel: count / sum / moment in a fixed order.  seg_offset[v] .. + seg_count[v] index into sorted_idx.
//     acc_out: [n_voxels][9] = sx sy sz | mxx mxy mxz myy myz mzz.  rec != nullptr: the points as records (above).
__global__ __launch_bounds__(256) void voxel_sums_kernel(const double* __restrict__ px, const double* __restrict__ py,
                                                         const double* __restrict__ pz, const double* __restrict__ rec,
                                                         const uint32_t* __restrict__ sorted_idx,
                                                         const uint32_t* __restrict__ seg_offset,
                                                         const uint32_t* __restrict__ seg_count, uint32_t n_voxels,
                                                         double* __restrict__ acc_out) {
  const uint32_t v = (blockIdx.x * 256 + threadIdx.x) / kWave;
  const int lane = threadIdx.x & (kWave - 1);
  if (v >= n_voxels) return;  // wave-uniform
  const uint32_t begin = seg_offset[v], count = seg_count[v];
  double acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (uint32_t k = lane; k < count; k += kWave) {
    const uint32_t i = sorted_idx[begin + k];
    double x, y, z;
    if (rec != nullptr) {  // kernel-uniform
      using V2 = double __attribute__((ext_vector_type(2)));
      const V2* r2 = reinterpret_cast<const V2*>(rec) + 2 * size_t(i);
      const V2 a = r2[0], b = r2[1];
      x = a[0], y = a[1], z = b[0];
    } else {
      x = px[i], y = py[i], z = pz[i];
    }
    acc[0] += x;
    acc[1] += y;
    acc[2] += z;
    acc[3] = fma(x, x, acc[3]);
    acc[4] = fma(x, y, acc[4]);
    acc[5] = fma(x, z, acc[5]);
    acc[6] = fma(y, y, acc[6]);
    acc[7] = fma(y, z, acc[7]);
    acc[8] = fma(z, z, acc[8]);
  }
#pragma unroll
  for (int k = 0; k < 9; ++k) acc[k] = wave_sum(acc[k]);
  if (lane < 9) {  // lane k stores sum k (every lane holds all nine after the butterfly)
    double mine = acc[0];
#pragma unroll
    for (int k = 1; k < 9; ++k) mine = lane == k ? acc[k] : mine;
    acc_out[9 * size_t(v) + lane] = mine;
  }
}

// (2) One LANE per voxel: mean, covariance (the moment starts at identity, MDM/types.h:14), eigen-decomposition, validity
//     rules, sqrt-information.
__global__ __launch_bounds__(256) void voxel_eigen_kernel(const double* __restrict__ acc_in,
                                                          const uint32_t* __restrict__ seg_count, uint32_t n_voxels,
                                                          MapBuildParams prm, double* __restrict__ mean_out,
                                                          double* __restrict__ sqrt_info_out,
                                                          unsigned char* __restrict__ valid_out) {
  const uint32_t v = blockIdx.x * 256 + threadIdx.x;
  if (v >= n_voxels) return;
  const uint32_t count = seg_count[v];
  double acc[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) acc[k] = acc_in[9 * size_t(v) + k];
  double S[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  double mean[3] = {0, 0, 0};
  unsigned char ok = 0;
  if (count >= uint32_t(prm.min_points)) {
    const double inv = 1.0 / double(count);
    for (int k = 0; k < 3; ++k) mean[k] = acc[k] * inv;
    // moment = Identity + sum p p^T
    double cov[9];
    cov[0] = (acc[3] + 1.0) * inv - mean[0] * mean[0];
    cov[1] = cov[3] = acc[4] * inv - mean[0] * mean[1];
    cov[2] = cov[6] = acc[5] * inv - mean[0] * mean[2];
    cov[4] = (acc[6] + 1.0) * inv - mean[1] * mean[1];
    cov[5] = cov[7] = acc[7] * inv - mean[1] * mean[2];
    cov[8] = (acc[8] + 1.0) * inv - mean[2] * mean[2];
    double w[3], V[9];
    symmetric_eigen3(cov, w, V);
    if (!(w[2] < prm.min_eigenvalue)) {
      w[0] = fmax(w[0], w[2] * prm.eig_floor);
      w[1] = fmax(w[1], w[2] * prm.eig_floor);
      for (int i = 0; i < 3; ++i) {
        const double sc = 1.0 / sqrt(w[i]);
        for (int j = 0; j < 3; ++j) S[3 * i + j] = sc * (prm.proper_transpose ? V[3 * j + i] : V[3 * i + j]);
      }
      ok = 1;
    }
  }
  for (int k = 0; k < 3; ++k) mean_out[3 * size_t(v) + k] = mean[k];
  for (int k = 0; k < 9; ++k) sqrt_info_out[9 * size_t(v) + k] = S[k];
  valid_out[v] = ok;
}

}  // namespace nos
