// assemble_kernels.hpp — hand-written gfx950 (CDNA4, wave64) kernels for the Gauss-Newton
// normal-equation assembly path.
//
// One streaming pass per LM iteration: every lane reads ITEMS consecutive correspondences
// from a tiled struct-of-arrays dataset (one wide load per field), evaluates residual,
// analytic Jacobian and robust weight in registers, and keeps the 28 (or 10) running sums
// of  H = Σ w JᵀJ (upper triangle),  g = Σ w Jᵀr,  cost = Σ ρ  in registers for the whole
// grid-stride loop.  The sums leave the lane exactly once: wave64 butterfly → LDS across
// the block's waves → one row of `partials`, then a fixed-order final pass → 28 doubles.
// No atomics, so results are bit-reproducible for a fixed launch geometry.
//
// Bound: HBM bandwidth (120 B fp64 / 60 B fp32 per correspondence, ≈200 VALU ops); the
// contraction is a fixed 6×6 outer product so MFMA is deliberately not used.
//
// Math restated from (reference paths relative to nonlinear_optimizer/):
//   6-DoF NDT     mahalanobis_distance_minimizer/mahalanobis_distance_minimizer_analytic.cc:159-185
//                 (lane form: ..._analytic_simd_various.cc:656-697)
//   3-DoF NDT     mahalanobis_distance_minimizer/mahalanobis_distance_minimizer_analytic_3dof.cc:110-139
//   reprojection  reprojection_error_minimizer/reprojection_error_minimizer_analytic.cc:107-162
//   robust loss   loss_function.h:28-33, 57-66
#pragma once

#include <hip/hip_runtime.h>
#include <limits>

#include "host/nos_lm.hpp"
#include <stdint.h>

namespace nos {

constexpr int kWave = 64;

enum LossKind : int { kLossNone = 0, kLossExponential = 1, kLossHuber = 2 };

// Tiled SoA addressing.  Correspondence i, field f lives at element offset
//   (i >> tile_shift) * tile_stride + f * field_stride + (i & (tile - 1)).
// tile == n_padded, tile_stride == 0 gives a plain planar layout.
struct TiledLayout {
  const void* base;
  uint64_t n;            // real correspondences
  uint64_t n_padded;     // multiple of tile (pads are all-zero records)
  uint64_t tile_stride;  // elements between consecutive tiles
  uint64_t field_stride; // elements between consecutive fields inside a tile
  uint32_t tile_shift;   // log2(tile)
  uint32_t tile_mask;    // tile - 1
};

template <typename T>
struct Ndt6Params {
  T R[9];
  T t[3];
  T la, lb, lc;  // loss: (c1, c2, 2*c1*c2) | (th, th*th, 2*th)
};

template <typename T>
struct Ndt3Params {
  T R2[4];
  T t2[2];
  T la, lb, lc;
};

template <typename T>
struct ReprojParams {
  T R[9];
  T t[3];
  T inv_fx, inv_fy, cx, cy;
  T min_depth;
  T la, lb, lc;
  // Validity rules on the depth z = (R X + t)_z, set by the launcher (set_reproj_rules):
  //   scalar class (REM/..._analytic.cc:111,119-123): a correspondence with z < min_depth contributes nothing at all
  //     → thr_w = min_depth, loss_everywhere = 0;
  //   fp32 class (REM/..._analytic_simd.cc:66-92,134): the WEIGHT counts where z > 0, residual and loss are evaluated for
  //     every correspondence → thr_w = smallest positive number, loss_everywhere = 1.  (z == 0 exactly then gives the same
  //     inf / NaN as in the reference; the damped solve reports the non-finite pivot instead of returning a pose.)
  // The second rule is a uniform flag combined with the lane mask by scalar instructions: no vector-ALU cost.
  T thr_w;
  int loss_everywhere;
};
template <typename T>
inline void set_reproj_rules(ReprojParams<T>& P, bool simd_class) {
  P.thr_w = simd_class ? std::numeric_limits<T>::min() : P.min_depth;
  P.loss_everywhere = simd_class ? 1 : 0;
}

// ---------------------------------------------------------------- math helpers

template <typename T>
__device__ __forceinline__ T fast_exp(T x);
template <>
__device__ __forceinline__ double fast_exp<double>(double x) {
  return exp(x);
}
template <>
__device__ __forceinline__ float fast_exp<float>(float x) {
  return __expf(x);
}
template <typename T>
__device__ __forceinline__ T fast_sqrt(T x);
template <>
__device__ __forceinline__ double fast_sqrt<double>(double x) {
  return sqrt(x);
}
template <>
__device__ __forceinline__ float fast_sqrt<float>(float x) {
  return sqrtf(x);
}

// 1/x and 1/sqrt(x) to full fp64 accuracy from the hardware seed plus two Newton steps (~5 / ~9 instructions
// instead of the ~20-instruction IEEE divide / sqrt sequences; the reprojection kernel is fp64-ALU bound).
// Callers pass x > 0 and finite.
template <typename T>
__device__ __forceinline__ T fast_inv(T x) {
  if constexpr (sizeof(T) == 8) {
    double y = __builtin_amdgcn_rcp(x);
    double e = fma(-x, y, 1.0);
    y = fma(y, e, y);
    e = fma(-x, y, 1.0);
    return fma(y, e, y);
  } else {
    return T(1) / x;
  }
}

template <typename T>
__device__ __forceinline__ T fast_rsqrt(T x) {
  if constexpr (sizeof(T) == 8) {
    double y = __builtin_amdgcn_rsq(x);
    // y <- y + y * (0.5 - 0.5 x y^2): quadratic convergence, twice
    double h = 0.5 * y;
    double e = fma(-x * y, h, 0.5);
    y = fma(y, e, y);
    h = 0.5 * y;
    e = fma(-x * y, h, 0.5);
    return fma(y, e, y);
  } else {
    return rsqrtf(x);
  }
}

// ---- value types of the item math.  The item functions below are written once for a value type V: the element type T
// itself (one correspondence per call) or — fp32 only — a packed pair of floats (two correspondences per call: gfx950
// has v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32; measured, the packed kernels are slower than the scalar ones, so the
// pair form is a compile-time experiment only, see assemble_kernel).  Pose, loss parameters and masks stay scalar.
using float2_t = float __attribute__((ext_vector_type(2)));

template <typename V>
struct Lanes {
  static constexpr int n = 1;
  using S = V;
};
template <>
struct Lanes<float2_t> {
  static constexpr int n = 2;
  using S = float;
};

template <typename V>
__device__ __forceinline__ V splat(typename Lanes<V>::S s) {
  if constexpr (Lanes<V>::n == 2)
    return V{s, s};
  else
    return s;
}
template <typename V>
__device__ __forceinline__ V vfma(V a, V b, V c) {
  if constexpr (Lanes<V>::n == 2)
    return __builtin_elementwise_fma(a, b, c);
  else
    return fma(a, b, c);
}
// scalar coefficient (pose / intrinsics entry) times value plus value
template <typename V>
__device__ __forceinline__ V sfma(typename Lanes<V>::S a, V b, V c) {
  return vfma<V>(splat<V>(a), b, c);
}
template <typename V>
__device__ __forceinline__ typename Lanes<V>::S lane_get(const V& v, int k) {
  if constexpr (Lanes<V>::n == 2)
    return v[k];
  else
    return v;
}
template <typename V>
__device__ __forceinline__ void lane_set(V& v, int k, typename Lanes<V>::S s) {
  if constexpr (Lanes<V>::n == 2)
    v[k] = s;
  else
    v = s;
}

// loss_function.h:28-33 / :57-66 ; LOSS == 0 is the `loss_function_ == nullptr` branch.  Scalar form:
template <typename T, int LOSS>
__device__ __forceinline__ void loss_eval(T s, T la, T lb, T lc, T& rho, T& w) {
  if constexpr (LOSS == kLossExponential) {
    const T ex = fast_exp<T>(-lb * s);
    rho = la - la * ex;
    w = lc * ex;
  } else if constexpr (LOSS == kLossHuber) {
    const bool outlier = s > lb;           // lb = th^2
    const T sc = outlier ? s : T(1);
    const T ir = fast_rsqrt<T>(sc);        // 1 / |r|
    rho = outlier ? (lc * (sc * ir) - lb) : s;  // lc = 2 th ;  |r| = s / |r|
    w = outlier ? (la * ir) : T(1);
  } else {
    rho = s;
    w = T(1);
  }
}
// value form: per lane through the scalar form (the transcendental / select part is not packable anyway)
template <typename V, int LOSS>
__device__ __forceinline__ void loss_eval_v(V s, typename Lanes<V>::S la, typename Lanes<V>::S lb, typename Lanes<V>::S lc,
                                            V& rho, V& w) {
  using S = typename Lanes<V>::S;
#pragma unroll
  for (int k = 0; k < Lanes<V>::n; ++k) {
    S r1, w1;
    loss_eval<S, LOSS>(lane_get<V>(s, k), la, lb, lc, r1, w1);
    lane_set<V>(rho, k, r1);
    lane_set<V>(w, k, w1);
  }
}

// acc += w * JᵀJ (upper), w * Jᵀr for a ROWS×6 Jacobian held as J[row][6].
template <typename V, int ROWS>
__device__ __forceinline__ void rank_update6(const V (&J)[ROWS][6], const V (&r)[ROWS], V w,
                                             V rho, V (&acc)[28]) {
  V wJ[ROWS][6];
#pragma unroll
  for (int a = 0; a < ROWS; ++a)
#pragma unroll
    for (int c = 0; c < 6; ++c) wJ[a][c] = w * J[a][c];
  int k = 0;
#pragma unroll
  for (int row = 0; row < 6; ++row)
#pragma unroll
    for (int col = row; col < 6; ++col) {
      V h = acc[k];
#pragma unroll
      for (int a = 0; a < ROWS; ++a) h = vfma<V>(wJ[a][row], J[a][col], h);
      acc[k] = h;
      ++k;
    }
#pragma unroll
  for (int c = 0; c < 6; ++c) {
    V gsum = acc[21 + c];
#pragma unroll
    for (int a = 0; a < ROWS; ++a) gsum = vfma<V>(wJ[a][c], r[a], gsum);
    acc[21 + c] = gsum;
  }
  acc[27] += rho;
}

// M = -R [p]x, column form of ..._analytic_simd_various.cc:677-687.
template <typename V>
__device__ __forceinline__ void minus_R_hat(const typename Lanes<V>::S (&R)[9], V px, V py, V pz, V (&M)[3][3]) {
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    M[i][0] = sfma<V>(R[3 * i + 2], py, -(splat<V>(R[3 * i + 1]) * pz));
    M[i][1] = sfma<V>(R[3 * i + 0], pz, -(splat<V>(R[3 * i + 2]) * px));
    M[i][2] = sfma<V>(R[3 * i + 1], px, -(splat<V>(R[3 * i + 0]) * py));
  }
}

// ---------------------------------------------------------------- problems

template <typename T, int LOSS>
struct Ndt6Problem {
  static constexpr int kFields = 15;
  static constexpr int kOut = 28;
  using Params = Ndt6Params<T>;
  // x = {p(3), mu(3), S row-major (9)}; V = T (one correspondence) or float2_t (two, fp32 only)
  template <typename V = T>
  __device__ static __forceinline__ void item(const V (&x)[15], const Params& P, const bool (&)[Lanes<V>::n] /*valid*/,
                                              V (&acc)[28]) {
    V e[3], r[3], M[3][3], J[3][6];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const V pw = sfma<V>(P.R[3 * i], x[0], sfma<V>(P.R[3 * i + 1], x[1], sfma<V>(P.R[3 * i + 2], x[2], splat<V>(P.t[i]))));
      e[i] = pw - x[3 + i];
    }
#ifndef NOS_NDT6_SFORM_F32
    if constexpr (sizeof(typename Lanes<V>::S) == 4) {
      // fp32: A = SᵀS first, then H = w [I|M]ᵀ A [I|M], g = w [I|M]ᵀ A e, s = eᵀ A e — ≈ 150 instead of ≈ 186 operations
      // per correspondence, the same sums.  Measured error against the fp64 oracle unchanged (1.09e-6 against 1.07e-6
      // scaled, of which 1.0e-6 is the rounding of the inputs; profiles/r02_fp32_error.jsonl), 2.5 % faster at 10 M.
      V A[3][3], Ae[3], wAe[3], B[3][3];
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = i; j < 3; ++j) {
          A[i][j] = vfma<V>(x[6 + i], x[6 + j], vfma<V>(x[9 + i], x[9 + j], x[12 + i] * x[12 + j]));
          A[j][i] = A[i][j];
        }
#pragma unroll
      for (int i = 0; i < 3; ++i) Ae[i] = vfma<V>(A[i][0], e[0], vfma<V>(A[i][1], e[1], A[i][2] * e[2]));
      const V s2 = vfma<V>(e[0], Ae[0], vfma<V>(e[1], Ae[1], e[2] * Ae[2]));
      V rho2, w2;
      loss_eval_v<V, LOSS>(s2, P.la, P.lb, P.lc, rho2, w2);
      minus_R_hat<V>(P.R, x[0], x[1], x[2], M);
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        wAe[i] = w2 * Ae[i];
#pragma unroll
        for (int j = i; j < 3; ++j) {
          A[i][j] = w2 * A[i][j];
          A[j][i] = A[i][j];
        }
      }
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int b = 0; b < 3; ++b) B[i][b] = vfma<V>(A[i][0], M[0][b], vfma<V>(A[i][1], M[1][b], A[i][2] * M[2][b]));
      acc[0] += A[0][0];
      acc[1] += A[0][1];
      acc[2] += A[0][2];
      acc[6] += A[1][1];
      acc[7] += A[1][2];
      acc[11] += A[2][2];
#pragma unroll
      for (int b = 0; b < 3; ++b) {
        acc[3 + b] += B[0][b];
        acc[8 + b] += B[1][b];
        acc[12 + b] += B[2][b];
        acc[21 + b] += wAe[b];
        acc[24 + b] = vfma<V>(M[0][b], wAe[0], vfma<V>(M[1][b], wAe[1], vfma<V>(M[2][b], wAe[2], acc[24 + b])));
      }
      int k = 15;
#pragma unroll
      for (int p = 0; p < 3; ++p)
#pragma unroll
        for (int q = p; q < 3; ++q) {
          acc[k] = vfma<V>(M[0][p], B[0][q], vfma<V>(M[1][p], B[1][q], vfma<V>(M[2][p], B[2][q], acc[k])));
          ++k;
        }
      acc[27] += rho2;
      return;
    }
#endif
#pragma unroll
    for (int a = 0; a < 3; ++a)
      r[a] = vfma<V>(x[6 + 3 * a], e[0], vfma<V>(x[7 + 3 * a], e[1], x[8 + 3 * a] * e[2]));
    minus_R_hat<V>(P.R, x[0], x[1], x[2], M);
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int b = 0; b < 3; ++b) {
        J[a][b] = x[6 + 3 * a + b];
        J[a][3 + b] = vfma<V>(x[6 + 3 * a], M[0][b], vfma<V>(x[7 + 3 * a], M[1][b], x[8 + 3 * a] * M[2][b]));
      }
    const V s = vfma<V>(r[0], r[0], vfma<V>(r[1], r[1], r[2] * r[2]));
    V rho, w;
    loss_eval_v<V, LOSS>(s, P.la, P.lb, P.lc, rho, w);
    // zero-padded records have S = 0 → r = 0, J = 0, rho(0) = 0: no mask needed
    rank_update6<V, 3>(J, r, w, rho, acc);
  }
  __device__ static __forceinline__ void item(const T (&x)[15], const Params& P, bool valid, T (&acc)[28]) {
    const bool v1[1] = {valid};
    item<T>(x, P, v1, acc);
  }

  // Voxel-indexed form: the voxel table holds A = SᵀS (a00 a01 a02 a11 a12 a22) instead of S.  With J = [S | S M]:
  //   s = rᵀr = eᵀAe,  g = w [A e ; Mᵀ A e],  H = w [A, A M ; · , Mᵀ A M]
  // — ≈ 144 instead of ≈ 190 operations per correspondence, 9 instead of 12 values per voxel record.
  __device__ static __forceinline__ void item_A(const T (&p)[3], const T (&mu)[3], const T (&A)[6], const Params& P,
                                                T (&acc)[28]) {
    T e[3], Ae[3], M[3][3], B[3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
      e[i] = fma(P.R[3 * i], p[0], fma(P.R[3 * i + 1], p[1], fma(P.R[3 * i + 2], p[2], P.t[i]))) - mu[i];
    const T a00 = A[0], a01 = A[1], a02 = A[2], a11 = A[3], a12 = A[4], a22 = A[5];
    Ae[0] = fma(a00, e[0], fma(a01, e[1], a02 * e[2]));
    Ae[1] = fma(a01, e[0], fma(a11, e[1], a12 * e[2]));
    Ae[2] = fma(a02, e[0], fma(a12, e[1], a22 * e[2]));
    const T s = fma(e[0], Ae[0], fma(e[1], Ae[1], e[2] * Ae[2]));
    T rho, w;
    loss_eval<T, LOSS>(s, P.la, P.lb, P.lc, rho, w);
    minus_R_hat<T>(P.R, p[0], p[1], p[2], M);
#pragma unroll
    for (int b = 0; b < 3; ++b) {
      B[0][b] = fma(a00, M[0][b], fma(a01, M[1][b], a02 * M[2][b]));
      B[1][b] = fma(a01, M[0][b], fma(a11, M[1][b], a12 * M[2][b]));
      B[2][b] = fma(a02, M[0][b], fma(a12, M[1][b], a22 * M[2][b]));
    }
    // upper triangle, row-major: rows 0-2 = [A | B], rows 3-5 = MᵀB
    acc[0] = fma(w, a00, acc[0]);
    acc[1] = fma(w, a01, acc[1]);
    acc[2] = fma(w, a02, acc[2]);
    acc[3] = fma(w, B[0][0], acc[3]);
    acc[4] = fma(w, B[0][1], acc[4]);
    acc[5] = fma(w, B[0][2], acc[5]);
    acc[6] = fma(w, a11, acc[6]);
    acc[7] = fma(w, a12, acc[7]);
    acc[8] = fma(w, B[1][0], acc[8]);
    acc[9] = fma(w, B[1][1], acc[9]);
    acc[10] = fma(w, B[1][2], acc[10]);
    acc[11] = fma(w, a22, acc[11]);
    acc[12] = fma(w, B[2][0], acc[12]);
    acc[13] = fma(w, B[2][1], acc[13]);
    acc[14] = fma(w, B[2][2], acc[14]);
    int k = 15;
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int b = a; b < 3; ++b) {
        const T c = fma(M[0][a], B[0][b], fma(M[1][a], B[1][b], M[2][a] * B[2][b]));
        acc[k] = fma(w, c, acc[k]);
        ++k;
      }
#pragma unroll
    for (int i = 0; i < 3; ++i) acc[21 + i] = fma(w, Ae[i], acc[21 + i]);
#pragma unroll
    for (int b = 0; b < 3; ++b) {
      const T gw = fma(M[0][b], Ae[0], fma(M[1][b], Ae[1], M[2][b] * Ae[2]));
      acc[24 + b] = fma(w, gw, acc[24 + b]);
    }
    acc[27] += rho;
  }
};

template <typename T, int LOSS>
struct Ndt3Problem {
  static constexpr int kFields = 15;
  static constexpr int kOut = 10;
  using Params = Ndt3Params<T>;
  template <typename V = T>
  __device__ static __forceinline__ void item(const V (&x)[15], const Params& P, const bool (&)[Lanes<V>::n] /*valid*/,
                                              V (&acc)[10]) {
    V e[3], r[3], J[3][3];
    const V ux = x[0], uy = x[1];
    e[0] = sfma<V>(P.R2[0], ux, sfma<V>(P.R2[1], uy, splat<V>(P.t2[0]))) - x[3];
    e[1] = sfma<V>(P.R2[2], ux, sfma<V>(P.R2[3], uy, splat<V>(P.t2[1]))) - x[4];
    e[2] = x[2] - x[5];
    const V d0 = sfma<V>(P.R2[1], ux, -(splat<V>(P.R2[0]) * uy));
    const V d1 = sfma<V>(P.R2[3], ux, -(splat<V>(P.R2[2]) * uy));
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      r[a] = vfma<V>(x[6 + 3 * a], e[0], vfma<V>(x[7 + 3 * a], e[1], x[8 + 3 * a] * e[2]));
      J[a][0] = x[6 + 3 * a];
      J[a][1] = x[7 + 3 * a];
      J[a][2] = vfma<V>(x[6 + 3 * a], d0, x[7 + 3 * a] * d1);
    }
    const V s = vfma<V>(r[0], r[0], vfma<V>(r[1], r[1], r[2] * r[2]));
    V rho, w;
    loss_eval_v<V, LOSS>(s, P.la, P.lb, P.lc, rho, w);
    V wJ[3][3];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int c = 0; c < 3; ++c) wJ[a][c] = w * J[a][c];
    int k = 0;
#pragma unroll
    for (int row = 0; row < 3; ++row)
#pragma unroll
      for (int col = row; col < 3; ++col) {
        acc[k] = vfma<V>(wJ[0][row], J[0][col], vfma<V>(wJ[1][row], J[1][col], vfma<V>(wJ[2][row], J[2][col], acc[k])));
        ++k;
      }
#pragma unroll
    for (int c = 0; c < 3; ++c)
      acc[6 + c] = vfma<V>(wJ[0][c], r[0], vfma<V>(wJ[1][c], r[1], vfma<V>(wJ[2][c], r[2], acc[6 + c])));
    acc[9] += rho;
  }
  __device__ static __forceinline__ void item(const T (&x)[15], const Params& P, bool valid, T (&acc)[10]) {
    const bool v1[1] = {valid};
    item<T>(x, P, v1, acc);
  }

  // Voxel-indexed form with A = SᵀS: J = [S(:,0) S(:,1) S(:,0:2)·d] ⇒ JᵀJ = [[a00, a01, q0], [·, a11, q1], [·, ·, dᵀq]]
  // with q = A(0:2,0:2)·d, and Jᵀr = [Ae₀, Ae₁, d·(Ae)(0:2)].
  __device__ static __forceinline__ void item_A(const T (&p)[3], const T (&mu)[3], const T (&A)[6], const Params& P,
                                                T (&acc)[10]) {
    const T ux = p[0], uy = p[1];
    T e[3];
    e[0] = fma(P.R2[0], ux, fma(P.R2[1], uy, P.t2[0])) - mu[0];
    e[1] = fma(P.R2[2], ux, fma(P.R2[3], uy, P.t2[1])) - mu[1];
    e[2] = p[2] - mu[2];
    const T d0 = fma(P.R2[1], ux, -(P.R2[0] * uy));
    const T d1 = fma(P.R2[3], ux, -(P.R2[2] * uy));
    const T a00 = A[0], a01 = A[1], a02 = A[2], a11 = A[3], a12 = A[4], a22 = A[5];
    const T Ae0 = fma(a00, e[0], fma(a01, e[1], a02 * e[2]));
    const T Ae1 = fma(a01, e[0], fma(a11, e[1], a12 * e[2]));
    const T Ae2 = fma(a02, e[0], fma(a12, e[1], a22 * e[2]));
    const T s = fma(e[0], Ae0, fma(e[1], Ae1, e[2] * Ae2));
    T rho, w;
    loss_eval<T, LOSS>(s, P.la, P.lb, P.lc, rho, w);
    const T q0 = fma(a00, d0, a01 * d1);
    const T q1 = fma(a01, d0, a11 * d1);
    acc[0] = fma(w, a00, acc[0]);
    acc[1] = fma(w, a01, acc[1]);
    acc[2] = fma(w, q0, acc[2]);
    acc[3] = fma(w, a11, acc[3]);
    acc[4] = fma(w, q1, acc[4]);
    acc[5] = fma(w, fma(d0, q0, d1 * q1), acc[5]);
    acc[6] = fma(w, Ae0, acc[6]);
    acc[7] = fma(w, Ae1, acc[7]);
    acc[8] = fma(w, fma(d0, Ae0, d1 * Ae1), acc[8]);
    acc[9] += rho;
  }
};

template <typename T, int LOSS>
struct ReprojProblem {
  static constexpr int kFields = 5;
  static constexpr int kOut = 28;
  using Params = ReprojParams<T>;
  // x = {X(3), pixel(2)}; V = T or float2_t
  template <typename V = T>
  __device__ static __forceinline__ void item(const V (&x)[5], const Params& P, const bool (&valid)[Lanes<V>::n],
                                              V (&acc)[28]) {
    using S = typename Lanes<V>::S;
    V Xw[3], J[2][6], r[2];
#pragma unroll
    for (int i = 0; i < 3; ++i)
      Xw[i] = sfma<V>(P.R[3 * i], x[0], sfma<V>(P.R[3 * i + 1], x[1], sfma<V>(P.R[3 * i + 2], x[2], splat<V>(P.t[i]))));
    // depth test of ..._analytic.cc:119-123; pads (valid == false) contribute nothing
    bool ok[Lanes<V>::n], okr[Lanes<V>::n];
    V iz;
#pragma unroll
    for (int k = 0; k < Lanes<V>::n; ++k) {
      const S z = lane_get<V>(Xw[2], k);
      ok[k] = valid[k] && !(z < P.thr_w);                          // the weight counts
      okr[k] = ok[k] || (valid[k] && P.loss_everywhere != 0);      // residual and loss are evaluated
      lane_set<V>(iz, k, fast_inv<S>(okr[k] ? z : S(1)));
    }
    const V iz2 = iz * iz;
    // (pixel − c) first: the difference is (nearly) exact, so fp32 keeps its digits in the residual
    r[0] = vfma<V>(Xw[0], iz, -(splat<V>(P.inv_fx) * (x[3] - splat<V>(P.cx))));
    r[1] = vfma<V>(Xw[1], iz, -(splat<V>(P.inv_fy) * (x[4] - splat<V>(P.cy))));
    const V k02 = -Xw[0] * iz2, k12 = -Xw[1] * iz2;
    J[0][0] = iz;
    J[0][1] = splat<V>(S(0));
    J[0][2] = k02;
    J[1][0] = splat<V>(S(0));
    J[1][1] = iz;
    J[1][2] = k12;
    // rotation block: row_a · (−R [X]x) = (X × u_a)ᵀ with u_a = R₀ᵀ/z + k_a2 R₂ᵀ (rows of R) — 24 operations instead
    // of the 30 that go through M = −R [X]x
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      const V ka = a == 0 ? k02 : k12;
      V u[3];
#pragma unroll
      for (int j = 0; j < 3; ++j) u[j] = sfma<V>(P.R[3 * a + j], iz, splat<V>(P.R[6 + j]) * ka);
      J[a][3] = vfma<V>(x[1], u[2], -(x[2] * u[1]));
      J[a][4] = vfma<V>(x[2], u[0], -(x[0] * u[2]));
      J[a][5] = vfma<V>(x[0], u[1], -(x[1] * u[0]));
    }
    V s = vfma<V>(r[0], r[0], r[1] * r[1]);
#pragma unroll
    for (int k = 0; k < Lanes<V>::n; ++k)
      if (!okr[k]) lane_set<V>(s, k, S(0));
    V rho, w;
    loss_eval_v<V, LOSS>(s, P.la, P.lb, P.lc, rho, w);
#pragma unroll
    for (int k = 0; k < Lanes<V>::n; ++k)
    {
      if (!ok[k]) lane_set<V>(w, k, S(0));
      if (!okr[k]) lane_set<V>(rho, k, S(0));
    }
    // acc += w JᵀJ (upper), w Jᵀr with the structure of this Jacobian spelled out — row 0 = [a 0 c d0 d1 d2],
    // row 1 = [0 a e f0 f1 f2] (a = 1/z): 49 operations instead of the 66 of the generic 2x6 update (the kernel is
    // fp64-VALU bound when the data is resident)
    {
      const V a = J[0][0], c = J[0][2], e = J[1][2];
      const V wa = w * a, wc = w * c, we = w * e;
      V wd[3], wf[3];
#pragma unroll
      for (int b = 0; b < 3; ++b) {
        wd[b] = w * J[0][3 + b];
        wf[b] = w * J[1][3 + b];
      }
      acc[0] = vfma<V>(wa, a, acc[0]);
      acc[2] = vfma<V>(wa, c, acc[2]);
      acc[6] = vfma<V>(wa, a, acc[6]);
      acc[7] = vfma<V>(wa, e, acc[7]);
      acc[11] = vfma<V>(wc, c, vfma<V>(we, e, acc[11]));
#pragma unroll
      for (int b = 0; b < 3; ++b) {
        acc[3 + b] = vfma<V>(wa, J[0][3 + b], acc[3 + b]);
        acc[8 + b] = vfma<V>(wa, J[1][3 + b], acc[8 + b]);
        acc[12 + b] = vfma<V>(wc, J[0][3 + b], vfma<V>(we, J[1][3 + b], acc[12 + b]));
      }
      int k = 15;
#pragma unroll
      for (int p = 0; p < 3; ++p)
#pragma unroll
        for (int q = p; q < 3; ++q) {
          acc[k] = vfma<V>(wd[p], J[0][3 + q], vfma<V>(wf[p], J[1][3 + q], acc[k]));
          ++k;
        }
      acc[21] = vfma<V>(wa, r[0], acc[21]);
      acc[22] = vfma<V>(wa, r[1], acc[22]);
      acc[23] = vfma<V>(wc, r[0], vfma<V>(we, r[1], acc[23]));
#pragma unroll
      for (int b = 0; b < 3; ++b) acc[24 + b] = vfma<V>(wd[b], r[0], vfma<V>(wf[b], r[1], acc[24 + b]));
      acc[27] += rho;
    }
  }
  __device__ static __forceinline__ void item(const T (&x)[5], const Params& P, bool valid, T (&acc)[28]) {
    const bool v1[1] = {valid};
    item<T>(x, P, v1, acc);
  }
};

// ---------------------------------------------------------------- loads

template <typename T, int N>
struct VecOf;
template <>
struct VecOf<double, 1> { using type = double; };
template <>
struct VecOf<double, 2> { using type = double __attribute__((ext_vector_type(2))); };
template <>
struct VecOf<double, 4> { using type = double __attribute__((ext_vector_type(4))); };  // two 16-byte loads
template <>
struct VecOf<double, 8> { using type = double __attribute__((ext_vector_type(8))); };
template <>
struct VecOf<float, 1> { using type = float; };
template <>
struct VecOf<float, 2> { using type = float __attribute__((ext_vector_type(2))); };
template <>
struct VecOf<float, 4> { using type = float __attribute__((ext_vector_type(4))); };
template <>
struct VecOf<float, 8> { using type = float __attribute__((ext_vector_type(8))); };

template <typename T, int N, bool NT>
__device__ __forceinline__ void load_items(const T* p, T (&dst)[N]) {
  using V = typename VecOf<T, N>::type;
  const V* vp = reinterpret_cast<const V*>(p);
  V v;
  if constexpr (NT)
    v = __builtin_nontemporal_load(vp);
  else
    v = *vp;
  if constexpr (N == 1) {
    dst[0] = v;
  } else {
#pragma unroll
    for (int k = 0; k < N; ++k) dst[k] = v[k];
  }
}

// ---------------------------------------------------------------- reduction

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = kWave / 2; off > 0; off >>= 1) v += __shfl_xor(v, off, kWave);
  return v;
}

// Sums NOUT values per lane over the 64 lanes of a wave with a reduce-scatter butterfly: at every step a lane
// keeps one half of its values and trades the other half with its partner (lane ^ 32, ^ 16, …), so the number of
// values halves each time — P + log2(64 / P) cross-lane exchanges in total (P = NOUT rounded up to a power of two)
// instead of 6·NOUT for NOUT independent butterflies.  The cross-lane exchanges (ds_bpermute) are what bounds this
// phase: at 28 values and 8 waves per CU the independent form kept the LDS crossbar busy for ≈ 6-12 µs at the end of
// every launch.  On return lane L holds the wave total of value number  L >> (6 - log2 P)  (lanes that share a value
// number hold the same total).  Fixed order of additions → bit-identical results run to run.
// v_permlane32_swap (rows16 = false): lanes 32-63 of `a` trade places with lanes 0-31 of `b`;
// v_permlane16_swap (rows16 = true): the odd 16-lane rows of `a` trade places with the even rows of `b`.
__device__ __forceinline__ void swap_lane_halves(double& a, double& b, bool rows16) {
  const unsigned long long ab = __double_as_longlong(a), bb = __double_as_longlong(b);
  unsigned int a0 = (unsigned int)ab, a1 = (unsigned int)(ab >> 32), b0 = (unsigned int)bb, b1 = (unsigned int)(bb >> 32);
  if (rows16) {
    const auto r0 = __builtin_amdgcn_permlane16_swap(a0, b0, false, false);
    const auto r1 = __builtin_amdgcn_permlane16_swap(a1, b1, false, false);
    a0 = r0[0], b0 = r0[1], a1 = r1[0], b1 = r1[1];
  } else {
    const auto r0 = __builtin_amdgcn_permlane32_swap(a0, b0, false, false);
    const auto r1 = __builtin_amdgcn_permlane32_swap(a1, b1, false, false);
    a0 = r0[0], b0 = r0[1], a1 = r1[0], b1 = r1[1];
  }
  a = __longlong_as_double(((unsigned long long)a1 << 32) | a0);
  b = __longlong_as_double(((unsigned long long)b1 << 32) | b0);
}

template <int NOUT>
struct WaveScatter {
  static constexpr int kP = NOUT > 16 ? 32 : (NOUT > 8 ? 16 : 8);
  static constexpr int kLog2P = kP == 32 ? 5 : (kP == 16 ? 4 : 3);
  static constexpr int kShift = 6 - kLog2P;  // value number of lane L is L >> kShift
  __device__ static __forceinline__ double run(const double (&acc)[NOUT]) {
    double v[kP];
#pragma unroll
    for (int k = 0; k < kP; ++k) v[k] = k < NOUT ? acc[k] : 0.0;
    const int lane = threadIdx.x & (kWave - 1);
#pragma unroll
    for (int s = 0; s < kLog2P; ++s) {
      const int mask = 32 >> s;
      const int half = kP >> (s + 1);
      const bool upper = (lane & mask) != 0;
#pragma unroll
      for (int j = 0; j < half; ++j) {
#ifndef NOS_SCATTER_BPERMUTE
        // gfx950 half exchanges: after the swap the two registers hold, in every lane, this lane's kept value and its
        // partner's copy of the same value — 2 swaps + 1 add per exchange instead of 2 ds_bpermute + 4 selects + 1 add,
        // same operands, same bits (the reduce was VALU-issue bound: ≈ 1.8 µs of every resident LM iteration)
        if (mask >= 16) {
          double a = v[j], b = v[j + half];
          swap_lane_halves(a, b, mask == 16);
          v[j] = a + b;
          continue;
        }
#endif
        const double send = upper ? v[j] : v[j + half];
        const double keep = upper ? v[j + half] : v[j];
        v[j] = keep + __shfl_xor(send, mask, kWave);
      }
    }
#pragma unroll
    for (int mask = (32 >> kLog2P); mask > 0; mask >>= 1) v[0] += __shfl_xor(v[0], mask, kWave);
    return v[0];
  }
};

// Sums acc[] over the block and writes one row of kOut doubles.  Fixed order:
// reduce-scatter butterfly inside a wave, then waves 0..W-1.
template <int NOUT, int BLOCK>
__device__ __forceinline__ void block_reduce_store(const double (&acc)[NOUT], double* row, bool write_through) {
  constexpr int kWaves = BLOCK / kWave;
  __shared__ double lds[kWaves][NOUT];
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = threadIdx.x / kWave;
  {
    const double s = WaveScatter<NOUT>::run(acc);
    constexpr int kShift = WaveScatter<NOUT>::kShift;
    const int k = lane >> kShift;
    if ((lane & ((1 << kShift) - 1)) == 0 && k < NOUT) lds[wave][k] = s;
  }
  __syncthreads();
  if (threadIdx.x < NOUT) {
    double s = 0.0;
#pragma unroll
    for (int wv = 0; wv < kWaves; ++wv) s += lds[wv][threadIdx.x];
    if (write_through)  // sc1 store: leaves the XCD's L2 at once (hand-off without a release fence)
      __hip_atomic_store(row + threadIdx.x, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else
      row[threadIdx.x] = s;
  }
}

// Same reduction, result returned instead of stored: thread k < NOUT of the block gets block total number k.
template <int NOUT, int BLOCK>
__device__ __forceinline__ double block_reduce_value(const double (&acc)[NOUT]) {
  constexpr int kWaves = BLOCK / kWave;
  __shared__ double lds_v[kWaves][NOUT];
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = threadIdx.x / kWave;
  {
    const double s = WaveScatter<NOUT>::run(acc);
    constexpr int kShift = WaveScatter<NOUT>::kShift;
    const int k = lane >> kShift;
    if ((lane & ((1 << kShift) - 1)) == 0 && k < NOUT) lds_v[wave][k] = s;
  }
  __syncthreads();
  double s = 0.0;
  if (threadIdx.x < NOUT) {
#pragma unroll
    for (int wv = 0; wv < kWaves; ++wv) s += lds_v[wv][threadIdx.x];
  }
  return s;
}

// A value and the sequence number it belongs to in ONE naturally aligned 16-byte unit, written and read with single
// 16-byte cache-bypassing accesses (global_store / global_load_dwordx4 sc1): the reader sees either the old pair or the
// new pair, so "has it arrived" and "what is it" are one memory round trip, and the writer needs no drain between data
// and flag (MI355X_MICROARCH.md lists 16-byte sc1 flag stores / polls among the measured-valid hand-off forms).
struct alignas(16) TaggedUnit {
  double value;
  unsigned long long seq;
};
__device__ __forceinline__ void tagged_store(TaggedUnit* p, double value, unsigned long long seq) {
  using V4 = unsigned int __attribute__((ext_vector_type(4)));
  const unsigned long long bits = __double_as_longlong(value);
  V4 w;
  w[0] = (unsigned int)(bits & 0xFFFFFFFFull);
  w[1] = (unsigned int)(bits >> 32);
  w[2] = (unsigned int)(seq & 0xFFFFFFFFull);
  w[3] = (unsigned int)(seq >> 32);
  // (s_nop 1 inside the string: a 16-byte store reads its data registers up to two states after issue and hipcc pads
  //  nothing around inline asm — without it the next instruction may overwrite them; cdna_hip_programming.md §5.7 item 1)
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(w) : "memory");
}
// The same unit with a PLAIN store: the line stays in the storing CU's XCD L2, where a reader on the SAME XCD finds it with its
// sc1 load (L1-bypassing, L2-served) without the trip through the fabric; a reader on another XCD never sees it.
__device__ __forceinline__ void tagged_store_plain(TaggedUnit* p, double value, unsigned long long seq) {
  using V4 = unsigned int __attribute__((ext_vector_type(4)));
  const unsigned long long bits = __double_as_longlong(value);
  V4 w;
  w[0] = (unsigned int)(bits & 0xFFFFFFFFull);
  w[1] = (unsigned int)(bits >> 32);
  w[2] = (unsigned int)(seq & 0xFFFFFFFFull);
  w[3] = (unsigned int)(seq >> 32);
  asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" ::"v"(p), "v"(w) : "memory");
}
// XCD (XCC) this wave runs on, 0…7
__device__ __forceinline__ unsigned int xcc_id() {
  return __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 0xFu;  // hwreg(HW_REG_XCC_ID, 0, 4)
}
__device__ __forceinline__ TaggedUnit tagged_load(const TaggedUnit* p) {
  using V4 = unsigned int __attribute__((ext_vector_type(4)));
  V4 w;
  asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(w) : "v"(p) : "memory");
  TaggedUnit u;
  u.value = __longlong_as_double(((unsigned long long)w[1] << 32) | w[0]);
  u.seq = ((unsigned long long)w[3] << 32) | w[2];
  return u;
}

// ---------------------------------------------------------------- in-launch final reduce

// When `counter` is set the grid finishes its own reduction: every block publishes its row,
// takes a ticket, and the block that draws the last ticket sums all rows in fixed order and
// writes the result (device pointer and/or host-mapped pinned pointer), then bumps a host
// visible sequence word.  This removes the dependent 1-block kernel and the D2H memcpy from
// the per-iteration critical path.  Hand-off protocol = /opt/skills/guides
// cdna_hip_programming.md Guideline 16: storing wave drains (vmcnt(0)) → one lane
// agent-scope release → asm vmcnt(0) → relaxed agent atomic ticket;  last block: ticket
// value is the "poll", one lane agent-scope acquire → vmcnt(0) → barrier → plain loads.
// Device-resident Levenberg-Marquardt loop (nos_*_solve): the pose lives in device memory, every launch reads it
// from there instead of from its kernel arguments, and the workgroup that finishes the reduction also runs the
// loop body of the reference (damped 6x6 solve, pose update, convergence tests, λ schedule — the same
// nos_host::LmAdvance6 / LmAdvance3 the host loop calls) and leaves the new pose for the next launch.  The host
// only keeps a few launches in flight and watches a log in pinned memory, so consecutive iterations run
// back-to-back on the GPU without a host round trip in between.
struct LmDevice {
  nos_host::LmState st;
  nos_host::LmSettings settings;
};

// Layout (in doubles) of one entry of the pinned host log the loop writes per iteration.
constexpr int kLogOut = 0;        // [0..27] the sums of this iteration
constexpr int kLogR = 32;         // [32..40] pose after the update
constexpr int kLogT = 41;         // [41..43]
constexpr int kLogLambda = 44;
constexpr int kLogPrevCost = 45;
constexpr int kLogCost = 46;
constexpr int kLogIteration = 47;
constexpr int kLogDone = 48;
constexpr int kLogOk = 49;
constexpr int kLogExecuted = 62;  // single-workgroup solve: iterations executed inside the launch
constexpr int kLogEntryDoubles = 64;

// In-kernel all-reduce of the per-GPU sums for one-process-per-GPU runs on one node (nos_ctx_comm_init_shm): a mailbox
// in host memory shared by the ranks (POSIX shm, mapped into every rank's GPU address space).  The workgroup that
// finished its GPU's sums stores them into its own slot followed by a round number (system-scope release), polls the
// round numbers of all ranks (one lane per rank) and adds the slots in rank order — every rank gets identical bits,
// with no extra kernel launch, no RCCL call and no host step in the iteration.  Slots are double buffered by round
// parity: a rank can be at most one round ahead of the slowest reader.  The wait is bounded (kMailboxTimeoutTicks = 8 s of
// the 100 MHz wall clock): on a time-out the launch flags an error instead of spinning for ever.
constexpr int kMailSlotDoubles = 64;                         // one slot: [0..27] sums, [32] round number; 512 bytes
constexpr unsigned long long kMailboxTimeoutTicks = 800000000ull;  // 8 s
struct Mailbox {
  double* base;                 // device address of the shared mailbox: [n_ranks][2][kMailSlotDoubles]; null = no exchange
  double* const* peers;         // device-memory form: peers[r] = rank r's [n_ranks][2][kMailSlotDoubles] buffer (fine-grained
                                // device memory, peers[rank] is local); null = the slots behind `base` (host memory)
  unsigned long long* round;    // device word: rounds completed by this rank (all ranks run the same sequence)
  unsigned int* error_host;     // host-mapped word set to 1 when a peer did not arrive in time
  int n_ranks;
  int rank;
};

struct FusedFinal {
  unsigned int* counter;           // device words (top counter at [0], 8 group counters at [32 * (1 + g)]), all 0
                                   // before the launch and reset to 0 by the blocks that complete them
  double* out_dev;                 // device result (may be null)
  double* out_host;                // host-mapped pinned result (may be null)
  unsigned long long* seq_host;    // host-mapped pinned sequence word (may be null)
  unsigned long long seq;          // value stored to *seq_host when the result is complete
  int write_through;               // 1: rows travel as sc1 stores / sc1 loads instead of release / acquire fences
  LmDevice* lm;                    // device-resident loop state: pose source of this launch (null = pose from arguments)
  int lm_step;                     // 1: the finishing workgroup also advances the loop; 0: a separate kernel does
  const Mailbox* mail;             // cross-rank exchange of the sums inside the launch: descriptor in device memory,
                                   // read by the finishing workgroup only (null: none) — kept out of the kernel
                                   // arguments proper because every argument stays in scalar registers through the loop
};

// The exchange itself; called by the first NOUT threads of one workgroup (wave 0 included: NOUT <= 64 and
// n_ranks <= 64) with `tot` = this GPU's sum number threadIdx.x.  Contains block-wide barriers: every thread of the
// block must call it.  Returns the sum over ranks (valid in threads < NOUT).
template <int NOUT>
__device__ __forceinline__ double mailbox_allreduce(const Mailbox& mb, double tot, bool* failed = nullptr) {
  __shared__ unsigned long long s_round;
  __shared__ int s_failed;
  if (threadIdx.x == 0) {
    s_round = *mb.round + 1ull;
    s_failed = 0;
  }
  __syncthreads();
  const unsigned long long round = s_round;
  const size_t parity = size_t(round & 1ull);
#ifdef NOS_LM_TIMING
  unsigned long long tm0 = wall_clock64(), tm1 = 0, tm2 = 0;
#endif
  // Host-memory form: every rank stores into ITS slot of the one shared segment and polls the others' slots there.
  // Device-memory form: every rank PUSHES its slot into every peer's buffer (remote stores over the fabric; its own buffer
  // included) and polls only its own, local memory — the same slots, the same round numbers, the same rank-order sum.
  const size_t my_slot = (size_t(mb.rank) * 2 + parity) * kMailSlotDoubles;
  const bool pushed = mb.peers != nullptr;
  double* const local = pushed ? mb.peers[mb.rank] : mb.base;  // where this rank polls and sums
  if (threadIdx.x < NOUT) {
    if (pushed) {
      for (int p = 0; p < mb.n_ranks; ++p)
        __hip_atomic_store(mb.peers[p] + my_slot + threadIdx.x, tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    } else {
      __hip_atomic_store(mb.base + my_slot + threadIdx.x, tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
  if (threadIdx.x < kWave) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the sums left through lanes of wave 0
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (pushed) {
      if (int(threadIdx.x) < mb.n_ranks)  // lane p raises this rank's flag in peer p's buffer
        __hip_atomic_store(reinterpret_cast<unsigned long long*>(mb.peers[threadIdx.x] + my_slot + 32), round, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_SYSTEM);
    } else if (threadIdx.x == 0) {
      __hip_atomic_store(reinterpret_cast<unsigned long long*>(mb.base + my_slot + 32), round, __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_SYSTEM);
    }
#ifdef NOS_LM_TIMING
    tm1 = wall_clock64();
#endif
    if (int(threadIdx.x) < mb.n_ranks) {
      const unsigned long long* flag = reinterpret_cast<const unsigned long long*>(
          local + (size_t(threadIdx.x) * 2 + parity) * kMailSlotDoubles + 32);
      const unsigned long long deadline = wall_clock64() + kMailboxTimeoutTicks;
      while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != round) {
        if (wall_clock64() > deadline) {  // a peer is missing: report, do not hang
          __hip_atomic_store(mb.error_host, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          s_failed = 1;
          break;
        }
        __builtin_amdgcn_s_sleep(2);
      }
      // No system-scope acquire here: on this part it invalidates the whole L2 (measured 45-110 µs per call); every
      // load of the exchanged values below is itself a system-scope (cache-bypassing) load issued after the barrier.
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
#ifdef NOS_LM_TIMING
    tm2 = wall_clock64();
#endif
  }
  __syncthreads();
  double sum = 0.0;
  if (threadIdx.x < NOUT) {
    for (int r = 0; r < mb.n_ranks; ++r)  // rank order: the same additions on every rank
      sum += __hip_atomic_load(local + (size_t(r) * 2 + parity) * kMailSlotDoubles + threadIdx.x, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_SYSTEM);
  }
  if (threadIdx.x == 0) *mb.round = round;
#ifdef NOS_LM_TIMING
  if (threadIdx.x == 0) {
    const unsigned long long tm3 = wall_clock64() + (unsigned long long)(sum * 0.0);
    mb.base[(size_t(mb.rank) * 2) * kMailSlotDoubles + 40] = double(tm1 - tm0);
    mb.base[(size_t(mb.rank) * 2) * kMailSlotDoubles + 41] = double(tm2 - tm1);
    mb.base[(size_t(mb.rank) * 2) * kMailSlotDoubles + 42] = double(tm3 - tm2);
  }
#endif
  if (failed != nullptr) *failed = s_failed != 0;
  return sum;
}

__device__ __forceinline__ double uniform_load(const double* p) {
  // the address is the same for every lane of the grid: keep the value in scalar registers
  const double v = *p;
  const unsigned long long u = __double_as_longlong(v);
  const unsigned int lo = __builtin_amdgcn_readfirstlane((unsigned int)(u & 0xFFFFFFFFull));
  const unsigned int hi = __builtin_amdgcn_readfirstlane((unsigned int)(u >> 32));
  return __longlong_as_double(((unsigned long long)hi << 32) | lo);
}

template <typename T>
__device__ __forceinline__ void set_pose(Ndt6Params<T>& P, const LmDevice* lm) {
#pragma unroll
  for (int k = 0; k < 9; ++k) P.R[k] = T(uniform_load(&lm->st.R[k]));
#pragma unroll
  for (int k = 0; k < 3; ++k) P.t[k] = T(uniform_load(&lm->st.t[k]));
}
template <typename T>
__device__ __forceinline__ void set_pose(ReprojParams<T>& P, const LmDevice* lm) {
#pragma unroll
  for (int k = 0; k < 9; ++k) P.R[k] = T(uniform_load(&lm->st.R[k]));
#pragma unroll
  for (int k = 0; k < 3; ++k) P.t[k] = T(uniform_load(&lm->st.t[k]));
}
template <typename T>
__device__ __forceinline__ void set_pose(Ndt3Params<T>& P, const LmDevice* lm) {
#pragma unroll
  for (int k = 0; k < 4; ++k) P.R2[k] = T(uniform_load(&lm->st.R[k]));
#pragma unroll
  for (int k = 0; k < 2; ++k) P.t2[k] = T(uniform_load(&lm->st.t[k]));
}

// Launch prologue of the device-resident loop.  Returns true if this launch has nothing to do (the loop already
// finished): block 0 then only forwards the sequence word so the host's wait completes.
template <typename Params>
__device__ __forceinline__ bool lm_prologue(const FusedFinal& fin, Params& P) {
  if (fin.lm == nullptr) return false;
  // pose and the done flag are fetched together (one memory round trip at the head of the launch)
  Params Q = P;
  set_pose(Q, fin.lm);
  const int done = __builtin_amdgcn_readfirstlane(*reinterpret_cast<const int*>(&fin.lm->st.done));
  if (done != 0) {
    if (blockIdx.x == 0 && threadIdx.x == 0 && fin.seq_host != nullptr)
      __hip_atomic_store(fin.seq_host, fin.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    return true;
  }
  P = Q;
  return false;
}

// ---------------------------------------------------------------- the loop body on the device: one lane, two real functions
//
// nos_host::LmAdvance6 / LmAdvance3 (csrc/host/nos_lm.hpp; the reference's loop body, MDM/..._analytic_simd.cc:78-102) as
// every device form of the loop runs it (launch per iteration, stand-alone step kernel, single workgroup, one-launch
// resident / streamed).  Round 2 had that function inlined into the kernels; unrolled for instruction-level parallelism it
// wanted ≈ 230 VGPRs (a 6x6 system, its factor, the sums, the state), which pinned every kernel that contained it at the
// 256-register ceiling and made the streaming kernels spill around it.  Now it is ONE NOINLINE function called by lane 0 —
// the damped solve (nos_host::DampedStep itself), a scheduling barrier, then the O(1) rest (pose update, convergence tests,
// λ schedule): 117 VGPRs — so a kernel's own allocation is set by its hot loop and what it keeps alive across the call
// (the streaming kernels: the prefetched first chunk of the next iteration).  A wave-parallel elimination (one matrix
// element per lane, pivots by v_readlane, operands by ds_bpermute) was built and measured first: it needs only ≈ 40
// registers but turns the step into ONE dependent chain — 2.45 µs against the 1.5 µs of the single lane's interleaved
// chains (profiles/r03_lm_step_forms.txt) — so the single lane stayed.
// `tot` (the NOUT sums) and `lmd` (loop state and settings) are LDS.
using LdsDouble = __attribute__((address_space(3))) double;
using LdsLmDevice = __attribute__((address_space(3))) LmDevice;
constexpr int kLmTotDoubles(int n_out) { return n_out; }

__device__ __forceinline__ void wave_sync_lds() {
  // LDS instructions of one wave execute in issue order; this only keeps the compiler from moving accesses across
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

#ifdef NOS_LM_TIMING
__shared__ unsigned long long s_step_cycles[4];  // probe build: shader-clock cycles of the two halves of the step
#define NOS_STEP_TICK(slot_)                     \
  {                                              \
    const unsigned long long now_ = clock64();  \
    s_step_cycles[slot_] += now_ - tick_;        \
    tick_ = now_;                                \
  }
#else
#define NOS_STEP_TICK(slot_)
#endif

// cos(x) and sin(x) / x as power series in v = x^2, for v < 1/256 (|x| < 1/16): six terms each, first omitted term < 1e-23
__device__ __forceinline__ void series_cos_sinc(double v, double* c_out, double* sinc_out) {
  double c = -1.0 / 3628800.0, sc = -1.0 / 39916800.0;
  c = __builtin_fma(c, v, 1.0 / 40320.0), sc = __builtin_fma(sc, v, 1.0 / 362880.0);
  c = __builtin_fma(c, v, -1.0 / 720.0), sc = __builtin_fma(sc, v, -1.0 / 5040.0);
  c = __builtin_fma(c, v, 1.0 / 24.0), sc = __builtin_fma(sc, v, 1.0 / 120.0);
  c = __builtin_fma(c, v, -0.5), sc = __builtin_fma(sc, v, -1.0 / 6.0);
  *c_out = __builtin_fma(c, v, 1.0);
  *sinc_out = __builtin_fma(sc, v, 1.0);
}

// First half: δ = -(H with its diagonal scaled by 1 + λ)^-1 g — nos_host::DampedStep, the host loop's own function
// (right-looking LDLT with reciprocal pivots; on the device the reciprocal is the hardware seed + two Newton steps).
template <int NOUT, int N>
__device__ __forceinline__ bool lm_solve_lane(const LdsDouble* tot, double lambda, double (&step)[N]) {
  double out[NOUT - 1];
#pragma unroll
  for (int k = 0; k < NOUT - 1; ++k) out[k] = tot[k];
  return nos_host::DampedStep<N>(out, lambda, step);
}

// Second half: pose update, the two convergence tests (after the update, as in the reference), λ schedule — the rest of
// nos_host::LmAdvance6 / LmAdvance3 with the transcendental part written for a lone GPU lane, where every fp64 instruction
// costs 8 cycles whatever it computes: the exponential map's two factors are even in θ and are summed as power series for
// θ < 1/8 (no square root, no argument reduction, no division; sincos beyond), normalisation by reciprocal square root (seed
// + two Newton steps), the tests on squared norms.  Within an ulp or two of the host loop's libm calls per operation.
template <int NOUT, int N>
__device__ __forceinline__ void lm_finish_lane(const LdsDouble* tot, LdsLmDevice* lmd, const double (&step)[N], bool solved) {
#ifdef NOS_LM_TIMING
  unsigned long long tick_ = clock64();
#endif
  const double lambda = lmd->st.lambda, previous_cost = lmd->st.previous_cost, cost = tot[NOUT - 1];
  const int iteration = lmd->st.iteration;
  const int max_iterations = lmd->settings.max_iterations, float_schedule = lmd->settings.float_schedule;
  const double gtol = lmd->settings.gradient_tolerance, ptol = lmd->settings.parameter_tolerance;
  double g2 = 0.0, s2 = 0.0;
#pragma unroll
  for (int r = 0; r < N; ++r) {
    const double gr = tot[N * (N + 1) / 2 + r];
    g2 = __builtin_fma(gr, gr, g2);
    s2 = __builtin_fma(step[r], step[r], s2);
  }
  lmd->st.cost = cost;
  if (!solved) {
    lmd->st.ok = 0;
    lmd->st.done = 1;
    NOS_STEP_TICK(1)
    return;
  }
  if constexpr (N == 6) {
#pragma unroll
    for (int r = 0; r < 3; ++r) lmd->st.t[r] += step[r];
    // ExpQuat (MahalanobisDistanceMinimizer::ComputeQuaternion, MDM/mahalanobis_distance_minimizer.cc:20-33):
    //   theta < 1e-6: (1, w / 2);  else (cos(theta / 2), sin(theta / 2) / theta * w)
    const double wx = step[3], wy = step[4], wz = step[5];
    const double th2 = __builtin_fma(wx, wx, __builtin_fma(wy, wy, wz * wz));
    double dw, kk;
    if (th2 < 1.0 / 64.0) {
      double c, sc;
      series_cos_sinc(0.25 * th2, &c, &sc);
      const bool tiny = !(th2 >= 1e-12);  // theta < 1e-6: the reference's un-normalised small-angle form
      dw = tiny ? 1.0 : c;
      kk = tiny ? 0.5 : 0.5 * sc;
    } else {
      const double inv_th = fast_rsqrt<double>(th2);  // 1 / theta
      double sn, cs;
      sincos(0.5 * (th2 * inv_th), &sn, &cs);
      dw = cs;
      kk = sn * inv_th;
    }
    const double dx = kk * wx, dy = kk * wy, dz = kk * wz;
    // q <- normalize(q (x) dq)
    const double aw = lmd->st.q.w, ax = lmd->st.q.x, ay = lmd->st.q.y, az = lmd->st.q.z;
    const double rw = aw * dw - ax * dx - ay * dy - az * dz;
    const double rx = aw * dx + ax * dw + ay * dz - az * dy;
    const double ry = aw * dy + ay * dw + az * dx - ax * dz;
    const double rz = aw * dz + az * dw + ax * dy - ay * dx;
    const double inv_n = fast_rsqrt<double>((rx * rx + ry * ry) + (rz * rz + rw * rw));
    nos_host::Quat q;
    q.w = rw * inv_n, q.x = rx * inv_n, q.y = ry * inv_n, q.z = rz * inv_n;
    lmd->st.q.w = q.w, lmd->st.q.x = q.x, lmd->st.q.y = q.y, lmd->st.q.z = q.z;
    double R[9];
    nos_host::QuatToMatrix(q, R);
#pragma unroll
    for (int r = 0; r < 9; ++r) lmd->st.R[r] = R[r];
  } else {
    lmd->st.t[0] += step[0];
    lmd->st.t[1] += step[1];
    double c, sn;
    if (step[2] * step[2] < 1.0 / 256.0) {  // |dtheta| < 1/16
      double sc;
      series_cos_sinc(step[2] * step[2], &c, &sc);
      sn = step[2] * sc;
    } else {
      sincos(step[2], &sn, &c);
    }
    const double a = lmd->st.R[0], b = lmd->st.R[1], dd = lmd->st.R[2], e = lmd->st.R[3];
    lmd->st.R[0] = a * c + b * sn;  // linear <- linear * Rot2(dtheta)   (Isometry2d::rotate)
    lmd->st.R[1] = b * c - a * sn;
    lmd->st.R[2] = dd * c + e * sn;
    lmd->st.R[3] = e * c - dd * sn;
  }
  if ((ptol > 0.0 && s2 < ptol * ptol) || (gtol > 0.0 && g2 < gtol * gtol)) {  // |step| < ptol || |g| < gtol
    lmd->st.done = 1;
  } else {
    if (float_schedule) {
      lmd->st.lambda = nos_host::NextLambdaFloat(lambda, cost, previous_cost);
      lmd->st.previous_cost = double(float(cost));
    } else {
      lmd->st.lambda = nos_host::NextLambda(lambda, cost, previous_cost);
      lmd->st.previous_cost = cost;
    }
    lmd->st.iteration = iteration + 1;
    if (iteration + 1 >= max_iterations) lmd->st.done = 1;
  }
  NOS_STEP_TICK(1)
}

// The loop body; call with ONE lane.
template <int NOUT>
__device__ __attribute__((noinline)) void lm_step_lane(const LdsDouble* tot, LdsLmDevice* lmd) {
  constexpr int N = NOUT == 28 ? 6 : 3;
#ifdef NOS_LM_TIMING
  unsigned long long tick_ = clock64();
#endif
  double step[N];
  const bool solved = lm_solve_lane<NOUT, N>(tot, lmd->st.lambda, step);
  NOS_STEP_TICK(0)
  // the scheduler must not weave the two halves into each other: together they would want ≈ 230 registers again
  __builtin_amdgcn_sched_barrier(0);
  lm_finish_lane<NOUT, N>(tot, lmd, step, solved);
}

// LDS address of a __shared__ object (the generic pointer HIP hands out, narrowed back to its address space)
template <typename T>
__device__ __forceinline__ __attribute__((address_space(3))) T* lds_ptr(T* p) {
  return (__attribute__((address_space(3))) T*)p;
}

#ifdef NOS_LM_TIMING
#define NOS_LM_STAMP(slot) \
  if (entry_host != nullptr) __hip_atomic_store(entry_host + 50 + (slot), double(wall_clock64()), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)
#else
#define NOS_LM_STAMP(slot)
#endif

// One lane: the state after a step → device memory for the next launch and, if `entry_host` is set, the pinned log entry
// the host is waiting for.
__device__ __forceinline__ void lm_publish(const nos_host::LmState& st, LmDevice* lm, double* entry_host) {
  lm->st = st;
  if (entry_host != nullptr) {
#pragma unroll
    for (int k = 0; k < 9; ++k)
      __hip_atomic_store(entry_host + kLogR + k, st.R[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
#pragma unroll
    for (int k = 0; k < 3; ++k)
      __hip_atomic_store(entry_host + kLogT + k, st.t[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(entry_host + kLogLambda, st.lambda, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(entry_host + kLogPrevCost, st.previous_cost, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(entry_host + kLogCost, st.cost, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(entry_host + kLogIteration, double(st.iteration), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(entry_host + kLogDone, double(st.done), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(entry_host + kLogOk, double(st.ok), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

template <int NOUT, int BLOCK>
__device__ __forceinline__ void finish_in_last_block(const double* partials, const FusedFinal& fin,
                                                     unsigned long long t_start = 0) {
  (void)t_start;
  __shared__ unsigned int s_last;
  constexpr int kCols = 32;
  constexpr int kSlices = BLOCK / kCols;
  __shared__ double red[kSlices][kCols];
  // the row was stored by lanes 0..NOUT-1 of wave 0; thread 0 is in that wave
  if (threadIdx.x < kWave) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  // Two forms of the same hand-off (both listed as valid in the guide):
  //   fences:        plain row stores → drain → agent release → ticket;  last block: agent acquire → plain loads
  //   write-through: sc1 row stores → drain → ticket;                     last block: sc1 loads (bypass L1), no fences
  // The second is used for the one-workgroup-per-CU geometry it was measured for (MI355X_MICROARCH.md,
  // "Hand-offs measured with sc1 loads in place of the acquire", row 1) and saves both fences (≈3 µs).
  const bool wt = fin.write_through != 0;
  if (threadIdx.x == 0) {
    if (!wt) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    // Two-level ticket: one device-scope counter saturates at ≈ 88 arrivals/µs (guide, "dequeue" / "fanin"
    // rows: 256 arrivals ≈ 2.9 µs), so blocks first arrive on one of 8 group counters (group = blockIdx mod 8,
    // i.e. blocks that share an XCD under round-robin placement — used for speed only, any grouping is correct);
    // the last arriver of a group resets it and arrives on the top counter; the last of those finishes.  Every
    // block has released (or written through and drained) its row BEFORE its first arrival, the atomics execute
    // in arrival order at the memory side and each later arrival is issued only after the earlier one returned
    // (data dependence), so when the top ticket reads "last" every row is already out of the writers' L2s.
    unsigned int last = 0u;
    const unsigned int group = blockIdx.x & 7u;
    const unsigned int group_size = (gridDim.x - group + 7u) >> 3;
    const unsigned int n_groups = gridDim.x < 8u ? gridDim.x : 8u;
    unsigned int* group_counter = fin.counter + 32u * (1u + group);  // 128 bytes apart
    const unsigned int t1 = __hip_atomic_fetch_add(group_counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (t1 == group_size - 1u) {
      __hip_atomic_store(group_counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // ready for the next launch
      const unsigned int t2 = __hip_atomic_fetch_add(fin.counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      last = (t2 == n_groups - 1u) ? 1u : 0u;
    }
    if (last && !wt) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    s_last = last;
  }
  __syncthreads();
  if (s_last == 0u) return;  // block-uniform
#ifdef NOS_LM_TIMING
  if (threadIdx.x == 0 && fin.out_host != nullptr && fin.lm != nullptr) {
    __hip_atomic_store(fin.out_host + 50, double(t_start), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(fin.out_host + 51, double(wall_clock64()), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
#endif
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");  // no instruction: keeps the loads below the ticket
  const int col = threadIdx.x % kCols;
  const int slice = threadIdx.x / kCols;
  // the loop state is requested now so that its latency hides behind the row sums
  const bool step_here = fin.lm != nullptr && fin.lm_step != 0;  // grid-uniform
  __shared__ double s_lmd_raw[(sizeof(LmDevice) + 7) / 8];       // the loop state while wave 0 advances it
  LmDevice& s_lmd = *reinterpret_cast<LmDevice*>(s_lmd_raw);
  constexpr int kLmdWords = int(sizeof(LmDevice) / sizeof(double));
  static_assert(sizeof(LmDevice) % sizeof(double) == 0, "LmDevice must be a whole number of doubles");
  double lmd_pre[kLmdWords];  // thread 0: loop state + settings, in flight while the rows are summed
  if (step_here && threadIdx.x == 0) {
    const double* src = reinterpret_cast<const double*>(fin.lm);
#pragma unroll
    for (int k = 0; k < kLmdWords; ++k) lmd_pre[k] = src[k];
  }
  // Thread (slice, col) adds rows slice, slice + S, slice + 2S, … in that order.  Sixteen row loads are put in flight
  // before the first add: the loop is latency bound (each row comes from another XCD's L2 / memory).
  constexpr int kUnroll = 16;
  double s = 0.0;
  if (col < NOUT) {
    const double* p = partials + col;
    auto sum_rows = [&](auto load) {
      for (uint32_t r = slice; r < gridDim.x; r += kUnroll * kSlices) {
        double v[kUnroll];
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) {
          const uint32_t rr = r + u * kSlices;
          const double x = load(p + size_t(rr < gridDim.x ? rr : r) * NOUT);  // clamped address, value masked below
          v[u] = rr < gridDim.x ? x : 0.0;
        }
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) s += v[u];
      }
    };
    if (wt)
      sum_rows([](const double* q) { return __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); });
    else
      sum_rows([](const double* q) { return *q; });
  }
  red[slice][col] = s;
  __syncthreads();
  __shared__ double s_tot[kLmTotDoubles(NOUT)];
  double tot = 0.0;
  if (threadIdx.x < NOUT) {
#pragma unroll
    for (int sl = 0; sl < kSlices; ++sl) tot += red[sl][threadIdx.x];
  }
  bool exchange_failed = false;
  if (fin.mail != nullptr) {  // grid-uniform branch
    const Mailbox mb = *fin.mail;
    tot = mailbox_allreduce<NOUT>(mb, tot, &exchange_failed);
  }
  if (threadIdx.x < NOUT) {
    if (fin.out_dev != nullptr) fin.out_dev[threadIdx.x] = tot;
    if (fin.out_host != nullptr)
      __hip_atomic_store(fin.out_host + threadIdx.x, tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (step_here) s_tot[threadIdx.x] = tot;
  }
  if (step_here) {
    if (threadIdx.x == 0) {
#pragma unroll
      for (int k = 0; k < kLmdWords; ++k) s_lmd_raw[k] = lmd_pre[k];
      if (exchange_failed) {  // a peer never arrived: stop the loop here, the host reports the error
        s_lmd.st.ok = 0;
        s_lmd.st.done = 1;
        fin.lm->st = s_lmd.st;
        if (fin.out_host != nullptr) {
          __hip_atomic_store(fin.out_host + kLogDone, 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          __hip_atomic_store(fin.out_host + kLogOk, 0.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
      }
    }
    __syncthreads();
    if (threadIdx.x == 0 && !exchange_failed) {
      double* entry_host = fin.out_host;
      (void)entry_host;
      NOS_LM_STAMP(2);
      lm_step_lane<NOUT>(lds_ptr(s_tot), lds_ptr(&s_lmd));
      NOS_LM_STAMP(3);
      lm_publish(s_lmd.st, fin.lm, fin.out_host);
      NOS_LM_STAMP(4);
    }
  }
  if (threadIdx.x < kWave) {
    // results leave through lanes 0..NOUT-1 of wave 0: drain them, then one lane publishes
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (threadIdx.x == 0) {
      __hip_atomic_store(fin.counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // ready for the next launch
      if (fin.seq_host != nullptr) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");  // system scope: results before the sequence word
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_store(fin.seq_host, fin.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
  }
}

// ---------------------------------------------------------------- the assemble kernel

// Grid-stride over chunks of BLOCK*ITEMS correspondences.  `n_chunks * BLOCK * ITEMS`
// must equal L.n_padded and the tile size must be a multiple of BLOCK*ITEMS (checked on
// the host before launch).
// PREFETCH: 0 = the loads of a chunk, then its math; 1 = the NEXT chunk's loads are issued before the current chunk is
// evaluated; 2 = two chunks ahead (twice the bytes in flight per lane while the item math runs).
template <typename Problem, typename T, int ITEMS, int BLOCK, int MINW, bool NT, int PREFETCH = 0>
__global__ __launch_bounds__(BLOCK, MINW) void assemble_kernel(TiledLayout L,
                                                              typename Problem::Params P,
                                                              uint32_t n_chunks,
                                                              double* __restrict__ partials,
                                                              FusedFinal fin) {
  constexpr int kF = Problem::kFields;
  constexpr int kOut = Problem::kOut;
  constexpr uint32_t kChunk = BLOCK * ITEMS;
  const T* __restrict__ base = static_cast<const T*>(L.base);

#ifdef NOS_LM_TIMING
  const unsigned long long t_start = wall_clock64();
#else
  const unsigned long long t_start = 0;
#endif
  // The pose of this launch (device-resident loop) is awaited only AFTER the loads of the first chunk have been issued:
  // they do not depend on it, and its memory round trip (≈ 1.5 µs at the head of every launch) hides behind them.
#ifdef NOS_LM_TIMING
  unsigned long long t_prologue = t_start;
#endif

  T acc[kOut];
#pragma unroll
  for (int k = 0; k < kOut; ++k) acc[k] = T(0);
  // fp32 with an even number of correspondences per lane CAN run the item math on pairs (packed v_pk_* instructions) — and
  // is SLOWER that way on gfx950: 0.0942 → 0.1068 ms per launch at 10 M (profiles/r02_tune_f32_packed.txt; the guide's
  // constants table prices one v_pk_fma_f32 above two v_fma_f32).  Kept as a compile-time experiment (-DNOS_PACKED_F32).
#ifdef NOS_PACKED_F32
  constexpr bool kPacked = sizeof(T) == 4 && (ITEMS % 2 == 0) && PREFETCH == 0;
#else
  constexpr bool kPacked = false;
#endif
  [[maybe_unused]] float2_t acc2[kPacked ? kOut : 1];
  if constexpr (kPacked) {
#pragma unroll
    for (int k = 0; k < kOut; ++k) acc2[k] = float2_t{0.0f, 0.0f};
  }

  auto chunk_offset = [&](uint32_t c, uint64_t& i0) {
    i0 = uint64_t(c) * kChunk + uint64_t(threadIdx.x) * ITEMS;
    return (i0 >> L.tile_shift) * L.tile_stride + (i0 & L.tile_mask);
  };
  if constexpr (PREFETCH == 3) {
    // ping-pong: two named buffers, the loop unrolled twice — while buffer A is evaluated the loads into B are in flight and
    // vice versa.  No register copies and NO branch around a load in the steady state (the tail is peeled), so the wait
    // before an evaluation is a counted one for the OLDER group of loads only: a wave always has a chunk in flight.
    T xa[kF][ITEMS], xb[kF][ITEMS];
    uint32_t c = blockIdx.x;
    const uint32_t G = gridDim.x;
    uint64_t ia = 0, ib = 0;
    auto issue = [&](uint32_t cc, T (&dst)[kF][ITEMS], uint64_t& i0) {
      const uint64_t off = chunk_offset(cc, i0);
#pragma unroll
      for (int f = 0; f < kF; ++f) load_items<T, ITEMS, NT>(base + off + uint64_t(f) * L.field_stride, dst[f]);
    };
    auto evaluate = [&](const T (&src)[kF][ITEMS], uint64_t i0) {
#pragma unroll
      for (int it = 0; it < ITEMS; ++it) {
        T xi[kF];
#pragma unroll
        for (int f = 0; f < kF; ++f) xi[f] = src[f][it];
        Problem::item(xi, P, (i0 + it) < L.n, acc);
      }
    };
    const bool any = c < n_chunks;  // block-uniform
    if (any) issue(c, xa, ia);
    if (lm_prologue(fin, P)) return;  // grid-uniform
    if (any) {
      while (uint64_t(c) + 2ull * G < n_chunks) {
        issue(c + G, xb, ib);
        __builtin_amdgcn_sched_barrier(0);
        evaluate(xa, ia);
        issue(c + 2 * G, xa, ia);
        __builtin_amdgcn_sched_barrier(0);
        evaluate(xb, ib);
        c += 2 * G;
      }
      const bool has_b = uint64_t(c) + G < n_chunks;  // block-uniform
      if (has_b) issue(c + G, xb, ib);
      evaluate(xa, ia);
      if (has_b) evaluate(xb, ib);
    }
  } else if constexpr (PREFETCH == 2) {
    // two chunks ahead: while chunk c is evaluated the loads of c + G and c + 2G are in flight
    T xa[kF][ITEMS], xb[kF][ITEMS];
    uint32_t c = blockIdx.x;
    uint64_t i0 = 0, i1 = 0;
    if (c < n_chunks) {
      const uint64_t off = chunk_offset(c, i0);
#pragma unroll
      for (int f = 0; f < kF; ++f) load_items<T, ITEMS, NT>(base + off + uint64_t(f) * L.field_stride, xa[f]);
    }
    if (c + gridDim.x < n_chunks) {
      const uint64_t off = chunk_offset(c + gridDim.x, i1);
#pragma unroll
      for (int f = 0; f < kF; ++f) load_items<T, ITEMS, NT>(base + off + uint64_t(f) * L.field_stride, xb[f]);
    }
    if (lm_prologue(fin, P)) return;  // grid-uniform
    for (; c < n_chunks; c += gridDim.x) {
      T xc[kF][ITEMS];
      uint64_t i2 = 0;
      const uint32_t cn = c + 2 * gridDim.x;
      if (cn < n_chunks) {
        const uint64_t off = chunk_offset(cn, i2);
#pragma unroll
        for (int f = 0; f < kF; ++f) load_items<T, ITEMS, NT>(base + off + uint64_t(f) * L.field_stride, xc[f]);
      }
#pragma unroll
      for (int it = 0; it < ITEMS; ++it) {
        T xi[kF];
#pragma unroll
        for (int f = 0; f < kF; ++f) xi[f] = xa[f][it];
        Problem::item(xi, P, (i0 + it) < L.n, acc);
      }
#pragma unroll
      for (int f = 0; f < kF; ++f)
#pragma unroll
        for (int it = 0; it < ITEMS; ++it) {
          xa[f][it] = xb[f][it];
          xb[f][it] = xc[f][it];
        }
      i0 = i1;
      i1 = i2;
    }
  } else if constexpr (PREFETCH == 1) {
    // software pipelined: the 15 loads of the NEXT chunk are issued before the current chunk is
    // evaluated, so a wave always has a chunk in flight while it computes
    T xa[kF][ITEMS];
    uint32_t c = blockIdx.x;
    uint64_t i0 = 0;
    if (c < n_chunks) {
      const uint64_t off = chunk_offset(c, i0);
#pragma unroll
      for (int f = 0; f < kF; ++f) load_items<T, ITEMS, NT>(base + off + uint64_t(f) * L.field_stride, xa[f]);
    }
    if (lm_prologue(fin, P)) return;  // grid-uniform
#ifdef NOS_LM_TIMING
    t_prologue = wall_clock64() + (unsigned long long)(*reinterpret_cast<const T*>(&P) * T(0));  // after the pose arrived
#endif
    for (; c < n_chunks; c += gridDim.x) {
      T xb[kF][ITEMS];
      uint64_t i1 = 0;
      const uint32_t cn = c + gridDim.x;
      if (cn < n_chunks) {
        const uint64_t off = chunk_offset(cn, i1);
#pragma unroll
        for (int f = 0; f < kF; ++f) load_items<T, ITEMS, NT>(base + off + uint64_t(f) * L.field_stride, xb[f]);
      }
#pragma unroll
      for (int it = 0; it < ITEMS; ++it) {
        T xi[kF];
#pragma unroll
        for (int f = 0; f < kF; ++f) xi[f] = xa[f][it];
        Problem::item(xi, P, (i0 + it) < L.n, acc);
      }
#pragma unroll
      for (int f = 0; f < kF; ++f)
#pragma unroll
        for (int it = 0; it < ITEMS; ++it) xa[f][it] = xb[f][it];
      i0 = i1;
    }
  } else {
    bool first = true;
    for (uint32_t c = blockIdx.x; c < n_chunks || first; c += gridDim.x) {
      uint64_t i0 = 0;
      T x[kF][ITEMS];
      const bool live = c < n_chunks;  // false only for a block without any chunk, which still has to pass the prologue
      if (live) {
        const uint64_t off = chunk_offset(c, i0);
#pragma unroll
        for (int f = 0; f < kF; ++f) load_items<T, ITEMS, NT>(base + off + uint64_t(f) * L.field_stride, x[f]);
      }
      // All loads of the chunk go out before any of the item math: the machine scheduler otherwise interleaves them
      // with their uses in groups of 4-6 (seen in the ISA), which cuts the bytes a wave keeps in flight and costs ≈ 7 %
      // of the streaming rate.
      __builtin_amdgcn_sched_barrier(0);
      if (first) {  // block-uniform
        first = false;
        if (lm_prologue(fin, P)) return;  // grid-uniform
#ifdef NOS_LM_TIMING
        t_prologue = wall_clock64() + (unsigned long long)(*reinterpret_cast<const T*>(&P) * T(0));  // after the pose arrived
#endif
        if (!live) break;
      }
      if constexpr (kPacked) {
#pragma unroll
        for (int it = 0; it < ITEMS; it += 2) {
          float2_t xi[kF];
#pragma unroll
          for (int f = 0; f < kF; ++f) xi[f] = float2_t{x[f][it], x[f][it + 1]};
          const bool valid2[2] = {(i0 + it) < L.n, (i0 + it + 1) < L.n};
          Problem::template item<float2_t>(xi, P, valid2, acc2);
        }
      } else {
#pragma unroll
        for (int it = 0; it < ITEMS; ++it) {
          T xi[kF];
#pragma unroll
          for (int f = 0; f < kF; ++f) xi[f] = x[f][it];
          Problem::item(xi, P, (i0 + it) < L.n, acc);
        }
      }
    }
  }
  if constexpr (kPacked) {
#pragma unroll
    for (int k = 0; k < kOut; ++k) acc[k] = acc2[k][0] + acc2[k][1];
  }

  double dacc[kOut];
#pragma unroll
  for (int k = 0; k < kOut; ++k) dacc[k] = double(acc[k]);
#ifdef NOS_LM_TIMING
  const unsigned long long t_loop = wall_clock64() + (unsigned long long)(dacc[0] * 0.0);  // after the item math
#endif
  block_reduce_store<kOut, BLOCK>(dacc, partials + size_t(blockIdx.x) * kOut, fin.write_through != 0);
#ifdef NOS_LM_TIMING
  if (threadIdx.x == 0 && fin.out_host != nullptr && fin.lm != nullptr) {
    // overwritten by every block; the last writer is (almost always) the finishing block
    __hip_atomic_store(fin.out_host + 56, double(t_prologue - t_start), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(fin.out_host + 57, double(t_loop - t_prologue), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(fin.out_host + 58, double(wall_clock64() - t_loop), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
#endif
  if (fin.counter != nullptr) finish_in_last_block<kOut, BLOCK>(partials, fin, t_start);
}

// ---------------------------------------------------------------- whole solve in one workgroup (small problems)
//
// At the reference's own test sizes (630 reprojection points, ≈ 2.9 k NDT correspondences) an LM iteration through the
// grid kernel costs ≈ 11-12 µs, nearly all of it launch, hand-off and dispatch latency.  Below kSingleBlockMaxElements
// plane-elements the whole loop runs inside ONE workgroup and ONE launch instead: the data (≤ 1 MB) stays in L2, the sums are
// reduced inside the block, one lane runs the same nos_host::LmAdvance* loop body on a state kept in LDS, and the
// next iteration starts after one barrier — no grid-wide hand-off, nothing to wait for, no way to hang.
// One CU evaluates a 512-correspondence NDT chunk in ≈ 0.9 µs, so the single-workgroup form only pays while the whole
// pass stays below the ≈ 6 µs that a launch with its hand-off costs: measured 11.3 → 5.3 µs per iteration at 630
// reprojection points, but no gain at 2 900 NDT correspondences (6 chunks) — hence a budget in plane-elements.
constexpr size_t kSingleBlockMaxElements = size_t(1024) * 15;  // n × fields: 1024 NDT or 3072 reprojection correspondences

template <typename Problem, typename T, int BLOCK, bool NT = false>
__global__ __launch_bounds__(BLOCK) void solve_single_block_kernel(TiledLayout L, typename Problem::Params P,
                                                                  uint32_t n_chunks, LmDevice* lm,
                                                                  double* __restrict__ cost_history, int history_capacity,
                                                                  double* entry_host, unsigned long long* seq_host,
                                                                  unsigned long long seq) {
  constexpr int kF = Problem::kFields;
  constexpr int kOut = Problem::kOut;
  const T* __restrict__ base = static_cast<const T*>(L.base);
  __shared__ double s_lm_raw[(sizeof(LmDevice) + 7) / 8];  // raw storage: the struct has default member initialisers
  LmDevice& s_lm = *reinterpret_cast<LmDevice*>(s_lm_raw);
  __shared__ double s_sum[kLmTotDoubles(kOut)];
  if (threadIdx.x == 0) s_lm = *lm;
  __syncthreads();
  int executed = 0;
  while (s_lm.st.done == 0) {  // block-uniform: every thread reads the same LDS word after a barrier
    set_pose(P, &s_lm);
    T acc[kOut];
#pragma unroll
    for (int k = 0; k < kOut; ++k) acc[k] = T(0);
    // several chunks per round, all their loads in flight before the first item is evaluated: with one workgroup there
    // are no other waves to hide the L2 latency behind
    constexpr uint32_t kRound = (kF * sizeof(T) > 64) ? 2 : 4;  // 15 fp64 planes: two chunks fill the register file
    for (uint32_t c = 0; c < n_chunks; c += kRound) {
      T x[kRound][kF][1];
      uint64_t i0[kRound];
#pragma unroll
      for (uint32_t u = 0; u < kRound; ++u) {
        const uint32_t cu = (c + u < n_chunks) ? c + u : c;  // clamped: re-reads chunk c, masked out below
        i0[u] = uint64_t(cu) * BLOCK + threadIdx.x;
        const uint64_t off = (i0[u] >> L.tile_shift) * L.tile_stride + (i0[u] & L.tile_mask);
#pragma unroll
        for (int f = 0; f < kF; ++f) load_items<T, 1, NT>(base + off + uint64_t(f) * L.field_stride, x[u][f]);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (uint32_t u = 0; u < kRound; ++u) {
        if (c + u < n_chunks) {  // block-uniform
          T xi[kF];
#pragma unroll
          for (int f = 0; f < kF; ++f) xi[f] = x[u][f][0];
          Problem::item(xi, P, i0[u] < L.n, acc);
        }
      }
    }
    double dacc[kOut];
#pragma unroll
    for (int k = 0; k < kOut; ++k) dacc[k] = double(acc[k]);
    block_reduce_store<kOut, BLOCK>(dacc, s_sum, false);
    __syncthreads();
    if (threadIdx.x == 0) {
      if (cost_history != nullptr && executed < history_capacity)
        __hip_atomic_store(cost_history + executed, s_sum[kOut - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      lm_step_lane<kOut>(lds_ptr(s_sum), lds_ptr(&s_lm));
    }
    ++executed;
    __syncthreads();
  }
  if (threadIdx.x < kOut && entry_host != nullptr && executed > 0)
    __hip_atomic_store(entry_host + kLogOut + threadIdx.x, s_sum[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  if (threadIdx.x == 0) {
    const nos_host::LmState st = s_lm.st;
    lm->st = st;
    if (entry_host != nullptr) {
#pragma unroll
      for (int k = 0; k < 9; ++k)
        __hip_atomic_store(entry_host + kLogR + k, st.R[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
#pragma unroll
      for (int k = 0; k < 3; ++k)
        __hip_atomic_store(entry_host + kLogT + k, st.t[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __hip_atomic_store(entry_host + kLogLambda, st.lambda, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __hip_atomic_store(entry_host + kLogPrevCost, st.previous_cost, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __hip_atomic_store(entry_host + kLogCost, st.cost, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __hip_atomic_store(entry_host + kLogIteration, double(st.iteration), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __hip_atomic_store(entry_host + kLogDone, double(st.done), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __hip_atomic_store(entry_host + kLogOk, double(st.ok), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __hip_atomic_store(entry_host + kLogExecuted, double(executed), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
  if (threadIdx.x < kWave) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (threadIdx.x == 0 && seq_host != nullptr) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __hip_atomic_store(seq_host, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

// ---------------------------------------------------------------- whole solve in one launch, data resident on chip
//
// Between the single-workgroup form above and the sizes where a launch is mostly streaming, an LM iteration through
// one launch per iteration costs ≈ 12 µs, nearly all of it kernel boundary, dispatch, first loads and hand-off.  Up to the
// on-chip capacity (ResidentShape below) the whole loop runs in ONE launch instead, one 512-thread workgroup per CU, all
// resident at once, every workgroup keeping its correspondences in REGISTERS + LDS for all iterations.
//
// Per iteration (an all-reduce, every workgroup for itself — nothing is broadcast):
//   1. item math over the resident correspondences, block reduction, the row of sums goes out as write-through (sc1)
//      stores into partials[iteration parity][workgroup];
//   2. the storing wave drains (vmcnt(0)), one lane ARRIVES: a no-return agent-scope add on one of 8 arrival counters
//      (workgroup index mod 8; counters are monotonic for the whole launch, each on a cache line of its own);
//   3. 8 lanes poll the 8 counters (sc1 loads) until all stand at (iteration + 1) x group size — every row of this
//      iteration has then left its writer (hand-off form "sc1 stores + drain + counter / sc1 loads", MI355X_MICROARCH.md);
//   4. EVERY workgroup adds all rows in the same fixed order (sc1 loads, 16 in flight per thread) and runs the same
//      nos_host::LmAdvance* on its own copy of the loop state: identical bits everywhere, so no state has to travel.
// Rows are double buffered by iteration parity: a workgroup can run at most one iteration ahead of the slowest reader,
// because arriving at iteration k + 1 happens after reading the rows of iteration k.
// Compared with round 1's form (one finishing workgroup: tickets with returned values, row sums, LM step, state written
// through, epoch word, everybody polls and re-reads the state) this removes two memory round trips and the state
// broadcast from the critical path of every iteration.
// Every wait is bounded (kClusterTimeoutTicks): a workgroup that waits longer — e.g. because another process holds CUs and
// the grid is not fully resident — raises `abort` and everybody leaves; the host then re-runs the solve with one launch
// per iteration.
constexpr uint32_t kClusterMaxBlocks = 256;
constexpr unsigned long long kClusterTimeoutTicks = 5000000ull;  // 50 ms of the 100 MHz wall clock per iteration

// Control words of one resident launch, zeroed by the host before the launch (hipMemsetAsync on the launch stream).
struct ClusterCtl {
  unsigned int abort;           // 1: a wait timed out, the launch gave up
  unsigned int pad0[31];
  struct alignas(128) Arrival {
    unsigned int count;         // arrivals of the workgroups with index mod 8 == this counter's index, all iterations
    unsigned int pad[31];
  } arrival[8];
};

__device__ __forceinline__ double sc1_load(const double* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void sc1_store(double* p, double v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// state <-> device memory through write-through stores / cache-bypassing loads (element-wise: the struct is plain data)
__device__ __forceinline__ void state_store_sc1(LmDevice* lm, const nos_host::LmState& st) {
  double* d = reinterpret_cast<double*>(&lm->st);
  const double* s = reinterpret_cast<const double*>(&st);
  constexpr int kWords = int(sizeof(nos_host::LmState) / sizeof(double));
  static_assert(sizeof(nos_host::LmState) % sizeof(double) == 0, "LmState must be a whole number of doubles");
#pragma unroll
  for (int k = 0; k < kWords; ++k) sc1_store(d + k, s[k]);
}
__device__ __forceinline__ void state_load_sc1(const LmDevice* lm, nos_host::LmState& st) {
  const double* d = reinterpret_cast<const double*>(&lm->st);
  double* s = reinterpret_cast<double*>(&st);
  constexpr int kWords = int(sizeof(nos_host::LmState) / sizeof(double));
#pragma unroll
  for (int k = 0; k < kWords; ++k) s[k] = sc1_load(d + k);
}

// How many correspondences a lane keeps resident for the whole solve: RI of them in REGISTERS (compile-time unrolled) and
// up to LI more in LDS (dynamic allocation, [slot][field][lane] so that lanes read consecutive addresses).  One
// 512-thread workgroup per CU → two waves per SIMD → 256 VGPRs per lane and ≈ 150 KB of the CU's 160 KB LDS:
//   NDT fp64 (resident form 96 B / correspondence): 3 + 3 → 6 per lane → 786 432 correspondences on 256 CUs
//   NDT fp32 (60 B, S kept)                        : 3 + 4 → 7         → 917 504
//   reprojection fp64 (40 B)          : 9 + 7 → 16        → 2 097 152  (BASELINE.json configs[2]: 2 M)
//   reprojection fp32 (20 B)          : 10 + 14 → 24      → 3 145 728
// i.e. at these sizes an LM iteration touches neither HBM nor the caches: its cost is the item math plus one grid-wide
// hand-off.  The first touch (one pass over the dataset) is paid once per solve.
// What a RESIDENT NDT correspondence consists of: the solvers only ever need A = SᵀS of the sqrt-information (with
// J = [S | S M]: s = eᵀAe, g = w [Ae ; MᵀAe], H = w [A, AM ; ·, MᵀAM] — Ndt6Problem::item_A / Ndt3Problem::item_A), so a
// correspondence that stays on chip for the whole solve is converted ONCE, at first touch, from {p, mu, S (9)} to
// {p, mu, A (6)}: 12 values instead of 15 per item (more items fit) and ≈ 35 % fewer instructions per item and iteration
// (fp64: 233 → ≈ 150).  Streamed data keeps the 15 planes: it is read
// once per iteration, the conversion would cost more than it saves.
// fp64 only: the fp32 item function already works from A and measured SLOWER through item_A (900 000: 8.57 → 9.24 µs).
template <int FIELDS, size_t ELEM>
constexpr int resident_fields() {
  return (FIELDS == 15 && ELEM == 8) ? 12 : FIELDS;
}
template <int FIELDS, int ELEM>
struct ResidentShape;
template <>
struct ResidentShape<15, 8> { static constexpr int RI = 3, LI = 3; };
template <>
struct ResidentShape<15, 4> { static constexpr int RI = 3, LI = 4; };
template <>
struct ResidentShape<5, 8> { static constexpr int RI = 9, LI = 7; };
template <>
struct ResidentShape<5, 4> { static constexpr int RI = 10, LI = 14; };

// SI > 0 selects the STREAMING form of the same kernel (instantiated with RI = LI = 0): the data set does not fit the
// register files and LDS of the chip, so every LM iteration streams it from HBM again, in chunks of BLOCK * SI
// correspondences taken grid-stride exactly like assemble_kernel does — but the loop still lives in ONE launch: no kernel
// boundary, no launch prologue and no ticket + last-block reduce per iteration (≈ 5 µs of every iteration at 10 M), the
// tagged all-reduce instead, and the first chunk of iteration k + 1 is already in flight while iteration k is being
// reduced and stepped (it does not depend on the pose).  `items_per_lane` then carries the number of chunks.  SPF: the
// next chunk's loads are issued before the current chunk is evaluated (fp32), NT: non-temporal loads.
template <typename Problem, typename T, int BLOCK, int RI, int LI, int PROTO = 1, int SI = 0, bool SPF = false, bool NT = false>
__global__ __launch_bounds__(BLOCK) void solve_cluster_kernel(TiledLayout L, typename Problem::Params P,
                                                             double* __restrict__ partials, LmDevice* lm, ClusterCtl* ctl,
                                                             double* __restrict__ cost_history, int history_capacity,
                                                             double* entry_host, unsigned long long* seq_host,
                                                             unsigned long long seq, uint32_t items_per_lane) {
  constexpr int kF = Problem::kFields;
  constexpr int kOut = Problem::kOut;
  constexpr int kCols = 32;
  constexpr int kSlices = BLOCK / kCols;
  const T* __restrict__ base = static_cast<const T*>(L.base);
  constexpr bool kAForm = kF == 15 && sizeof(T) == 8 && SI == 0;     // resident fp64 NDT items hold A = SᵀS (6) instead of S (9)
  constexpr int kRF = kAForm ? resident_fields<kF, sizeof(T)>() : kF;  // values per resident item
  extern __shared__ __align__(16) unsigned char resident_raw[];  // [items_per_lane - RI][kRF][BLOCK] of T
  T* resident = reinterpret_cast<T*>(resident_raw);
  __shared__ int s_flag;  // 0 go on, 1 loop finished, 2 abort
  __shared__ int s_fast;  // 1 once every group has been seen to sit on one XCD: stage-1 units then stay in that XCD's L2
  __shared__ double red[kSlices][kCols];
  __shared__ double s_tot[kLmTotDoubles(kOut)];
  __shared__ double s_lmd_raw[(sizeof(LmDevice) + 7) / 8];  // this workgroup's copy of the loop state and settings
  LmDevice& s_lmd = *reinterpret_cast<LmDevice*>(s_lmd_raw);
  nos_host::LmState& s_state = s_lmd.st;

  // This workgroup's correspondences, read ONCE: slot j of lane l is item  block_base + j * BLOCK + l  (a wave reads
  // consecutive items of one field per load).  Slots beyond n are zero records (contribute exactly nothing) and are
  // flagged invalid for the problems that mask.
  const uint32_t J = items_per_lane & 0x7fffffffu;  // grid-uniform, 1 … RI + LI (streaming form: the number of chunks)
  [[maybe_unused]] const bool allow_fast = (items_per_lane >> 31) == 0u;  // bit 31: keep stage 1 of the all-reduce on sc1 stores
  [[maybe_unused]] constexpr int kXccCol = 28;
  [[maybe_unused]] const unsigned int my_xcc = xcc_id();
  const uint64_t block_base = uint64_t(blockIdx.x) * BLOCK * J;
  auto fetch = [&](uint32_t j, T (&dst)[kRF]) -> bool {
    const uint64_t i = block_base + uint64_t(j) * BLOCK + threadIdx.x;
    const bool ok = i < L.n;
    const uint64_t ic = ok ? i : 0;  // clamped address; the value is zeroed below
    const uint64_t off = (ic >> L.tile_shift) * L.tile_stride + (ic & L.tile_mask);
    T xt[kF][1];
#pragma unroll
    for (int f = 0; f < kF; ++f) load_items<T, 1, false>(base + off + uint64_t(f) * L.field_stride, xt[f]);
    if constexpr (kAForm) {
#pragma unroll
      for (int f = 0; f < 6; ++f) dst[f] = ok ? xt[f][0] : T(0);
      int q = 6;
#pragma unroll
      for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = a; b < 3; ++b) {  // A(a, b) = sum over rows k of S(k, a) S(k, b);  a00 a01 a02 a11 a12 a22
          const T v = fma(xt[6 + a][0], xt[6 + b][0], fma(xt[9 + a][0], xt[9 + b][0], xt[12 + a][0] * xt[12 + b][0]));
          dst[q++] = ok ? v : T(0);
        }
    } else {
#pragma unroll
      for (int f = 0; f < kF; ++f) dst[f] = ok ? xt[f][0] : T(0);
    }
    return ok;
  };
  // one resident item → the sums (NDT: the A form; reprojection: the item as it is)
  auto evaluate_resident = [&](const T (&xi)[kRF], bool ok, T (&acc_)[kOut]) {
    if constexpr (kAForm) {
      const T p3[3] = {xi[0], xi[1], xi[2]}, mu3[3] = {xi[3], xi[4], xi[5]};
      const T A6[6] = {xi[6], xi[7], xi[8], xi[9], xi[10], xi[11]};
      (void)ok;  // pads are all-zero records: they contribute exactly nothing
      Problem::item_A(p3, mu3, A6, P, acc_);
    } else {
      T xf[kF];
#pragma unroll
      for (int f = 0; f < kF; ++f) xf[f] = xi[f < kRF ? f : 0];
      Problem::item(xf, P, ok, acc_);
    }
  };
  T x[RI > 0 ? RI : 1][kRF];
  bool valid[RI > 0 ? RI : 1];
  static_assert(SI == 0 || (RI == 0 && LI == 0), "the streaming form keeps nothing resident");
  // streaming form: the chunk being evaluated next (the first one of every iteration is fetched ahead of time)
  [[maybe_unused]] T xs[kF][SI > 0 ? SI : 1];
  [[maybe_unused]] uint64_t xs_i0 = 0;
  [[maybe_unused]] auto fetch_chunk = [&](uint32_t c, T (&dst)[kF][SI > 0 ? SI : 1]) -> uint64_t {
    const uint64_t i0 = uint64_t(c) * (uint64_t(BLOCK) * (SI > 0 ? SI : 1)) + uint64_t(threadIdx.x) * (SI > 0 ? SI : 1);
    const uint64_t off = (i0 >> L.tile_shift) * L.tile_stride + (i0 & L.tile_mask);
#pragma unroll
    for (int f = 0; f < kF; ++f) load_items<T, (SI > 0 ? SI : 1), NT>(base + off + uint64_t(f) * L.field_stride, dst[f]);
    return i0;
  };
  if constexpr (SI > 0) {
    if (blockIdx.x < J) xs_i0 = fetch_chunk(blockIdx.x, xs);
  }
#pragma unroll
  for (int j = 0; j < RI; ++j) {
    valid[j] = false;
    if (uint32_t(j) < J) {
      valid[j] = fetch(uint32_t(j), x[j]);
    } else {
#pragma unroll
      for (int f = 0; f < kRF; ++f) x[j][f] = T(0);
    }
  }
  if constexpr (LI > 0) {
    for (uint32_t j = RI; j < J; ++j) {
      T xi[kRF];
      (void)fetch(j, xi);
#pragma unroll
      for (int f = 0; f < kRF; ++f) resident[(size_t(j - RI) * kRF + f) * BLOCK + threadIdx.x] = xi[f];
    }
  }
#ifdef NOS_LM_TIMING
  if (threadIdx.x < 4) s_step_cycles[threadIdx.x] = 0ull;
#endif
  if (threadIdx.x == 0) {
    s_lmd.settings = lm->settings;  // constant during the launch
    s_state = lm->st;  // written by lm_init_kernel before this launch
    s_flag = s_state.done != 0 ? 1 : 0;
    s_fast = 0;
    // a launch that finds `abort` raised (the test hook raises it beforehand) gives up at once, like one whose wait timed out
    if (__hip_atomic_load(&ctl->abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) s_flag = 2;
  }
  __syncthreads();
  const unsigned int group = blockIdx.x & 7u;
  const unsigned int n_groups = gridDim.x < 8u ? gridDim.x : 8u;
  unsigned int it = 0;
  int executed = 0;
#ifdef NOS_LM_TIMING
  unsigned long long tq[6] = {0, 0, 0, 0, 0, 0}, tp = 0;
#define NOS_RES_STAMP(slot_)                                  \
  {                                                           \
    const unsigned long long now_ = wall_clock64();           \
    tq[slot_] += now_ - tp;                                   \
    tp = now_;                                                \
  }
#else
#define NOS_RES_STAMP(slot_)
#endif
  while (s_flag == 0) {
#ifdef NOS_LM_TIMING
    tp = wall_clock64();
#endif
    // pose of this iteration from LDS → scalar registers
    if constexpr (kOut == 28) {
#pragma unroll
      for (int k = 0; k < 9; ++k) P.R[k] = T(uniform_load(&s_state.R[k]));
#pragma unroll
      for (int k = 0; k < 3; ++k) P.t[k] = T(uniform_load(&s_state.t[k]));
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k) P.R2[k] = T(uniform_load(&s_state.R[k]));
#pragma unroll
      for (int k = 0; k < 2; ++k) P.t2[k] = T(uniform_load(&s_state.t[k]));
    }
    T acc[kOut];
#pragma unroll
    for (int k = 0; k < kOut; ++k) acc[k] = T(0);
    if constexpr (SI > 0) {
      const uint32_t n_chunks = J;
      auto evaluate = [&](const T (&xc)[kF][SI > 0 ? SI : 1], uint64_t i0) {
#pragma unroll
        for (int it = 0; it < (SI > 0 ? SI : 1); ++it) {
          T xi[kF];
#pragma unroll
          for (int f = 0; f < kF; ++f) xi[f] = xc[f][it];
          Problem::item(xi, P, (i0 + it) < L.n, acc);
        }
      };
      uint32_t c = blockIdx.x;
      if constexpr (SPF) {
        for (; c < n_chunks; c += gridDim.x) {
          T xb[kF][SI > 0 ? SI : 1];
          uint64_t i1 = 0;
          const uint32_t cn = c + gridDim.x;
          if (cn < n_chunks) i1 = fetch_chunk(cn, xb);
          evaluate(xs, xs_i0);
#pragma unroll
          for (int f = 0; f < kF; ++f)
#pragma unroll
            for (int it = 0; it < (SI > 0 ? SI : 1); ++it) xs[f][it] = xb[f][it];
          xs_i0 = i1;
        }
      } else {
        while (c < n_chunks) {
          __builtin_amdgcn_sched_barrier(0);  // all loads of a chunk before any of its math (see assemble_kernel)
          evaluate(xs, xs_i0);
          c += gridDim.x;
          if (c < n_chunks) xs_i0 = fetch_chunk(c, xs);
        }
      }
      // the first chunk of the NEXT iteration: in flight during the all-reduce and the step below
      // (every wave does, the polling ones too: letting only the other half prefetch measured 2 % slower at 10 M)
      if (blockIdx.x < n_chunks) xs_i0 = fetch_chunk(blockIdx.x, xs);
    } else
    // (fp64 only: the fp32 kernels spill when their items are interleaved)
    if (sizeof(T) == 8 && J >= uint32_t(RI)) {  // grid-uniform; one straight-line block, so the scheduler can interleave the items
#pragma unroll
      for (int j = 0; j < RI; ++j) evaluate_resident(x[j], valid[j], acc);
    } else {
#pragma unroll
      for (int j = 0; j < RI; ++j)
        if (uint32_t(j) < J) evaluate_resident(x[j], valid[j], acc);
    }
    if constexpr (LI > 0) {
      // (fetching item j + 1 from LDS before item j is evaluated was tried and is SLOWER: reprojection 2 M 14.4 -> 15.4 us
      //  per iteration, profiles/r03_ab_resident.txt — the second buffer costs the register items their interleaving)
      for (uint32_t j = RI; j < J; ++j) {
        T xi[kRF];
#pragma unroll
        for (int f = 0; f < kRF; ++f) xi[f] = resident[(size_t(j - RI) * kRF + f) * BLOCK + threadIdx.x];
        evaluate_resident(xi, (block_base + uint64_t(j) * BLOCK + threadIdx.x) < L.n, acc);
      }
    }
    double dacc[kOut];
#pragma unroll
    for (int k = 0; k < kOut; ++k) dacc[k] = double(acc[k]);
    NOS_RES_STAMP(0)  // item math
    if constexpr (PROTO == 1) {
      // ---- tagged two-stage all-reduce (round 2, second form): every sum travels as a 16-byte {value, iteration} unit.
      //   stage 1: each workgroup publishes its 28 block sums; the LEADER of its group (workgroups 0..7 lead the groups
      //            "index mod 8") spins on the units of its ≤ 32 members, adds them in member order, publishes 28 group sums;
      //   stage 2: every workgroup spins on the 8 x 28 group units and adds them in group order.
      // No counters, no drain between data and flag, two memory round trips on the critical path, ≈ 1 MB of polling
      // traffic per iteration chip-wide instead of the 14.7 MB of "everybody reads every row".  Block rows need no double
      // buffering (a workgroup publishes iteration k + 1 only after all group sums of k, i.e. after every leader has read
      // the rows of k); group rows are double buffered by parity (a leader can run one iteration ahead of a reader in
      // another group, not two).
      TaggedUnit* const block_units = reinterpret_cast<TaggedUnit*>(partials);                       // [blocks][32]
      TaggedUnit* const group_units = block_units + size_t(kClusterMaxBlocks) * 32;                 // [2][8][32]
      // the tag is unique across launches too (the host's sequence number of this launch in the upper bits): no memset
      const unsigned long long tag = (seq << 24) | ((unsigned long long)it + 1ull);
      const double mine = block_reduce_value<kOut, BLOCK>(dacc);
      // Stage 1 stays inside an XCD when the placement allows it.  HIP promises nothing about which XCD a workgroup lands on
      // (observed: round-robin, so the members of group "index mod 8" share one), so iteration 0 goes the placement-independent
      // way (sc1 stores) and carries every workgroup's XCC id in unit 28; each leader counts the members that are NOT on its
      // own XCD, the counts travel with the group sums, and only if all eight are zero do the following iterations use plain
      // stage-1 stores (line kept in the shared L2: 2.9 -> 2.3 µs for both stages).  Stage 2 is cross-XCD by nature: sc1.
      const bool probe = it == 0u && allow_fast;  // block-uniform
      if (threadIdx.x < kOut) {
        if (s_fast != 0)
          tagged_store_plain(block_units + size_t(blockIdx.x) * 32 + threadIdx.x, mine, tag);
        else
          tagged_store(block_units + size_t(blockIdx.x) * 32 + threadIdx.x, mine, tag);
      } else if (probe && threadIdx.x == kXccCol) {
        tagged_store(block_units + size_t(blockIdx.x) * 32 + kXccCol, double(my_xcc), tag);
      }
      NOS_RES_STAMP(1)  // block reduce + units issued
      const unsigned long long deadline = wall_clock64() + kClusterTimeoutTicks;
      // bounded spin on one unit; returns false when the launch is being abandoned
      auto await = [&](const TaggedUnit* u, double* value) -> bool {
        unsigned int polls = 0;
        for (;;) {
          const TaggedUnit got = tagged_load(u);
          if (got.seq == tag) {
            *value = got.value;
            return true;
          }
          // (no back-off between polls: s_sleep 4 / 16 measured slower at 100 k — 6.2 → 6.3 / 6.8 µs — and no help at 10 M)
          if ((++polls & 63u) == 0u &&
              (__hip_atomic_load(&ctl->abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u || wall_clock64() > deadline)) {
            __hip_atomic_store(&ctl->abort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_flag = 2;
            return false;
          }
        }
      };
      const int col = threadIdx.x % kCols;    // which sum
      const int slice = threadIdx.x / kCols;  // which member / group
      if (blockIdx.x < n_groups) {            // block-uniform: this workgroup leads group blockIdx.x
        const unsigned int g_size = (gridDim.x - blockIdx.x + 7u) >> 3;
        double gsum = 0.0;
        for (unsigned int m0 = 0; m0 < g_size; m0 += kSlices) {  // ≤ 2 passes of 16 members
          const unsigned int m = m0 + slice;
          double v = 0.0;
          if (m < g_size && (col < kOut || (probe && col == kXccCol))) {
            (void)await(block_units + size_t(blockIdx.x + 8u * m) * 32 + col, &v);
            if (col == kXccCol) v = v == double(my_xcc) ? 0.0 : 1.0;  // a member on another XCD
          }
          red[slice][col] = v;
          __syncthreads();
          if (threadIdx.x < kOut || (probe && threadIdx.x == kXccCol)) {
#pragma unroll
            for (int sl = 0; sl < kSlices; ++sl) gsum += red[sl][threadIdx.x];  // members in index order
          }
          __syncthreads();
        }
        if ((threadIdx.x < kOut || (probe && threadIdx.x == kXccCol)) && s_flag != 2)
          tagged_store(group_units + (size_t(it & 1u) * 8 + blockIdx.x) * 32 + threadIdx.x, gsum, tag);
      }
      {
        double v = 0.0;
        if (slice < int(n_groups) && (col < kOut || (probe && col == kXccCol)) && s_flag != 2)
          (void)await(group_units + (size_t(it & 1u) * 8 + slice) * 32 + col, &v);
        if (slice < 8) red[slice][col] = v;
      }
      __syncthreads();
      NOS_RES_STAMP(2)  // both stages arrived
      if (s_flag == 2) break;  // block-uniform
      if (threadIdx.x < kOut) {
        double tot = 0.0;
        for (unsigned int g = 0; g < n_groups; ++g) tot += red[g][threadIdx.x];  // groups in index order
        s_tot[threadIdx.x] = tot;
      } else if (probe && threadIdx.x == kXccCol) {
        double strangers = 0.0;
        for (unsigned int g = 0; g < n_groups; ++g) strangers += red[g][kXccCol];
        s_fast = strangers == 0.0 ? 1 : 0;  // the same verdict in every workgroup
      }
      __syncthreads();
    } else {
    double* rows = partials + size_t(it & 1u) * size_t(kClusterMaxBlocks) * kOut;  // this iteration's buffer
    block_reduce_store<kOut, BLOCK>(dacc, rows + size_t(blockIdx.x) * kOut, true);  // sc1 row
    NOS_RES_STAMP(1)  // block reduce + row store issued
    if (threadIdx.x < kWave) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the row left through lanes of wave 0
      if (threadIdx.x == 0)  // arrive (no value returned: nothing waits for this add)
        (void)__hip_atomic_fetch_add(&ctl->arrival[group].count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      // wait for everybody's arrival: lane g watches counter g
      const unsigned int g = threadIdx.x & 7u;
      const unsigned int g_size = (gridDim.x - g + 7u) >> 3;
      const unsigned int target = (it + 1u) * g_size;
      const unsigned long long deadline = wall_clock64() + kClusterTimeoutTicks;
      int flag = 0;
      unsigned int polls = 0;
      for (;;) {
        const unsigned int seen = (g < n_groups && threadIdx.x < 8u)
                                      ? __hip_atomic_load(&ctl->arrival[g].count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                      : target;
        if (__ballot(seen < target) == 0ull) break;  // wave-uniform
        // the abort word and the clock are looked at on the first and then every 16th poll
        if ((polls++ & 15u) == 0u &&
            (__hip_atomic_load(&ctl->abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u || wall_clock64() > deadline)) {
          flag = 2;
          break;
        }
      }
      if (threadIdx.x == 0 && flag == 2) {
        __hip_atomic_store(&ctl->abort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_flag = 2;
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");  // no instruction: keeps the row loads below the poll
    }
    __syncthreads();
    NOS_RES_STAMP(2)  // drain + arrive + everybody arrived
    if (s_flag == 2) break;  // block-uniform
    {
      // every workgroup adds the rows of this iteration in the same fixed order
      const int col = threadIdx.x % kCols;
      const int slice = threadIdx.x / kCols;
      constexpr int kUnroll = 16;
      double sum = 0.0;
      if (col < kOut) {
        const double* p = rows + col;
        for (uint32_t r = slice; r < gridDim.x; r += kUnroll * kSlices) {
          double v[kUnroll];
#pragma unroll
          for (int u = 0; u < kUnroll; ++u) {
            const uint32_t rr = r + u * kSlices;
            const double q = sc1_load(p + size_t(rr < gridDim.x ? rr : r) * kOut);
            v[u] = rr < gridDim.x ? q : 0.0;
          }
#pragma unroll
          for (int u = 0; u < kUnroll; ++u) sum += v[u];
        }
      }
      red[slice][col] = sum;
      __syncthreads();
      if (threadIdx.x < kOut) {
        double tot = 0.0;
#pragma unroll
        for (int sl = 0; sl < kSlices; ++sl) tot += red[sl][threadIdx.x];
        s_tot[threadIdx.x] = tot;
      }
      __syncthreads();
    }
    }  // PROTO
    {
      NOS_RES_STAMP(3)  // rows → sums
      if (threadIdx.x == 0) {  // lane 0 of EVERY workgroup advances its own copy of the loop (identical bits everywhere)
        if (blockIdx.x == 0 && cost_history != nullptr && executed < history_capacity)
          __hip_atomic_store(cost_history + executed, s_tot[kOut - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        lm_step_lane<kOut>(lds_ptr(s_tot), lds_ptr(&s_lmd));
        s_flag = s_state.done != 0 ? 1 : 0;
      }
    }
    ++executed;
    ++it;
    __syncthreads();
    NOS_RES_STAMP(4)  // LM step + barrier
  }
#ifdef NOS_LM_TIMING
  if (blockIdx.x == 0 && threadIdx.x == 0 && entry_host != nullptr) {
    for (int k = 0; k < 5; ++k)
      __hip_atomic_store(entry_host + 50 + k, double(tq[k]) / double(executed > 0 ? executed : 1), __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_SYSTEM);
    for (int k = 0; k < 3; ++k)
      __hip_atomic_store(entry_host + 56 + k, double(s_step_cycles[k]) / double(executed > 0 ? executed : 1), __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_SYSTEM);
  }
#endif
  // workgroup 0 reports (on abort nobody does: the host sees the missing sequence word)
  if (s_flag == 1 && blockIdx.x == 0) {
    if (threadIdx.x < kOut && entry_host != nullptr && executed > 0)
      __hip_atomic_store(entry_host + kLogOut + threadIdx.x, s_tot[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (threadIdx.x == 0) {
      const nos_host::LmState st = s_state;
      lm->st = st;
      if (entry_host != nullptr) {
#pragma unroll
        for (int k = 0; k < 9; ++k)
          __hip_atomic_store(entry_host + kLogR + k, st.R[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
#pragma unroll
        for (int k = 0; k < 3; ++k)
          __hip_atomic_store(entry_host + kLogT + k, st.t[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(entry_host + kLogLambda, st.lambda, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(entry_host + kLogPrevCost, st.previous_cost, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(entry_host + kLogCost, st.cost, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(entry_host + kLogIteration, double(st.iteration), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(entry_host + kLogDone, double(st.done), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(entry_host + kLogOk, double(st.ok), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(entry_host + kLogExecuted, double(executed), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
    if (threadIdx.x < kWave) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (threadIdx.x == 0 && seq_host != nullptr) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_store(seq_host, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
  }
}

// ---------------------------------------------------------------- voxel-indexed variant
//
// The reference's data model copies the whole NDT into every correspondence (MDM/types.h:23-26), which
// is what the flat 120-byte layout above streams.  When many points share a voxel (10 M points over
// 200 k voxels = 50 per voxel) the same sums can be formed from  point (3 values) + voxel id(s)  and a
// table of voxel records {mean(3), A = SᵀS (6), pad}: 24 B + 4 B·K per point instead of
// 120 B·K, with the table (≈ 25 MB at 200 k voxels) served from L2 / Infinity Cache.  Points are
// stored sorted by voxel id (done once at dataset creation), so the lanes of a wave hit a handful
// of table records that stay in L1.  The kernel is then fp64-ALU bound, not HBM bound; it is reported
// separately from the 120-byte roofline.
struct IndexedLayout {
  const void* points;      // 3 planes of n_padded (element type T)
  const int32_t* index;    // K planes of n_padded voxel ids, -1 = no correspondence in that slot
  const void* table;       // [n_voxels][16] of T: mean(3), A = SᵀS upper triangle (6), pad(7)
  uint64_t n_padded;       // multiple of the kernel chunk; pads carry index -1
};

// Only the nine values in use are loaded (fp64: four 16-byte loads + one 8-byte, fp32: two 16-byte + one 4-byte).  A
// wider last load would fetch a pad element into a register the compiler knows to be dead: it re-uses that register as a
// temporary inside the item math while the load is still in flight, and the write-after-write hazard costs an
// `s_waitcnt vmcnt(0)` in the middle of every evaluation (round 3's ISA) — i.e. the whole software pipeline.
template <typename T>
__device__ __forceinline__ void load_voxel_record(const T* table, int32_t v, T (&rec)[12]) {
  const T* p = table + size_t(16) * size_t(v);
  if constexpr (sizeof(T) == 8) {
    using V2 = double __attribute__((ext_vector_type(2)));
    const V2* q = reinterpret_cast<const V2*>(p);
#pragma unroll
    for (int k = 0; k < 4; ++k) {  // mean (3) + A = SᵀS (6) = 9 values
      const V2 t = q[k];
      rec[2 * k] = t[0];
      rec[2 * k + 1] = t[1];
    }
    rec[8] = p[8];
  } else {
    using V4 = float __attribute__((ext_vector_type(4)));
    const V4* q = reinterpret_cast<const V4*>(p);
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const V4 t = q[k];
#pragma unroll
      for (int m = 0; m < 4; ++m) rec[4 * k + m] = t[m];
    }
    rec[8] = p[8];
  }
}

// Problem = Ndt6Problem / Ndt3Problem (their item() takes the same 15 values).  K = voxel slots per point.
// The kernel is latency / ALU bound, so it is software pipelined by hand: while chunk c is being evaluated
// the voxel records of chunk c+1 (ids already in registers) and the points + ids of chunk c+2 are in flight.
// One large workgroup per CU (768 or 1024 threads) keeps the in-launch reduction at 256 tickets.
template <typename Problem, typename T, int K, int BLOCK, int MINW>
__global__ __launch_bounds__(BLOCK, MINW) void assemble_indexed_kernel(IndexedLayout L, typename Problem::Params P,
                                                                      uint32_t n_chunks,
                                                                      double* __restrict__ partials,
                                                                      FusedFinal fin) {
  constexpr int kOut = Problem::kOut;
  const T* __restrict__ pts = static_cast<const T*>(L.points);
  const T* __restrict__ table = static_cast<const T*>(L.table);
  if (lm_prologue(fin, P)) return;  // grid-uniform

  T acc[kOut];
#pragma unroll
  for (int k = 0; k < kOut; ++k) acc[k] = T(0);

  // No load sits behind a branch: a chunk index past the end is clamped to the last chunk (its loads are issued and their
  // results ignored — `live` below), because a branch around a load makes the compiler's wait-count bookkeeping fall back
  // to `s_waitcnt vmcnt(0)` at the join, which serialised the three stages (round 3's ISA: six vmcnt(0) in the loop body).
  const uint32_t last_chunk = n_chunks - 1u;
  auto load_point = [&](uint32_t c, T (&p)[3], int32_t (&vid)[K]) {
    const uint64_t i = uint64_t(c < n_chunks ? c : last_chunk) * BLOCK + threadIdx.x;
    p[0] = __builtin_nontemporal_load(pts + i);
    p[1] = __builtin_nontemporal_load(pts + L.n_padded + i);
    p[2] = __builtin_nontemporal_load(pts + 2 * L.n_padded + i);
#pragma unroll
    for (int k = 0; k < K; ++k) vid[k] = __builtin_nontemporal_load(L.index + uint64_t(k) * L.n_padded + i);
  };
  auto load_records = [&](const int32_t (&vid)[K], T (&rec)[K][12]) {
#pragma unroll
    for (int k = 0; k < K; ++k) load_voxel_record<T>(table, vid[k] < 0 ? 0 : vid[k], rec[k]);  // id 0 is always readable
  };

  // Software pipeline with two prefetch distances.  The point stream comes from HBM (28 bytes per lane and chunk): by
  // Little's law its rate is (bytes in flight) / latency, and round 3's two chunks in flight — 28 KB per CU — were what held
  // the kernel at 3.1 TB/s of its own bytes with the vector ALUs 26 % busy (profiles/r04pre_indexed_summary.json).  So the
  // points and ids run kPointAhead chunks ahead of the evaluation (7 registers per chunk), the voxel records (L1 / L2
  // hits: the points are sorted by voxel) one chunk ahead.  Buffers are rings indexed by stage number; the loop body is
  // unrolled over one full rotation of both rings (kUnroll stages), so every index is a constant and the rings live in
  // registers, rotating by NAME — no copies, and the compiler's wait counts stay exact (vmcnt(N), never vmcnt(0)).
  // (fp32: 12 waves per CU, half the bytes per chunk — three chunks ahead; unroll = lcm of the two ring lengths)
  constexpr int kPointAhead = sizeof(T) == 8 ? 4 : 3, kPointRing = kPointAhead + 1, kRecordRing = 2,
                kUnroll = (kPointRing % kRecordRing == 0) ? kPointRing : kPointRing * kRecordRing;
  T pt[kPointRing][3];
  int32_t id[kPointRing][K];
  T rc[kRecordRing][K][12];
  auto evaluate = [&](bool live, const T (&p)[3], const int32_t (&vid)[K], const T (&rec)[K][12]) {
#pragma unroll
    for (int k = 0; k < K; ++k) {
      if (live && vid[k] >= 0) {
        const T mu[3] = {rec[k][0], rec[k][1], rec[k][2]};
        const T A[6] = {rec[k][3], rec[k][4], rec[k][5], rec[k][6], rec[k][7], rec[k][8]};
        Problem::item_A(p, mu, A, P, acc);
      }
    }
  };
  uint32_t c = blockIdx.x;
  const uint32_t g = gridDim.x;
#pragma unroll
  for (int s = 0; s < kPointAhead; ++s) load_point(c + uint32_t(s) * g, pt[s], id[s]);
  load_records(id[0], rc[0]);
  for (; c < n_chunks; c += uint32_t(kUnroll) * g) {
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) {  // stage u: chunk c + u g
      load_point(c + uint32_t(u + kPointAhead) * g, pt[(u + kPointAhead) % kPointRing], id[(u + kPointAhead) % kPointRing]);
      load_records(id[(u + 1) % kPointRing], rc[(u + 1) % kRecordRing]);
      __builtin_amdgcn_sched_barrier(0);
      evaluate(c + uint32_t(u) * g < n_chunks, pt[u % kPointRing], id[u % kPointRing], rc[u % kRecordRing]);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  double dacc[kOut];
#pragma unroll
  for (int k = 0; k < kOut; ++k) dacc[k] = double(acc[k]);
  block_reduce_store<kOut, BLOCK>(dacc, partials + size_t(blockIdx.x) * kOut, fin.write_through != 0);
  if (fin.counter != nullptr) finish_in_last_block<kOut, BLOCK>(partials, fin);
}

// dst[j] = src[perm[j]] for planes of T / int32 (dataset creation: apply the voxel-sort permutation)
template <typename SRC, typename DST>
__global__ __launch_bounds__(256) void gather_plane_kernel(const SRC* __restrict__ src, const uint32_t* __restrict__ perm,
                                                           uint64_t n, uint64_t n_padded, DST pad_value,
                                                           DST* __restrict__ dst) {
  const uint64_t j = uint64_t(blockIdx.x) * 256 + threadIdx.x;
  if (j >= n_padded) return;
  dst[j] = j < n ? DST(src[perm ? perm[j] : j]) : pad_value;
}

// voxel table: [V][3] means + [V][9] sqrt-informations (double) → [V][16] records of T = {mean, SᵀS upper triangle}
template <typename T>
__global__ __launch_bounds__(256) void build_voxel_table_kernel(const double* __restrict__ means,
                                                                const double* __restrict__ sqrt_infos, uint64_t n_voxels,
                                                                T* __restrict__ table) {
  const uint64_t t = uint64_t(blockIdx.x) * 256 + threadIdx.x;
  const uint64_t v = t >> 4;
  const int k = int(t & 15);
  if (v >= n_voxels) return;
  T val = T(0);
  if (k < 3) {
    val = T(means[3 * v + k]);
  } else if (k < 9) {  // A = SᵀS, upper triangle row-major: (0,0) (0,1) (0,2) (1,1) (1,2) (2,2)
    const int ii[6] = {0, 0, 0, 1, 1, 2}, jj[6] = {0, 1, 2, 1, 2, 2};
    const int i = ii[k - 3], j = jj[k - 3];
    const double* S = sqrt_infos + 9 * v;
    val = T(S[i] * S[j] + S[3 + i] * S[3 + j] + S[6 + i] * S[6 + j]);
  }
  table[t] = val;
}

// sort keys for the voxel ordering: slot-0 voxel id, absent (-1) last
__attribute__((unused)) static __global__ __launch_bounds__(256) void index_sort_key_kernel(const int32_t* __restrict__ idx0, uint64_t n,
                                                             uint32_t* __restrict__ keys, uint32_t* __restrict__ ids) {
  const uint64_t i = uint64_t(blockIdx.x) * 256 + threadIdx.x;
  if (i >= n) return;
  keys[i] = idx0[i] < 0 ? 0xFFFFFFFFu : uint32_t(idx0[i]);
  ids[i] = uint32_t(i);
}

// Device-resident loop: initial state (one lane; the arguments travel by value, no copy is needed).
struct LmInitArgs {
  double R[9];
  double t[3];
  nos_host::LmSettings settings;
  int dof;  // 6 or 3
};
__attribute__((unused)) static __global__ void lm_init_kernel(LmDevice* lm, LmInitArgs a) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  nos_host::LmState st;
  if (a.dof == 6)
    nos_host::LmInit6(&st, a.R, a.t, a.settings.max_iterations, a.settings.float_schedule);
  else
    nos_host::LmInit3(&st, a.R, a.t, a.settings.max_iterations, a.settings.float_schedule);
  lm->st = st;
  lm->settings = a.settings;
}

// Device-resident loop, stand-alone step (one wave): used when the sums come out of an RCCL all-reduce (or when
// the in-launch step is switched off).  Reads the sums from `sums`, publishes them and the new state to the pinned
// log entry, then the sequence word.
template <int NOUT>
__global__ __launch_bounds__(64) void lm_step_kernel(const double* __restrict__ sums, LmDevice* lm, double* entry_host,
                                                     unsigned long long* seq_host, unsigned long long seq) {
  __shared__ double s_tot[kLmTotDoubles(NOUT)];
  __shared__ double s_lmd_raw[(sizeof(LmDevice) + 7) / 8];
  LmDevice& s_lmd = *reinterpret_cast<LmDevice*>(s_lmd_raw);
  const int done = *reinterpret_cast<const volatile int*>(&lm->st.done);  // loop finished earlier: forward seq only
  if (done == 0) {
    if (threadIdx.x < NOUT) {
      const double v = sums[threadIdx.x];
      s_tot[threadIdx.x] = v;
      if (entry_host != nullptr)
        __hip_atomic_store(entry_host + kLogOut + threadIdx.x, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    if (threadIdx.x == 0) {
      s_lmd.st = lm->st;
      s_lmd.settings = lm->settings;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      lm_step_lane<NOUT>(lds_ptr(s_tot), lds_ptr(&s_lmd));
      lm_publish(s_lmd.st, lm, entry_host);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (threadIdx.x == 0 && seq_host != nullptr) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __hip_atomic_store(seq_host, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// nos_ctx_comm_allreduce over the mailbox: values[0..count) (device) → sums over ranks, in place.  One workgroup.
__attribute__((unused)) static __global__ __launch_bounds__(64) void mailbox_allreduce_kernel(Mailbox mb, double* values,
                                                                                              int count) {
  const double mine = int(threadIdx.x) < count ? values[threadIdx.x] : 0.0;
  const double sum = mailbox_allreduce<28>(mb, mine);
  if (int(threadIdx.x) < count) values[threadIdx.x] = sum;
}

// Fixed-order sum of the block rows: thread (slice, col) adds rows slice, slice+S, …;
// then the S slice sums are added in slice order.  One block, 1024 threads.
template <int NOUT>
__global__ __launch_bounds__(1024) void final_reduce_kernel(const double* __restrict__ partials,
                                                            uint32_t n_rows,
                                                            double* __restrict__ out) {
  constexpr int kCols = 32;
  constexpr int kSlices = 1024 / kCols;
  __shared__ double lds[kSlices][kCols];
  const int col = threadIdx.x % kCols;
  const int slice = threadIdx.x / kCols;
  double s = 0.0;
  if (col < NOUT)
    for (uint32_t r = slice; r < n_rows; r += kSlices) s += partials[size_t(r) * NOUT + col];
  lds[slice][col] = s;
  __syncthreads();
  if (threadIdx.x < NOUT) {
    double tot = 0.0;
#pragma unroll
    for (int sl = 0; sl < kSlices; ++sl) tot += lds[sl][threadIdx.x];
    out[threadIdx.x] = tot;
  }
}

// ---------------------------------------------------------------- ingestion kernels

// planar source planes (15 or 5 pointers, element type SRC) → tiled layout of DST, zero pads.
struct PlanePtrs {
  const void* p[15];
};

template <typename SRC, typename DST>
__global__ __launch_bounds__(256) void retile_kernel(PlanePtrs src, int n_fields, TiledLayout L,
                                                     DST* __restrict__ dst) {
  const uint64_t i = uint64_t(blockIdx.x) * 256 + threadIdx.x;
  const int f = blockIdx.y;
  if (i >= L.n_padded || f >= n_fields) return;
  const uint64_t off = (i >> L.tile_shift) * L.tile_stride + uint64_t(f) * L.field_stride + (i & L.tile_mask);
  DST v = DST(0);
  if (i < L.n) v = DST(static_cast<const SRC*>(src.p[f])[i]);
  dst[off] = v;
}

// array-of-structures records (double fields at byte offsets) → tiled layout.
// `first` is the index of records[0] inside the dataset; count records are unpacked.
struct FieldOffsets {
  uint32_t off[15];
};

template <typename DST>
__global__ __launch_bounds__(256) void unpack_records_kernel(const unsigned char* __restrict__ records,
                                                             uint64_t stride_bytes, FieldOffsets fo,
                                                             int n_fields, uint64_t first,
                                                             uint64_t count, TiledLayout L,
                                                             DST* __restrict__ dst) {
  const uint64_t j = uint64_t(blockIdx.x) * 256 + threadIdx.x;
  if (j >= count) return;
  const unsigned char* rec = records + j * stride_bytes;
  const uint64_t i = first + j;
  const uint64_t o = (i >> L.tile_shift) * L.tile_stride + (i & L.tile_mask);
  for (int f = 0; f < n_fields; ++f) {
    const double v = *reinterpret_cast<const double*>(rec + fo.off[f]);
    dst[o + uint64_t(f) * L.field_stride] = DST(v);
  }
}

template <typename DST>
__global__ __launch_bounds__(256) void zero_pad_kernel(int n_fields, TiledLayout L, DST* __restrict__ dst) {
  const uint64_t i = L.n + uint64_t(blockIdx.x) * 256 + threadIdx.x;
  if (i >= L.n_padded) return;
  const uint64_t o = (i >> L.tile_shift) * L.tile_stride + (i & L.tile_mask);
  for (int f = 0; f < n_fields; ++f) dst[o + uint64_t(f) * L.field_stride] = DST(0);
}

}  // namespace nos
