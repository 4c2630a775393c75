// assemble_items.hpp — layouts, parameter blocks, robust loss, and the per-correspondence item functions of the three problems.
// Part of the hand-written gfx950 kernels of the Gauss-Newton normal-equation assembly path; see assemble_kernels.hpp
// (the umbrella header every translation unit includes) for the overview and the reference citations.
#pragma once

#include <hip/hip_runtime.h>
#include <limits>

#include "host/nos_lm.hpp"
#include <stdint.h>

// -DNOS_LM_TIMING is the probe build of tools/ (device time stamps of the phases of a launch).  Everything it adds to the
// kernels sits inside NOS_PROBE(...) or one of the three stamp macros defined from it; the product build compiles none of it.
#ifdef NOS_LM_TIMING
#define NOS_PROBE(...) __VA_ARGS__
#else
#define NOS_PROBE(...)
#endif

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

}  // namespace nos
