/*
 * nos_oracle_avx.c — TEST / BENCH-BASELINE INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * AVX2 + FMA fp32 restatement of the reference's fastest CPU variant,
 * MahalanobisDistanceMinimizerAnalyticSIMDVarious::SolveFloatIntrinsicAligned
 * (MDM/mahalanobis_distance_minimizer_analytic_simd_various.cc:1300-1447): 15 aligned
 * float planes, 8 correspondences per step, 28 lane accumulators, robust loss evaluated
 * per lane through the scalar double path (:1396-1409), horizontal sum at the end
 * (:1430-1447).  Threads take the reference's contiguous multiple-of-8 ranges and the
 * caller sums the partials in thread order (MDM/..._analytic_simd.cc:55-76).
 * Used only as the "repo's own AVX path" timing baseline beside the GPU numbers and as
 * a coarse fp32 cross-check of the scalar oracle.
 *
 * The same treatment for the two other SIMD classes of the reference (both single-threaded there: neither has
 * a SetMultiThreadExecutor path; the thread fan-out offered here reuses the 6-DoF partition and is an
 * extrapolation, stated as such wherever it is reported):
 *   oracle_avx_ndt3_accumulate    MDM/mahalanobis_distance_minimizer_analytic_3dof_simd.cc:85-158 (+ lane sums :160-177)
 *   oracle_avx_reproj_accumulate  REM/reprojection_error_minimizer_analytic_simd.cc:55-138 (+ lane sums :140-157):
 *                                 depth mask `Xw.z > 0` multiplying the weight (:66,92), 1/fx evaluated in float (:29-30)
 *
 * The SAME-PRECISION baseline of the fp64 headline (round 3):
 *   oracle_avx_ndt6_accumulate_f64  the inner loop of MahalanobisDistanceMinimizerAnalyticSIMDVarious::SolveDouble
 *                                 (MDM/..._analytic_simd_various.cc:42-134; lane sums :135-149): 4-lane fp64 (the in-tree
 *                                 ScalarD = __m256d, NO/simd_helper/simd_scalar_amd.h), 4 correspondences per step, only
 *                                 floor(N/4)*4 used (:41-43), pw = R p + t, e, r = S e, M = -R [p]x (:69-77), S M (:78),
 *                                 loss per lane through the scalar virtual (:107-121), g += (J^T r) w (:124),
 *                                 H(ii, jj) += w (J0i J0j + J1i J1j + J2i J2j) (:127-133), cost += loss (:135).
 *                                 The reference GATHERS its four 304-byte records into lanes every iteration (:48-58) and
 *                                 runs on one thread; this restatement reads planar fp64 planes (the CPU is spared the
 *                                 gather: an upper bound of that variant's speed) and offers the 6-DoF SIMD class's thread
 *                                 partition on multiples of 4 (an extrapolation, stated wherever it is reported).
 */
#include <immintrin.h>
#include <math.h>
#include <pthread.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>

#include "nos_oracle.h"

typedef struct avx_job {
  const float* const* planes;
  size_t begin, end; /* multiples of 8 */
  float R[9], t[3];
  float intr[4]; /* reprojection: 1/fx, 1/fy, cx, cy (floats, as the SIMD class broadcasts them) */
  int kind;      /* 0: 6-DoF NDT, 1: 3-DoF NDT, 2: reprojection */
  oracle_loss loss;
  double out28[28];
  int created; /* its thread exists (pthread_create succeeded) */
} avx_job;

static inline float hsum8(__m256 v) {
  float buf[8];
  _mm256_storeu_ps(buf, v);
  return buf[0] + buf[1] + buf[2] + buf[3] + buf[4] + buf[5] + buf[6] + buf[7];
}

static void avx_range(avx_job* job) {
  __m256 R[9], t[3], H[21], g[6], cost;
  const float* const* pl = job->planes;
  size_t i;
  int a, b, k;
  for (k = 0; k < 9; ++k) R[k] = _mm256_set1_ps(job->R[k]);
  for (k = 0; k < 3; ++k) t[k] = _mm256_set1_ps(job->t[k]);
  for (k = 0; k < 21; ++k) H[k] = _mm256_setzero_ps();
  for (k = 0; k < 6; ++k) g[k] = _mm256_setzero_ps();
  cost = _mm256_setzero_ps();
  for (i = job->begin; i < job->end; i += 8) {
    __m256 p[3], mu[3], S[9], e[3], r[3], M[9], J[18], s, rho, w;
    for (k = 0; k < 3; ++k) p[k] = _mm256_loadu_ps(pl[k] + i);
    for (k = 0; k < 3; ++k) mu[k] = _mm256_loadu_ps(pl[3 + k] + i);
    for (k = 0; k < 9; ++k) S[k] = _mm256_loadu_ps(pl[6 + k] + i);
    for (a = 0; a < 3; ++a) {
      __m256 pw = _mm256_fmadd_ps(
          R[3 * a], p[0], _mm256_fmadd_ps(R[3 * a + 1], p[1], _mm256_fmadd_ps(R[3 * a + 2], p[2], t[a])));
      e[a] = _mm256_sub_ps(pw, mu[a]);
    }
    for (a = 0; a < 3; ++a)
      r[a] = _mm256_fmadd_ps(S[3 * a], e[0],
                             _mm256_fmadd_ps(S[3 * a + 1], e[1], _mm256_mul_ps(S[3 * a + 2], e[2])));
    for (a = 0; a < 3; ++a) {
      M[3 * a + 0] = _mm256_fmsub_ps(R[3 * a + 2], p[1], _mm256_mul_ps(R[3 * a + 1], p[2]));
      M[3 * a + 1] = _mm256_fmsub_ps(R[3 * a + 0], p[2], _mm256_mul_ps(R[3 * a + 2], p[0]));
      M[3 * a + 2] = _mm256_fmsub_ps(R[3 * a + 1], p[0], _mm256_mul_ps(R[3 * a + 0], p[1]));
    }
    for (a = 0; a < 3; ++a)
      for (b = 0; b < 3; ++b) {
        J[6 * a + b] = S[3 * a + b];
        J[6 * a + 3 + b] = _mm256_fmadd_ps(
            S[3 * a], M[b], _mm256_fmadd_ps(S[3 * a + 1], M[3 + b], _mm256_mul_ps(S[3 * a + 2], M[6 + b])));
      }
    s = _mm256_fmadd_ps(r[0], r[0], _mm256_fmadd_ps(r[1], r[1], _mm256_mul_ps(r[2], r[2])));
    rho = s;
    w = _mm256_set1_ps(1.0f);
    if (job->loss.kind != 0) {
      float sb[8], lb[8], wb[8];
      _mm256_storeu_ps(sb, s);
      for (k = 0; k < 8; ++k) {
        double lr, lw;
        oracle_loss_evaluate(&job->loss, (double)sb[k], &lr, &lw);
        lb[k] = (float)lr;
        wb[k] = (float)lw;
      }
      rho = _mm256_loadu_ps(lb);
      w = _mm256_loadu_ps(wb);
    }
    for (a = 0; a < 6; ++a)
      g[a] = _mm256_add_ps(
          g[a],
          _mm256_mul_ps(w, _mm256_fmadd_ps(J[a], r[0],
                                           _mm256_fmadd_ps(J[6 + a], r[1], _mm256_mul_ps(J[12 + a], r[2])))));
    k = 0;
    for (a = 0; a < 6; ++a)
      for (b = a; b < 6; ++b) {
        H[k] = _mm256_add_ps(
            H[k], _mm256_mul_ps(w, _mm256_fmadd_ps(J[a], J[b],
                                                   _mm256_fmadd_ps(J[6 + a], J[6 + b],
                                                                   _mm256_mul_ps(J[12 + a], J[12 + b])))));
        ++k;
      }
    cost = _mm256_add_ps(cost, rho);
  }
  for (k = 0; k < 21; ++k) job->out28[k] = (double)hsum8(H[k]);
  for (k = 0; k < 6; ++k) job->out28[21 + k] = (double)hsum8(g[k]);
  job->out28[27] = (double)hsum8(cost);
}


static inline void lane_loss(const oracle_loss* loss, __m256 s, __m256* rho, __m256* w) {
  /* per lane through the scalar double virtual, as the reference does (…_3dof_simd.cc:131-141, REM/…_simd.cc:78-91) */
  float sb[8], lb[8], wb[8];
  int k;
  _mm256_storeu_ps(sb, s);
  for (k = 0; k < 8; ++k) {
    double lr, lw;
    oracle_loss_evaluate(loss, (double)sb[k], &lr, &lw);
    lb[k] = (float)lr;
    wb[k] = (float)lw;
  }
  *rho = _mm256_loadu_ps(lb);
  *w = _mm256_loadu_ps(wb);
}

/* MDM/mahalanobis_distance_minimizer_analytic_3dof_simd.cc:85-158; R = {R00 R01 R10 R11}, t = {tx ty}. out: H upper (6), g (3), cost. */
static void avx_range3(avx_job* job) {
  __m256 R[4], t[2], H[6], g[3], cost;
  const float* const* pl = job->planes;
  size_t i;
  int a, b, k;
  for (k = 0; k < 4; ++k) R[k] = _mm256_set1_ps(job->R[k]);
  for (k = 0; k < 2; ++k) t[k] = _mm256_set1_ps(job->t[k]);
  for (k = 0; k < 6; ++k) H[k] = _mm256_setzero_ps();
  for (k = 0; k < 3; ++k) g[k] = _mm256_setzero_ps();
  cost = _mm256_setzero_ps();
  for (i = job->begin; i < job->end; i += 8) {
    __m256 p[3], mu[3], S[9], e[3], r[3], J[9], d0, d1, s, rho, w;
    for (k = 0; k < 3; ++k) p[k] = _mm256_loadu_ps(pl[k] + i);
    for (k = 0; k < 3; ++k) mu[k] = _mm256_loadu_ps(pl[3 + k] + i);
    for (k = 0; k < 9; ++k) S[k] = _mm256_loadu_ps(pl[6 + k] + i);
    e[0] = _mm256_sub_ps(_mm256_fmadd_ps(R[0], p[0], _mm256_fmadd_ps(R[1], p[1], t[0])), mu[0]);
    e[1] = _mm256_sub_ps(_mm256_fmadd_ps(R[2], p[0], _mm256_fmadd_ps(R[3], p[1], t[1])), mu[1]);
    e[2] = _mm256_sub_ps(p[2], mu[2]);
    for (a = 0; a < 3; ++a)
      r[a] = _mm256_fmadd_ps(S[3 * a], e[0], _mm256_fmadd_ps(S[3 * a + 1], e[1], _mm256_mul_ps(S[3 * a + 2], e[2])));
    d0 = _mm256_fmsub_ps(R[1], p[0], _mm256_mul_ps(R[0], p[1]));
    d1 = _mm256_fmsub_ps(R[3], p[0], _mm256_mul_ps(R[2], p[1]));
    for (a = 0; a < 3; ++a) {
      J[3 * a + 0] = S[3 * a + 0];
      J[3 * a + 1] = S[3 * a + 1];
      J[3 * a + 2] = _mm256_fmadd_ps(S[3 * a], d0, _mm256_mul_ps(S[3 * a + 1], d1));
    }
    s = _mm256_fmadd_ps(r[0], r[0], _mm256_fmadd_ps(r[1], r[1], _mm256_mul_ps(r[2], r[2])));
    rho = s;
    w = _mm256_set1_ps(1.0f);
    if (job->loss.kind != 0) lane_loss(&job->loss, s, &rho, &w);
    for (a = 0; a < 3; ++a)
      g[a] = _mm256_add_ps(g[a], _mm256_mul_ps(w, _mm256_fmadd_ps(J[a], r[0], _mm256_fmadd_ps(J[3 + a], r[1],
                                                                                             _mm256_mul_ps(J[6 + a], r[2])))));
    k = 0;
    for (a = 0; a < 3; ++a)
      for (b = a; b < 3; ++b) {
        H[k] = _mm256_add_ps(H[k], _mm256_mul_ps(w, _mm256_fmadd_ps(J[a], J[b], _mm256_fmadd_ps(J[3 + a], J[3 + b],
                                                                                               _mm256_mul_ps(J[6 + a], J[6 + b])))));
        ++k;
      }
    cost = _mm256_add_ps(cost, rho);
  }
  memset(job->out28, 0, sizeof(job->out28));
  for (k = 0; k < 6; ++k) job->out28[k] = (double)hsum8(H[k]);
  for (k = 0; k < 3; ++k) job->out28[6 + k] = (double)hsum8(g[k]);
  job->out28[9] = (double)hsum8(cost);
}

/* REM/reprojection_error_minimizer_analytic_simd.cc:55-138; planes X Y Z px py. */
static void avx_range_reproj(avx_job* job) {
  __m256 R[9], t[3], H[21], g[6], cost, inv_fx, inv_fy, cx, cy;
  const float* const* pl = job->planes;
  const __m256 one = _mm256_set1_ps(1.0f), zero = _mm256_setzero_ps();
  size_t i;
  int a, b, k;
  for (k = 0; k < 9; ++k) R[k] = _mm256_set1_ps(job->R[k]);
  for (k = 0; k < 3; ++k) t[k] = _mm256_set1_ps(job->t[k]);
  inv_fx = _mm256_set1_ps(job->intr[0]);
  inv_fy = _mm256_set1_ps(job->intr[1]);
  cx = _mm256_set1_ps(job->intr[2]);
  cy = _mm256_set1_ps(job->intr[3]);
  for (k = 0; k < 21; ++k) H[k] = _mm256_setzero_ps();
  for (k = 0; k < 6; ++k) g[k] = _mm256_setzero_ps();
  cost = _mm256_setzero_ps();
  for (i = job->begin; i < job->end; i += 8) {
    __m256 X[3], px, py, Xw[3], M[9], J[12], r[2], mask, iz, iz2, xz, yz, s, rho, w;
    for (k = 0; k < 3; ++k) X[k] = _mm256_loadu_ps(pl[k] + i);
    px = _mm256_loadu_ps(pl[3] + i);
    py = _mm256_loadu_ps(pl[4] + i);
    for (a = 0; a < 3; ++a)
      Xw[a] = _mm256_fmadd_ps(R[3 * a], X[0], _mm256_fmadd_ps(R[3 * a + 1], X[1], _mm256_fmadd_ps(R[3 * a + 2], X[2], t[a])));
    mask = _mm256_and_ps(_mm256_cmp_ps(Xw[2], zero, _CMP_GT_OQ), one); /* 0/1 mask, :66 */
    iz = _mm256_div_ps(one, Xw[2]);
    r[0] = _mm256_fmsub_ps(Xw[0], iz, _mm256_mul_ps(inv_fx, _mm256_sub_ps(px, cx)));
    r[1] = _mm256_fmsub_ps(Xw[1], iz, _mm256_mul_ps(inv_fy, _mm256_sub_ps(py, cy)));
    s = _mm256_fmadd_ps(r[0], r[0], _mm256_mul_ps(r[1], r[1]));
    rho = s;
    w = one;
    if (job->loss.kind != 0) lane_loss(&job->loss, s, &rho, &w);
    w = _mm256_mul_ps(w, mask); /* :92 */
    for (a = 0; a < 3; ++a) {
      M[3 * a + 0] = _mm256_fmsub_ps(R[3 * a + 2], X[1], _mm256_mul_ps(R[3 * a + 1], X[2]));
      M[3 * a + 1] = _mm256_fmsub_ps(R[3 * a + 0], X[2], _mm256_mul_ps(R[3 * a + 2], X[0]));
      M[3 * a + 2] = _mm256_fmsub_ps(R[3 * a + 1], X[0], _mm256_mul_ps(R[3 * a + 0], X[1]));
    }
    iz2 = _mm256_mul_ps(iz, iz);
    xz = _mm256_mul_ps(Xw[0], iz2);
    yz = _mm256_mul_ps(Xw[1], iz2);
    J[0] = iz;
    J[1] = zero;
    J[2] = _mm256_sub_ps(zero, xz);
    J[6] = zero;
    J[7] = iz;
    J[8] = _mm256_sub_ps(zero, yz);
    for (b = 0; b < 3; ++b) {
      J[3 + b] = _mm256_fmsub_ps(iz, M[b], _mm256_mul_ps(xz, M[6 + b]));
      J[9 + b] = _mm256_fmsub_ps(iz, M[3 + b], _mm256_mul_ps(yz, M[6 + b]));
    }
    for (a = 0; a < 6; ++a)
      g[a] = _mm256_add_ps(g[a], _mm256_mul_ps(_mm256_fmadd_ps(J[a], r[0], _mm256_mul_ps(J[6 + a], r[1])), w));
    k = 0;
    for (a = 0; a < 6; ++a)
      for (b = a; b < 6; ++b) {
        H[k] = _mm256_add_ps(H[k], _mm256_mul_ps(_mm256_mul_ps(J[a], J[b]), w));
        H[k] = _mm256_add_ps(H[k], _mm256_mul_ps(_mm256_mul_ps(J[6 + a], J[6 + b]), w));
        ++k;
      }
    cost = _mm256_add_ps(cost, rho);
  }
  for (k = 0; k < 21; ++k) job->out28[k] = (double)hsum8(H[k]);
  for (k = 0; k < 6; ++k) job->out28[21 + k] = (double)hsum8(g[k]);
  job->out28[27] = (double)hsum8(cost);
}

typedef struct avx_job_f64 {
  const double* const* planes;
  size_t begin, end; /* multiples of 4 */
  double R[9], t[3];
  oracle_loss loss;
  double out28[28];
  int created;
} avx_job_f64;

static inline double hsum4(__m256d v) {
  double buf[4];
  _mm256_storeu_pd(buf, v);
  return buf[0] + buf[1] + buf[2] + buf[3]; /* `buf[0] + buf[1] + buf[2] + buf[3]`, MDM/..._simd_various.cc:139,144,147 */
}

/* MDM/mahalanobis_distance_minimizer_analytic_simd_various.cc:42-134 (SolveDouble's inner loop), 4-lane fp64.  The
 * helper's operators are plain mul / add / sub on __m256d (NO/simd_helper/simd_scalar_amd.h: no fused forms), what the
 * compiler then contracts under the reference's -O2 -march=native is its business; here the products feed explicit FMAs
 * like the fp32 restatement above (both are timing baselines, checked against the scalar oracle to 1e-12). */
static void avx_range_f64(avx_job_f64* job) {
  __m256d R[9], t[3], H[21], g[6], cost;
  const double* const* pl = job->planes;
  size_t i;
  int a, b, k;
  for (k = 0; k < 9; ++k) R[k] = _mm256_set1_pd(job->R[k]);
  for (k = 0; k < 3; ++k) t[k] = _mm256_set1_pd(job->t[k]);
  for (k = 0; k < 21; ++k) H[k] = _mm256_setzero_pd();
  for (k = 0; k < 6; ++k) g[k] = _mm256_setzero_pd();
  cost = _mm256_setzero_pd();
  for (i = job->begin; i < job->end; i += 4) {
    __m256d p[3], mu[3], S[9], e[3], r[3], M[9], J[18], s, rho, w;
    for (k = 0; k < 3; ++k) p[k] = _mm256_loadu_pd(pl[k] + i);
    for (k = 0; k < 3; ++k) mu[k] = _mm256_loadu_pd(pl[3 + k] + i);
    for (k = 0; k < 9; ++k) S[k] = _mm256_loadu_pd(pl[6 + k] + i);
    for (a = 0; a < 3; ++a) {
      __m256d pw = _mm256_fmadd_pd(
          R[3 * a], p[0], _mm256_fmadd_pd(R[3 * a + 1], p[1], _mm256_fmadd_pd(R[3 * a + 2], p[2], t[a])));
      e[a] = _mm256_sub_pd(pw, mu[a]);
    }
    for (a = 0; a < 3; ++a)
      r[a] = _mm256_fmadd_pd(S[3 * a], e[0],
                             _mm256_fmadd_pd(S[3 * a + 1], e[1], _mm256_mul_pd(S[3 * a + 2], e[2])));
    for (a = 0; a < 3; ++a) { /* minus_R_skewp, :69-77 */
      M[3 * a + 0] = _mm256_fmsub_pd(R[3 * a + 2], p[1], _mm256_mul_pd(R[3 * a + 1], p[2]));
      M[3 * a + 1] = _mm256_fmsub_pd(R[3 * a + 0], p[2], _mm256_mul_pd(R[3 * a + 2], p[0]));
      M[3 * a + 2] = _mm256_fmsub_pd(R[3 * a + 1], p[0], _mm256_mul_pd(R[3 * a + 0], p[1]));
    }
    for (a = 0; a < 3; ++a)
      for (b = 0; b < 3; ++b) {
        J[6 * a + b] = S[3 * a + b];
        J[6 * a + 3 + b] = _mm256_fmadd_pd(
            S[3 * a], M[b], _mm256_fmadd_pd(S[3 * a + 1], M[3 + b], _mm256_mul_pd(S[3 * a + 2], M[6 + b])));
      }
    s = _mm256_fmadd_pd(r[0], r[0], _mm256_fmadd_pd(r[1], r[1], _mm256_mul_pd(r[2], r[2])));
    rho = s;
    w = _mm256_set1_pd(1.0);
    if (job->loss.kind != 0) { /* per lane through the scalar virtual, :107-121 */
      double sb[4], lb[4], wb[4];
      _mm256_storeu_pd(sb, s);
      for (k = 0; k < 4; ++k) oracle_loss_evaluate(&job->loss, sb[k], &lb[k], &wb[k]);
      rho = _mm256_loadu_pd(lb);
      w = _mm256_loadu_pd(wb);
    }
    for (a = 0; a < 6; ++a)
      g[a] = _mm256_add_pd(
          g[a],
          _mm256_mul_pd(_mm256_fmadd_pd(J[a], r[0], _mm256_fmadd_pd(J[6 + a], r[1], _mm256_mul_pd(J[12 + a], r[2]))),
                        w));
    k = 0;
    for (a = 0; a < 6; ++a)
      for (b = a; b < 6; ++b) {
        H[k] = _mm256_add_pd(
            H[k], _mm256_mul_pd(w, _mm256_fmadd_pd(J[a], J[b],
                                                   _mm256_fmadd_pd(J[6 + a], J[6 + b],
                                                                   _mm256_mul_pd(J[12 + a], J[12 + b])))));
        ++k;
      }
    cost = _mm256_add_pd(cost, rho);
  }
  for (k = 0; k < 21; ++k) job->out28[k] = hsum4(H[k]);
  for (k = 0; k < 6; ++k) job->out28[21 + k] = hsum4(g[k]);
  job->out28[27] = hsum4(cost);
}

static void* avx_thread_f64(void* arg) {
  avx_range_f64((avx_job_f64*)arg);
  return NULL;
}

static void* avx_thread(void* arg) {
  avx_job* job = (avx_job*)arg;
  if (job->kind == 1)
    avx_range3(job);
  else if (job->kind == 2)
    avx_range_reproj(job);
  else
    avx_range(job);
  return NULL;
}

/* loss evaluation lives in the scalar oracle; keep this library self-contained */
void oracle_loss_evaluate(const oracle_loss* loss, double s, double* rho, double* w) {
  if (loss == NULL || loss->kind == 0) {
    *rho = s;
    *w = 1.0;
  } else if (loss->kind == 1) {
    const double ex = exp(-loss->b * s);
    *rho = loss->a - loss->a * ex;
    *w = 2.0 * loss->a * loss->b * ex;
  } else {
    const double th = loss->a, th2 = th * th;
    if (s > th2) {
      const double rr = sqrt(s);
      *rho = 2.0 * th * rr - th2;
      *w = th / rr;
    } else {
      *rho = s;
      *w = 1.0;
    }
  }
}

/* Processes floor(n/8)*8 items split into `threads` contiguous batches of floor(floor(n/8)/threads)*8 (last batch
 * clipped), exactly the reference's partition (MDM/..._analytic_simd.cc:57-69) — so with T threads up to 8*T-8 further
 * tail items are dropped, as in the reference. */
static int avx_run(int kind, size_t n, const float* const* planes, const double* R, int nR, const double* t, int nt,
                   const double* intr, const oracle_loss* loss, int threads, double* out, int n_out) {
  const size_t num_stride = n / 8;
  avx_job* jobs;
  pthread_t* tids;
  int i, k;
  if (threads < 1) threads = 1;
  jobs = (avx_job*)calloc((size_t)threads, sizeof(avx_job));
  tids = (pthread_t*)calloc((size_t)threads, sizeof(pthread_t));
  if (!jobs || !tids) {
    free(jobs);
    free(tids);
    return 1;
  }
  {
    const size_t num_batch = (threads == 1) ? num_stride * 8 : (num_stride / (size_t)threads) * 8;
    for (i = 0; i < threads; ++i) {
      size_t b = (size_t)i * num_batch, e = ((size_t)i + 1) * num_batch;
      if (e > num_stride * 8) e = num_stride * 8;
      if (b > e) b = e;
      jobs[i].kind = kind;
      jobs[i].planes = planes;
      jobs[i].begin = b;
      jobs[i].end = e;
      for (k = 0; k < nR; ++k) jobs[i].R[k] = (float)R[k];
      for (k = 0; k < nt; ++k) jobs[i].t[k] = (float)t[k];
      if (intr) {
        /* the SIMD class broadcasts 1.0f / fx computed in float, and cx, cy cast to float (REM/..._simd.cc:29-32);
         * intr here = {inv_fx, inv_fy, cx, cy} in double like the C ABI: recover fx as 1/inv_fx first */
        jobs[i].intr[0] = 1.0f / (float)(1.0 / intr[0]);
        jobs[i].intr[1] = 1.0f / (float)(1.0 / intr[1]);
        jobs[i].intr[2] = (float)intr[2];
        jobs[i].intr[3] = (float)intr[3];
      }
      if (loss) jobs[i].loss = *loss;
    }
  }
  if (threads == 1) {
    avx_thread(&jobs[0]);
  } else {
    /* a thread that cannot be created (process / thread limit of the box) has its batch run inline: the sums are complete
     * either way, and no join is attempted on a thread id that was never filled in */
    for (i = 0; i < threads; ++i) {
      jobs[i].created = pthread_create(&tids[i], NULL, avx_thread, &jobs[i]) == 0;
      if (!jobs[i].created) avx_thread(&jobs[i]);
    }
    for (i = 0; i < threads; ++i)
      if (jobs[i].created) pthread_join(tids[i], NULL);
  }
  memset(out, 0, (size_t)n_out * sizeof(double));
  for (i = 0; i < threads; ++i)
    for (k = 0; k < n_out; ++k) out[k] += jobs[i].out28[k];
  free(jobs);
  free(tids);
  return 0;
}

int oracle_avx_ndt6_accumulate(size_t n, const float* const planes[15], const double R[9],
                               const double t[3], const oracle_loss* loss, int threads,
                               double out28[28]) {
  return avx_run(0, n, planes, R, 9, t, 3, NULL, loss, threads, out28, 28);
}

int oracle_avx_ndt3_accumulate(size_t n, const float* const planes[15], const double R2[4],
                               const double t2[2], const oracle_loss* loss, int threads,
                               double out10[10]) {
  return avx_run(1, n, planes, R2, 4, t2, 2, NULL, loss, threads, out10, 10);
}

int oracle_avx_reproj_accumulate(size_t n, const float* const planes[5], const double R[9],
                                 const double t[3], const double intr[4], const oracle_loss* loss,
                                 int threads, double out28[28]) {
  return avx_run(2, n, planes, R, 9, t, 3, intr, loss, threads, out28, 28);
}

/* floor(n/4)*4 items (MDM/..._simd_various.cc:41-43) in `threads` contiguous batches of floor(floor(n/4)/threads)*4 — the
 * 6-DoF SIMD class's partition rule (MDM/..._analytic_simd.cc:57-69) on this variant's stride; threads = 1 is what the
 * reference's SolveDouble does. */
int oracle_avx_ndt6_accumulate_f64(size_t n, const double* const planes[15], const double R[9], const double t[3],
                                   const oracle_loss* loss, int threads, double out28[28]) {
  const size_t num_stride = n / 4;
  avx_job_f64* jobs;
  pthread_t* tids;
  int i, k;
  if (threads < 1) threads = 1;
  jobs = (avx_job_f64*)calloc((size_t)threads, sizeof(avx_job_f64));
  tids = (pthread_t*)calloc((size_t)threads, sizeof(pthread_t));
  if (!jobs || !tids) {
    free(jobs);
    free(tids);
    return 1;
  }
  {
    const size_t num_batch = (threads == 1) ? num_stride * 4 : (num_stride / (size_t)threads) * 4;
    for (i = 0; i < threads; ++i) {
      size_t b = (size_t)i * num_batch, e = ((size_t)i + 1) * num_batch;
      if (e > num_stride * 4) e = num_stride * 4;
      if (b > e) b = e;
      jobs[i].planes = planes;
      jobs[i].begin = b;
      jobs[i].end = e;
      for (k = 0; k < 9; ++k) jobs[i].R[k] = R[k];
      for (k = 0; k < 3; ++k) jobs[i].t[k] = t[k];
      if (loss) jobs[i].loss = *loss;
    }
  }
  if (threads == 1) {
    avx_thread_f64(&jobs[0]);
  } else {
    for (i = 0; i < threads; ++i) {  /* as above: a batch whose thread cannot be created runs inline */
      jobs[i].created = pthread_create(&tids[i], NULL, avx_thread_f64, &jobs[i]) == 0;
      if (!jobs[i].created) avx_thread_f64(&jobs[i]);
    }
    for (i = 0; i < threads; ++i)
      if (jobs[i].created) pthread_join(tids[i], NULL);
  }
  memset(out28, 0, 28 * sizeof(double));
  for (i = 0; i < threads; ++i)
    for (k = 0; k < 28; ++k) out28[k] += jobs[i].out28[k];
  free(jobs);
  free(tids);
  return 0;
}

/* ---------------------------------------------------------------- AoS -> SoA pack (CPU baseline of the ingestion stage)
 *
 * Restates the loop every SIMD-class Solve() opens with (MDM/mahalanobis_distance_minimizer_analytic_simd.cc:19-28):
 *   for each correspondence: points.Append(corr.point.cast<float>()); means.Append(corr.ndt.mean.cast<float>());
 *                            sqrt_infos.Append(corr.ndt.sqrt_information.cast<float>());
 * i.e. 15 doubles gathered out of every 304-byte record (MDM/types.h:11-26) into 15 planar float arrays, single-threaded,
 * in index order.  field_offset[f] = byte offset of plane f's double inside a record.  Returns 0. */
int oracle_pack_records_f32(size_t n, const unsigned char* records, size_t stride, const size_t field_offset[15],
                            float* const planes[15]) {
  size_t i;
  int f;
  for (i = 0; i < n; ++i) {
    const unsigned char* rec = records + i * stride;
    for (f = 0; f < 15; ++f) {
      double v;
      memcpy(&v, rec + field_offset[f], sizeof v);
      planes[f][i] = (float)v;
    }
  }
  return 0;
}
