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
  oracle_loss loss;
  double out28[28];
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

static void* avx_thread(void* arg) {
  avx_range((avx_job*)arg);
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

/* planes: 15 float arrays.  Processes floor(n/8)*8 items split into `threads` contiguous
 * batches of floor(floor(n/8)/threads)*8 (last batch clipped), exactly the reference's
 * partition (MDM/..._analytic_simd.cc:57-69) — so with T threads up to 8*T-8 further tail
 * items are dropped, as in the reference. */
int oracle_avx_ndt6_accumulate(size_t n, const float* const planes[15], const double R[9],
                               const double t[3], const oracle_loss* loss, int threads,
                               double out28[28]) {
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
      jobs[i].planes = planes;
      jobs[i].begin = b;
      jobs[i].end = e;
      for (k = 0; k < 9; ++k) jobs[i].R[k] = (float)R[k];
      for (k = 0; k < 3; ++k) jobs[i].t[k] = (float)t[k];
      if (loss) jobs[i].loss = *loss;
    }
  }
  if (threads == 1) {
    avx_range(&jobs[0]);
  } else {
    for (i = 0; i < threads; ++i) pthread_create(&tids[i], NULL, avx_thread, &jobs[i]);
    for (i = 0; i < threads; ++i) pthread_join(tids[i], NULL);
  }
  memset(out28, 0, 28 * sizeof(double));
  for (i = 0; i < threads; ++i)
    for (k = 0; k < 28; ++k) out28[k] += jobs[i].out28[k];
  free(jobs);
  free(tids);
  return 0;
}
