/*
 * scene_oracle.c — TEST INFRASTRUCTURE, NOT PRODUCT CODE (same rules as nos_oracle.h).
 *
 * Restatement of the part of the reference's NDT test harness whose ROUNDING decides the
 * captured COST lines (results/maha_amd64_simple.txt, maha_3_vs_6_amd64.txt, maha_amd64.txt):
 *
 *   GenerateGlobalPoints  MDM/tests/simple_optimization_test.cc:170-204
 *   ComputeVoxelKey       …:283-294
 *   UpdateNdtMap          …:236-281   count / sum / moment in point order, mean, covariance,
 *                                     Eigen::SelfAdjointEigenSolver<Matrix3d>, eigenvalue floor,
 *                                     sqrt_information = D^-1/2 · V   (NOT V^T — harness formula)
 *
 * Why rounding matters: a full 1 m x 1 m floor / wall patch of the 1 cm grid has two EQUAL
 * in-plane variances in exact arithmetic, so the 2x2 block [[a, e],[e, a']] the eigen-solver
 * sees consists of the rounding noise of the accumulation (a - a' and e are both ~1e-15), its
 * eigenvectors sit at a noise-determined angle, and because the harness stores D^-1/2·V (rows
 * of V, not columns, meet the eigenvalues) that angle decides which direction the voxel's
 * strong constraint points to.  Reproducing the captured runs therefore needs the
 * accumulation and Eigen's solver bit for bit.
 *
 * Third-party code restated here (absent from /root/reference, un-pinned there:
 * `find_package(Eigen3 REQUIRED)`, MDM/CMakeLists.txt:6): Eigen 3.3.x / 3.4.0
 *   SelfAdjointEigenSolver<Matrix3d>::compute()      (Eigen/src/Eigenvalues/SelfAdjointEigenSolver.h)
 *     scale by max |coeff| of the lower triangle, tridiagonalization_inplace (3x3 real special
 *     case, Tridiagonalization.h), computeFromTridiagonal_impl (deflation test, implicit
 *     symmetric QR step with Wilkinson shift, selection sort ascending with the vectors),
 *     JacobiRotation::makeGivens (Jacobi.h), numext::hypot (MathFunctions.h).
 * The two Eigen releases differ in the deflation test and in the shift's guard; both are here
 * (`eigen_version`), and every multiply-add site can be evaluated fused or unfused
 * (`fma_mask`), because the reference is built with `-O2 -march=native` (root
 * CMakeLists.txt:1-30) where GCC contracts a*b+c into FMA.  tests/golden/make_ndt_scene_golden.py
 * records which setting reproduces the captured COST lines.
 *
 * Compile with -ffp-contract=off: fusing is explicit (fma()) so the library computes the same
 * bits on every host.
 */
#include <float.h>
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* multiply-add sites, one bit each in fma_mask; the two accumulation sites have one bit per matrix element
 * (row-major index e: bit e of the 9-bit field) because Eigen evaluates a Matrix3d in packets of two doubles
 * plus one scalar tail element, and only what stays in registers can be contracted. */
enum {
  SITE_TRIDIAG = 4, /* tridiagonalization_inplace 3x3 */
  SITE_SQRT1P = 8,  /* sqrt(1 + t*t) in makeGivens and hypot */
  SITE_QR = 16,     /* tridiagonal_qr_step T = G' T G */
  SITE_Q = 32,      /* eivec.applyOnTheRight(k, k+1, rot) */
  SITE_MOMENT_SHIFT = 8, /* bits 8..16:  ndt.moment(e) += point * point^T      (…test.cc:247) */
  SITE_COV_SHIFT = 17    /* bits 17..25: moment(e) / count - mean * mean^T     (…test.cc:258-259) */
};

static inline double madd(int fused, double a, double b, double c) { return fused ? fma(a, b, c) : a * b + c; }

/* ----------------------------------------------------------------- scene ---- */

/* MDM/tests/simple_optimization_test.cc:170-204; loop variables accumulate in floating point
 * exactly as there.  Returns the number of points; writes at most cap of them (xyz triples). */
size_t scene_generate_global_points(double* out, size_t cap) {
  const double width = 5.0, length = 7.0, height = 2.5, point_step = 0.01;
  size_t n = 0;
  double x, y, z;
#define PUSH(a, b, c)          \
  do {                         \
    if (n < cap) {             \
      out[3 * n + 0] = (a);    \
      out[3 * n + 1] = (b);    \
      out[3 * n + 2] = (c);    \
    }                          \
    ++n;                       \
  } while (0)
  z = 0.0;
  for (x = -length / 2.0; x <= length / 2.0; x += point_step)
    for (y = -width / 2.0; y <= width / 2.0; y += point_step) PUSH(x, y, z);
  y = -width / 2.0;
  for (x = -length / 2.0; x <= length / 2.0; x += point_step)
    for (z = 0.0; z <= height; z += point_step) {
      PUSH(x, y, z);
      PUSH(x, -y, z);
    }
  x = -length / 2.0;
  for (y = -width / 2.0; y <= width / 2.0; y += point_step)
    for (z = 0.0; z <= height; z += point_step) {
      PUSH(-x, y, z);
      PUSH(x, y, z);
    }
#undef PUSH
  return n;
}

/* …test.cc:283-294 (Cantor pairing; the first pairing is evaluated in `int`). */
uint64_t scene_voxel_key(const double p[3], double inv_res) {
  int xk = (int)floor(p[0] * inv_res), yk = (int)floor(p[1] * inv_res), zk = (int)floor(p[2] * inv_res);
  xk = xk >= 0 ? 2 * xk : -2 * xk - 1;
  yk = yk >= 0 ? 2 * yk : -2 * yk - 1;
  zk = zk >= 0 ? 2 * zk : -2 * zk - 1;
  uint64_t xy = (uint64_t)(int64_t)((xk + yk) * (xk + yk + 1) / 2 + yk);
  return (xy + (uint64_t)(int64_t)zk) * (xy + (uint64_t)(int64_t)zk + 1) / 2 + (uint64_t)(int64_t)zk;
}

/* ---------------------------------------------- Eigen 3x3 self-adjoint solver ---- */

typedef struct {
  double c, s;
} givens;

/* Eigen/src/Jacobi/Jacobi.h, JacobiRotation<double>::makeGivens(p, q) (real case). */
static givens make_givens(double p, double q, int fm) {
  givens g;
  if (q == 0.0) {
    g.c = p < 0.0 ? -1.0 : 1.0;
    g.s = 0.0;
  } else if (p == 0.0) {
    g.c = 0.0;
    g.s = q < 0.0 ? 1.0 : -1.0;
  } else if (fabs(p) > fabs(q)) {
    double t = q / p;
    double u = sqrt(madd(fm & SITE_SQRT1P, t, t, 1.0));
    if (p < 0.0) u = -u;
    g.c = 1.0 / u;
    g.s = -t * g.c;
  } else {
    double t = p / q;
    double u = sqrt(madd(fm & SITE_SQRT1P, t, t, 1.0));
    if (q < 0.0) u = -u;
    g.s = -1.0 / u;
    g.c = -t * g.s;
  }
  return g;
}

/* Eigen/src/Core/MathFunctions.h, numext::hypot (Eigen's own scaled form, both releases). */
static double eigen_hypot(double x, double y, int fm) {
  double ax = fabs(x), ay = fabs(y), p, qp;
  if (ax > ay) {
    p = ax;
    qp = ay / p;
  } else {
    p = ay;
    qp = ax / p;
  }
  if (p == 0.0) return 0.0;
  return p * sqrt(madd(fm & SITE_SQRT1P, qp, qp, 1.0));
}

/* tridiagonal_qr_step (SelfAdjointEigenSolver.h); Q is column-major 3x3 (Eigen's default). */
static void qr_step(double* diag, double* sub, int start, int end, double* Q, int n, int version, int fm) {
  double td = (diag[end - 1] - diag[end]) * 0.5;
  double e = sub[end - 1];
  double mu = diag[end];
  int k, i;
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
  const int fq = fm & SITE_QR;
  for (k = start; k < end && (version != 34 || z != 0.0); ++k) {
    givens r = make_givens(x, z, fm);
    const double c = r.c, s = r.s;
    /* T = G' T G */
    const double c_sub = c * sub[k];
    /* contraction shapes as g++ 11 -O2 -mfma chooses them for Eigen's statements: the products with c are
     * shared (plain multiplies), the products with s are the fused ones */
    double sdk = madd(fq, s, diag[k], c_sub);             /* s*diag[k] + c*subdiag[k]   */
    double dkp1 = madd(fq, s, sub[k], c * diag[k + 1]);   /* s*subdiag[k] + c*diag[k+1] */
    double in1 = madd(fq, -s, sub[k], c * diag[k]);       /* c*diag[k] - s*subdiag[k]   */
    double in2 = madd(fq, -s, diag[k + 1], c_sub);        /* c*subdiag[k] - s*diag[k+1] */
    diag[k] = madd(fq, c, in1, -(s * in2));
    diag[k + 1] = madd(fq, s, sdk, c * dkp1);
    sub[k] = madd(fq, c, sdk, -(s * dkp1));
    if (k > start) sub[k - 1] = madd(fq, c, sub[k - 1], -(s * z));
    x = sub[k];
    if (k < end - 1) {
      z = -s * sub[k + 1];
      sub[k + 1] = c * sub[k + 1];
    }
    /* Q = Q * G : applyOnTheRight(k, k+1, rot) → x_i = c x_i - s y_i ; y_i = s x_i + c y_i */
    if (Q) {
      const int fqq = fm & SITE_Q;
      for (i = 0; i < n; ++i) {
        double xi = Q[k * n + i], yi = Q[(k + 1) * n + i];
        Q[k * n + i] = madd(fqq, c, xi, -(s * yi));
        Q[(k + 1) * n + i] = madd(fqq, s, xi, c * yi);
      }
    }
  }
}

/* SelfAdjointEigenSolver<Matrix3d>::compute(A, ComputeEigenvectors).  A row-major symmetric (only the
 * lower triangle is read, as in Eigen).  evals ascending, evecs COLUMN-major (evecs[3*k + i] = component i
 * of eigenvector k).  Returns 0 on success (Eigen::Success), 1 on NoConvergence. */
int scene_eigen_selfadjoint3(const double A[9], int version, int fm, double evals[3], double evecs[9]) {
  double m[3][3];
  double diag[3], sub[2];
  double Q[9];
  int i, j;
  /* mat = A.triangularView<Lower>() ; scale = max |mat| ; lower /= scale */
  double scale = 0.0;
  for (i = 0; i < 3; ++i)
    for (j = 0; j < 3; ++j) {
      m[i][j] = j <= i ? A[3 * i + j] : 0.0;
      if (fabs(m[i][j]) > scale) scale = fabs(m[i][j]);
    }
  if (scale == 0.0) scale = 1.0;
  for (i = 0; i < 3; ++i)
    for (j = 0; j <= i; ++j) m[i][j] /= scale;

  /* tridiagonalization_inplace_selector<MatrixType,3,false>::run */
  {
    const double tol = DBL_MIN;
    const int ft = fm & SITE_TRIDIAG;
    diag[0] = m[0][0];
    double v1norm2 = m[2][0] * m[2][0];
    if (v1norm2 <= tol) {
      diag[1] = m[1][1];
      diag[2] = m[2][2];
      sub[0] = m[1][0];
      sub[1] = m[2][1];
      for (i = 0; i < 9; ++i) Q[i] = 0.0;
      Q[0] = Q[4] = Q[8] = 1.0;
    } else {
      double beta = sqrt(madd(ft, m[1][0], m[1][0], v1norm2));
      double invBeta = 1.0 / beta;
      double m01 = m[1][0] * invBeta;
      double m02 = m[2][0] * invBeta;
      double q = madd(ft, 2.0 * m01, m[2][1], m02 * (m[2][2] - m[1][1]));
      diag[1] = madd(ft, m02, q, m[1][1]);
      diag[2] = madd(ft, -m02, q, m[2][2]);
      sub[0] = beta;
      sub[1] = madd(ft, -m01, q, m[2][1]);
      /* mat << 1,0,0, 0,m01,m02, 0,m02,-m01 (symmetric, so storage order is irrelevant) */
      for (i = 0; i < 9; ++i) Q[i] = 0.0;
      Q[0] = 1.0;
      Q[4] = m01;
      Q[5] = m02;
      Q[7] = m02;
      Q[8] = -m01;
    }
  }

  /* computeFromTridiagonal_impl */
  {
    const int n = 3, max_iterations = 30;
    int end = n - 1, start = 0, iter = 0;
    const double consider_as_zero = DBL_MIN;
    const double precision_inv = 1.0 / DBL_EPSILON;
    const double precision = 2.0 * DBL_EPSILON;
    while (end > 0) {
      for (i = start; i < end; ++i) {
        if (version == 34) {
          if (fabs(sub[i]) < consider_as_zero) {
            sub[i] = 0.0;
          } else {
            const double scaled = precision_inv * sub[i];
            if (scaled * scaled <= (fabs(diag[i]) + fabs(diag[i + 1]))) sub[i] = 0.0;
          }
        } else {
          /* internal::isMuchSmallerThan(|sub|, |d_i| + |d_i+1|, precision) || |sub| <= considerAsZero */
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
      qr_step(diag, sub, start, end, Q, n, version, fm);
    }
    if (iter > max_iterations * n) return 1;
    /* selection sort, ascending, vectors follow */
    for (i = 0; i < n - 1; ++i) {
      int k = 0;
      for (j = 1; j < n - i; ++j)
        if (diag[i + j] < diag[i + k]) k = j;
      if (k > 0) {
        double tmp = diag[i];
        diag[i] = diag[k + i];
        diag[k + i] = tmp;
        for (j = 0; j < n; ++j) {
          tmp = Q[i * n + j];
          Q[i * n + j] = Q[(k + i) * n + j];
          Q[(k + i) * n + j] = tmp;
        }
      }
    }
  }
  for (i = 0; i < 3; ++i) evals[i] = diag[i] * scale;
  memcpy(evecs, Q, sizeof(Q));
  return 0;
}

/* ------------------------------------------------------------- UpdateNdtMap ---- */

typedef struct {
  uint64_t key;
  int count;
  double sum[3];
  double moment[9];
} voxel_acc;

/* Open-addressing table keyed by voxel key; voxels come out in first-seen order. */
typedef struct {
  voxel_acc* v;
  size_t n, cap;
  int64_t* slots;
  size_t n_slots;
} voxel_table;

static size_t table_find_or_add(voxel_table* t, uint64_t key) {
  size_t h = (size_t)(key * 0x9E3779B97F4A7C15ull) & (t->n_slots - 1);
  for (;;) {
    int64_t s = t->slots[h];
    if (s < 0) break;
    if (t->v[s].key == key) return (size_t)s;
    h = (h + 1) & (t->n_slots - 1);
  }
  if (t->n == t->cap) return (size_t)-1;
  t->slots[h] = (int64_t)t->n;
  voxel_acc* a = &t->v[t->n];
  memset(a, 0, sizeof(*a));
  a->key = key;
  a->moment[0] = a->moment[4] = a->moment[8] = 1.0; /* NDT::moment starts at Identity, MDM/types.h:14 */
  return t->n++;
}

/* UpdateNdtMap (…test.cc:236-281).  Outputs (first-seen voxel order, capacity max_voxels):
 *   keys[V], counts[V], means[V*3], sqrt_info[V*9] (row-major), valid[V], evals[V*3] (un-floored),
 *   evecs[V*9] (row-major V: evecs[9v + 3i + k] = component i of eigenvector k).
 * A voxel with count < 5 is invalid.  A voxel whose solve fails or whose largest eigenvalue is < 0.01 is
 * invalid too; the reference `return`s there (harness bug) — not reproduced, and no voxel of
 * the reference scene takes that branch.  Returns V, or -1 when max_voxels is too small. */
long scene_build_ndt_map(const double* points, size_t n_points, double voxel_resolution, int eigen_version,
                         int fma_mask, size_t max_voxels, uint64_t* keys, int* counts, double* means,
                         double* sqrt_info, int* valid, double* evals_out, double* evecs_out) {
  const double inv_res = 1.0 / voxel_resolution;
  voxel_table t;
  size_t i, v;
  int a, b;
  t.cap = max_voxels;
  t.n = 0;
  t.v = (voxel_acc*)malloc(sizeof(voxel_acc) * (max_voxels ? max_voxels : 1));
  t.n_slots = 64;
  while (t.n_slots < 4 * max_voxels) t.n_slots <<= 1;
  t.slots = (int64_t*)malloc(sizeof(int64_t) * t.n_slots);
  for (i = 0; i < t.n_slots; ++i) t.slots[i] = -1;

  const int fmom = (fma_mask >> SITE_MOMENT_SHIFT) & 0x1ff, fcov = (fma_mask >> SITE_COV_SHIFT) & 0x1ff;
  for (i = 0; i < n_points; ++i) {
    const double* p = points + 3 * i;
    size_t s = table_find_or_add(&t, scene_voxel_key(p, inv_res));
    if (s == (size_t)-1) {
      free(t.v);
      free(t.slots);
      return -1;
    }
    voxel_acc* acc = &t.v[s];
    ++acc->count;
    for (a = 0; a < 3; ++a) acc->sum[a] += p[a];
    for (a = 0; a < 3; ++a)
      for (b = 0; b < 3; ++b) acc->moment[3 * a + b] = madd((fmom >> (3 * a + b)) & 1, p[a], p[b], acc->moment[3 * a + b]);
  }

  for (v = 0; v < t.n; ++v) {
    const voxel_acc* acc = &t.v[v];
    double mean[3] = {0, 0, 0}, cov[9], ev[3], U[9];
    keys[v] = acc->key;
    counts[v] = acc->count;
    valid[v] = 0;
    for (a = 0; a < 3; ++a) means[3 * v + a] = 0.0;
    for (a = 0; a < 9; ++a) {
      sqrt_info[9 * v + a] = (a % 4 == 0) ? 1.0 : 0.0;
      evecs_out[9 * v + a] = 0.0;
    }
    for (a = 0; a < 3; ++a) evals_out[3 * v + a] = 0.0;
    if (acc->count < 5) continue;
    const double cnt = (double)acc->count;
    for (a = 0; a < 3; ++a) mean[a] = acc->sum[a] / cnt;
    for (a = 0; a < 3; ++a)
      for (b = 0; b < 3; ++b) cov[3 * a + b] = madd((fcov >> (3 * a + b)) & 1, -mean[a], mean[b], acc->moment[3 * a + b] / cnt);
    int info = scene_eigen_selfadjoint3(cov, eigen_version, fma_mask, ev, U);
    for (a = 0; a < 3; ++a) {
      evals_out[3 * v + a] = ev[a];
      for (b = 0; b < 3; ++b) evecs_out[9 * v + 3 * a + b] = U[3 * b + a];
    }
    if (info != 0 || ev[2] < 0.01) continue;
    const double ratio = 0.01;
    double d[3] = {ev[0], ev[1], ev[2]};
    d[0] = d[0] > d[2] * ratio ? d[0] : d[2] * ratio; /* std::max(eigvals(0), eigvals(2) * ratio) */
    d[1] = d[1] > d[2] * ratio ? d[1] : d[2] * ratio;
    for (a = 0; a < 3; ++a) means[3 * v + a] = mean[a];
    for (a = 0; a < 3; ++a) {
      const double w = sqrt(1.0 / d[a]); /* eigvals.cwiseInverse().cwiseSqrt() */
      for (b = 0; b < 3; ++b) sqrt_info[9 * v + 3 * a + b] = w * U[3 * b + a]; /* (D^-1/2 · V)(a, b) = w_a V(a, b) */
    }
    valid[v] = 1;
  }
  long V = (long)t.n;
  free(t.v);
  free(t.slots);
  return V;
}
