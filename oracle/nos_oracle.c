/*
 * nos_oracle.c — TEST INFRASTRUCTURE, NOT PRODUCT CODE (see nos_oracle.h).
 *
 * Plain C, scalar fp64, index-order summation.  Compile with -ffp-contract=off so the
 * operation order written here is the operation order executed.
 *
 * Every function cites the reference lines it restates.  Third-party arithmetic the
 * reference takes from Eigen3 (un-pinned version, absent from /root/reference) is restated
 * from Eigen's published algorithms: Quaternion(Matrix3), toRotationMatrix(), quaternion
 * product/normalize, 3x3 cofactor inverse, partial-pivot LU inverse (6x6), pivoted LDLT.
 */
#include "nos_oracle.h"

#include <float.h>
#include <math.h>
#include <string.h>

/* ------------------------------------------------------------------ loss ---- */

/* NO/loss_function.h:28-33 (ExponentialLossFunction::Evaluate, scalar overload),
 * NO/loss_function.h:57-66 (HuberLossFunction::Evaluate, scalar overload),
 * kind 0 = `loss_function_ == nullptr` branch of MDM/..._analytic.cc:44-48. */
void oracle_loss_evaluate(const oracle_loss* loss, double s, double* rho, double* w) {
  if (loss == NULL || loss->kind == 0) {
    *rho = s;
    *w = 1.0;
    return;
  }
  if (loss->kind == 1) {
    const double c1 = loss->a, c2 = loss->b;
    const double two_c1c2 = 2.0 * c1 * c2;
    const double exp_term = exp(-c2 * s);
    *rho = c1 - c1 * exp_term;
    *w = two_c1c2 * exp_term;
    return;
  }
  {
    const double th = loss->a;
    const double th2 = th * th;
    if (s > th2) {
      const double residual = sqrt(s);
      *rho = 2.0 * th * residual - th2;
      *w = th / residual;
    } else {
      *rho = s;
      *w = 1.0;
    }
  }
}

/* ------------------------------------------------------- 6-DoF NDT item ---- */

/* MDM/mahalanobis_distance_minimizer_analytic.cc:159-185 (ComputeJacobianAndResidual).
 * x = {p(3), mu(3), S row-major(9)}. */
void oracle_ndt6_item(const double x[15], const double R[9], const double t[3], double r[3],
                      double J[18]) {
  const double* p = x;
  const double* mu = x + 3;
  const double* S = x + 6;
  double pw[3], e[3], Rs[9];
  int i, j;
  for (i = 0; i < 3; ++i) {
    pw[i] = (R[3 * i + 0] * p[0] + R[3 * i + 1] * p[1]) + R[3 * i + 2] * p[2];
    pw[i] = pw[i] + t[i];
    e[i] = pw[i] - mu[i];
  }
  for (i = 0; i < 3; ++i) r[i] = (S[3 * i + 0] * e[0] + S[3 * i + 1] * e[1]) + S[3 * i + 2] * e[2];
  /* R * skew(p), skew(p) = [0 -pz py; pz 0 -px; -py px 0]  (:171-179) */
  for (i = 0; i < 3; ++i) {
    Rs[3 * i + 0] = R[3 * i + 1] * p[2] + R[3 * i + 2] * (-p[1]);
    Rs[3 * i + 1] = R[3 * i + 0] * (-p[2]) + R[3 * i + 2] * p[0];
    Rs[3 * i + 2] = R[3 * i + 0] * p[1] + R[3 * i + 1] * (-p[0]);
  }
  /* J = [S | -S * R*skew(p)]  (:183-184) */
  for (i = 0; i < 3; ++i) {
    for (j = 0; j < 3; ++j) {
      J[6 * i + j] = S[3 * i + j];
      J[6 * i + 3 + j] =
          ((-S[3 * i + 0]) * Rs[0 + j] + (-S[3 * i + 1]) * Rs[3 + j]) + (-S[3 * i + 2]) * Rs[6 + j];
    }
  }
}

static void pack_out(int dim, const double* H /* dim x dim, upper valid */, const double* g,
                     double cost, double* out) {
  int k = 0, row, col;
  for (row = 0; row < dim; ++row)
    for (col = row; col < dim; ++col) out[k++] = H[dim * row + col];
  for (row = 0; row < dim; ++row) out[k++] = g[row];
  out[k] = cost;
}

/* MDM/mahalanobis_distance_minimizer_analytic.cc:12-52 (ComputeCostAndDerivatives) with
 * :187-199 (ComputeHessianOnlyUpperTriangle), :201-208 (MultiplyWeight...), :210-218
 * (AddHessian...). */
void oracle_ndt6_accumulate(size_t n, const double* const planes[15], const double R[9],
                            const double t[3], const oracle_loss* loss, double out28[28]) {
  double H[36], g[6], cost = 0.0;
  size_t idx;
  int row, col, k;
  const int has_loss = (loss != NULL && loss->kind != 0);
  memset(H, 0, sizeof H);
  memset(g, 0, sizeof g);
  for (idx = 0; idx < n; ++idx) {
    double x[15], r[3], J[18], lg[6], lH[36], s;
    for (k = 0; k < 15; ++k) x[k] = planes[k][idx];
    oracle_ndt6_item(x, R, t, r, J);
    /* local_gradient = J^T r  (:28) */
    for (row = 0; row < 6; ++row) lg[row] = (J[row] * r[0] + J[6 + row] * r[1]) + J[12 + row] * r[2];
    /* local hessian, upper triangle (:187-199) */
    for (row = 0; row < 6; ++row)
      for (col = row; col < 6; ++col) {
        double h = 0.0;
        for (k = 0; k < 3; ++k) h += J[6 * k + row] * J[6 * k + col];
        lH[6 * row + col] = h;
      }
    s = (r[0] * r[0] + r[1] * r[1]) + r[2] * r[2]; /* (:35) */
    if (has_loss) {
      double rho, w;
      oracle_loss_evaluate(loss, s, &rho, &w);
      for (row = 0; row < 6; ++row) g[row] += w * lg[row];
      for (row = 0; row < 6; ++row)
        for (col = row; col < 6; ++col) {
          lH[6 * row + col] *= w;
          H[6 * row + col] += lH[6 * row + col];
        }
      cost += rho;
    } else {
      for (row = 0; row < 6; ++row) g[row] += lg[row];
      for (row = 0; row < 6; ++row)
        for (col = row; col < 6; ++col) H[6 * row + col] += lH[6 * row + col];
      cost += s;
    }
  }
  pack_out(6, H, g, cost, out28);
}

/* -------------------------------------------------------------- 3-DoF NDT ---- */

/* MDM/mahalanobis_distance_minimizer_analytic_3dof.cc:110-139 (J, r) and :36-69 (loop).
 * All n items are processed (the reference truncates to floor(n/4)*4, :33-36). */
void oracle_ndt3_accumulate(size_t n, const double* const planes[15], const double R2[4],
                            const double t2[2], const oracle_loss* loss, double out10[10]) {
  double H[9], g[3], cost = 0.0;
  size_t idx;
  int row, col, k;
  const int has_loss = (loss != NULL && loss->kind != 0);
  memset(H, 0, sizeof H);
  memset(g, 0, sizeof g);
  for (idx = 0; idx < n; ++idx) {
    double p[3], mu[3], S[9], uw[2], e[3], r[3], d[2], J[9], lg[3], lH[9], s;
    for (k = 0; k < 3; ++k) p[k] = planes[k][idx];
    for (k = 0; k < 3; ++k) mu[k] = planes[3 + k][idx];
    for (k = 0; k < 9; ++k) S[k] = planes[6 + k][idx];
    uw[0] = (R2[0] * p[0] + R2[1] * p[1]) + t2[0];
    uw[1] = (R2[2] * p[0] + R2[3] * p[1]) + t2[1];
    e[0] = uw[0] - mu[0];
    e[1] = uw[1] - mu[1];
    e[2] = p[2] - mu[2]; /* z is not transformed (:123) */
    for (k = 0; k < 3; ++k) r[k] = (S[3 * k] * e[0] + S[3 * k + 1] * e[1]) + S[3 * k + 2] * e[2];
    d[0] = -R2[0] * p[1] + R2[1] * p[0]; /* (:131-132) */
    d[1] = -R2[2] * p[1] + R2[3] * p[0];
    /* J = [A | A d ; c | c d] with A = S(0:2,0:2), c = S(2,0:2)  (:133-136) */
    for (k = 0; k < 3; ++k) {
      J[3 * k + 0] = S[3 * k + 0];
      J[3 * k + 1] = S[3 * k + 1];
      J[3 * k + 2] = S[3 * k + 0] * d[0] + S[3 * k + 1] * d[1];
    }
    for (row = 0; row < 3; ++row) lg[row] = (J[row] * r[0] + J[3 + row] * r[1]) + J[6 + row] * r[2];
    for (row = 0; row < 3; ++row)
      for (col = row; col < 3; ++col) {
        double h = 0.0;
        for (k = 0; k < 3; ++k) h += J[3 * k + row] * J[3 * k + col];
        lH[3 * row + col] = h;
      }
    s = (r[0] * r[0] + r[1] * r[1]) + r[2] * r[2];
    if (has_loss) {
      double rho, w;
      oracle_loss_evaluate(loss, s, &rho, &w);
      for (row = 0; row < 3; ++row) g[row] += w * lg[row];
      for (row = 0; row < 3; ++row)
        for (col = row; col < 3; ++col) {
          lH[3 * row + col] *= w;
          H[3 * row + col] += lH[3 * row + col];
        }
      cost += rho;
    } else {
      for (row = 0; row < 3; ++row) g[row] += lg[row];
      for (row = 0; row < 3; ++row)
        for (col = row; col < 3; ++col) H[3 * row + col] += lH[3 * row + col];
      cost += s;
    }
  }
  pack_out(3, H, g, cost, out10);
}

/* ----------------------------------------------------------- reprojection ---- */

/* REM/reprojection_error_minimizer_analytic.cc:107-162 (J, r) and :31-64 (loop),
 * :164-171 (upper-triangle hessian).  intr = {inv_fx, inv_fy, cx, cy}. */
void oracle_reproj_accumulate(size_t n, const double* const planes[5], const double R[9],
                              const double t[3], const double intr[4], const oracle_loss* loss,
                              double min_depth, double out28[28]) {
  double H[36], g[6], cost = 0.0;
  size_t idx;
  int row, col, i, j;
  const int has_loss = (loss != NULL && loss->kind != 0);
  memset(H, 0, sizeof H);
  memset(g, 0, sizeof g);
  for (idx = 0; idx < n; ++idx) {
    double X[3], px[2], Xw[3], r[2], J[12], lg[6], lH[36], s;
    X[0] = planes[0][idx];
    X[1] = planes[1][idx];
    X[2] = planes[2][idx];
    px[0] = planes[3][idx];
    px[1] = planes[4][idx];
    for (i = 0; i < 3; ++i) {
      Xw[i] = (R[3 * i + 0] * X[0] + R[3 * i + 1] * X[1]) + R[3 * i + 2] * X[2];
      Xw[i] = Xw[i] + t[i];
    }
    if (Xw[2] < min_depth) { /* (:119-123) */
      memset(J, 0, sizeof J);
      r[0] = r[1] = 0.0;
    } else {
      double Rs[9], dK[6];
      const double iz = 1.0 / Xw[2];
      const double iz2 = iz * iz;
      r[0] = Xw[0] * iz - intr[0] * (px[0] - intr[2]);
      r[1] = Xw[1] * iz - intr[1] * (px[1] - intr[3]);
      dK[0] = iz;
      dK[1] = 0.0;
      dK[2] = -Xw[0] * iz2;
      dK[3] = 0.0;
      dK[4] = iz;
      dK[5] = -Xw[1] * iz2;
      for (i = 0; i < 3; ++i) {
        Rs[3 * i + 0] = R[3 * i + 1] * X[2] + R[3 * i + 2] * (-X[1]);
        Rs[3 * i + 1] = R[3 * i + 0] * (-X[2]) + R[3 * i + 2] * X[0];
        Rs[3 * i + 2] = R[3 * i + 0] * X[1] + R[3 * i + 1] * (-X[0]);
      }
      for (i = 0; i < 2; ++i)
        for (j = 0; j < 3; ++j) {
          J[6 * i + j] = dK[3 * i + j];
          J[6 * i + 3 + j] =
              ((-dK[3 * i + 0]) * Rs[0 + j] + (-dK[3 * i + 1]) * Rs[3 + j]) + (-dK[3 * i + 2]) * Rs[6 + j];
        }
    }
    for (row = 0; row < 6; ++row) lg[row] = J[row] * r[0] + J[6 + row] * r[1];
    for (row = 0; row < 6; ++row)
      for (col = row; col < 6; ++col) lH[6 * row + col] = 0.0 + (J[row] * J[col] + J[6 + row] * J[6 + col]);
    s = r[0] * r[0] + r[1] * r[1];
    if (has_loss) {
      double rho, w;
      oracle_loss_evaluate(loss, s, &rho, &w);
      for (row = 0; row < 6; ++row) g[row] += w * lg[row];
      for (row = 0; row < 6; ++row)
        for (col = row; col < 6; ++col) {
          lH[6 * row + col] *= w;
          H[6 * row + col] += lH[6 * row + col];
        }
      cost += rho;
    } else {
      for (row = 0; row < 6; ++row) g[row] += lg[row];
      for (row = 0; row < 6; ++row)
        for (col = row; col < 6; ++col) H[6 * row + col] += lH[6 * row + col];
      cost += s;
    }
  }
  pack_out(6, H, g, cost, out28);
}

/* ------------------------------------------ fp32 8-lane restatement (SIMD) ---- */

/* MDM/mahalanobis_distance_minimizer_analytic_simd.cc:113-177: operands cast to float
 * (:25-27,117-118), 8 lanes accumulate independently in fp32, lanes summed into double in
 * the epilogue (:158-174).  simd::exp of the external simd_helper is approximated by expf
 * (library absent: parity unpinned for its last bits). */
void oracle_ndt6_accumulate_f32lanes(size_t n, const double* const planes[15], const double R[9],
                                     const double t[3], const oracle_loss* loss, int drop_tail,
                                     double out28[28]) {
  enum { L = 8 };
  static const int tri[21][2] = {{0, 0}, {0, 1}, {0, 2}, {0, 3}, {0, 4}, {0, 5}, {1, 1},
                                 {1, 2}, {1, 3}, {1, 4}, {1, 5}, {2, 2}, {2, 3}, {2, 4},
                                 {2, 5}, {3, 3}, {3, 4}, {3, 5}, {4, 4}, {4, 5}, {5, 5}};
  float accH[21][L], accg[6][L], accc[L];
  float Rf[9], tf[3];
  const size_t n_used = drop_tail ? (n / L) * L : n;
  size_t base;
  int k, lane, a;
  const int kind = (loss == NULL) ? 0 : loss->kind;
  memset(accH, 0, sizeof accH);
  memset(accg, 0, sizeof accg);
  memset(accc, 0, sizeof accc);
  for (k = 0; k < 9; ++k) Rf[k] = (float)R[k];
  for (k = 0; k < 3; ++k) tf[k] = (float)t[k];
  for (base = 0; base < n_used; base += L) {
    for (lane = 0; lane < L; ++lane) {
      const size_t idx = base + (size_t)lane;
      float p[3], mu[3], S[9], pw[3], e[3], r[3], M[9], J[18], s, rho, w;
      int i, j;
      if (idx >= n_used) break;
      for (k = 0; k < 3; ++k) p[k] = (float)planes[k][idx];
      for (k = 0; k < 3; ++k) mu[k] = (float)planes[3 + k][idx];
      for (k = 0; k < 9; ++k) S[k] = (float)planes[6 + k][idx];
      for (i = 0; i < 3; ++i) {
        pw[i] = (Rf[3 * i] * p[0] + Rf[3 * i + 1] * p[1]) + Rf[3 * i + 2] * p[2] + tf[i];
        e[i] = pw[i] - mu[i];
      }
      for (i = 0; i < 3; ++i) r[i] = (S[3 * i] * e[0] + S[3 * i + 1] * e[1]) + S[3 * i + 2] * e[2];
      /* -R * hat(p): columns as in MDM/..._simd_various.cc:677-687 */
      for (i = 0; i < 3; ++i) {
        M[3 * i + 0] = Rf[3 * i + 2] * p[1] - Rf[3 * i + 1] * p[2];
        M[3 * i + 1] = Rf[3 * i + 0] * p[2] - Rf[3 * i + 2] * p[0];
        M[3 * i + 2] = Rf[3 * i + 1] * p[0] - Rf[3 * i + 0] * p[1];
      }
      for (i = 0; i < 3; ++i)
        for (j = 0; j < 3; ++j) {
          J[6 * i + j] = S[3 * i + j];
          J[6 * i + 3 + j] = (S[3 * i] * M[j] + S[3 * i + 1] * M[3 + j]) + S[3 * i + 2] * M[6 + j];
        }
      s = (r[0] * r[0] + r[1] * r[1]) + r[2] * r[2];
      if (kind == 1) {
        const float ex = expf((float)(-loss->b) * s);
        rho = (float)loss->a - (float)loss->a * ex;
        w = (float)(2.0 * loss->a * loss->b) * ex;
      } else if (kind == 2) {
        const float th = (float)loss->a;
        if (s > th * th) {
          const float rr = sqrtf(s);
          rho = 2.0f * th * rr - th * th;
          w = th / rr;
        } else {
          rho = s;
          w = 1.0f;
        }
      } else {
        rho = s;
        w = 1.0f;
      }
      for (a = 0; a < 6; ++a)
        accg[a][lane] += ((J[a] * r[0] + J[6 + a] * r[1]) + J[12 + a] * r[2]) * w;
      for (a = 0; a < 21; ++a) {
        const int ii = tri[a][0], jj = tri[a][1];
        accH[a][lane] += w * ((J[ii] * J[jj] + J[6 + ii] * J[6 + jj]) + J[12 + ii] * J[12 + jj]);
      }
      accc[lane] += rho;
    }
  }
  for (a = 0; a < 21; ++a) {
    double sum = 0.0;
    for (lane = 0; lane < L; ++lane) sum += accH[a][lane];
    out28[a] = sum;
  }
  for (a = 0; a < 6; ++a) {
    double sum = 0.0;
    for (lane = 0; lane < L; ++lane) sum += accg[a][lane];
    out28[21 + a] = sum;
  }
  {
    double sum = 0.0;
    for (lane = 0; lane < L; ++lane) sum += accc[lane];
    out28[27] = sum;
  }
}

/* --------------------------------------------- Eigen restatements (host) ---- */

/* Eigen::Quaternion(const Matrix3&) — Eigen/src/Geometry/Quaternion.h,
 * quaternionbase_assign_impl<Other,3,3>::run (Shoemake's method).  Call site:
 * `Orientation optimized_orientation(initial_pose.rotation())`, MDM/..._analytic.cc:87. */
void oracle_quat_from_matrix(const double R[9], double q[4]) {
  double tr = R[0] + R[4] + R[8];
  if (tr > 0.0) {
    tr = sqrt(tr + 1.0);
    q[0] = 0.5 * tr;
    tr = 0.5 / tr;
    q[1] = (R[7] - R[5]) * tr;
    q[2] = (R[2] - R[6]) * tr;
    q[3] = (R[3] - R[1]) * tr;
  } else {
    int i = 0, j, k;
    double tt;
    if (R[4] > R[0]) i = 1;
    if (R[8] > R[4 * i]) i = 2;
    j = (i + 1) % 3;
    k = (j + 1) % 3;
    tt = sqrt(R[4 * i] - R[4 * j] - R[4 * k] + 1.0);
    q[1 + i] = 0.5 * tt;
    tt = 0.5 / tt;
    q[0] = (R[3 * k + j] - R[3 * j + k]) * tt;
    q[1 + j] = (R[3 * j + i] + R[3 * i + j]) * tt;
    q[1 + k] = (R[3 * k + i] + R[3 * i + k]) * tt;
  }
}

/* Eigen::QuaternionBase::toRotationMatrix().  Call site MDM/..._analytic.cc:99,154. */
void oracle_quat_to_matrix(const double q[4], double R[9]) {
  const double w = q[0], x = q[1], y = q[2], z = q[3];
  const double tx = 2.0 * x, ty = 2.0 * y, tz = 2.0 * z;
  const double twx = tx * w, twy = ty * w, twz = tz * w;
  const double txx = tx * x, txy = ty * x, txz = tz * x;
  const double tyy = ty * y, tyz = tz * y, tzz = tz * z;
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

/* MahalanobisDistanceMinimizer::ComputeQuaternion, MDM/mahalanobis_distance_minimizer.cc:20-33
 * (same body inline in REM/reprojection_error_minimizer.h:35-52). */
void oracle_exp_quat(const double w[3], double q[4]) {
  const double theta = sqrt((w[0] * w[0] + w[1] * w[1]) + w[2] * w[2]);
  if (theta < 1e-6) {
    q[0] = 1.0;
    q[1] = 0.5 * w[0];
    q[2] = 0.5 * w[1];
    q[3] = 0.5 * w[2];
  } else {
    const double half_theta = theta * 0.5;
    const double sdt = sin(half_theta) / theta;
    q[0] = cos(half_theta);
    q[1] = sdt * w[0];
    q[2] = sdt * w[1];
    q[3] = sdt * w[2];
  }
}

/* Eigen quaternion product a*b then normalize(): `optimized_orientation *= dq;
 * optimized_orientation.normalize();`  MDM/..._analytic.cc:135-136. */
static void quat_mul_normalize(double a[4], const double b[4]) {
  const double w = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
  const double x = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
  const double y = a[0] * b[2] + a[2] * b[0] + a[3] * b[1] - a[1] * b[3];
  const double z = a[0] * b[3] + a[3] * b[0] + a[1] * b[2] - a[2] * b[1];
  const double nrm = sqrt(((x * x + y * y) + z * z) + w * w);
  a[0] = w / nrm;
  a[1] = x / nrm;
  a[2] = y / nrm;
  a[3] = z / nrm;
}

/* Partial-pivot LU inverse — the algorithm behind Eigen's Matrix6d::inverse()
 * (PartialPivLU for sizes > 4).  Call site MDM/..._analytic.cc:129. */
static void inverse_lu(int dim, const double* A, double* inv) {
  double a[36], b[36];
  int i, j, k;
  for (i = 0; i < dim * dim; ++i) a[i] = A[i];
  for (i = 0; i < dim; ++i)
    for (j = 0; j < dim; ++j) b[dim * i + j] = (i == j) ? 1.0 : 0.0;
  for (k = 0; k < dim; ++k) {
    int piv = k;
    double best = fabs(a[dim * k + k]);
    for (i = k + 1; i < dim; ++i)
      if (fabs(a[dim * i + k]) > best) {
        best = fabs(a[dim * i + k]);
        piv = i;
      }
    if (piv != k)
      for (j = 0; j < dim; ++j) {
        double tmp = a[dim * k + j];
        a[dim * k + j] = a[dim * piv + j];
        a[dim * piv + j] = tmp;
        tmp = b[dim * k + j];
        b[dim * k + j] = b[dim * piv + j];
        b[dim * piv + j] = tmp;
      }
    for (i = k + 1; i < dim; ++i) {
      const double f = a[dim * i + k] / a[dim * k + k];
      for (j = k; j < dim; ++j) a[dim * i + j] -= f * a[dim * k + j];
      for (j = 0; j < dim; ++j) b[dim * i + j] -= f * b[dim * k + j];
    }
  }
  for (j = 0; j < dim; ++j)
    for (i = dim - 1; i >= 0; --i) {
      double sum = b[dim * i + j];
      for (k = i + 1; k < dim; ++k) sum -= a[dim * i + k] * inv[dim * k + j];
      inv[dim * i + j] = sum / a[dim * i + i];
    }
}

/* Eigen's 3x3 inverse is the cofactor formula (compute_inverse<Matrix3d>).  Call site
 * MDM/..._analytic_3dof.cc:77. */
static void inverse3_cofactor(const double* m, double* inv) {
  const double c00 = m[4] * m[8] - m[5] * m[7];
  const double c10 = m[5] * m[6] - m[3] * m[8];
  const double c20 = m[3] * m[7] - m[4] * m[6];
  const double det = (m[0] * c00 + m[1] * c10) + m[2] * c20;
  const double id = 1.0 / det;
  inv[0] = c00 * id;
  inv[1] = (m[2] * m[7] - m[1] * m[8]) * id;
  inv[2] = (m[1] * m[5] - m[2] * m[4]) * id;
  inv[3] = c10 * id;
  inv[4] = (m[0] * m[8] - m[2] * m[6]) * id;
  inv[5] = (m[2] * m[3] - m[0] * m[5]) * id;
  inv[6] = c20 * id;
  inv[7] = (m[1] * m[6] - m[0] * m[7]) * id;
  inv[8] = (m[0] * m[4] - m[1] * m[3]) * id;
}

/* Diagonally pivoted LDL^T solve — the algorithm behind Eigen's `H.ldlt().solve(b)`.
 * Call site MDM/..._analytic_simd.cc:85. */
static void solve_ldlt6(const double* Hin, const double* b, double* x) {
  double A[36], y[6];
  int perm[6];
  int i, j, k;
  for (i = 0; i < 36; ++i) A[i] = Hin[i];
  for (i = 0; i < 6; ++i) perm[i] = i;
  for (k = 0; k < 6; ++k) {
    int piv = k;
    double best = fabs(A[6 * k + k]);
    for (i = k + 1; i < 6; ++i)
      if (fabs(A[6 * i + i]) > best) {
        best = fabs(A[6 * i + i]);
        piv = i;
      }
    if (piv != k) {
      int tp;
      for (j = 0; j < 6; ++j) {
        double tmp = A[6 * k + j];
        A[6 * k + j] = A[6 * piv + j];
        A[6 * piv + j] = tmp;
      }
      for (j = 0; j < 6; ++j) {
        double tmp = A[6 * j + k];
        A[6 * j + k] = A[6 * j + piv];
        A[6 * j + piv] = tmp;
      }
      tp = perm[k];
      perm[k] = perm[piv];
      perm[piv] = tp;
    }
    /* column k of L below the diagonal, D(k) stays in A(k,k) */
    for (i = k + 1; i < 6; ++i) A[6 * i + k] /= A[6 * k + k];
    for (i = k + 1; i < 6; ++i)
      for (j = k + 1; j <= i; ++j) {
        A[6 * i + j] -= A[6 * i + k] * A[6 * k + k] * A[6 * j + k];
        A[6 * j + i] = A[6 * i + j];
      }
  }
  for (i = 0; i < 6; ++i) y[i] = b[perm[i]];
  for (i = 0; i < 6; ++i)
    for (j = 0; j < i; ++j) y[i] -= A[6 * i + j] * y[j];
  for (i = 0; i < 6; ++i) y[i] /= A[6 * i + i];
  for (i = 5; i >= 0; --i)
    for (j = i + 1; j < 6; ++j) y[i] -= A[6 * j + i] * y[j];
  for (i = 0; i < 6; ++i) x[perm[i]] = y[i];
}

/* ReflectHessian + multiplicative damping + solve: MDM/..._analytic.cc:122-129
 * (inverse) or MDM/..._analytic_simd.cc:78-85 (ldlt). */
void oracle_lm_step6(const double out28[28], double lambda, int linear_solver, double step[6]) {
  double H[36], mg[6];
  int row, col, k = 0;
  for (row = 0; row < 6; ++row)
    for (col = row; col < 6; ++col) {
      H[6 * row + col] = out28[k];
      H[6 * col + row] = out28[k];
      ++k;
    }
  for (row = 0; row < 6; ++row) H[6 * row + row] *= 1.0 + lambda;
  for (row = 0; row < 6; ++row) mg[row] = -out28[21 + row];
  if (linear_solver == 1) {
    solve_ldlt6(H, mg, step);
  } else {
    double inv[36];
    inverse_lu(6, H, inv);
    for (row = 0; row < 6; ++row) {
      double sum = 0.0;
      for (col = 0; col < 6; ++col) sum += inv[6 * row + col] * mg[col];
      step[row] = sum;
    }
  }
}

/* MDM/..._analytic_3dof.cc:70-77. */
void oracle_lm_step3(const double out10[10], double lambda, double step[3]) {
  double H[9], inv[9], mg[3];
  int row, col, k = 0;
  for (row = 0; row < 3; ++row)
    for (col = row; col < 3; ++col) {
      H[3 * row + col] = out10[k];
      H[3 * col + row] = out10[k];
      ++k;
    }
  for (row = 0; row < 3; ++row) H[3 * row + row] *= 1.0 + lambda;
  for (row = 0; row < 3; ++row) mg[row] = -out10[6 + row];
  inverse3_cofactor(H, inv);
  for (row = 0; row < 3; ++row)
    step[row] = (inv[3 * row] * mg[0] + inv[3 * row + 1] * mg[1]) + inv[3 * row + 2] * mg[2];
}

static double norm_n(const double* v, int n) {
  double s = 0.0;
  int i;
  for (i = 0; i < n; ++i) s += v[i] * v[i];
  return sqrt(s);
}

static double clampd(double v, double lo, double hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* ------------------------------------------------------------- LM loops ---- */

typedef void (*accum6_fn)(void* user, const double R[9], const double t[3], double out28[28]);

/* The loop shared by MDM/..._analytic.cc:81-157 and REM/..._analytic.cc:15-105. */
static void lm_loop6(accum6_fn accumulate, void* user, const oracle_options* opt, double t[3],
                     double R[9], oracle_report* rep) {
  const double min_lambda = 1e-6, max_lambda = 1e-2; /* constexpr in code, not Options */
  double q[4], lambda = 0.001, previous_cost = DBL_MAX, cost = 0.0;
  int iteration = 0;
  oracle_quat_from_matrix(R, q);
  for (; iteration < opt->max_iterations; ++iteration) {
    double out28[28], Rcur[9], step[6], dq[4];
    oracle_quat_to_matrix(q, Rcur);
    accumulate(user, Rcur, t, out28);
    cost = out28[27];
    oracle_lm_step6(out28, lambda, opt->linear_solver, step);
    t[0] += step[0];
    t[1] += step[1];
    t[2] += step[2];
    oracle_exp_quat(step + 3, dq);
    quat_mul_normalize(q, dq);
    if (norm_n(step, 6) < opt->parameter_tolerance) break;
    if (norm_n(out28 + 21, 6) < opt->gradient_tolerance) break;
    lambda *= (cost > previous_cost ? 2.0 : 0.6);
    lambda = clampd(lambda, min_lambda, max_lambda);
    previous_cost = cost;
  }
  oracle_quat_to_matrix(q, R);
  if (rep) {
    rep->iterations = iteration;
    rep->printed_cost = previous_cost;
    rep->last_cost = cost;
    rep->final_lambda = lambda;
  }
}

struct ndt_user {
  size_t n;
  const double* const* planes;
  const oracle_loss* loss;
};

static void ndt6_cb(void* user, const double R[9], const double t[3], double out28[28]) {
  const struct ndt_user* u = (const struct ndt_user*)user;
  oracle_ndt6_accumulate(u->n, u->planes, R, t, u->loss, out28);
}

void oracle_ndt6_solve(size_t n, const double* const planes[15], const oracle_options* opt,
                       const oracle_loss* loss, double t[3], double R[9], oracle_report* rep) {
  struct ndt_user u;
  u.n = n;
  u.planes = planes;
  u.loss = loss;
  lm_loop6(ndt6_cb, &u, opt, t, R, rep);
}

struct reproj_user {
  size_t n;
  const double* const* planes;
  const double* intr;
  const oracle_loss* loss;
  double min_depth;
};

static void reproj_cb(void* user, const double R[9], const double t[3], double out28[28]) {
  const struct reproj_user* u = (const struct reproj_user*)user;
  oracle_reproj_accumulate(u->n, u->planes, R, t, u->intr, u->loss, u->min_depth, out28);
}

void oracle_reproj_solve(size_t n, const double* const planes[5], const double intr[4],
                         const oracle_options* opt, const oracle_loss* loss, double min_depth,
                         double t[3], double R[9], oracle_report* rep) {
  struct reproj_user u;
  u.n = n;
  u.planes = planes;
  u.intr = intr;
  u.loss = loss;
  u.min_depth = min_depth;
  lm_loop6(reproj_cb, &u, opt, t, R, rep);
}

/* MDM/mahalanobis_distance_minimizer_analytic_3dof.cc:17-108.  The planar pose is the
 * top-left 2x2 of R and (t_x, t_y) (:23-25); `optimized_pose.rotate(dtheta)` is
 * linear <- linear * Rot2(dtheta) (:83); only x, y and the 2x2 block are written back
 * (:104-105). */
void oracle_ndt3_solve(size_t n, const double* const planes[15], const oracle_options* opt,
                       const oracle_loss* loss, double t[3], double R[9], oracle_report* rep) {
  const double min_lambda = 1e-6, max_lambda = 1e-2;
  double R2[4], t2[2], lambda = 0.001, previous_cost = DBL_MAX, cost = 0.0;
  int iteration = 0;
  R2[0] = R[0];
  R2[1] = R[1];
  R2[2] = R[3];
  R2[3] = R[4];
  t2[0] = t[0];
  t2[1] = t[1];
  for (; iteration < opt->max_iterations; ++iteration) {
    double out10[10], step[3], c, s, n00, n01, n10, n11;
    oracle_ndt3_accumulate(n, planes, R2, t2, loss, out10);
    cost = out10[9];
    oracle_lm_step3(out10, lambda, step);
    t2[0] += step[0];
    t2[1] += step[1];
    c = cos(step[2]);
    s = sin(step[2]);
    n00 = R2[0] * c + R2[1] * s;
    n01 = R2[0] * (-s) + R2[1] * c;
    n10 = R2[2] * c + R2[3] * s;
    n11 = R2[2] * (-s) + R2[3] * c;
    R2[0] = n00;
    R2[1] = n01;
    R2[2] = n10;
    R2[3] = n11;
    if (norm_n(step, 3) < opt->parameter_tolerance) break;
    if (norm_n(out10 + 6, 3) < opt->gradient_tolerance) break;
    lambda *= (cost > previous_cost ? 2.0 : 0.6);
    lambda = clampd(lambda, min_lambda, max_lambda);
    previous_cost = cost;
  }
  t[0] = t2[0];
  t[1] = t2[1];
  R[0] = R2[0];
  R[1] = R2[1];
  R[3] = R2[2];
  R[4] = R2[3];
  if (rep) {
    rep->iterations = iteration;
    rep->printed_cost = previous_cost;
    rep->last_cost = cost;
    rep->final_lambda = lambda;
  }
}
