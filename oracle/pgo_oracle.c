/* pgo_oracle.c — CPU restatement of the pose-graph linearisation in plain C.  TEST INFRASTRUCTURE: only tests/ and
 * bench.py's cpu_baseline leg load it; the product never does.
 *
 * Residual: nonlinear_optimizer/pose_graph_optimizer/ceres_cost_functor.h:17-53 (plain) and :55-98 (switchable: residual
 * scaled by s, seventh residual (1 - s) * 1e-9).  The reference evaluates it through Ceres autodiff only (its analytic
 * Solve is an empty loop, pose_graph_optimizer_analytic.cc:21-42) and holds no captured PGO run: parity UNPINNED against
 * reference outputs.  This file follows oracle/oracle_pgo.py statement by statement (edge_residual, edge_jacobians,
 * Graph.linearize) and is pinned to it by tests/test_oracle_golden.py; it exists so that the CPU baseline of the bench's
 * pose-graph rows is compiled code, not an interpreter.
 *
 * One pass over the constraints in index order; per constraint the 6-vector residual, the two 6x6 Jacobians under
 * p <- p + dp, q <- q (x) Exp(dw), and their contribution to the diagonal blocks (21 upper entries, row-major), the
 * gradient (6 per pose, 1 per free switch), the switch curvature and the cost. */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <string.h>

static void qmul(const double* a, const double* b, double* o) {
  o[0] = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
  o[1] = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
  o[2] = a[0] * b[2] + a[2] * b[0] + a[3] * b[1] - a[1] * b[3];
  o[3] = a[0] * b[3] + a[3] * b[0] + a[1] * b[2] - a[2] * b[1];
}

static void qrot(const double* q, double R[9]) {
  const double w = q[0], x = q[1], y = q[2], z = q[3];
  R[0] = 1 - 2 * (y * y + z * z), R[1] = 2 * (x * y - w * z), R[2] = 2 * (x * z + w * y);
  R[3] = 2 * (x * y + w * z), R[4] = 1 - 2 * (x * x + z * z), R[5] = 2 * (y * z - w * x);
  R[6] = 2 * (x * z - w * y), R[7] = 2 * (y * z + w * x), R[8] = 1 - 2 * (x * x + y * y);
}

static void hat(const double* v, double H[9]) {
  H[0] = 0, H[1] = -v[2], H[2] = v[1];
  H[3] = v[2], H[4] = 0, H[5] = -v[0];
  H[6] = -v[1], H[7] = v[0], H[8] = 0;
}

static void mat3_mul(const double* A, const double* B, double* C) { /* C = A B */
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) C[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
}

#define C_SWITCH 1e-9 /* ceres_cost_functor.h:93 */

/* poses [n][7] = p (3), q wxyz (4); meas [m][7] likewise; sw [m]; sw_free, fixed: bytes (may be NULL = none).
 * hdiag [n][21], grad [n][6], hs [m], gs [m] (switch curvature / gradient; 1 / 0 for switches that are not free),
 * cost: sum of squares.  Fixed poses get identity blocks and zero gradient, as Graph.linearize gives them. */
int oracle_pgo_linearize(size_t n, const double* poses, size_t m, const int32_t* ref, const int32_t* qry, const double* meas,
                         const double* sw, const unsigned char* sw_free, const unsigned char* fixed, double* hdiag,
                         double* grad, double* hs, double* gs, double* cost_out) {
  memset(hdiag, 0, n * 21 * sizeof(double));
  memset(grad, 0, n * 6 * sizeof(double));
  double cost = 0.0;
  for (size_t e = 0; e < m; ++e) {
    const int ir = ref[e], iq = qry[e];
    if (ir < 0 || iq < 0 || (size_t)ir >= n || (size_t)iq >= n) return 1;
    const double *pr = poses + 7 * (size_t)ir, *qr = pr + 3, *pq = poses + 7 * (size_t)iq, *qq = pq + 3;
    const double *tm = meas + 7 * e, *qm = tm + 3;
    /* edge_residual */
    double Rr[9], r[6], eq[4], t1[4];
    qrot(qr, Rr);
    for (int i = 0; i < 3; ++i) r[i] = (pq[i] - pr[i]) - (Rr[3 * i] * tm[0] + Rr[3 * i + 1] * tm[1] + Rr[3 * i + 2] * tm[2]);
    const double qqc[4] = {qq[0], -qq[1], -qq[2], -qq[3]};
    qmul(qqc, qr, t1);
    qmul(t1, qm, eq);
    for (int i = 0; i < 3; ++i) r[3 + i] = 2.0 * eq[1 + i];
    /* edge_jacobians */
    double Jr[36], Jq[36], Hm[9], He[9], Rm[9], Ep[9], Em[9], A[9], Bm[9];
    memset(Jr, 0, sizeof Jr);
    memset(Jq, 0, sizeof Jq);
    hat(tm, Hm);
    hat(eq + 1, He);
    qrot(qm, Rm);
    for (int i = 0; i < 9; ++i) Ep[i] = He[i], Em[i] = He[i];
    for (int i = 0; i < 3; ++i) Ep[4 * i] += eq[0], Em[4 * i] -= eq[0];
    mat3_mul(Rr, Hm, A); /* R(q_r) [t_m]x */
    double RmT[9];
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) RmT[3 * i + j] = Rm[3 * j + i];
    mat3_mul(Ep, RmT, Bm); /* E+ R(q_m)^T */
    for (int i = 0; i < 3; ++i) {
      Jr[6 * i + i] = -1.0;
      Jq[6 * i + i] = 1.0;
      for (int j = 0; j < 3; ++j) {
        Jr[6 * i + 3 + j] = A[3 * i + j];
        Jr[6 * (3 + i) + 3 + j] = Bm[3 * i + j];
        Jq[6 * (3 + i) + 3 + j] = Em[3 * i + j];
      }
    }
    const double s = sw[e];
    const int free_sw = sw_free != NULL && sw_free[e] != 0;
    double rr = 0.0;
    for (int i = 0; i < 6; ++i) rr += r[i] * r[i];
    cost += s * s * rr;
    if (free_sw) cost += (C_SWITCH * (1.0 - s)) * (C_SWITCH * (1.0 - s));
    /* J^T J diagonal blocks and J^T f for both ends (f = s r; the Jacobians carry the factor s as well) */
    const double* J2[2] = {Jr, Jq};
    const int end[2] = {ir, iq};
    for (int k = 0; k < 2; ++k) {
      const size_t i = (size_t)end[k];
      if (fixed != NULL && fixed[i]) continue;
      const double* J = J2[k];
      double* H = hdiag + 21 * i;
      double* g = grad + 6 * i;
      int u = 0;
      for (int a = 0; a < 6; ++a) {
        double ga = 0.0;
        for (int c = 0; c < 6; ++c) ga += J[6 * c + a] * r[c];
        g[a] += s * s * ga;
        for (int b = a; b < 6; ++b) {
          double hab = 0.0;
          for (int c = 0; c < 6; ++c) hab += J[6 * c + a] * J[6 * c + b];
          H[u++] += s * s * hab;
        }
      }
    }
    if (hs != NULL) hs[e] = free_sw ? rr + C_SWITCH * C_SWITCH : 1.0;
    if (gs != NULL) gs[e] = free_sw ? s * rr - C_SWITCH * C_SWITCH * (1.0 - s) : 0.0;
  }
  if (fixed != NULL)
    for (size_t i = 0; i < n; ++i)
      if (fixed[i]) {
        const int dg[6] = {0, 6, 11, 15, 18, 20};
        for (int k = 0; k < 6; ++k) hdiag[21 * i + dg[k]] = 1.0;
      }
  *cost_out = cost;
  return 0;
}
