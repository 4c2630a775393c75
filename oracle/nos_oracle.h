/*
 * nos_oracle.h — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Scalar fp64 CPU restatement of the reference's Gauss-Newton assembly path and of the
 * Levenberg-Marquardt loop around it.  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load this library; the product (libnos_hip.so and the
 * C++ host layer above it) never links or calls it.
 *
 * Pinning status (see DESIGN.md "Oracle"):
 *   - reprojection path: PINNED end-to-end by the reference's captured run
 *     results/reproj_amd64.txt:5,8,10 ("COST: 2.33228e-11, iter: 6", final pose equal to
 *     the true pose to the printed digits) — tests/test_oracle_golden.py.
 *   - 6-DoF / 3-DoF NDT paths: PINNED end-to-end by the reference's captured runs
 *     results/maha_amd64_simple.txt:10-13,24, results/maha_3_vs_6_amd64.txt:7-10,19-23,33,35 and
 *     results/maha_amd64.txt:4-7,55 — 17 `COST: ..., iter: ...` lines, 4 outer_iter counts and 4 final
 *     poses, every printed digit (tests/test_reference_ndt_runs.py).  That needed the harness's map
 *     bit for bit: oracle/scene_oracle.c restates UpdateNdtMap's accumulation and Eigen's
 *     SelfAdjointEigenSolver<Matrix3d> (incl. which multiply-adds the reference binary fuses), and the
 *     floor(N/4)*4 truncation of the captured revision is applied by the caller
 *     (oracle_scene.compact_correspondences).  The reference itself cannot be compiled here (Eigen,
 *     Ceres, FLANN, simd_helper absent), so there is no oracle/_ref.
 *
 * Plane order and output order are those of include/nos.h.
 */
#ifndef NOS_ORACLE_H_
#define NOS_ORACLE_H_

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct oracle_loss {
  int kind; /* 0 none, 1 exponential(c1=a, c2=b), 2 huber(threshold=a) */
  double a;
  double b;
} oracle_loss;

typedef struct oracle_options {
  int max_iterations;          /* NO/options.h:16 (default 40) */
  double gradient_tolerance;   /* NO/options.h:21 (1e-6) */
  double parameter_tolerance;  /* NO/options.h:22 (1e-6) */
  int linear_solver;           /* 0: H.inverse()*(-g) (scalar classes), 1: LDLT (SIMD 6-DoF class) */
} oracle_options;

typedef struct oracle_report {
  int iterations;        /* loop index at exit — the "iter:" the reference prints */
  double printed_cost;   /* previous_cost at exit — the "COST:" the reference prints */
  double last_cost;      /* cost of the last evaluated iteration */
  double final_lambda;
} oracle_report;

/* NO/loss_function.h:28-33 (exponential), :57-66 (huber); kind 0 = nullptr branch. */
void oracle_loss_evaluate(const oracle_loss* loss, double s, double* rho, double* w);

/* MDM/mahalanobis_distance_minimizer_analytic.cc:12-52 + :159-218, index order. */
void oracle_ndt6_accumulate(size_t n, const double* const planes[15], const double R[9],
                            const double t[3], const oracle_loss* loss, double out28[28]);
/* Per-correspondence pieces for fine-grained tests (one item of the loop above). */
void oracle_ndt6_item(const double x[15], const double R[9], const double t[3], double r[3],
                      double J[18] /* row-major 3x6 */);

/* MDM/mahalanobis_distance_minimizer_analytic_3dof.cc:36-69 + :110-139 (all N items). */
void oracle_ndt3_accumulate(size_t n, const double* const planes[15], const double R2[4],
                            const double t2[2], const oracle_loss* loss, double out10[10]);

/* REM/reprojection_error_minimizer_analytic.cc:31-64 + :107-162.
 * intr = {inv_fx, inv_fy, cx, cy}. */
void oracle_reproj_accumulate(size_t n, const double* const planes[5], const double R[9],
                              const double t[3], const double intr[4], const oracle_loss* loss,
                              double min_depth, double out28[28]);

/* fp32-lane restatement of the SIMD classes' arithmetic (lane width 8, per-lane fp32
 * accumulators summed into double at the end; MDM/..._analytic_simd.cc:113-177).
 * Processes floor(n/8)*8 items like the reference when drop_tail != 0, else all n. */
void oracle_ndt6_accumulate_f32lanes(size_t n, const double* const planes[15], const double R[9],
                                     const double t[3], const oracle_loss* loss, int drop_tail,
                                     double out28[28]);

/* LM loops.  pose is in/out: t[3] then row-major R[9].
 * 6-DoF: MDM/..._analytic.cc:81-157 ; 3-DoF: MDM/..._analytic_3dof.cc:17-108 ;
 * reprojection: REM/..._analytic.cc:15-105. */
void oracle_ndt6_solve(size_t n, const double* const planes[15], const oracle_options* opt,
                       const oracle_loss* loss, double t[3], double R[9], oracle_report* rep);
void oracle_ndt3_solve(size_t n, const double* const planes[15], const oracle_options* opt,
                       const oracle_loss* loss, double t[3], double R[9], oracle_report* rep);
void oracle_reproj_solve(size_t n, const double* const planes[5], const double intr[4],
                         const oracle_options* opt, const oracle_loss* loss, double min_depth,
                         double t[3], double R[9], oracle_report* rep);

/* One host LM step given the accumulated 28 (or 10) numbers — exposed so tests can
 * compare the product's host step with the oracle's on identical inputs. */
void oracle_lm_step6(const double out28[28], double lambda, int linear_solver, double step[6]);
void oracle_lm_step3(const double out10[10], double lambda, double step[3]);

/* Small Eigen restatements used by the loops (exposed for unit tests). */
void oracle_quat_from_matrix(const double R[9], double q_wxyz[4]); /* Eigen Quaternion(Matrix3) */
void oracle_quat_to_matrix(const double q_wxyz[4], double R[9]);  /* toRotationMatrix()        */
void oracle_exp_quat(const double w[3], double q_wxyz[4]);        /* MDM/..._minimizer.cc:20-33 */

#ifdef __cplusplus
}
#endif

#endif /* NOS_ORACLE_H_ */
