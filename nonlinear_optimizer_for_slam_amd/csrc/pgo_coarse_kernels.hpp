// pgo_coarse_kernels.hpp — coarse level of the two-level preconditioner of the pose-graph PCG (configs[4]).
//
// Why: a SLAM trajectory is a long chain with local extra constraints; block-Jacobi PCG on its normal equations needs a
// number of iterations that grows with the length of the chain (the slow modes are "bendings" of the whole trajectory): at
// 1 M poses the round-1 solver ran into its 300-iteration cap from the third LM iteration on.  The coarse space below makes
// the iteration count independent of the length (measured on the CPU prototype: 2 080 → 66 iterations at 6 k and at 24 k
// poses, λ = 1e-6; tools/measure_pgo.py for the GPU).
//
// Coarse space: the poses are cut into aggregates of `agg` consecutive indices (trajectory order).  Aggregate I carries six
// unknowns (a, θ): a translation and a WORLD-frame rotation about its first pose's position c_I — a rigid motion of the
// whole aggregate.  Under the solvers' update rule (p ← p + δp, q ← q ⊗ Exp(δω), δω in the pose's own frame) pose i of the
// aggregate moves by
//        δp_i = a + θ x (p_i − c_I),   δω_i = R_iᵀ θ           i.e.  δx_i = B_i (a, θ),  B_i = [ I  −[p_i − c_I]x ; 0  R_iᵀ ]
// (fixed poses: B_i = 0).  Prolongation P stacks the B_i; the preconditioner is additive,
//        M⁻¹ r = blockdiag(H_ii)⁻¹ r  +  P A_c⁻¹ Pᵀ r,        A_c = Pᵀ H' P,
// both terms symmetric positive definite.  H' is the damped normal matrix with the coupling of constraints that span more
// than neighbouring aggregates dropped (their diagonal blocks stay): H' is then block tridiagonal over aggregates whatever
// loop closures the graph has, and within a factor two of H on those constraints.
//
// A_c is never assembled from blocks — the library stores nothing of H but its diagonal: it is PROBED.  Aggregates are
// three-coloured (I mod 3); for each colour and each of the six coarse unknowns one masked product y = H' P e is formed with
// the matrix-free sweep of pgo_kernels.hpp and restricted (Pᵀ y): 18 products per LM iteration give all three block
// diagonals.  The block-tridiagonal system is solved by parallel cyclic reduction (PCR): ⌈log2 n_c⌉ levels, one thread per
// aggregate per level, the elimination factors of every level are kept so that applying A_c⁻¹ to a right-hand side costs
// one small kernel per level.  Everything runs in a fixed order: results are bit-reproducible.
#pragma once

#include "pgo_kernels.hpp"

namespace nos {

// ---- 6x6 helpers (row-major, in registers)

__device__ __forceinline__ void m6_load(const double* __restrict__ p, double (&A)[36]) {
#pragma unroll
  for (int k = 0; k < 36; ++k) A[k] = p[k];
}
__device__ __forceinline__ void m6_store(double* __restrict__ p, const double (&A)[36]) {
#pragma unroll
  for (int k = 0; k < 36; ++k) p[k] = A[k];
}
// C = A B
__device__ __forceinline__ void m6_mul(const double (&A)[36], const double (&B)[36], double (&C)[36]) {
#pragma unroll
  for (int r = 0; r < 6; ++r)
#pragma unroll
    for (int c = 0; c < 6; ++c) {
      double v = 0.0;
#pragma unroll
      for (int m = 0; m < 6; ++m) v = fma(A[6 * r + m], B[6 * m + c], v);
      C[6 * r + c] = v;
    }
}
// y += A x
__device__ __forceinline__ void m6_mulvec_add(const double (&A)[36], const double (&x)[6], double (&y)[6]) {
#pragma unroll
  for (int r = 0; r < 6; ++r) {
    double v = y[r];
#pragma unroll
    for (int m = 0; m < 6; ++m) v = fma(A[6 * r + m], x[m], v);
    y[r] = v;
  }
}
// Inverse of a symmetric positive definite 6x6 (Cholesky; pivots floored so that an empty aggregate cannot poison the
// recurrence).  Only the upper triangle of A is read.
__device__ __forceinline__ void m6_spd_inverse(const double (&A)[36], double (&Ai)[36]) {
  double L[6][6], Li[6][6];
#pragma unroll
  for (int r = 0; r < 6; ++r)
#pragma unroll
    for (int c = 0; c < 6; ++c) {
      L[r][c] = 0.0;
      Li[r][c] = 0.0;
    }
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    double d = A[6 * j + j];
#pragma unroll
    for (int m = 0; m < 6; ++m)
      if (m < j) d -= L[j][m] * L[j][m];
    d = d > 1e-300 ? d : 1e-300;
    const double lj = sqrt(d);
    L[j][j] = lj;
#pragma unroll
    for (int r = 0; r < 6; ++r)
      if (r > j) {
        double v = A[6 * j + r];  // upper triangle: A(j, r) = A(r, j)
#pragma unroll
        for (int m = 0; m < 6; ++m)
          if (m < j) v -= L[r][m] * L[j][m];
        L[r][j] = v / lj;
      }
  }
#pragma unroll
  for (int c = 0; c < 6; ++c) {
    Li[c][c] = 1.0 / L[c][c];
#pragma unroll
    for (int r = 0; r < 6; ++r)
      if (r > c) {
        double v = 0.0;
#pragma unroll
        for (int m = 0; m < 6; ++m)
          if (m >= c && m < r) v -= L[r][m] * Li[m][c];
        Li[r][c] = v / L[r][r];
      }
  }
#pragma unroll
  for (int r = 0; r < 6; ++r)
#pragma unroll
    for (int c = 0; c < 6; ++c) {
      double v = 0.0;
#pragma unroll
      for (int m = 0; m < 6; ++m)
        if (m >= r && m >= c) v += Li[m][r] * Li[m][c];
      Ai[6 * r + c] = v;
    }
}

// B_i of the header comment for pose i of aggregate I (c = position of the aggregate's first pose).
__device__ __forceinline__ void coarse_basis(const PgoView& G, uint32_t i, uint32_t agg, double (&B)[36]) {
  double pi[8], pc[8];
  load_record(G.pose, i, pi);
  load_record(G.pose, size_t(i / agg) * agg, pc);
  double R[9];
  qrot_matrix(Quat4{pi[3], pi[4], pi[5], pi[6]}, R);
  const double d[3] = {pi[0] - pc[0], pi[1] - pc[1], pi[2] - pc[2]};
#pragma unroll
  for (int k = 0; k < 36; ++k) B[k] = 0.0;
  B[0] = B[7] = B[14] = 1.0;
  // −[d]x = [0 dz −dy; −dz 0 dx; dy −dx 0]
  B[4] = d[2];
  B[5] = -d[1];
  B[9] = -d[2];
  B[11] = d[0];
  B[15] = d[1];
  B[16] = -d[0];
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) B[6 * (3 + r) + 3 + c] = R[3 * c + r];  // R_iᵀ
}

// Probe vector: x_i = column `dof` of B_i for the free poses of the aggregates with I mod 3 == colour, 0 elsewhere.
__global__ __launch_bounds__(256) void pgo_coarse_probe_kernel(PgoView G, uint32_t agg, int colour, int dof,
                                                               double* __restrict__ x) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i >= G.n_poses) return;
  double out[6] = {0, 0, 0, 0, 0, 0};
  if (!G.fixed[i] && int((i / agg) % 3u) == colour) {
    double B[36];
    coarse_basis(G, i, agg, B);
#pragma unroll
    for (int r = 0; r < 6; ++r) out[r] = B[6 * r + dof];
  }
#pragma unroll
  for (int r = 0; r < 6; ++r) x[size_t(6) * i + r] = out[r];
}

// Restriction rc_J = Σ_{i in J} B_iᵀ y_i: one WAVE per aggregate (lane l takes poses lo + l, lo + l + 64, …), the six sums
// by the fixed butterfly of wave_sum — deterministic.  Launch with 256-thread blocks, ceil(n_agg / 4) of them.
__global__ __launch_bounds__(256) void pgo_coarse_restrict_kernel(PgoView G, uint32_t agg, uint32_t n_agg,
                                                                  const double* __restrict__ y, double* __restrict__ rc) {
  const uint32_t J = blockIdx.x * 4 + (threadIdx.x >> 6);
  const uint32_t lane = threadIdx.x & 63u;
  if (J >= n_agg) return;  // wave-uniform
  double acc[6] = {0, 0, 0, 0, 0, 0};
  const uint32_t lo = J * agg, hi = (lo + agg < G.n_poses) ? lo + agg : G.n_poses;
  for (uint32_t i = lo + lane; i < hi; i += 64u) {
    if (G.fixed[i]) continue;
    double B[36], yi[6];
    coarse_basis(G, i, agg, B);
#pragma unroll
    for (int r = 0; r < 6; ++r) yi[r] = y[size_t(6) * i + r];
#pragma unroll
    for (int c = 0; c < 6; ++c) {
      double v = acc[c];
#pragma unroll
      for (int r = 0; r < 6; ++r) v = fma(B[6 * r + c], yi[r], v);
      acc[c] = v;
    }
  }
#pragma unroll
  for (int c = 0; c < 6; ++c) {
    const double tot = wave_sum(acc[c]);
    if (lane == 0) rc[size_t(6) * J + c] = tot;
  }
}

// The vector update of a PCG iteration and the restriction of the new residual in ONE pass (device-resident CG scalars):
//     x += alpha p,  r -= alpha q            (alpha = scalars[5], written by the product's tail)
//     rc_J = Σ_{i in J} B_iᵀ r_i
// Same lane ↔ pose assignment, same order of additions as pgo_cg_update_dev_kernel followed by pgo_coarse_restrict_kernel
// (one wave per aggregate): identical bits, one sweep over r instead of two and one launch instead of two.  Workgroups
// [0, ceil(n_agg / 4)) take the pose rows; the ones behind them the switch rows [6 n_poses, n) element-wise.
__global__ __launch_bounds__(256) void pgo_update_restrict_kernel(PgoView G, uint32_t agg, uint32_t n_agg, size_t n,
                                                                  const double* __restrict__ scalars,
                                                                  const double* __restrict__ p, const double* __restrict__ q,
                                                                  double* __restrict__ x, double* __restrict__ r,
                                                                  double* __restrict__ rc) {
  const double alpha = scalars[5];
  const uint32_t agg_blocks = (n_agg + 3u) / 4u;
  if (blockIdx.x >= agg_blocks) {
    const size_t i = size_t(6) * G.n_poses + size_t(blockIdx.x - agg_blocks) * 256 + threadIdx.x;
    if (i < n) {
      x[i] += alpha * p[i];
      r[i] -= alpha * q[i];
    }
    return;
  }
  const uint32_t J = blockIdx.x * 4 + (threadIdx.x >> 6);
  const uint32_t lane = threadIdx.x & 63u;
  if (J >= n_agg) return;  // wave-uniform
  double acc[6] = {0, 0, 0, 0, 0, 0};
  const uint32_t lo = J * agg, hi = (lo + agg < G.n_poses) ? lo + agg : G.n_poses;
  using V2 = double __attribute__((ext_vector_type(2)));
  for (uint32_t i = lo + lane; i < hi; i += 64u) {
    // a pose's six entries as three 16-byte pieces (48 i bytes is 16-byte aligned): half the memory instructions of six
    // 8-byte accesses at a 48-byte stride, which the texture path serves at the same cycles per instruction
    double yi[6];
    const V2 *p2 = reinterpret_cast<const V2*>(p) + size_t(3) * i, *q2 = reinterpret_cast<const V2*>(q) + size_t(3) * i;
    V2 *x2 = reinterpret_cast<V2*>(x) + size_t(3) * i, *r2 = reinterpret_cast<V2*>(r) + size_t(3) * i;
    V2 pv[3], qv[3], xv[3], rv[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      pv[k] = p2[k];
      qv[k] = q2[k];
      xv[k] = x2[k];
      rv[k] = r2[k];
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      V2 xn, rn;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        xn[h] = xv[k][h] + alpha * pv[k][h];
        rn[h] = rv[k][h] - alpha * qv[k][h];
        yi[2 * k + h] = rn[h];
      }
      x2[k] = xn;
      r2[k] = rn;
    }
    if (G.fixed[i]) continue;
    double B[36];
    coarse_basis(G, i, agg, B);
#pragma unroll
    for (int c = 0; c < 6; ++c) {
      double v = acc[c];
#pragma unroll
      for (int rr = 0; rr < 6; ++rr) v = fma(B[6 * rr + c], yi[rr], v);
      acc[c] = v;
    }
  }
#pragma unroll
  for (int c = 0; c < 6; ++c) {
    const double tot = wave_sum(acc[c]);
    if (lane == 0) rc[size_t(6) * J + c] = tot;
  }
}

// One probing product gives column `dof` of the blocks A_c(J, I) with I the aggregate of colour `colour` among J−1, J, J+1.
// L / D / U: [n_agg][36] row-major blocks A_c(J, J−1), A_c(J, J), A_c(J, J+1).
__global__ __launch_bounds__(128) void pgo_coarse_scatter_kernel(uint32_t n_agg, int colour, int dof,
                                                                 const double* __restrict__ rc, double* __restrict__ L,
                                                                 double* __restrict__ D, double* __restrict__ U) {
  const uint32_t J = blockIdx.x * 128 + threadIdx.x;
  if (J >= n_agg) return;
  const int cj = int(J % 3u);
  double* dst = cj == colour ? D : ((cj + 1) % 3 == colour ? U : L);  // colour of J+1 is cj+1, of J−1 is cj+2 (mod 3)
  if ((dst == U && J + 1 >= n_agg) || (dst == L && J == 0)) return;
#pragma unroll
  for (int r = 0; r < 6; ++r) dst[size_t(36) * J + 6 * r + dof] = rc[size_t(6) * J + r];
}

// ---- the coarse operator assembled directly (round 4; replaces the 18 probing products per solve)
//
// A_c(I, J) = Σ_{i in I} Σ_{j in J} B_iᵀ H'_ij B_j with H'_ii = Σ_e s² J_iᵀ J_i + λ diag(H_ii), H'_ij = Σ_e s² J_iᵀ J_j for the
// constraints e = (i, j) whose ends lie at most one aggregate apart and are both free.  With B = [I X; 0 Rᵀ] (X = −[p − c]x)
// and the constraint Jacobians J_ref = [−I A; 0 B_m], J_qry = [I 0; 0 C] the product W = J B is again [σ I, P; 0, Q]:
//     reference end: σ = −1, P = −X + A Rᵀ, Q = B_m Rᵀ          query end: σ = +1, P = X, Q = C Rᵀ
// and W_iᵀ W_j = [σ_i σ_j I, σ_i P_j; σ_j P_iᵀ, P_iᵀ P_j + Q_iᵀ Q_j] — one scalar and three 3x3 blocks (28 numbers).
// Owner computes, like every other sweep: one wave per aggregate, lane ↔ pose, each lane walks its pose's constraints in
// adjacency order, the 28 sums go through the fixed butterfly.  WHICH = 0: A_c(I, I); 1: A_c(I, I−1); 2: A_c(I, I+1) —
// three launches of one sweep each instead of 18 probing products (each a full matrix-free product plus three small kernels).
struct CoarseW {
  double sigma, P[9], Q[9];
};

__device__ __forceinline__ void coarse_w(const EdgeTerms& T, int role, const double (&d)[3], const double (&R)[9], CoarseW& W) {
  // X = −[d]x = [0 dz −dy; −dz 0 dx; dy −dx 0];  Rt = Rᵀ
  const double X[9] = {0.0, d[2], -d[1], -d[2], 0.0, d[0], d[1], -d[0], 0.0};
  if (role == 0) {
    W.sigma = -1.0;
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        // (A Rᵀ)(r, c) = Σ_k A(r, k) R(c, k)
        W.P[3 * r + c] = -X[3 * r + c] + (T.A[3 * r] * R[3 * c] + T.A[3 * r + 1] * R[3 * c + 1] + T.A[3 * r + 2] * R[3 * c + 2]);
        W.Q[3 * r + c] = T.B[3 * r] * R[3 * c] + T.B[3 * r + 1] * R[3 * c + 1] + T.B[3 * r + 2] * R[3 * c + 2];
      }
  } else {
    W.sigma = 1.0;
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        W.P[3 * r + c] = X[3 * r + c];
        W.Q[3 * r + c] = T.C[3 * r] * R[3 * c] + T.C[3 * r + 1] * R[3 * c + 1] + T.C[3 * r + 2] * R[3 * c + 2];
      }
  }
}

// acc (28) += w · Wiᵀ Wj:  [0] the scalar of the top-left block, [1..9] top-right, [10..18] bottom-left, [19..27] bottom-right
__device__ __forceinline__ void coarse_accumulate(double w, const CoarseW& Wi, const CoarseW& Wj, double (&acc)[28]) {
  acc[0] += w * (Wi.sigma * Wj.sigma);
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      acc[1 + 3 * r + c] += w * (Wi.sigma * Wj.P[3 * r + c]);
      acc[10 + 3 * r + c] += w * (Wj.sigma * Wi.P[3 * c + r]);
      double v = 0.0;
#pragma unroll
      for (int k = 0; k < 3; ++k) v += Wi.P[3 * k + r] * Wj.P[3 * k + c] + Wi.Q[3 * k + r] * Wj.Q[3 * k + c];
      acc[19 + 3 * r + c] += w * v;
    }
}

template <int WHICH>
__global__ __launch_bounds__(256) void pgo_coarse_assemble_kernel(PgoView G, const double* __restrict__ hdiag, double lambda,
                                                                  uint32_t agg, uint32_t n_agg, double* __restrict__ out) {
  const uint32_t I = blockIdx.x * 4 + (threadIdx.x >> 6);
  const uint32_t lane = threadIdx.x & 63u;
  if (I >= n_agg) return;  // wave-uniform
  if ((WHICH == 1 && I == 0) || (WHICH == 2 && I + 1 >= n_agg)) {  // no such neighbour: the block is zero
    if (lane < 36) out[size_t(36) * I + lane] = 0.0;
    return;
  }
  double acc[28];
#pragma unroll
  for (int k = 0; k < 28; ++k) acc[k] = 0.0;
  const uint32_t lo = I * agg, hi = (lo + agg < G.n_poses) ? lo + agg : G.n_poses;
  const uint32_t want = WHICH == 0 ? I : (WHICH == 1 ? I - 1 : I + 1);  // aggregate of the neighbour whose coupling is summed
  double pc[8];
  load_record(G.pose, lo, pc);
  for (uint32_t i = lo + lane; i < hi; i += 64u) {
    if (G.fixed[i]) continue;  // B_i = 0
    double pi[8], Ri[9];
    load_record(G.pose, i, pi);
    qrot_matrix(Quat4{pi[3], pi[4], pi[5], pi[6]}, Ri);
    const double di[3] = {pi[0] - pc[0], pi[1] - pc[1], pi[2] - pc[2]};
    if (WHICH == 0) {  // damping: B_iᵀ λ diag(H_ii) B_i = [Dt, Dt X; Xᵀ Dt, Xᵀ Dt X + R Dr Rᵀ]
      const int dg[6] = {0, 6, 11, 15, 18, 20};
      double h[6];
#pragma unroll
      for (int k = 0; k < 6; ++k) h[k] = lambda * hdiag[size_t(dg[k]) * G.n_poses + i];
      const double X[9] = {0.0, di[2], -di[1], -di[2], 0.0, di[0], di[1], -di[0], 0.0};
      // the top-left block is diag(h_t), not a multiple of I: it is kept in the three diagonal slots of a 3x3 by folding
      // it into the accumulators after the wave sum is impossible — so the damping goes through its own small block below
      // (dt[3]) and is added at the store
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          acc[1 + 3 * r + c] += h[r] * X[3 * r + c];
          acc[10 + 3 * r + c] += X[3 * c + r] * h[c];
          double v = 0.0;
#pragma unroll
          for (int k = 0; k < 3; ++k) v += X[3 * k + r] * h[k] * X[3 * k + c] + Ri[3 * r + k] * h[3 + k] * Ri[3 * c + k];
          acc[19 + 3 * r + c] += v;
        }
    }
    for (uint32_t a = G.adj_off[i]; a < G.adj_off[i + 1]; ++a) {
      const uint32_t e = G.adj[a] >> 1;
      const int role = int(G.adj[a] & 1u);
      const uint32_t j = G.adj_nbr[a];
      const uint32_t aj = j / agg;
      const bool cross = !G.fixed[j] && aj == want;  // WHICH = 0: the neighbour lies in the same aggregate
      if (WHICH != 0 && !cross) continue;
      EdgeTerms T;
      const double s = edge_terms(G, e, role == 0 ? i : j, role == 0 ? j : i, T);
      CoarseW Wi;
      coarse_w(T, role, di, Ri, Wi);
      if (WHICH == 0) coarse_accumulate(s * s, Wi, Wi, acc);
      if (cross) {
        double pj[8], pcj[8], Rj[9];
        load_record(G.pose, j, pj);
        load_record(G.pose, size_t(aj) * agg, pcj);
        qrot_matrix(Quat4{pj[3], pj[4], pj[5], pj[6]}, Rj);
        const double dj[3] = {pj[0] - pcj[0], pj[1] - pcj[1], pj[2] - pcj[2]};
        CoarseW Wj;
        coarse_w(T, 1 - role, dj, Rj, Wj);
        coarse_accumulate(s * s, Wi, Wj, acc);
      }
    }
  }
  // the damping's top-left block diag(lambda h_t) is not a multiple of the identity: its three numbers travel separately
  double dt[3] = {0.0, 0.0, 0.0};
  if (WHICH == 0) {
    for (uint32_t i = lo + lane; i < hi; i += 64u) {
      if (G.fixed[i]) continue;
      dt[0] += lambda * hdiag[size_t(0) * G.n_poses + i];
      dt[1] += lambda * hdiag[size_t(6) * G.n_poses + i];
      dt[2] += lambda * hdiag[size_t(11) * G.n_poses + i];
    }
  }
#pragma unroll
  for (int k = 0; k < 28; ++k) acc[k] = wave_sum(acc[k]);
#pragma unroll
  for (int k = 0; k < 3; ++k) dt[k] = wave_sum(dt[k]);
  if (lane != 0) return;
  double M[36];
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      M[6 * r + c] = r == c ? acc[0] + dt[r] : 0.0;
      M[6 * r + 3 + c] = acc[1 + 3 * r + c];
      M[6 * (3 + r) + c] = acc[10 + 3 * r + c];
      M[6 * (3 + r) + 3 + c] = acc[19 + 3 * r + c];
    }
  m6_store(out + size_t(36) * I, M);
}

// Make the probed operator exactly symmetric (D ← (D + Dᵀ)/2, L_J ← (L_J + U_{J−1}ᵀ)/2, U_{J−1} ← L_Jᵀ) and give
// aggregates without a free pose an identity block.
__global__ __launch_bounds__(128) void pgo_coarse_symmetrize_kernel(uint32_t n_agg, double* __restrict__ L,
                                                                    double* __restrict__ D, double* __restrict__ U) {
  const uint32_t J = blockIdx.x * 128 + threadIdx.x;
  if (J >= n_agg) return;
  double d[36];
  m6_load(D + size_t(36) * J, d);
  bool empty = true;
#pragma unroll
  for (int r = 0; r < 6; ++r)
#pragma unroll
    for (int c = r; c < 6; ++c) {
      const double v = 0.5 * (d[6 * r + c] + d[6 * c + r]);
      d[6 * r + c] = v;
      d[6 * c + r] = v;
      empty = empty && v == 0.0;
    }
  if (empty) {
#pragma unroll
    for (int r = 0; r < 6; ++r) d[7 * r] = 1.0;
  }
  m6_store(D + size_t(36) * J, d);
  if (J > 0) {
    double l[36], u[36];
    m6_load(L + size_t(36) * J, l);
    m6_load(U + size_t(36) * (J - 1), u);
#pragma unroll
    for (int r = 0; r < 6; ++r)
#pragma unroll
      for (int c = 0; c < 6; ++c) {
        const double v = 0.5 * (l[6 * r + c] + u[6 * c + r]);
        L[size_t(36) * J + 6 * r + c] = v;
        U[size_t(36) * (J - 1) + 6 * c + r] = v;
      }
  } else {
#pragma unroll
    for (int k = 0; k < 36; ++k) L[k] = 0.0;
  }
  if (J + 1 == n_agg) {
#pragma unroll
    for (int k = 0; k < 36; ++k) U[size_t(36) * J + k] = 0.0;
  }
}

// Storage of the per-level elimination factors alpha / gamma (36 doubles per row and level each).  They are only ever read
// by pgo_pcr_span_kernel, whose lanes walk the rows of ONE residue class J ≡ c (mod 2^shift) in order, so the factors of a
// level are kept as 36 planes over a class-major row index: plane k of level l at (l·36 + k)·pitch + slot(J),
//     slot(J) = (J mod 2^shift)·rows + (J >> shift),    rows = ceil(n_agg / 2^shift),   pitch >= 2^shift · rows
// (shift = first level of the span the level belongs to; shift = 0: slot = J).  A wave then reads 64 consecutive doubles per
// load instruction; the row-major [J][36] storage of round 3 made every load instruction touch 64 cache lines.
struct PcrLevelLayout {
  uint32_t shift, rows;
  __host__ __device__ size_t slot(uint32_t J) const { return size_t(J & ((1u << shift) - 1u)) * rows + (J >> shift); }
};

__device__ __forceinline__ void m6_store_planes(double* __restrict__ p, size_t pitch, const double (&A)[36]) {
#pragma unroll
  for (int k = 0; k < 36; ++k) p[size_t(k) * pitch] = A[k];
}
__device__ __forceinline__ void m6_load_planes(const double* __restrict__ p, size_t pitch, double (&A)[36]) {
#pragma unroll
  for (int k = 0; k < 36; ++k) A[k] = p[size_t(k) * pitch];
}

// One level of parallel cyclic reduction with stride d: row J is combined with rows J − d and J + d,
//   alpha = −L_J D_{J−d}⁻¹,  gamma = −U_J D_{J+d}⁻¹,
//   L'_J = alpha L_{J−d},  D'_J = D_J + alpha U_{J−d} + gamma L_{J+d},  U'_J = gamma U_{J+d};
// alpha / gamma are kept (per level) for the right-hand sides.  in / out are different buffers.
__global__ __launch_bounds__(128) void pgo_pcr_setup_kernel(uint32_t n_agg, uint32_t d, const double* __restrict__ Lin,
                                                            const double* __restrict__ Din, const double* __restrict__ Uin,
                                                            double* __restrict__ Lout, double* __restrict__ Dout,
                                                            double* __restrict__ Uout, double* __restrict__ alpha,
                                                            double* __restrict__ gamma, PcrLevelLayout lay, size_t pitch) {
  // alpha / gamma: the planes of THIS level (see PcrLevelLayout)
  const uint32_t J = blockIdx.x * 128 + threadIdx.x;
  if (J >= n_agg) return;
  double Dn[36], Ln[36], Un[36], a[36], g[36], t[36], inv[36], blk[36];
  m6_load(Din + size_t(36) * J, Dn);
#pragma unroll
  for (int k = 0; k < 36; ++k) {
    Ln[k] = 0.0;
    Un[k] = 0.0;
    a[k] = 0.0;
    g[k] = 0.0;
  }
  if (J >= d) {
    m6_load(Din + size_t(36) * (J - d), blk);
    m6_spd_inverse(blk, inv);
    m6_load(Lin + size_t(36) * J, blk);
    m6_mul(blk, inv, a);
#pragma unroll
    for (int k = 0; k < 36; ++k) a[k] = -a[k];
    m6_load(Uin + size_t(36) * (J - d), blk);
    m6_mul(a, blk, t);
#pragma unroll
    for (int k = 0; k < 36; ++k) Dn[k] += t[k];
    m6_load(Lin + size_t(36) * (J - d), blk);
    m6_mul(a, blk, Ln);
  }
  if (J + d < n_agg) {
    m6_load(Din + size_t(36) * (J + d), blk);
    m6_spd_inverse(blk, inv);
    m6_load(Uin + size_t(36) * J, blk);
    m6_mul(blk, inv, g);
#pragma unroll
    for (int k = 0; k < 36; ++k) g[k] = -g[k];
    m6_load(Lin + size_t(36) * (J + d), blk);
    m6_mul(g, blk, t);
#pragma unroll
    for (int k = 0; k < 36; ++k) Dn[k] += t[k];
    m6_load(Uin + size_t(36) * (J + d), blk);
    m6_mul(g, blk, Un);
  }
  m6_store(Lout + size_t(36) * J, Ln);
  m6_store(Dout + size_t(36) * J, Dn);
  m6_store(Uout + size_t(36) * J, Un);
  m6_store_planes(alpha + lay.slot(J), pitch, a);
  m6_store_planes(gamma + lay.slot(J), pitch, g);
}

// Inverse of the decoupled diagonal blocks after the last level.
__global__ __launch_bounds__(128) void pgo_pcr_finish_kernel(uint32_t n_agg, const double* __restrict__ D,
                                                             double* __restrict__ Dinv) {
  const uint32_t J = blockIdx.x * 128 + threadIdx.x;
  if (J >= n_agg) return;
  double d[36], inv[36];
  m6_load(D + size_t(36) * J, d);
  m6_spd_inverse(d, inv);
  m6_store(Dinv + size_t(36) * J, inv);
}

// Several PCR levels of the right-hand side in ONE launch (round 3: one launch per level — 15 at 1 M poses, each a few µs
// of work behind a kernel boundary — plus one for the block solves).  Level l couples row J with rows J ± 2^l only, so after
// the levels [0, l0) the rows fall into 2^l0 independent residue classes J ≡ c (mod 2^l0); a workgroup takes T consecutive
// rows of one class (position m ↔ row c + m 2^l0), keeps the right-hand side in LDS (ping-pong) and runs the levels
// [l0, l1):   b'_J = b_J + alpha_J b_{J−d} + gamma_J b_{J+d},  d = 2^l
// What bounds it is what ONE CU can load (a level is 576 bytes of factors per row; ≈ 30 GB/s per CU, measured 7.5 µs per
// level with 512 rows per workgroup on 81 CUs): small workgroups (T = 128), few levels per span (4: halo 15) → every span
// runs on 200+ CUs.
//   * halo > 0: the T positions include `halo` = 2^(l1-l0) - 1 positions on either side whose values go wrong level by
//     level (their own neighbours are outside the workgroup) and are dropped: T - 2 halo positions are written;
//   * halo = 0: the whole class fits the workgroup (rows of a class <= T): no position is dropped, and with Dinv != nullptr
//     the decoupled blocks are solved in the same launch (x_J = D_J^-1 b_J).
// The factors of the NEXT level are requested as soon as the current ones have been used, so their latency overlaps the LDS
// exchange and the barrier; the loads sit behind no branch (a level index past the span re-reads the last level).
template <int T>
__global__ __launch_bounds__(T) void pgo_pcr_span_kernel(uint32_t n_agg, uint32_t l0, uint32_t l1, uint32_t halo,
                                                         const double* __restrict__ alpha, const double* __restrict__ gamma,
                                                         size_t pitch, PcrLevelLayout lay, const double* __restrict__ Dinv,
                                                         const double* __restrict__ bin, double* __restrict__ bout) {
  // (the right-hand side of a level is exchanged through LDS as 6 planes of T doubles: conflict-free 8-byte accesses)
  __shared__ double lds[2][6][T];
  const uint32_t cls = blockIdx.x & ((1u << l0) - 1u), chunk = blockIdx.x >> l0;
  const int tid = int(threadIdx.x);
  const long long m = (long long)chunk * (T - 2 * int(halo)) + tid - int(halo);  // position inside the class
  const long long J = (long long)cls + m * (1ll << l0);
  const bool exists = m >= 0 && J < (long long)n_agg;
  const size_t slot = lay.slot(exists ? uint32_t(J) : 0u);  // a lane without a row reads row 0's factors and drops them
  double b[6];
#pragma unroll
  for (int k = 0; k < 6; ++k) b[k] = exists ? bin[size_t(6) * size_t(J) + k] : 0.0;
  double fa[36], fg[36];
  if (l0 < l1) {
    m6_load_planes(alpha + size_t(l0) * 36 * pitch + slot, pitch, fa);
    m6_load_planes(gamma + size_t(l0) * 36 * pitch + slot, pitch, fg);
  }
  int cur = 0;
  for (uint32_t l = l0; l < l1; ++l) {
#pragma unroll
    for (int k = 0; k < 6; ++k) lds[cur][k][tid] = b[k];
    __syncthreads();
    const long long d = 1ll << l;
    const int dm = 1 << (l - l0);
    const uint32_t ln = l + 1 < l1 ? l + 1 : l;
    double xl[6], xr[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      xl[k] = tid - dm >= 0 ? lds[cur][k][tid - dm] : 0.0;
      xr[k] = tid + dm < T ? lds[cur][k][tid + dm] : 0.0;
    }
    if (exists && J >= d) m6_mulvec_add(fa, xl, b);
    m6_load_planes(alpha + size_t(ln) * 36 * pitch + slot, pitch, fa);
    if (exists && J + d < (long long)n_agg) m6_mulvec_add(fg, xr, b);
    m6_load_planes(gamma + size_t(ln) * 36 * pitch + slot, pitch, fg);
    cur ^= 1;
  }
  if (!exists || tid < int(halo) || tid >= T - int(halo)) return;
  if (Dinv != nullptr) {
    double mat[36], xi[6] = {0, 0, 0, 0, 0, 0};
    m6_load(Dinv + size_t(36) * size_t(J), mat);
    m6_mulvec_add(mat, b, xi);
#pragma unroll
    for (int k = 0; k < 6; ++k) b[k] = xi[k];
  }
#pragma unroll
  for (int k = 0; k < 6; ++k) bout[size_t(6) * size_t(J) + k] = b[k];
}

}  // namespace nos
