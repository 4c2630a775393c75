// pgo_kernels.hpp — pose-graph optimisation on the GPU (BASELINE.json configs[4]).
//
// Residual of one relative-pose constraint (reference: nonlinear_optimizer/pose_graph_optimizer/
// ceres_cost_functor.h:44-51; switchable form :83-94):
//     r_t = (p_q - p_r) - q_r (x) t_m
//     r_R = 2 vec( q_q^* (x) q_r (x) q_m )
//     loop constraints:  r <- s r,  r_7 = (1 - s) * 1e-9   with a free switch variable s
// The reference evaluates this through Ceres autodiff only — its analytic solver is an empty loop
// (pose_graph_optimizer_analytic.cc:21-42, "Make sparse Hessian / Solve normal equation using Sparse
// Cholesky / Update poses" are TODO comments).  Here the Jacobians are analytic, under the same
// right-multiplicative update the other analytic solvers use (p <- p + dp, q <- q (x) Exp(dw)):
//     d r / d x_r = [ -I   R_r [t_m]x              ]      d r / d x_q = [ I   0                ]
//                   [  0   (e_w I + [e_v]x) R_m^T  ]                    [ 0   -e_w I + [e_v]x  ]
// with e = q_q^* q_r q_m.
//
// MI355X mapping: the normal matrix is block sparse (6x6 block per pose, one per constraint).  Nothing
// of it is stored except the diagonal blocks: every sweep is "owner computes" — one lane per pose walks
// that pose's incident constraints (CSR adjacency built once), gathers the neighbour pose, re-derives
// the 3x3 Jacobian blocks in registers and accumulates its own row.  No atomics, no scatter, results
// bit-reproducible; poses (56 B) and vectors (48 B) of a million-pose graph stay in the Infinity
// Cache, so the sweeps are fp64-ALU / latency bound rather than HBM bound.  The damped system is solved
// by block-Jacobi preconditioned CG built from three such sweeps per iteration.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "assemble_kernels.hpp"

namespace nos {

constexpr double kSwitchPrior = 1e-9;  // ceres_cost_functor.h:93

// Layout rule: data a lane GATHERS from another pose / constraint is stored as one 64-byte record (one cache
// line per gather); data a lane reads for ITS OWN row is stored plane-wise (coalesced across the wave).
struct PgoView {
  const double* pose;          // [n_poses][8]: px py pz qw qx qy qz pad  (64-byte records)
  const double* edge;          // [n_edges][8]: t_m (3), q_m wxyz (4), switch value (1)
  const int32_t* ref;
  const int32_t* qry;
  const uint8_t* sw_free;      // 1: the switch is an optimisation variable (loop constraints)
  // adjacency: constraints incident to pose i are adj[adj_off[i] .. adj_off[i+1]); entry = 2*edge + role
  // (role 0: pose is the reference end, 1: the query end); adj_nbr holds the pose at the other end
  const uint32_t* adj_off;
  const uint32_t* adj;
  const uint32_t* adj_nbr;
  const uint8_t* fixed;        // per pose
  uint32_t n_poses;
  uint32_t n_edges;
};

struct Quat4 {
  double w, x, y, z;
};

__device__ __forceinline__ Quat4 qmul(const Quat4& a, const Quat4& b) {
  return {a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z, a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y,
          a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z, a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x};
}

__device__ __forceinline__ void qrot_matrix(const Quat4& q, double R[9]) {
  const double tx = 2 * q.x, ty = 2 * q.y, tz = 2 * q.z;
  const double twx = tx * q.w, twy = ty * q.w, twz = tz * q.w;
  const double txx = tx * q.x, txy = ty * q.x, txz = tz * q.x;
  const double tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
  R[0] = 1 - (tyy + tzz);
  R[1] = txy - twz;
  R[2] = txz + twy;
  R[3] = txy + twz;
  R[4] = 1 - (txx + tzz);
  R[5] = tyz - twx;
  R[6] = txz - twy;
  R[7] = tyz + twx;
  R[8] = 1 - (txx + tyy);
}

// Everything one constraint contributes, in registers.
struct EdgeTerms {
  double r[6];   // unscaled residual (r_t, r_R)
  double A[9];   // R_r [t_m]x                      (d r_t / d w_r)
  double B[9];   // (e_w I + [e_v]x) R_m^T          (d r_R / d w_r)
  double C[9];   // -e_w I + [e_v]x                 (d r_R / d w_q)
};

// 64-byte record load as four 16-byte loads
__device__ __forceinline__ void load_record(const double* base, size_t index, double (&v)[8]) {
  using V2 = double __attribute__((ext_vector_type(2)));
  const V2* p = reinterpret_cast<const V2*>(base + 8 * index);
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const V2 t = p[k];
    v[2 * k] = t[0];
    v[2 * k + 1] = t[1];
  }
}

// Terms of a constraint given the 64-byte records of its two ends (pr = reference pose, pq = query pose) and its own.
__device__ __forceinline__ double edge_terms_rec(const double (&pr)[8], const double (&pq)[8], const double (&ed)[8],
                                                 EdgeTerms& T);

// Terms of constraint e, records gathered from global memory (ir = reference pose, iq = query pose).
__device__ __forceinline__ double edge_terms(const PgoView& G, uint32_t e, uint32_t ir, uint32_t iq, EdgeTerms& T) {
  double pr[8], pq[8], ed[8];
  load_record(G.pose, ir, pr);
  load_record(G.pose, iq, pq);
  load_record(G.edge, e, ed);
  return edge_terms_rec(pr, pq, ed, T);
}

__device__ __forceinline__ double edge_terms_rec(const double (&pr)[8], const double (&pq)[8], const double (&ed)[8],
                                                 EdgeTerms& T) {
  const Quat4 qr{pr[3], pr[4], pr[5], pr[6]};
  const Quat4 qq{pq[3], pq[4], pq[5], pq[6]};
  const Quat4 qm{ed[3], ed[4], ed[5], ed[6]};
  const double tm[3] = {ed[0], ed[1], ed[2]};
  double Rr[9], Rm[9];
  qrot_matrix(qr, Rr);
  qrot_matrix(qm, Rm);
#pragma unroll
  for (int i = 0; i < 3; ++i)
    T.r[i] = (pq[i] - pr[i]) - (Rr[3 * i] * tm[0] + Rr[3 * i + 1] * tm[1] + Rr[3 * i + 2] * tm[2]);
  const Quat4 qqc{qq.w, -qq.x, -qq.y, -qq.z};
  const Quat4 eq = qmul(qmul(qqc, qr), qm);
  T.r[3] = 2 * eq.x;
  T.r[4] = 2 * eq.y;
  T.r[5] = 2 * eq.z;
  // A = R_r [t_m]x : column j = R_r (e_j-th column of [t]x);  [t]x = [0 -tz ty; tz 0 -tx; -ty tx 0]
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    T.A[3 * i + 0] = Rr[3 * i + 1] * tm[2] - Rr[3 * i + 2] * tm[1];
    T.A[3 * i + 1] = Rr[3 * i + 2] * tm[0] - Rr[3 * i + 0] * tm[2];
    T.A[3 * i + 2] = Rr[3 * i + 0] * tm[1] - Rr[3 * i + 1] * tm[0];
  }
  // E+ = e_w I + [e_v]x ,  E- = -e_w I + [e_v]x
  const double Ep[9] = {eq.w, -eq.z, eq.y, eq.z, eq.w, -eq.x, -eq.y, eq.x, eq.w};
  T.C[0] = -eq.w;
  T.C[1] = -eq.z;
  T.C[2] = eq.y;
  T.C[3] = eq.z;
  T.C[4] = -eq.w;
  T.C[5] = -eq.x;
  T.C[6] = -eq.y;
  T.C[7] = eq.x;
  T.C[8] = -eq.w;
  // B = E+ R_m^T
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j)
      T.B[3 * i + j] = Ep[3 * i] * Rm[3 * j] + Ep[3 * i + 1] * Rm[3 * j + 1] + Ep[3 * i + 2] * Rm[3 * j + 2];
  return ed[7];  // current switch value
}

// y = J_role x  (6-vector) for the unscaled Jacobian of the given role.
__device__ __forceinline__ void apply_J(const EdgeTerms& T, int role, const double x[6], double y[6]) {
  if (role == 0) {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      y[i] = -x[i] + T.A[3 * i] * x[3] + T.A[3 * i + 1] * x[4] + T.A[3 * i + 2] * x[5];
      y[3 + i] = T.B[3 * i] * x[3] + T.B[3 * i + 1] * x[4] + T.B[3 * i + 2] * x[5];
    }
  } else {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      y[i] = x[i];
      y[3 + i] = T.C[3 * i] * x[3] + T.C[3 * i + 1] * x[4] + T.C[3 * i + 2] * x[5];
    }
  }
}

// out += scale * J_role^T y
__device__ __forceinline__ void add_JT(const EdgeTerms& T, int role, const double y[6], double scale, double out[6]) {
  if (role == 0) {
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      out[j] += scale * (-y[j]);
      out[3 + j] += scale * (T.A[j] * y[0] + T.A[3 + j] * y[1] + T.A[6 + j] * y[2] + T.B[j] * y[3] + T.B[3 + j] * y[4] +
                             T.B[6 + j] * y[5]);
    }
  } else {
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      out[j] += scale * y[j];
      out[3 + j] += scale * (T.C[j] * y[3] + T.C[3 + j] * y[4] + T.C[6 + j] * y[5]);
    }
  }
}

// Linearisation sweep.  Per pose: H_ii (21 upper, row-major) and g_i; per launch: the cost
// (sum over constraints of |s r|^2 + (1-s)^2 c^2, counted at the reference end) as block partials.
// hdiag: 21 planes of n_poses (own-row data), grad: [n_poses][6] records.
__global__ __launch_bounds__(256) void pgo_linearize_kernel(PgoView G, double* __restrict__ hdiag,
                                                            double* __restrict__ grad,
                                                            double* __restrict__ cost_partials) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  double cost = 0.0;
  if (i < G.n_poses) {
    double H[21], g[6];
#pragma unroll
    for (int k = 0; k < 21; ++k) H[k] = 0.0;
#pragma unroll
    for (int k = 0; k < 6; ++k) g[k] = 0.0;
    const bool fixed = G.fixed[i] != 0;
    for (uint32_t a = G.adj_off[i]; a < G.adj_off[i + 1]; ++a) {
      const uint32_t e = G.adj[a] >> 1;
      const int role = int(G.adj[a] & 1u);
      const uint32_t j = G.adj_nbr[a];
      EdgeTerms T;
      const double s = edge_terms(G, e, role == 0 ? i : j, role == 0 ? j : i, T);
      if (role == 0) {
        double rr = 0.0;
#pragma unroll
        for (int k = 0; k < 6; ++k) rr += T.r[k] * T.r[k];
        cost += s * s * rr;
        if (G.sw_free[e]) cost += (1.0 - s) * (1.0 - s) * kSwitchPrior * kSwitchPrior;
      }
      if (fixed) continue;
      // g_i += (s J)^T (s r)
      add_JT(T, role, T.r, s * s, g);
      // H_ii += s^2 J^T J, column by column of J
      double Jcol[6][6];  // Jcol[c] = J e_c
#pragma unroll
      for (int c = 0; c < 6; ++c) {
        double ec[6] = {0, 0, 0, 0, 0, 0};
        ec[c] = 1.0;
        apply_J(T, role, ec, Jcol[c]);
      }
      int k = 0;
#pragma unroll
      for (int r0 = 0; r0 < 6; ++r0)
#pragma unroll
        for (int c0 = r0; c0 < 6; ++c0) {
          double d = 0.0;
#pragma unroll
          for (int m = 0; m < 6; ++m) d += Jcol[r0][m] * Jcol[c0][m];
          H[k++] += s * s * d;
        }
    }
    if (fixed) {  // identity block keeps the system SPD; the gradient of a fixed pose is zero
      const int dg[6] = {0, 6, 11, 15, 18, 20};
#pragma unroll
      for (int k = 0; k < 6; ++k) H[dg[k]] = 1.0;
    }
#pragma unroll
    for (int k = 0; k < 21; ++k) hdiag[size_t(k) * G.n_poses + i] = H[k];
#pragma unroll
    for (int k = 0; k < 6; ++k) grad[size_t(6) * i + k] = g[k];
  }
  // cost: block sum in fixed order
  __shared__ double lds[4];
  const double ws = wave_sum(cost);
  if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = ws;
  __syncthreads();
  if (threadIdx.x == 0) cost_partials[blockIdx.x] = (lds[0] + lds[1]) + (lds[2] + lds[3]);
}

// Switch variables (one lane per constraint): gradient s |r|^2 - c^2 (1 - s) and curvature |r|^2 + c^2.
__global__ __launch_bounds__(256) void pgo_switch_linearize_kernel(PgoView G, double* __restrict__ g_s,
                                                                   double* __restrict__ h_s) {
  const uint32_t e = blockIdx.x * 256 + threadIdx.x;
  if (e >= G.n_edges) return;
  if (!G.sw_free[e]) {
    g_s[e] = 0.0;
    h_s[e] = 1.0;
    return;
  }
  EdgeTerms T;
  const double s = edge_terms(G, e, uint32_t(G.ref[e]), uint32_t(G.qry[e]), T);
  double rr = 0.0;
#pragma unroll
  for (int k = 0; k < 6; ++k) rr += T.r[k] * T.r[k];
  g_s[e] = s * rr - kSwitchPrior * kSwitchPrior * (1.0 - s);
  h_s[e] = rr + kSwitchPrior * kSwitchPrior;
}

// ---- block partials → one sum, in ONE fixed order whatever kernel does it
//
// The order is that of a 1024-lane workgroup: lane t strides the rows t, t + 1024, … with four independent accumulators
// (rows t + 4096 k, + 1024, + 2048, + 3072; the ≤ 3 left-over rows go into the first), combines them as (s0 + s1) + (s2 + s3),
// and the 1024 lane values are folded by halving (lane t += lane t + o, o = 512 … 1).  NT threads (1024, or 256 when the
// sum runs in the tail of a sweep kernel) play the 1024 lanes, so the stand-alone kernel and the in-launch tails give the
// same bits.  The result is returned to thread 0 only.
template <int NT, int WIDTH>
__device__ __forceinline__ void pgo_sum_rows(const double* __restrict__ partials, uint32_t count, double* __restrict__ lds,
                                             double (&total)[WIDTH]) {
  // lds: WIDTH x 1024 doubles.  All WIDTH columns go through the loads and the fold together (one pass, one set of barriers)
  for (uint32_t t = threadIdx.x; t < 1024u; t += NT) {
    double s0[WIDTH], s1[WIDTH], s2[WIDTH], s3[WIDTH];
#pragma unroll
    for (int w = 0; w < WIDTH; ++w) s0[w] = s1[w] = s2[w] = s3[w] = 0.0;
    uint32_t i = t;
    for (; i + 3 * 1024 < count; i += 4 * 1024) {
#pragma unroll
      for (int w = 0; w < WIDTH; ++w) {
        s0[w] += partials[size_t(i) * WIDTH + w];
        s1[w] += partials[size_t(i + 1024) * WIDTH + w];
        s2[w] += partials[size_t(i + 2 * 1024) * WIDTH + w];
        s3[w] += partials[size_t(i + 3 * 1024) * WIDTH + w];
      }
    }
    for (; i < count; i += 1024) {
#pragma unroll
      for (int w = 0; w < WIDTH; ++w) s0[w] += partials[size_t(i) * WIDTH + w];
    }
#pragma unroll
    for (int w = 0; w < WIDTH; ++w) lds[w * 1024 + t] = (s0[w] + s1[w]) + (s2[w] + s3[w]);
  }
  __syncthreads();
  for (int o = 512; o >= 64; o >>= 1) {
    for (int t = int(threadIdx.x); t < o; t += NT) {
#pragma unroll
      for (int w = 0; w < WIDTH; ++w) lds[w * 1024 + t] += lds[w * 1024 + t + o];
    }
    __syncthreads();
  }
  // lanes t < o of the last six levels are the lanes of one wave: the same additions through shuffles, no barriers
  if (threadIdx.x < 64) {
#pragma unroll
    for (int w = 0; w < WIDTH; ++w) {
      double v = lds[w * 1024 + threadIdx.x];
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        const double u = __shfl_down(v, o, kWave);
        if (int(threadIdx.x) < o) v += u;
      }
      total[w] = v;
    }
  }
  __syncthreads();
}

// In-launch tail of a sweep whose workgroups each publish `width` partial sums: the workgroup that arrives last adds all
// rows up (pgo_sum_rows) and hands the totals to `finish` on its thread 0 — no dependent one-workgroup kernel and no
// one-lane scalar kernel behind the sweep.  Hand-off = cdna_hip_programming.md Guideline 16, recipe R1: the partials are
// stored write-through (agent-scope relaxed atomic stores = `global_store … sc1`) by thread 0, drained (`s_waitcnt
// vmcnt(0)`), then thread 0 takes a ticket (agent-scope atomic add); the workgroup whose ticket is the last one acquires
// (agent scope, L1 invalidate), waits for the invalidate, passes the workgroup barrier and reads the rows with plain
// loads.  The ticket word is reset by the finisher: launches of one stream are ordered, so the next sweep finds it at 0.
struct PgoTail {
  double* partials;      // [gridDim.x][width]
  unsigned int* ticket;  // one word per tail, zero between launches
  double* scalars;       // the device-resident CG scalars (layout at pgo_cg_alpha_kernel)
};

template <int WIDTH>
__device__ __forceinline__ void pgo_publish_partials(const PgoTail& T, const double (&v)[WIDTH]) {
  // thread 0 only
#pragma unroll
  for (int w = 0; w < WIDTH; ++w)
    __hip_atomic_store(T.partials + size_t(WIDTH) * blockIdx.x + w, v[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// → true on every thread of the workgroup that drew the last ticket (after the acquire).  `flag` = one LDS word.
__device__ __forceinline__ bool pgo_tail_is_last(const PgoTail& T, unsigned int* flag) {
  if (threadIdx.x == 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the write-through partials have left before the ticket is drawn
    const unsigned int old = __hip_atomic_fetch_add(T.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const bool last = old == gridDim.x - 1u;
    if (last) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // holds the barrier below until the invalidate has completed
      __hip_atomic_store(T.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    *flag = last ? 1u : 0u;
  }
  __syncthreads();
  return *flag != 0u;
}

// CG scalar updates (the bodies of pgo_cg_alpha_kernel / pgo_cg_beta_kernel, shared with the in-launch tails).
__device__ __forceinline__ void pgo_cg_alpha(double* __restrict__ s) {
  const double pap = s[0];
  if (s[7] != 0.0 || !(pap > 0.0) || !(pap <= 1.79769313486231570e308)) {
    s[5] = 0.0;
    s[7] = 1.0;
  } else {
    s[5] = s[4] / pap;
  }
}
__device__ __forceinline__ void pgo_cg_beta(double* __restrict__ s) {
  const double rz_new = s[0];
  s[3] = s[1];
  const double rz = s[4];
  s[6] = (s[7] != 0.0 || !(rz > 0.0)) ? 0.0 : rz_new / rz;
  if (s[7] == 0.0) s[4] = rz_new;
}

// y = (H + lambda diag(H)) x for the pose rows, matrix free, plus block partials of x.y (the p.Ap of CG).
// x / y: [n_poses][6] records; xs: per-constraint switch components (0 where the switch is not free).
// → this thread's share of x.y
__device__ __forceinline__ double pgo_matvec_pose_row(const PgoView& G, const double* __restrict__ hdiag, double lambda,
                                                      const double* __restrict__ x, const double* __restrict__ xs,
                                                      double* __restrict__ y, uint32_t mask_agg, uint32_t i) {
  // mask_agg != 0: the coupling of constraints whose ends lie more than one aggregate (of mask_agg consecutive poses)
  // apart is dropped, their diagonal part stays — the operator the coarse level is probed with (pgo_coarse_kernels.hpp)
  const size_t N = G.n_poses;
  double xy = 0.0;
  if (i < G.n_poses) {
  double xi[6], out[6];
#pragma unroll
  for (int k = 0; k < 6; ++k) xi[k] = x[size_t(6) * i + k];
  if (G.fixed[i]) {
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      out[k] = (1.0 + lambda) * xi[k];
      y[size_t(6) * i + k] = out[k];
      xy += xi[k] * out[k];
    }
  } else {
#pragma unroll
  for (int k = 0; k < 6; ++k) out[k] = 0.0;
  for (uint32_t a = G.adj_off[i]; a < G.adj_off[i + 1]; ++a) {
    const uint32_t e = G.adj[a] >> 1;
    const int role = int(G.adj[a] & 1u);
    const uint32_t j = G.adj_nbr[a];
    EdgeTerms T;
    const double s = edge_terms(G, e, role == 0 ? i : j, role == 0 ? j : i, T);
    double xj[6], v[6], vj[6];
    bool jfixed = G.fixed[j] != 0;
    if (mask_agg != 0) {
      const uint32_t ai = i / mask_agg, aj = j / mask_agg;
      jfixed = jfixed || ai > aj + 1u || aj > ai + 1u;
    }
#pragma unroll
    for (int k = 0; k < 6; ++k) xj[k] = jfixed ? 0.0 : x[size_t(6) * j + k];
    apply_J(T, role, xi, v);
    apply_J(T, 1 - role, xj, vj);
    const double xse = G.sw_free[e] ? xs[e] : 0.0;
#pragma unroll
    for (int k = 0; k < 6; ++k) v[k] = s * (v[k] + vj[k]) + T.r[k] * xse;  // (J_full x) restricted to the 6 rows
    add_JT(T, role, v, s, out);
  }
  const int dg[6] = {0, 6, 11, 15, 18, 20};
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    const double v = out[k] + lambda * hdiag[size_t(dg[k]) * N + i] * xi[k];
    y[size_t(6) * i + k] = v;
    xy += xi[k] * v;
  }
  }
  }
  return xy;
}

// Switch row of the same product → this thread's share of xs.ys.
__device__ __forceinline__ double pgo_matvec_switch_row(const PgoView& G, const double* __restrict__ h_s, double lambda,
                                                        const double* __restrict__ x, const double* __restrict__ xs,
                                                        double* __restrict__ ys, uint32_t e) {
  double xy = 0.0;
  if (e < G.n_edges) {
    double out;
    if (!G.sw_free[e]) {
      out = (1.0 + lambda) * xs[e];
    } else {
      const uint32_t ir = uint32_t(G.ref[e]), iq = uint32_t(G.qry[e]);
      EdgeTerms T;
      const double s = edge_terms(G, e, ir, iq, T);
      double xr[6], xq[6], vr[6], vq[6];
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        xr[k] = G.fixed[ir] ? 0.0 : x[size_t(6) * ir + k];
        xq[k] = G.fixed[iq] ? 0.0 : x[size_t(6) * iq + k];
      }
      apply_J(T, 0, xr, vr);
      apply_J(T, 1, xq, vq);
      double acc = 0.0;
#pragma unroll
      for (int k = 0; k < 6; ++k) acc += T.r[k] * (s * (vr[k] + vq[k]));
      out = acc + h_s[e] * (1.0 + lambda) * xs[e];
    }
    ys[e] = out;
    xy = xs[e] * out;
  }
  return xy;
}

// one value per thread → the workgroup's sum on thread 0 (4 waves, fixed order)
__device__ __forceinline__ double pgo_block_sum256(double v, double* __restrict__ lds4) {
  const double ws = wave_sum(v);
  if ((threadIdx.x & 63) == 0) lds4[threadIdx.x >> 6] = ws;
  __syncthreads();
  return (lds4[0] + lds4[1]) + (lds4[2] + lds4[3]);
}

// The pose rows alone, block partials of x.y to dot_partials (the coarse level's probing products; host-scalar PCG).
__global__ __launch_bounds__(256) void pgo_matvec_pose_kernel(PgoView G, const double* __restrict__ hdiag,
                                                              double lambda, const double* __restrict__ x,
                                                              const double* __restrict__ xs,
                                                              double* __restrict__ y,
                                                              double* __restrict__ dot_partials,
                                                              uint32_t mask_agg = 0) {
  const double xy = pgo_matvec_pose_row(G, hdiag, lambda, x, xs, y, mask_agg, blockIdx.x * 256 + threadIdx.x);
  __shared__ double lds[4];
  const double bs = pgo_block_sum256(xy, lds);
  if (threadIdx.x == 0) dot_partials[blockIdx.x] = bs;
}

// Switch rows alone; block partials of xs.ys (appended after the pose blocks' partials by the caller).
__global__ __launch_bounds__(256) void pgo_matvec_switch_kernel(PgoView G, const double* __restrict__ h_s, double lambda,
                                                                const double* __restrict__ x,
                                                                const double* __restrict__ xs,
                                                                double* __restrict__ ys,
                                                                double* __restrict__ dot_partials) {
  const double xy = pgo_matvec_switch_row(G, h_s, lambda, x, xs, ys, blockIdx.x * 256 + threadIdx.x);
  __shared__ double lds[4];
  const double bs = pgo_block_sum256(xy, lds);
  if (threadIdx.x == 0) dot_partials[blockIdx.x] = bs;
}

// The product of one PCG iteration in ONE launch: workgroups [0, pose_blocks) take the pose rows, the following ones the
// switch rows (only launched when a switch is free), and the workgroup that finishes last adds the partials of p.Ap up
// and computes the step length (pgo_cg_alpha) — what used to be product [+ switch product] + sum + alpha kernels.
__global__ __launch_bounds__(256) void pgo_matvec_cg_kernel(PgoView G, const double* __restrict__ hdiag,
                                                            const double* __restrict__ h_s, double lambda,
                                                            const double* __restrict__ x, double* __restrict__ y,
                                                            uint32_t pose_blocks, PgoTail tail) {
  const size_t N6 = size_t(6) * G.n_poses;
  double xy;
  if (blockIdx.x < pose_blocks)
    xy = pgo_matvec_pose_row(G, hdiag, lambda, x, x + N6, y, 0u, blockIdx.x * 256 + threadIdx.x);
  else
    xy = pgo_matvec_switch_row(G, h_s, lambda, x, x + N6, y + N6, (blockIdx.x - pose_blocks) * 256 + threadIdx.x);
  __shared__ double lds[1024];
  __shared__ unsigned int last_flag;
  const double bs = pgo_block_sum256(xy, lds);
  if (threadIdx.x == 0) {
    const double v[1] = {bs};
    pgo_publish_partials<1>(tail, v);
  }
  if (!pgo_tail_is_last(tail, &last_flag)) return;
  double total[1];
  pgo_sum_rows<256, 1>(tail.partials, gridDim.x, lds, total);
  if (threadIdx.x == 0) {
    tail.scalars[0] = total[0];
    pgo_cg_alpha(tail.scalars);
  }
}

// ---- block-local product (round 4)
//
// What the counters say about the owner-computes product above (profiles/r04pre_pgo_summary.json): 329 µs, 1.0-2.0 GB of
// L2-miss traffic for ≈ 0.53 GB of distinct bytes, L2 hit rate 0.41, vector ALUs 11 % busy — every constraint is gathered
// and its Jacobian blocks re-derived TWICE (once by the lane of either end), and the second visit comes long after the first
// has left the XCD's 4 MB L2 (the workgroups resident on an XCD touch ≈ 40 MB).  The block-local form visits every
// constraint once per workgroup:
//   * the poses are cut into blocks of P consecutive indices, one workgroup each; at creation every block gets the list of
//     the constraints that touch it ("entries", 80-byte records: the constraint's own 64 bytes + its id, its two pose ids and
//     the two adjacency slots it feeds; pose ids LOCAL to the block: own pose or halo index), in constraint order — a block's entries are ONE contiguous, coalesced stream, and a
//     constraint whose ends lie in two blocks is listed in both (16 % of the constraints at P = 128 on a trajectory graph);
//   * phase A, one lane per entry: the block's own poses and vector entries come from LDS (staged once, coalesced), poses
//     of other blocks from global memory; residual and Jacobian blocks once, v = s (J_r x_r + J_q x_q) + r x_s, and the two
//     6-vectors s J_r^T v, s J_q^T v go to the LDS slots of the two adjacency positions (switch rows: computed here too, by
//     the block that owns the reference end);
//   * phase B, one lane per pose: its slots are added up in adjacency order — the order the owner-computes sweep uses, so
//     the result is deterministic and differs from it only by the rounding of s J^T v (formed before the sum instead of
//     fused into it) — plus the damping term; block partials of x.y, and the in-launch tail of pgo_matvec_cg_kernel.
// Graphs whose blocks need more LDS slots than fit (hub poses) keep the owner-computes kernels (nos_pgo_create decides).
struct PgoBlockView {
  const double* entries;        // [n_entries][10]: t_m (3), q_m wxyz (4), switch, {e | ir_local << 32 | iq_local << 48},
                                //                  {slot_r | slot_q << 16}; *_local: < P = the block's own pose, else P + halo index
  const uint32_t* entry_off;    // [n_blocks + 1]
  const uint32_t* halo;         // pose ids of the other blocks' poses a block's entries refer to, block after block, ascending
  const uint32_t* halo_off;     // [n_blocks + 1]
  uint32_t n_blocks;
  uint32_t halo_cap;            // LDS room for halo poses per workgroup (>= halo poses of any block)
  uint32_t slot_cap;            // LDS contribution slots per workgroup (>= adjacency entries of any block)
};
constexpr uint32_t kPgoNoSlot = 0xFFFFu;

// Persistent workgroups: the grid is sized to what is resident (LDS-limited), every workgroup walks pose blocks
// blockIdx.x, + gridDim.x, … and publishes ONE partial of x.y and draws ONE ticket at its end.  Per block:
//   staging   own poses + vector entries (coalesced) and the HALO — the other blocks' poses this block's constraints touch,
//             listed at creation — into LDS; the lane of pose t also requests its six diagonal entries and its adjacency
//             range now, so that their latency is gone by phase B;
//   phase A   (above) every operand from LDS or from the entry stream: no dependent global gather left in the loop;
//   phase B   one lane per pose.
// SW: some switch is free (the vectors carry switch rows).  Its own instantiation because the switch rows gather their
// operands by constraint id inside phase A, and a load behind a branch there costs the counted waits of the ping-pong —
// in every instantiation that contains it, taken or not (read in the ISA).
template <int P, int T, bool SW>
__global__ __launch_bounds__(T) void pgo_matvec_block_kernel(PgoView G, PgoBlockView B, const double* __restrict__ hdiag,
                                                             const double* __restrict__ h_s, double lambda,
                                                             const double* __restrict__ x, double* __restrict__ y,
                                                             PgoTail tail, const double* __restrict__ z = nullptr,
                                                             double* __restrict__ x_new = nullptr) {
  constexpr bool switch_rows = SW;
  // z != nullptr: the PCG's direction update rides in the staging — the vector multiplied is x_new = z + beta x (beta =
  // scalars[6] of the preconditioner sweep's tail; x untouched after a breakdown, scalars[7]), formed where x is read and
  // written to x_new by the block that owns the row: the separate p = z + beta p pass (144 MB at 1 M poses) is gone.
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const bool fused = z != nullptr;
  const double beta = fused ? tail.scalars[6] : 0.0;
  const bool frozen = fused && tail.scalars[7] != 0.0;
  auto direction = [&](size_t idx) -> double {
    const double xv = x[idx];
    return (!fused || frozen) ? xv : z[idx] + beta * xv;
  };
  const uint32_t rows = uint32_t(P) + B.halo_cap;
  double* pose_s = smem;                  // [P + halo_cap][8]
  double* x_s = pose_s + size_t(rows) * 8;  // [P + halo_cap][6]
  double* contrib = x_s + size_t(rows) * 6;  // [slot_cap][6]; re-used by the tail's sum (>= 1024 doubles)
  const size_t N6 = size_t(6) * G.n_poses;
  double* ys = y + N6;
  using V2 = double __attribute__((ext_vector_type(2)));
  static_assert((P * 4) % T == 0, "the block's pose records are staged as whole rounds of 16-byte pieces");
  constexpr int kPoseRounds = P * 4 / T, kVecRounds = (P * 3 + T - 1) / T;
  V2* pose_s2 = reinterpret_cast<V2*>(pose_s);
  V2* x_s2 = reinterpret_cast<V2*>(x_s);
  const V2* pose2 = reinterpret_cast<const V2*>(G.pose);
  const V2* x2 = reinterpret_cast<const V2*>(x);
  const V2* z2 = reinterpret_cast<const V2*>(fused ? z : x);
  V2* xn2 = reinterpret_cast<V2*>(x_new);
  auto direction2 = [&](V2 xv, V2 zv) -> V2 { return (!fused || frozen) ? xv : V2{zv[0] + beta * xv[0], zv[1] + beta * xv[1]}; };
  double xy = 0.0;
  // what a block's staging needs before it can ask for anything else: read one block ahead
  struct Meta { uint32_t h0, n_halo, e0, n_ent, a0; };
  auto load_meta = [&](uint32_t b) -> Meta {
    const uint32_t h0 = B.halo_off[b], e0 = B.entry_off[b];
    return Meta{h0, B.halo_off[b + 1] - h0, e0, B.entry_off[b + 1] - e0, G.adj_off[b * P]};
  };
  Meta next = load_meta(blockIdx.x < B.n_blocks ? blockIdx.x : 0u);
  for (uint32_t blk = blockIdx.x; blk < B.n_blocks; blk += gridDim.x) {
    const Meta M = next;
    next = load_meta(blk + gridDim.x < B.n_blocks ? blk + gridDim.x : blk);
    const uint32_t base = blk * P;
    const uint32_t n_in = G.n_poses - base < uint32_t(P) ? G.n_poses - base : uint32_t(P);
    const uint32_t h0 = M.h0, n_halo = M.n_halo, e0 = M.e0, n_ent = M.n_ent, a0 = M.a0;
    // Staging.  EVERY load of the block's start is issued before the first wait (a loop of "load, store to LDS" pays one
    // memory round trip per turn — measured: 16 µs per block, 250 µs per product; profiles/r04c_pgo_summary.json): indices
    // past the end are clamped and their values dropped, so no load sits behind a branch.  Order = the order the results
    // are needed in: halo ids (the halo loads depend on them), own poses and vector entries, halo pieces, the first
    // entry of phase A, phase B's own-row operands.
    //   halo pieces: 7 per halo pose (4 of its record, 3 of its vector entries), two per lane and round
    const uint32_t n_pieces = n_halo * 7u;
    uint32_t hj[2], hid[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const uint32_t j = threadIdx.x + u * T;
      hj[u] = j < n_pieces ? j : (n_pieces ? n_pieces - 1 : 0u);
      hid[u] = B.halo[h0 + hj[u] / 7u];  // the list ends with one spare id: valid for an empty halo as well
    }
    V2 pv[kPoseRounds], xv[kVecRounds], zv[kVecRounds];
#pragma unroll
    for (int m = 0; m < kPoseRounds; ++m) {
      const uint32_t idx = threadIdx.x + m * T;
      pv[m] = pose2[size_t(4) * base + (idx < n_in * 4 ? idx : n_in * 4 - 1)];
    }
#pragma unroll
    for (int m = 0; m < kVecRounds; ++m) {
      const uint32_t idx = threadIdx.x + m * T;
      const size_t src = size_t(3) * base + (idx < n_in * 3 ? idx : n_in * 3 - 1);
      xv[m] = x2[src];
      zv[m] = z2[src];
    }
    V2 hv[2], hz[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const uint32_t c = hj[u] % 7u;
      const size_t vec = size_t(3) * hid[u] + (c < 4 ? 0u : c - 4u);
      const V2* src = c < 4 ? pose2 + size_t(4) * hid[u] + c : x2 + vec;  // one load through a selected address
      hv[u] = *src;
      hz[u] = z2[vec];
    }
    auto load_entry = [&](uint32_t k, double (&rec)[10]) {  // unconditional: an index past the end re-reads the last entry
      const uint32_t kk = k < n_ent ? k : (n_ent ? n_ent - 1 : 0u);  // (the array ends with one spare record)
      const V2* q = reinterpret_cast<const V2*>(B.entries + size_t(10) * (e0 + kk));
#pragma unroll
      for (int m = 0; m < 5; ++m) {
        const V2 v = q[m];
        rec[2 * m] = v[0];
        rec[2 * m + 1] = v[1];
      }
    };
    double recA[10], recB[10];
    load_entry(threadIdx.x, recA);
    double hd[6];
    uint32_t a_lo, a_hi;
    {
      const uint32_t i = base + (threadIdx.x < n_in ? threadIdx.x : n_in - 1);
      const int dg[6] = {0, 6, 11, 15, 18, 20};
#pragma unroll
      for (int m = 0; m < 6; ++m) hd[m] = hdiag[size_t(dg[m]) * G.n_poses + i];
      a_lo = G.adj_off[i] - a0;
      a_hi = G.adj_off[i + 1] - a0;
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int m = 0; m < kPoseRounds; ++m) {
      const uint32_t idx = threadIdx.x + m * T;
      if (idx < n_in * 4) pose_s2[idx] = pv[m];
    }
#pragma unroll
    for (int m = 0; m < kVecRounds; ++m) {
      const uint32_t idx = threadIdx.x + m * T;
      if (idx < n_in * 3) {
        const V2 v = direction2(xv[m], zv[m]);
        x_s2[idx] = v;
        if (fused) xn2[size_t(3) * base + idx] = v;
      }
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const uint32_t j = threadIdx.x + u * T, h = hj[u] / 7u, c = hj[u] % 7u;
      if (j < n_pieces) {
        if (c < 4)
          pose_s2[size_t(P + h) * 4 + c] = hv[u];
        else
          x_s2[size_t(P + h) * 3 + (c - 4)] = direction2(hv[u], hz[u]);
      }
    }
    for (uint32_t j = threadIdx.x + 2 * T; j < n_pieces; j += T) {  // halos beyond 2 T / 7 poses: round after round
      const uint32_t h = j / 7u, c = j % 7u, id = B.halo[h0 + h];
      if (c < 4)
        pose_s2[size_t(P + h) * 4 + c] = pose2[size_t(4) * id + c];
      else
        x_s2[size_t(P + h) * 3 + (c - 4)] = direction2(x2[size_t(3) * id + (c - 4)], z2[size_t(3) * id + (c - 4)]);
    }
    __syncthreads();

    auto evaluate = [&](const double (&rec)[10]) {
      const unsigned long long w8 = (unsigned long long)__double_as_longlong(rec[8]), w9 = (unsigned long long)__double_as_longlong(rec[9]);
      const uint32_t e = uint32_t(w8), lr = uint32_t(w8 >> 32) & 0xFFFFu, lq = uint32_t(w8 >> 48) & 0xFFFFu;
      const uint32_t slot_r = uint32_t(w9) & 0xFFFFu, slot_q = uint32_t(w9 >> 16) & 0xFFFFu;
      // The upper half of w9 carries nothing and, without switch rows, nobody reads e — and a loaded register nobody reads
      // is free for the allocator the moment the load is ISSUED: it was handed out as an address register while the record
      // was still in flight, and the write-after-write hazard turned the counted wait of the ping-pong into
      // s_waitcnt vmcnt(0) (read in the ISA).  Reading both here keeps them occupied until the record is consumed.
      asm volatile("" ::"v"(uint32_t(w9 >> 32)), "v"(e));
      double pr[8], pq[8], ed[8], xr[6], xq[6];
#pragma unroll
      for (int m = 0; m < 8; ++m) {
        ed[m] = rec[m];
        pr[m] = pose_s[8 * lr + m];
        pq[m] = pose_s[8 * lq + m];
      }
      // a fixed pose does not move: its vector entries count as zero (pose record element 7 = fixed flag)
#pragma unroll
      for (int m = 0; m < 6; ++m) {
        xr[m] = pr[7] != 0.0 ? 0.0 : x_s[6 * lr + m];
        xq[m] = pq[7] != 0.0 ? 0.0 : x_s[6 * lq + m];
      }
      EdgeTerms Tm;
      const double s = edge_terms_rec(pr, pq, ed, Tm);
      double vr[6], vq[6], v[6];
      apply_J(Tm, 0, xr, vr);
      apply_J(Tm, 1, xq, vq);
      bool free_sw = false;
      double xs_e = 0.0;  // this constraint's switch component of the vector
      if constexpr (switch_rows) {
        free_sw = G.sw_free[e] != 0;
        xs_e = direction(N6 + e);
      }
      const double xse = free_sw ? xs_e : 0.0;
#pragma unroll
      for (int m = 0; m < 6; ++m) v[m] = s * (vr[m] + vq[m]) + Tm.r[m] * xse;
      if (slot_r != kPgoNoSlot) {
        double c[6] = {0, 0, 0, 0, 0, 0};
        add_JT(Tm, 0, v, s, c);
#pragma unroll
        for (int m = 0; m < 6; ++m) contrib[6 * slot_r + m] = c[m];
        if constexpr (switch_rows) {  // the switch row of this constraint, by the block that owns its reference end
          double out;
          if (free_sw) {
            double acc = 0.0;
#pragma unroll
            for (int m = 0; m < 6; ++m) acc += Tm.r[m] * (s * (vr[m] + vq[m]));
            out = acc + h_s[e] * (1.0 + lambda) * xs_e;
          } else {
            out = (1.0 + lambda) * xs_e;
          }
          ys[e] = out;
          if (fused) x_new[N6 + e] = xs_e;
          xy += xs_e * out;
        }
      }
      if (slot_q != kPgoNoSlot) {
        double c[6] = {0, 0, 0, 0, 0, 0};
        add_JT(Tm, 1, v, s, c);
#pragma unroll
        for (int m = 0; m < 6; ++m) contrib[6 * slot_q + m] = c[m];
      }
    };
    // phase A: two named record buffers, the next entry's 80 bytes in flight while the current one is evaluated (the first
    // entry was requested with the staging)
    for (uint32_t k = threadIdx.x; k < n_ent; k += 2 * T) {
      load_entry(k + T, recB);
      __builtin_amdgcn_sched_barrier(0);
      evaluate(recA);
      __builtin_amdgcn_sched_barrier(0);
      load_entry(k + 2 * T, recA);
      __builtin_amdgcn_sched_barrier(0);
      if (k + T < n_ent) evaluate(recB);
      __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
    // phase B: one lane per pose, its slots in adjacency order
    if (threadIdx.x < n_in) {
      const uint32_t i = base + threadIdx.x;
      double xi[6], out[6];
#pragma unroll
      for (int m = 0; m < 6; ++m) xi[m] = x_s[6 * threadIdx.x + m];
      if (pose_s[8 * threadIdx.x + 7] != 0.0) {
#pragma unroll
        for (int m = 0; m < 6; ++m) out[m] = (1.0 + lambda) * xi[m];
      } else {
#pragma unroll
        for (int m = 0; m < 6; ++m) out[m] = 0.0;
        for (uint32_t a = a_lo; a < a_hi; ++a) {
#pragma unroll
          for (int m = 0; m < 6; ++m) out[m] += contrib[6 * a + m];
        }
#pragma unroll
        for (int m = 0; m < 6; ++m) out[m] += lambda * hd[m] * xi[m];
      }
#pragma unroll
      for (int m = 0; m < 6; ++m) {
        y[size_t(6) * i + m] = out[m];
        xy += xi[m] * out[m];
      }
    }
    __syncthreads();  // the staged rows and the slots are free for the next block (and, after the last one, for the tail)
  }
  __shared__ unsigned int last_flag;
  double* lds = contrib;
  const double ws = wave_sum(xy);
  if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = ws;
  __syncthreads();
  if (threadIdx.x == 0) {
    double bs = 0.0;
    for (int w = 0; w < T / 64; ++w) bs += lds[w];
    const double v1[1] = {bs};
    pgo_publish_partials<1>(tail, v1);
  }
  if (!pgo_tail_is_last(tail, &last_flag)) return;
  double total[1];
  pgo_sum_rows<T, 1>(tail.partials, gridDim.x, lds, total);
  if (threadIdx.x == 0) {
    tail.scalars[0] = total[0];
    pgo_cg_alpha(tail.scalars);
  }
}

// after a retract: the block entries' copies of the switch values follow the constraint records
__global__ __launch_bounds__(256) void pgo_refresh_entries_kernel(uint32_t n_entries, const uint32_t* __restrict__ entry_edge,
                                                                  const double* __restrict__ edge, double* __restrict__ entries) {
  const uint32_t k = blockIdx.x * 256 + threadIdx.x;
  if (k >= n_entries) return;
  entries[size_t(10) * k + 7] = edge[size_t(8) * entry_edge[k] + 7];
}

// Block-Jacobi preconditioner: inverse of the damped 6x6 diagonal block (Cholesky), stored as 21 upper
// entries of the symmetric inverse.
__global__ __launch_bounds__(256) void pgo_precond_kernel(const double* __restrict__ hdiag, double lambda, uint32_t n,
                                                          double* __restrict__ minv) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  double A[6][6];
  int k = 0;
#pragma unroll
  for (int r = 0; r < 6; ++r)
#pragma unroll
    for (int c = r; c < 6; ++c) {
      const double v = hdiag[size_t(k++) * n + i];
      A[r][c] = v;
      A[c][r] = v;
    }
#pragma unroll
  for (int r = 0; r < 6; ++r) A[r][r] *= 1.0 + lambda;
  // Cholesky A = L L^T (in place, lower)
  double L[6][6];
#pragma unroll
  for (int r = 0; r < 6; ++r)
#pragma unroll
    for (int c = 0; c < 6; ++c) L[r][c] = 0.0;
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    double d = A[j][j];
#pragma unroll
    for (int m = 0; m < 6; ++m)
      if (m < j) d -= L[j][m] * L[j][m];
    d = d > 1e-300 ? d : 1e-300;
    const double lj = sqrt(d);
    L[j][j] = lj;
#pragma unroll
    for (int r = 0; r < 6; ++r)
      if (r > j) {
        double v = A[r][j];
#pragma unroll
        for (int m = 0; m < 6; ++m)
          if (m < j) v -= L[r][m] * L[j][m];
        L[r][j] = v / lj;
      }
  }
  // inverse of L (lower), then A^-1 = L^-T L^-1
  double Li[6][6];
#pragma unroll
  for (int r = 0; r < 6; ++r)
#pragma unroll
    for (int c = 0; c < 6; ++c) Li[r][c] = 0.0;
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
  k = 0;
#pragma unroll
  for (int r = 0; r < 6; ++r)
#pragma unroll
    for (int c = r; c < 6; ++c) {
      double v = 0.0;
#pragma unroll
      for (int m = 0; m < 6; ++m)
        if (m >= c) v += Li[m][r] * Li[m][c];
      minv[size_t(k++) * n + i] = v;
    }
}

// z = M^-1 r on pose rows and switch rows, plus block partials of r.z and r.r.
// partials: [gridDim.x][2]
__global__ __launch_bounds__(256) void pgo_apply_precond_kernel(const double* __restrict__ minv,
                                                                const double* __restrict__ h_s, double lambda,
                                                                uint32_t n_poses, uint32_t n_edges,
                                                                const double* __restrict__ r,
                                                                const double* __restrict__ rs, double* __restrict__ z,
                                                                double* __restrict__ zs,
                                                                double* __restrict__ partials,
                                                                const double* __restrict__ pose_rec = nullptr,
                                                                const uint8_t* __restrict__ fixed = nullptr,
                                                                const double* __restrict__ xc = nullptr, uint32_t agg = 0,
                                                                PgoTail tail = PgoTail{nullptr, nullptr, nullptr}) {
  // xc != nullptr: two-level form, z_i = M_ii^-1 r_i + B_i xc[aggregate of i] (coarse correction, pgo_coarse_kernels.hpp)
  const uint32_t t = blockIdx.x * 256 + threadIdx.x;
  double rz = 0.0, rr = 0.0;
  if (t < n_poses) {
    double ri[6], M[6][6];
    int k = 0;
    using V2 = double __attribute__((ext_vector_type(2)));
    {  // the row's six entries as three 16-byte pieces
      const V2* r2 = reinterpret_cast<const V2*>(r) + size_t(3) * t;
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        const V2 v = r2[a];
        ri[2 * a] = v[0];
        ri[2 * a + 1] = v[1];
      }
    }
#pragma unroll
    for (int a = 0; a < 6; ++a)
#pragma unroll
      for (int b = a; b < 6; ++b) {
        const double v = minv[size_t(k++) * n_poses + t];
        M[a][b] = v;
        M[b][a] = v;
      }
    double corr[6] = {0, 0, 0, 0, 0, 0};
    if (xc != nullptr && !fixed[t]) {
      // B_t = [ I  -[p_t - c]x ; 0  R_t^T ] applied to the aggregate's (a, theta)
      double pi[8], pc[8], R[9], xa[6];
      load_record(pose_rec, t, pi);
      load_record(pose_rec, size_t(t / agg) * agg, pc);
      qrot_matrix(Quat4{pi[3], pi[4], pi[5], pi[6]}, R);
#pragma unroll
      for (int a = 0; a < 6; ++a) xa[a] = xc[size_t(6) * (t / agg) + a];
      const double d[3] = {pi[0] - pc[0], pi[1] - pc[1], pi[2] - pc[2]};
      corr[0] = xa[0] + (xa[4] * d[2] - xa[5] * d[1]);  // a + theta x d
      corr[1] = xa[1] + (xa[5] * d[0] - xa[3] * d[2]);
      corr[2] = xa[2] + (xa[3] * d[1] - xa[4] * d[0]);
#pragma unroll
      for (int a = 0; a < 3; ++a) corr[3 + a] = R[a] * xa[3] + R[3 + a] * xa[4] + R[6 + a] * xa[5];  // R^T theta
    }
    double zi[6];
#pragma unroll
    for (int a = 0; a < 6; ++a) {
      double v = corr[a];
#pragma unroll
      for (int b = 0; b < 6; ++b) v += M[a][b] * ri[b];
      zi[a] = v;
      rz += ri[a] * v;
      rr += ri[a] * ri[a];
    }
    V2* z2 = reinterpret_cast<V2*>(z) + size_t(3) * t;
#pragma unroll
    for (int a = 0; a < 3; ++a) z2[a] = V2{zi[2 * a], zi[2 * a + 1]};
  } else if (t - n_poses < n_edges) {
    const uint32_t e = t - n_poses;
    const double v = rs[e] / (h_s[e] * (1.0 + lambda));
    zs[e] = v;
    rz = rs[e] * v;
    rr = rs[e] * rs[e];
  }
  __shared__ double lds[2048];
  __shared__ unsigned int last_flag;
  const double a = wave_sum(rz), b = wave_sum(rr);
  if ((threadIdx.x & 63) == 0) {
    lds[2 * (threadIdx.x >> 6)] = a;
    lds[2 * (threadIdx.x >> 6) + 1] = b;
  }
  __syncthreads();
  const double v[2] = {(lds[0] + lds[2]) + (lds[4] + lds[6]), (lds[1] + lds[3]) + (lds[5] + lds[7])};
  if (tail.ticket == nullptr) {  // host-read scalars: a separate sum kernel follows
    if (threadIdx.x == 0) {
      partials[2 * blockIdx.x] = v[0];
      partials[2 * blockIdx.x + 1] = v[1];
    }
    return;
  }
  // device-resident CG scalars: the last workgroup sums r.z and r.r and computes beta (pgo_cg_beta)
  if (threadIdx.x == 0) pgo_publish_partials<2>(tail, v);
  if (!pgo_tail_is_last(tail, &last_flag)) return;
  double totals[2];
  pgo_sum_rows<256, 2>(tail.partials, gridDim.x, lds, totals);
  if (threadIdx.x == 0) {
    tail.scalars[0] = totals[0];
    tail.scalars[1] = totals[1];
    pgo_cg_beta(tail.scalars);
  }
}

// Generic vector helpers over the concatenated unknown vector (6 n_poses + n_edges entries).
__global__ __launch_bounds__(256) void pgo_dot_kernel(const double* __restrict__ a, const double* __restrict__ b,
                                                      size_t n, double* __restrict__ partials) {
  double s = 0.0;
  for (size_t i = size_t(blockIdx.x) * 256 + threadIdx.x; i < n; i += size_t(gridDim.x) * 256) s += a[i] * b[i];
  __shared__ double lds[4];
  const double ws = wave_sum(s);
  if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = ws;
  __syncthreads();
  if (threadIdx.x == 0) partials[blockIdx.x] = (lds[0] + lds[1]) + (lds[2] + lds[3]);
}

// x += alpha p ; r -= alpha q
__global__ __launch_bounds__(256) void pgo_cg_update_kernel(size_t n, double alpha, const double* __restrict__ p,
                                                            const double* __restrict__ q, double* __restrict__ x,
                                                            double* __restrict__ r) {
  const size_t i = size_t(blockIdx.x) * 256 + threadIdx.x;
  if (i >= n) return;
  x[i] += alpha * p[i];
  r[i] -= alpha * q[i];
}

// p = z + beta p
__global__ __launch_bounds__(256) void pgo_cg_direction_kernel(size_t n, double beta, const double* __restrict__ z,
                                                               double* __restrict__ p) {
  const size_t i = size_t(blockIdx.x) * 256 + threadIdx.x;
  if (i >= n) return;
  p[i] = z[i] + beta * p[i];
}

// sums `count` block partials (`width` interleaved values each) in a fixed order: 1024 lanes stride the
// rows with four independent accumulators, then a fixed tree over the lanes
// Device-resident CG scalars (no host round trip inside a PCG iteration).  s[0], s[1] receive the sums of the partials
// kernels as before; s[3] = |r|^2 of the latest iterate, s[4] = r·z of the current direction, s[5] = alpha, s[6] = beta,
// s[7] = 1 after a breakdown (p·Ap <= 0 or not finite: the iterate is kept from then on).
__global__ void pgo_cg_alpha_kernel(double* __restrict__ s) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  pgo_cg_alpha(s);
}
__global__ void pgo_cg_beta_kernel(double* __restrict__ s) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  pgo_cg_beta(s);
}
__global__ __launch_bounds__(256) void pgo_cg_update_dev_kernel(size_t n, const double* __restrict__ s,
                                                                const double* __restrict__ p, const double* __restrict__ q,
                                                                double* __restrict__ x, double* __restrict__ r) {
  const size_t i = size_t(blockIdx.x) * 256 + threadIdx.x;
  if (i >= n) return;
  const double alpha = s[5];
  x[i] += alpha * p[i];
  r[i] -= alpha * q[i];
}
__global__ __launch_bounds__(256) void pgo_cg_direction_dev_kernel(size_t n, const double* __restrict__ s,
                                                                   const double* __restrict__ z, double* __restrict__ p) {
  const size_t i = size_t(blockIdx.x) * 256 + threadIdx.x;
  if (i >= n) return;
  if (s[7] != 0.0) return;  // after a breakdown the direction is frozen (alpha is 0 anyway)
  p[i] = z[i] + s[6] * p[i];
}

__global__ __launch_bounds__(1024) void pgo_sum_partials_kernel(const double* __restrict__ partials, uint32_t count,
                                                                int width, double* __restrict__ out) {
  __shared__ double lds[2048];
  if (width == 1) {
    double total[1];
    pgo_sum_rows<1024, 1>(partials, count, lds, total);
    if (threadIdx.x == 0) out[0] = total[0];
  } else {  // width 2: r.z and r.r of the preconditioner sweep
    double total[2];
    pgo_sum_rows<1024, 2>(partials, count, lds, total);
    if (threadIdx.x == 0) {
      out[0] = total[0];
      out[1] = total[1];
    }
  }
}

// poses <- poses [+] step ;  switches += step
__global__ __launch_bounds__(256) void pgo_retract_kernel(uint32_t n_poses, uint32_t n_edges,
                                                          const uint8_t* __restrict__ fixed,
                                                          const uint8_t* __restrict__ sw_free,
                                                          const double* __restrict__ dx, const double* __restrict__ dxs,
                                                          double* __restrict__ pose, double* __restrict__ edge) {
  const uint32_t t = blockIdx.x * 256 + threadIdx.x;
  if (t < n_poses) {
    if (fixed[t]) return;
    double* P = pose + size_t(8) * t;
    const double* d6 = dx + size_t(6) * t;
    P[0] += d6[0];
    P[1] += d6[1];
    P[2] += d6[2];
    const double w[3] = {d6[3], d6[4], d6[5]};
    const double theta = sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
    Quat4 d;
    if (theta < 1e-6) {  // ComputeQuaternion, pose_graph_optimizer.h:70-86
      d = {1.0, 0.5 * w[0], 0.5 * w[1], 0.5 * w[2]};
    } else {
      const double k = sin(0.5 * theta) / theta;
      d = {cos(0.5 * theta), k * w[0], k * w[1], k * w[2]};
    }
    const Quat4 qn = qmul(Quat4{P[3], P[4], P[5], P[6]}, d);
    const double inv = 1.0 / sqrt(qn.w * qn.w + qn.x * qn.x + qn.y * qn.y + qn.z * qn.z);
    P[3] = qn.w * inv;
    P[4] = qn.x * inv;
    P[5] = qn.y * inv;
    P[6] = qn.z * inv;
  } else if (t - n_poses < n_edges) {
    const uint32_t e = t - n_poses;
    if (sw_free[e]) edge[size_t(8) * e + 7] += dxs[e];
  }
}

}  // namespace nos
