// assemble_indexed.hpp — the voxel-indexed layout's assemble kernel and the kernels that build its datasets.
// Part of the hand-written gfx950 kernels of the Gauss-Newton normal-equation assembly path; see assemble_kernels.hpp
// (the umbrella header every translation unit includes) for the overview and the reference citations.
#pragma once

#include "assemble_one_launch.hpp"

namespace nos {

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

// After a one-launch solve with an in-launch exchange gave up: the rounds it may have used are never used again (every rank
// skips the same number).  round[1] = the last round of a PATIENT exchange: the ranks abandon a launch up to one in-launch
// time-out apart, so the first exchanges of the loop that redoes the solve wait four times as long for each other as usual
// (mailbox_allreduce) before they call a peer missing.
__attribute__((unused)) static __global__ void mailbox_skip_rounds_kernel(unsigned long long* round, unsigned long long n) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    round[0] += n;
    round[1] = round[0] + 2ull;
  }
}

}  // namespace nos
