// assemble_one_launch.hpp — the whole LM loop in one launch: one workgroup (small problems), and the grid form with data resident on chip or streamed.
// Part of the hand-written gfx950 kernels of the Gauss-Newton normal-equation assembly path; see assemble_kernels.hpp
// (the umbrella header every translation unit includes) for the overview and the reference citations.
#pragma once

#include "assemble_pass.hpp"

namespace nos {

// ---------------------------------------------------------------- whole solve in one workgroup (small problems)
//
// At the reference's own test sizes (630 reprojection points, ≈ 2.9 k NDT correspondences) an LM iteration through the
// grid kernel costs ≈ 11-12 µs, nearly all of it launch, hand-off and dispatch latency.  Below kSingleBlockMaxElements
// plane-elements the whole loop runs inside ONE workgroup and ONE launch instead: the data (≤ 1 MB) stays in L2, the sums are
// reduced inside the block, one lane runs the same nos_host::LmAdvance* loop body on a state kept in LDS, and the
// next iteration starts after one barrier — no grid-wide hand-off, nothing to wait for, no way to hang.
// One CU evaluates a 512-correspondence NDT chunk in ≈ 0.9 µs, so the single-workgroup form only pays while the whole
// pass stays below the ≈ 6 µs that a launch with its hand-off costs: measured 11.3 → 5.3 µs per iteration at 630
// reprojection points, but no gain at 2 900 NDT correspondences (6 chunks) — hence a budget in plane-elements.
constexpr size_t kSingleBlockMaxElements = size_t(1024) * 15;  // n × fields: 1024 NDT or 3072 reprojection correspondences

template <typename Problem, typename T, int BLOCK, bool NT = false>
__global__ __launch_bounds__(BLOCK) void solve_single_block_kernel(TiledLayout L, typename Problem::Params P,
                                                                  uint32_t n_chunks, LmDevice* lm,
                                                                  double* __restrict__ cost_history, int history_capacity,
                                                                  double* entry_host, unsigned long long* seq_host,
                                                                  unsigned long long seq) {
  constexpr int kF = Problem::kFields;
  constexpr int kOut = Problem::kOut;
  const T* __restrict__ base = static_cast<const T*>(L.base);
  __shared__ double s_lm_raw[(sizeof(LmDevice) + 7) / 8];  // raw storage: the struct has default member initialisers
  LmDevice& s_lm = *reinterpret_cast<LmDevice*>(s_lm_raw);
  __shared__ double s_sum[kLmTotDoubles(kOut)];
  if (threadIdx.x == 0) s_lm = *lm;
  __syncthreads();
  int executed = 0;
  while (s_lm.st.done == 0) {  // block-uniform: every thread reads the same LDS word after a barrier
    set_pose(P, &s_lm);
    T acc[kOut];
#pragma unroll
    for (int k = 0; k < kOut; ++k) acc[k] = T(0);
    // several chunks per round, all their loads in flight before the first item is evaluated: with one workgroup there
    // are no other waves to hide the L2 latency behind
    constexpr uint32_t kRound = (kF * sizeof(T) > 64) ? 2 : 4;  // 15 fp64 planes: two chunks fill the register file
    for (uint32_t c = 0; c < n_chunks; c += kRound) {
      T x[kRound][kF][1];
      uint64_t i0[kRound];
#pragma unroll
      for (uint32_t u = 0; u < kRound; ++u) {
        const uint32_t cu = (c + u < n_chunks) ? c + u : c;  // clamped: re-reads chunk c, masked out below
        i0[u] = uint64_t(cu) * BLOCK + threadIdx.x;
        const uint64_t off = (i0[u] >> L.tile_shift) * L.tile_stride + (i0[u] & L.tile_mask);
#pragma unroll
        for (int f = 0; f < kF; ++f) load_items<T, 1, NT>(base + off + uint64_t(f) * L.field_stride, x[u][f]);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (uint32_t u = 0; u < kRound; ++u) {
        if (c + u < n_chunks) {  // block-uniform
          T xi[kF];
#pragma unroll
          for (int f = 0; f < kF; ++f) xi[f] = x[u][f][0];
          Problem::item(xi, P, i0[u] < L.n, acc);
        }
      }
    }
    double dacc[kOut];
#pragma unroll
    for (int k = 0; k < kOut; ++k) dacc[k] = double(acc[k]);
    block_reduce_store<kOut, BLOCK>(dacc, s_sum, false);
    __syncthreads();
    if (threadIdx.x == 0) {
      if (cost_history != nullptr && executed < history_capacity)
        __hip_atomic_store(cost_history + executed, s_sum[kOut - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      lm_step_lane<kOut>(lds_ptr(s_sum), lds_ptr(&s_lm));
    }
    ++executed;
    __syncthreads();
  }
  if (threadIdx.x < kOut && entry_host != nullptr && executed > 0)
    __hip_atomic_store(entry_host + kLogOut + threadIdx.x, s_sum[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  if (threadIdx.x == 0) {
    const nos_host::LmState st = s_lm.st;
    lm->st = st;
    if (entry_host != nullptr) {
#pragma unroll
      for (int k = 0; k < 9; ++k)
        __hip_atomic_store(entry_host + kLogR + k, st.R[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
#pragma unroll
      for (int k = 0; k < 3; ++k)
        __hip_atomic_store(entry_host + kLogT + k, st.t[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __hip_atomic_store(entry_host + kLogLambda, st.lambda, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __hip_atomic_store(entry_host + kLogPrevCost, st.previous_cost, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __hip_atomic_store(entry_host + kLogCost, st.cost, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __hip_atomic_store(entry_host + kLogIteration, double(st.iteration), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __hip_atomic_store(entry_host + kLogDone, double(st.done), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __hip_atomic_store(entry_host + kLogOk, double(st.ok), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __hip_atomic_store(entry_host + kLogExecuted, double(executed), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
  if (threadIdx.x < kWave) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (threadIdx.x == 0 && seq_host != nullptr) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __hip_atomic_store(seq_host, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

// ---------------------------------------------------------------- whole solve in one launch, data resident on chip
//
// Between the single-workgroup form above and the sizes where a launch is mostly streaming, an LM iteration through
// one launch per iteration costs ≈ 12 µs, nearly all of it kernel boundary, dispatch, first loads and hand-off.  Up to the
// on-chip capacity (ResidentShape below) the whole loop runs in ONE launch instead, one 512-thread workgroup per CU, all
// resident at once, every workgroup keeping its correspondences in REGISTERS + LDS for all iterations.
//
// Per iteration (an all-reduce, every workgroup for itself — nothing is broadcast):
//   1. item math over the resident correspondences, block reduction, the row of sums goes out as write-through (sc1)
//      stores into partials[iteration parity][workgroup];
//   2. the storing wave drains (vmcnt(0)), one lane ARRIVES: a no-return agent-scope add on one of 8 arrival counters
//      (workgroup index mod 8; counters are monotonic for the whole launch, each on a cache line of its own);
//   3. 8 lanes poll the 8 counters (sc1 loads) until all stand at (iteration + 1) x group size — every row of this
//      iteration has then left its writer (hand-off form "sc1 stores + drain + counter / sc1 loads", MI355X_MICROARCH.md);
//   4. EVERY workgroup adds all rows in the same fixed order (sc1 loads, 16 in flight per thread) and runs the same
//      nos_host::LmAdvance* on its own copy of the loop state: identical bits everywhere, so no state has to travel.
// Rows are double buffered by iteration parity: a workgroup can run at most one iteration ahead of the slowest reader,
// because arriving at iteration k + 1 happens after reading the rows of iteration k.
// Compared with round 1's form (one finishing workgroup: tickets with returned values, row sums, LM step, state written
// through, epoch word, everybody polls and re-reads the state) this removes two memory round trips and the state
// broadcast from the critical path of every iteration.
// Every wait is bounded (kClusterTimeoutTicks): a workgroup that waits longer — e.g. because another process holds CUs and
// the grid is not fully resident — raises `abort` and everybody leaves; the host then re-runs the solve with one launch
// per iteration.
constexpr uint32_t kClusterMaxBlocks = 256;
constexpr unsigned long long kClusterTimeoutTicks = 5000000ull;  // 50 ms of the 100 MHz wall clock per iteration
// With the cross-rank exchange inside the launch (stage 3) a wait also covers the peers' skew — processes reach their first
// launch seconds apart (3 ranks of the test suite: more than 2 s) — so it is as long as the launch-per-iteration exchange's.
// After a give-up the ranks enter the fall-back loop up to this far apart; its first exchanges therefore wait four times as
// long (mailbox_skip_rounds_kernel / mailbox_allreduce).
constexpr unsigned long long kClusterMailboxTimeoutTicks = kMailboxTimeoutTicks;

// Control words of one resident launch, zeroed by the host before the launch (hipMemsetAsync on the launch stream).
struct ClusterCtl {
  unsigned int abort;           // 1: a wait timed out, the launch gave up
  unsigned int pad0[31];
  struct alignas(128) Arrival {
    unsigned int count;         // arrivals of the workgroups with index mod 8 == this counter's index, all iterations
    unsigned int pad[31];
  } arrival[8];
};

__device__ __forceinline__ double sc1_load(const double* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void sc1_store(double* p, double v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// state <-> device memory through write-through stores / cache-bypassing loads (element-wise: the struct is plain data)
__device__ __forceinline__ void state_store_sc1(LmDevice* lm, const nos_host::LmState& st) {
  double* d = reinterpret_cast<double*>(&lm->st);
  const double* s = reinterpret_cast<const double*>(&st);
  constexpr int kWords = int(sizeof(nos_host::LmState) / sizeof(double));
  static_assert(sizeof(nos_host::LmState) % sizeof(double) == 0, "LmState must be a whole number of doubles");
#pragma unroll
  for (int k = 0; k < kWords; ++k) sc1_store(d + k, s[k]);
}
__device__ __forceinline__ void state_load_sc1(const LmDevice* lm, nos_host::LmState& st) {
  const double* d = reinterpret_cast<const double*>(&lm->st);
  double* s = reinterpret_cast<double*>(&st);
  constexpr int kWords = int(sizeof(nos_host::LmState) / sizeof(double));
#pragma unroll
  for (int k = 0; k < kWords; ++k) s[k] = sc1_load(d + k);
}

// How many correspondences a lane keeps resident for the whole solve: RI of them in REGISTERS (compile-time unrolled) and
// up to LI more in LDS (dynamic allocation, [slot][field][lane] so that lanes read consecutive addresses).  One
// 512-thread workgroup per CU → two waves per SIMD → 256 VGPRs per lane and ≈ 150 KB of the CU's 160 KB LDS:
//   NDT fp64 (resident form 96 B / correspondence): 3 + 3 → 6 per lane → 786 432 correspondences on 256 CUs
//   NDT fp32 (60 B, S kept)                        : 3 + 4 → 7         → 917 504
//   reprojection fp64 (40 B)          : 9 + 7 → 16        → 2 097 152  (BASELINE.json configs[2]: 2 M)
//   reprojection fp32 (20 B)          : 10 + 14 → 24      → 3 145 728
// i.e. at these sizes an LM iteration touches neither HBM nor the caches: its cost is the item math plus one grid-wide
// hand-off.  The first touch (one pass over the dataset) is paid once per solve.
// What a RESIDENT NDT correspondence consists of: the solvers only ever need A = SᵀS of the sqrt-information (with
// J = [S | S M]: s = eᵀAe, g = w [Ae ; MᵀAe], H = w [A, AM ; ·, MᵀAM] — Ndt6Problem::item_A / Ndt3Problem::item_A), so a
// correspondence that stays on chip for the whole solve is converted ONCE, at first touch, from {p, mu, S (9)} to
// {p, mu, A (6)}: 12 values instead of 15 per item (more items fit) and ≈ 35 % fewer instructions per item and iteration
// (fp64: 233 → ≈ 150).  Streamed data keeps the 15 planes: it is read
// once per iteration, the conversion would cost more than it saves.
// fp64 only: the fp32 item function already works from A and measured SLOWER through item_A (900 000: 8.57 → 9.24 µs).
template <int FIELDS, size_t ELEM>
constexpr int resident_fields() {
  return (FIELDS == 15 && ELEM == 8) ? 12 : FIELDS;
}
template <int FIELDS, int ELEM>
struct ResidentShape;
template <>
struct ResidentShape<15, 8> { static constexpr int RI = 3, LI = 3; };
template <>
struct ResidentShape<15, 4> { static constexpr int RI = 3, LI = 4; };
template <>
struct ResidentShape<5, 8> { static constexpr int RI = 9, LI = 7; };
template <>
struct ResidentShape<5, 4> { static constexpr int RI = 10, LI = 14; };

// SI > 0 selects the STREAMING form of the same kernel (instantiated with RI = LI = 0): the data set does not fit the
// register files and LDS of the chip, so every LM iteration streams it from HBM again, in chunks of BLOCK * SI
// correspondences taken grid-stride exactly like assemble_kernel does — but the loop still lives in ONE launch: no kernel
// boundary, no launch prologue and no ticket + last-block reduce per iteration (≈ 5 µs of every iteration at 10 M), the
// tagged all-reduce instead, and the first chunk of iteration k + 1 is already in flight while iteration k is being
// reduced and stepped (it does not depend on the pose).  `items_per_lane` then carries the number of chunks.  SPF: the
// next chunk's loads are issued before the current chunk is evaluated (fp32), NT: non-temporal loads.
template <typename Problem, typename T, int BLOCK, int RI, int LI, int SI = 0, bool SPF = false, bool NT = false>
__global__ __launch_bounds__(BLOCK) void solve_cluster_kernel(TiledLayout L, typename Problem::Params P,
                                                             double* __restrict__ partials, LmDevice* lm, ClusterCtl* ctl,
                                                             double* __restrict__ cost_history, int history_capacity,
                                                             double* entry_host, unsigned long long* seq_host,
                                                             unsigned long long seq, uint32_t items_per_lane,
                                                             const Mailbox* mail = nullptr) {
  // mail != nullptr (device-memory mailbox communicator, one process per GPU): the sums of every iteration are exchanged with
  // the other ranks INSIDE this launch — a third stage behind the two of the tagged all-reduce (below)
  constexpr int kF = Problem::kFields;
  constexpr int kOut = Problem::kOut;
  constexpr int kCols = 32;
  constexpr int kSlices = BLOCK / kCols;
  const T* __restrict__ base = static_cast<const T*>(L.base);
  constexpr bool kAForm = kF == 15 && sizeof(T) == 8 && SI == 0;     // resident fp64 NDT items hold A = SᵀS (6) instead of S (9)
  constexpr int kRF = kAForm ? resident_fields<kF, sizeof(T)>() : kF;  // values per resident item
  extern __shared__ __align__(16) unsigned char resident_raw[];  // [items_per_lane - RI][kRF][BLOCK] of T
  T* resident = reinterpret_cast<T*>(resident_raw);
  __shared__ int s_flag;  // 0 go on, 1 loop finished, 2 abort
  __shared__ int s_fast;  // 1 once every group has been seen to sit on one XCD: stage-1 units then stay in that XCD's L2
  __shared__ double red[kSlices][kCols];
  __shared__ double s_tot[kLmTotDoubles(kOut)];
  __shared__ double s_lmd_raw[(sizeof(LmDevice) + 7) / 8];  // this workgroup's copy of the loop state and settings
  LmDevice& s_lmd = *reinterpret_cast<LmDevice*>(s_lmd_raw);
  nos_host::LmState& s_state = s_lmd.st;
  [[maybe_unused]] __shared__ double s_local[kCols];        // workgroup 0, multi-rank: this GPU's sums, then the sums over ranks
  [[maybe_unused]] __shared__ unsigned long long s_round0;  // workgroup 0, multi-rank: exchange rounds completed before this launch
  const bool multi = mail != nullptr;                       // grid-uniform

  // This workgroup's correspondences, read ONCE: slot j of lane l is item  block_base + j * BLOCK + l  (a wave reads
  // consecutive items of one field per load).  Slots beyond n are zero records (contribute exactly nothing) and are
  // flagged invalid for the problems that mask.
  const uint32_t J = items_per_lane & 0x7fffffffu;  // grid-uniform, 1 … RI + LI (streaming form: the number of chunks)
  [[maybe_unused]] const bool allow_fast = (items_per_lane >> 31) == 0u;  // bit 31: keep stage 1 of the all-reduce on sc1 stores
  [[maybe_unused]] constexpr int kXccCol = 28;
  [[maybe_unused]] const unsigned int my_xcc = xcc_id();
  const uint64_t block_base = uint64_t(blockIdx.x) * BLOCK * J;
  auto fetch = [&](uint32_t j, T (&dst)[kRF]) -> bool {
    const uint64_t i = block_base + uint64_t(j) * BLOCK + threadIdx.x;
    const bool ok = i < L.n;
    const uint64_t ic = ok ? i : 0;  // clamped address; the value is zeroed below
    const uint64_t off = (ic >> L.tile_shift) * L.tile_stride + (ic & L.tile_mask);
    T xt[kF][1];
#pragma unroll
    for (int f = 0; f < kF; ++f) load_items<T, 1, false>(base + off + uint64_t(f) * L.field_stride, xt[f]);
    if constexpr (kAForm) {
#pragma unroll
      for (int f = 0; f < 6; ++f) dst[f] = ok ? xt[f][0] : T(0);
      int q = 6;
#pragma unroll
      for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = a; b < 3; ++b) {  // A(a, b) = sum over rows k of S(k, a) S(k, b);  a00 a01 a02 a11 a12 a22
          const T v = fma(xt[6 + a][0], xt[6 + b][0], fma(xt[9 + a][0], xt[9 + b][0], xt[12 + a][0] * xt[12 + b][0]));
          dst[q++] = ok ? v : T(0);
        }
    } else {
#pragma unroll
      for (int f = 0; f < kF; ++f) dst[f] = ok ? xt[f][0] : T(0);
    }
    return ok;
  };
  // one resident item → the sums (NDT: the A form; reprojection: the item as it is)
  auto evaluate_resident = [&](const T (&xi)[kRF], bool ok, T (&acc_)[kOut]) {
    if constexpr (kAForm) {
      const T p3[3] = {xi[0], xi[1], xi[2]}, mu3[3] = {xi[3], xi[4], xi[5]};
      const T A6[6] = {xi[6], xi[7], xi[8], xi[9], xi[10], xi[11]};
      (void)ok;  // pads are all-zero records: they contribute exactly nothing
      Problem::item_A(p3, mu3, A6, P, acc_);
    } else {
      T xf[kF];
#pragma unroll
      for (int f = 0; f < kF; ++f) xf[f] = xi[f < kRF ? f : 0];
      Problem::item(xf, P, ok, acc_);
    }
  };
  T x[RI > 0 ? RI : 1][kRF];
  bool valid[RI > 0 ? RI : 1];
  static_assert(SI == 0 || (RI == 0 && LI == 0), "the streaming form keeps nothing resident");
  // streaming form: the chunk being evaluated next (the first one of every iteration is fetched ahead of time)
  [[maybe_unused]] T xs[kF][SI > 0 ? SI : 1];
  [[maybe_unused]] uint64_t xs_i0 = 0;
  [[maybe_unused]] auto fetch_chunk = [&](uint32_t c, T (&dst)[kF][SI > 0 ? SI : 1]) -> uint64_t {
    const uint64_t i0 = uint64_t(c) * (uint64_t(BLOCK) * (SI > 0 ? SI : 1)) + uint64_t(threadIdx.x) * (SI > 0 ? SI : 1);
    const uint64_t off = (i0 >> L.tile_shift) * L.tile_stride + (i0 & L.tile_mask);
#pragma unroll
    for (int f = 0; f < kF; ++f) load_items<T, (SI > 0 ? SI : 1), NT>(base + off + uint64_t(f) * L.field_stride, dst[f]);
    return i0;
  };
  if constexpr (SI > 0) {
    if (blockIdx.x < J) xs_i0 = fetch_chunk(blockIdx.x, xs);
  }
#pragma unroll
  for (int j = 0; j < RI; ++j) {
    valid[j] = false;
    if (uint32_t(j) < J) {
      valid[j] = fetch(uint32_t(j), x[j]);
    } else {
#pragma unroll
      for (int f = 0; f < kRF; ++f) x[j][f] = T(0);
    }
  }
  if constexpr (LI > 0) {
    for (uint32_t j = RI; j < J; ++j) {
      T xi[kRF];
      (void)fetch(j, xi);
#pragma unroll
      for (int f = 0; f < kRF; ++f) resident[(size_t(j - RI) * kRF + f) * BLOCK + threadIdx.x] = xi[f];
    }
  }
  NOS_PROBE(if (threadIdx.x < 4) s_step_cycles[threadIdx.x] = 0ull;)
  if (threadIdx.x == 0) {
    s_lmd.settings = lm->settings;  // constant during the launch
    s_state = lm->st;  // written by lm_init_kernel before this launch
    s_flag = s_state.done != 0 ? 1 : 0;
    s_fast = 0;
    // a launch that finds `abort` raised (the test hook raises it beforehand) gives up at once, like one whose wait timed out
    if (__hip_atomic_load(&ctl->abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) s_flag = 2;
    if (multi && blockIdx.x == 0) s_round0 = *mail->round;
  }
  __syncthreads();
  [[maybe_unused]] const unsigned int group = blockIdx.x & 7u;
  const unsigned int n_groups = gridDim.x < 8u ? gridDim.x : 8u;
  unsigned int it = 0;
  int executed = 0;
  NOS_PROBE(unsigned long long tq[6] = {0, 0, 0, 0, 0, 0}, tp = 0;)
#define NOS_RES_STAMP(slot_) NOS_PROBE({ const unsigned long long now_ = wall_clock64(); tq[slot_] += now_ - tp; tp = now_; })
  while (s_flag == 0) {
    NOS_PROBE(tp = wall_clock64();)
    // pose of this iteration from LDS → scalar registers
    if constexpr (kOut == 28) {
#pragma unroll
      for (int k = 0; k < 9; ++k) P.R[k] = T(uniform_load(&s_state.R[k]));
#pragma unroll
      for (int k = 0; k < 3; ++k) P.t[k] = T(uniform_load(&s_state.t[k]));
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k) P.R2[k] = T(uniform_load(&s_state.R[k]));
#pragma unroll
      for (int k = 0; k < 2; ++k) P.t2[k] = T(uniform_load(&s_state.t[k]));
    }
    T acc[kOut];
#pragma unroll
    for (int k = 0; k < kOut; ++k) acc[k] = T(0);
    if constexpr (SI > 0) {
      const uint32_t n_chunks = J;
      auto evaluate = [&](const T (&xc)[kF][SI > 0 ? SI : 1], uint64_t i0) {
#pragma unroll
        for (int it = 0; it < (SI > 0 ? SI : 1); ++it) {
          T xi[kF];
#pragma unroll
          for (int f = 0; f < kF; ++f) xi[f] = xc[f][it];
          Problem::item(xi, P, (i0 + it) < L.n, acc);
        }
      };
      uint32_t c = blockIdx.x;
      if constexpr (SPF) {
        for (; c < n_chunks; c += gridDim.x) {
          T xb[kF][SI > 0 ? SI : 1];
          uint64_t i1 = 0;
          const uint32_t cn = c + gridDim.x;
          if (cn < n_chunks) i1 = fetch_chunk(cn, xb);
          evaluate(xs, xs_i0);
#pragma unroll
          for (int f = 0; f < kF; ++f)
#pragma unroll
            for (int it = 0; it < (SI > 0 ? SI : 1); ++it) xs[f][it] = xb[f][it];
          xs_i0 = i1;
        }
      } else {
        while (c < n_chunks) {
          __builtin_amdgcn_sched_barrier(0);  // all loads of a chunk before any of its math (see assemble_kernel)
          evaluate(xs, xs_i0);
          c += gridDim.x;
          if (c < n_chunks) xs_i0 = fetch_chunk(c, xs);
        }
      }
      // the first chunk of the NEXT iteration: in flight during the all-reduce and the step below
      // (every wave does, the polling ones too: letting only the other half prefetch measured 2 % slower at 10 M)
      if (blockIdx.x < n_chunks) xs_i0 = fetch_chunk(blockIdx.x, xs);
    } else
    // (fp64 only: the fp32 kernels spill when their items are interleaved)
    if (sizeof(T) == 8 && J >= uint32_t(RI)) {  // grid-uniform; one straight-line block, so the scheduler can interleave the items
#pragma unroll
      for (int j = 0; j < RI; ++j) evaluate_resident(x[j], valid[j], acc);
    } else {
#pragma unroll
      for (int j = 0; j < RI; ++j)
        if (uint32_t(j) < J) evaluate_resident(x[j], valid[j], acc);
    }
    if constexpr (LI > 0) {
      // (fetching item j + 1 from LDS before item j is evaluated was tried and is SLOWER: reprojection 2 M 14.4 -> 15.4 us
      //  per iteration, profiles/r03_ab_resident.txt — the second buffer costs the register items their interleaving)
      for (uint32_t j = RI; j < J; ++j) {
        T xi[kRF];
#pragma unroll
        for (int f = 0; f < kRF; ++f) xi[f] = resident[(size_t(j - RI) * kRF + f) * BLOCK + threadIdx.x];
        evaluate_resident(xi, (block_base + uint64_t(j) * BLOCK + threadIdx.x) < L.n, acc);
      }
    }
    double dacc[kOut];
#pragma unroll
    for (int k = 0; k < kOut; ++k) dacc[k] = double(acc[k]);
    NOS_RES_STAMP(0)  // item math
    {
      // ---- tagged two-stage all-reduce (round 2, second form): every sum travels as a 16-byte {value, iteration} unit.
      //   stage 1: each workgroup publishes its 28 block sums; the LEADER of its group (workgroups 0..7 lead the groups
      //            "index mod 8") spins on the units of its ≤ 32 members, adds them in member order, publishes 28 group sums;
      //   stage 2: every workgroup spins on the 8 x 28 group units and adds them in group order.
      // No counters, no drain between data and flag, two memory round trips on the critical path, ≈ 1 MB of polling
      // traffic per iteration chip-wide instead of the 14.7 MB of "everybody reads every row".  Block rows need no double
      // buffering (a workgroup publishes iteration k + 1 only after all group sums of k, i.e. after every leader has read
      // the rows of k); group rows are double buffered by parity (a leader can run one iteration ahead of a reader in
      // another group, not two).
      TaggedUnit* const block_units = reinterpret_cast<TaggedUnit*>(partials);                       // [blocks][32]
      TaggedUnit* const group_units = block_units + size_t(kClusterMaxBlocks) * 32;                 // [2][8][32]
      // the tag is unique across launches too (the host's sequence number of this launch in the upper bits): no memset
      const unsigned long long tag = (seq << 24) | ((unsigned long long)it + 1ull);
      const double mine = block_reduce_value<kOut, BLOCK>(dacc);
      // Stage 1 stays inside an XCD when the placement allows it.  HIP promises nothing about which XCD a workgroup lands on
      // (observed: round-robin, so the members of group "index mod 8" share one), so iteration 0 goes the placement-independent
      // way (sc1 stores) and carries every workgroup's XCC id in unit 28; each leader counts the members that are NOT on its
      // own XCD, the counts travel with the group sums, and only if all eight are zero do the following iterations use plain
      // stage-1 stores (line kept in the shared L2: 2.9 -> 2.3 µs for both stages).  Stage 2 is cross-XCD by nature: sc1.
      const bool probe = it == 0u && allow_fast;  // block-uniform
      if (threadIdx.x < kOut) {
        if (s_fast != 0)
          tagged_store_plain(block_units + size_t(blockIdx.x) * 32 + threadIdx.x, mine, tag);
        else
          tagged_store(block_units + size_t(blockIdx.x) * 32 + threadIdx.x, mine, tag);
      } else if (probe && threadIdx.x == kXccCol) {
        tagged_store(block_units + size_t(blockIdx.x) * 32 + kXccCol, double(my_xcc), tag);
      }
      NOS_RES_STAMP(1)  // block reduce + units issued
      // (several ranks: a wait inside this GPU also covers the time the slowest peer needs to get here — the exchange's bound)
      const unsigned long long deadline = wall_clock64() + (multi ? kClusterMailboxTimeoutTicks : kClusterTimeoutTicks);
      // bounded spin on one unit; returns false when the launch is being abandoned
      auto await = [&](const TaggedUnit* u, double* value) -> bool {
        unsigned int polls = 0;
        for (;;) {
          const TaggedUnit got = tagged_load(u);
          if (got.seq == tag) {
            *value = got.value;
            return true;
          }
          // (no back-off between polls: s_sleep 4 / 16 measured slower at 100 k — 6.2 → 6.3 / 6.8 µs — and no help at 10 M)
          if ((++polls & 63u) == 0u &&
              (__hip_atomic_load(&ctl->abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u || wall_clock64() > deadline)) {
            __hip_atomic_store(&ctl->abort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_flag = 2;
            return false;
          }
        }
      };
      const int col = threadIdx.x % kCols;    // which sum
      const int slice = threadIdx.x / kCols;  // which member / group
      if (blockIdx.x < n_groups) {            // block-uniform: this workgroup leads group blockIdx.x
        const unsigned int g_size = (gridDim.x - blockIdx.x + 7u) >> 3;
        double gsum = 0.0;
        for (unsigned int m0 = 0; m0 < g_size; m0 += kSlices) {  // ≤ 2 passes of 16 members
          const unsigned int m = m0 + slice;
          double v = 0.0;
          if (m < g_size && (col < kOut || (probe && col == kXccCol))) {
            (void)await(block_units + size_t(blockIdx.x + 8u * m) * 32 + col, &v);
            if (col == kXccCol) v = v == double(my_xcc) ? 0.0 : 1.0;  // a member on another XCD
          }
          red[slice][col] = v;
          __syncthreads();
          if (threadIdx.x < kOut || (probe && threadIdx.x == kXccCol)) {
#pragma unroll
            for (int sl = 0; sl < kSlices; ++sl) gsum += red[sl][threadIdx.x];  // members in index order
          }
          __syncthreads();
        }
        if ((threadIdx.x < kOut || (probe && threadIdx.x == kXccCol)) && s_flag != 2)
          tagged_store(group_units + (size_t(it & 1u) * 8 + blockIdx.x) * 32 + threadIdx.x, gsum, tag);
      }
      if (!multi || blockIdx.x == 0) {  // stage 2: every workgroup (one rank) / workgroup 0 only (several ranks)
        double v = 0.0;
        if (slice < int(n_groups) && (col < kOut || (probe && col == kXccCol)) && s_flag != 2)
          (void)await(group_units + (size_t(it & 1u) * 8 + slice) * 32 + col, &v);
        if (slice < 8) red[slice][col] = v;
      }
      __syncthreads();
      if (multi) {
        // ---- stage 3, several ranks (one process per GPU, device-memory mailbox): workgroup 0 holds this GPU's sums after
        // stage 2; it PUSHES them into every peer's fine-grained buffer (its own included), waits until every rank's sums
        // of this round have landed in its own buffer, adds them in rank order — identical bits on every rank — and hands
        // the result to the other workgroups as 28 tagged units, which they await instead of the eight group rows.
        // Across the fabric a value travels as two 8-byte granules {round tag (32) | half of the double (32)}: 8-byte
        // system-scope stores are single transactions on every path, so a granule is either the old or the new one and
        // needs no separate flag (cdna_hip_programming.md Guideline 16, R2).  Slots are double buffered by round parity
        // (a rank can be one round ahead of a peer that is still reading, not two).  The wait is bounded like the
        // launch-per-iteration exchange's (kClusterMailboxTimeoutTicks, above); a rank that has to give up says so to its peers through
        // granule 63 of its slot, so that they give up with it instead of waiting for sums that will not come.
        TaggedUnit* const global_units = group_units + size_t(2) * 8 * 32;  // [2][32]
        if (blockIdx.x == 0) {
          if (threadIdx.x < kCols && s_flag != 2) {
            double tot = 0.0;
            if (threadIdx.x < kOut || (probe && threadIdx.x == kXccCol))
              for (unsigned int g = 0; g < n_groups; ++g) tot += red[g][threadIdx.x];  // groups in index order
            s_local[threadIdx.x] = tot;
          }
          __syncthreads();
          if (threadIdx.x < kWave) {
            const Mailbox mb = *mail;
            const unsigned long long round = s_round0 + (unsigned long long)it + 1ull;
            const unsigned int xtag = (unsigned int)(round % 0xFFFFFFFFull) + 1u;  // never 0: fresh buffers are all zero
            const size_t parity = size_t(round & 1ull);
            const size_t units_base = size_t(mb.n_ranks) * 2 * kMailSlotDoubles;  // the granule slots follow the round-protocol slots
            const int lane = int(threadIdx.x);
            const bool giving_up = s_flag == 2;  // wave-uniform (LDS word)
            unsigned long long* const own = reinterpret_cast<unsigned long long*>(mb.peers[mb.rank] + units_base);
            if (giving_up) {
              if (lane < mb.n_ranks)
                __hip_atomic_store(reinterpret_cast<unsigned long long*>(mb.peers[lane] + units_base) +
                                       (size_t(mb.rank) * 2 + parity) * kMailSlotDoubles + 63,
                                   (unsigned long long)xtag << 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            } else {
              unsigned long long gran = 0ull;
              if (lane < 2 * kOut) {
                const unsigned long long bits = (unsigned long long)__double_as_longlong(s_local[lane >> 1]);
                gran = ((unsigned long long)xtag << 32) | ((lane & 1) ? (bits >> 32) : (bits & 0xFFFFFFFFull));
                for (int p = 0; p < mb.n_ranks; ++p)
                  __hip_atomic_store(reinterpret_cast<unsigned long long*>(mb.peers[p] + units_base) +
                                         (size_t(mb.rank) * 2 + parity) * kMailSlotDoubles + lane,
                                     gran, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
              }
              double sum = 0.0;
              bool lost = false;
              const unsigned long long give_up_at = wall_clock64() + kClusterMailboxTimeoutTicks;
              for (int r = 0; r < mb.n_ranks && !lost; ++r) {
                const unsigned long long* slot_r = own + (size_t(r) * 2 + parity) * kMailSlotDoubles;
                unsigned long long got = 0ull;
                unsigned int polls = 0;
                for (;;) {
                  if (lane < 2 * kOut) got = __hip_atomic_load(slot_r + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                  const bool here = lane >= 2 * kOut || (unsigned int)(got >> 32) == xtag;
                  if (__ballot(!here) == 0ull) break;  // wave-uniform
                  if ((++polls & 15u) == 0u) {
                    const unsigned long long peer_gave_up =
                        __hip_atomic_load(slot_r + 63, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    if ((unsigned int)(peer_gave_up >> 32) == xtag || wall_clock64() > give_up_at ||
                        __hip_atomic_load(&ctl->abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
                      lost = true;
                      break;
                    }
                  }
                  __builtin_amdgcn_s_sleep(1);
                }
                const unsigned int half = (unsigned int)(got & 0xFFFFFFFFull);
                const unsigned int other = (unsigned int)__shfl_xor(int(half), 1, kWave);
                const double value = __longlong_as_double((long long)(((unsigned long long)((lane & 1) ? half : other) << 32) |
                                                                      (unsigned long long)((lane & 1) ? other : half)));
                sum += value;  // rank order: the same additions on every rank (even lanes carry sum number lane / 2)
              }
              if (lost) {
                if (lane == 0) {
                  __hip_atomic_store(&ctl->abort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                  s_flag = 2;
                }
                if (lane < mb.n_ranks)  // tell the peers that wait for this rank's later rounds
                  __hip_atomic_store(reinterpret_cast<unsigned long long*>(mb.peers[lane] + units_base) +
                                         (size_t(mb.rank) * 2 + parity) * kMailSlotDoubles + 63,
                                     (unsigned long long)xtag << 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
              } else if (lane < 2 * kOut && (lane & 1) == 0) {
                s_local[lane >> 1] = sum;
              }
            }
          }
          __syncthreads();
          if (s_flag != 2) {
            if (threadIdx.x < kOut || (probe && threadIdx.x == kXccCol))
              tagged_store(global_units + size_t(it & 1u) * 32 + threadIdx.x, s_local[threadIdx.x], tag);
          }
        }
        {
          double v = 0.0;
          if (threadIdx.x < kCols && (col < kOut || (probe && col == kXccCol)) && s_flag != 2)
            (void)await(global_units + size_t(it & 1u) * 32 + col, &v);
          if (threadIdx.x < kCols) red[0][col] = v;
        }
        __syncthreads();
      }
      NOS_RES_STAMP(2)  // both stages arrived
      if (s_flag == 2) break;  // block-uniform
      if (multi) {
        if (threadIdx.x < kOut)
          s_tot[threadIdx.x] = red[0][threadIdx.x];
        else if (probe && threadIdx.x == kXccCol)
          s_fast = red[0][kXccCol] == 0.0 ? 1 : 0;
      } else
      if (threadIdx.x < kOut) {
        double tot = 0.0;
        for (unsigned int g = 0; g < n_groups; ++g) tot += red[g][threadIdx.x];  // groups in index order
        s_tot[threadIdx.x] = tot;
      } else if (probe && threadIdx.x == kXccCol) {
        double strangers = 0.0;
        for (unsigned int g = 0; g < n_groups; ++g) strangers += red[g][kXccCol];
        s_fast = strangers == 0.0 ? 1 : 0;  // the same verdict in every workgroup
      }
      __syncthreads();
    }
    {
      NOS_RES_STAMP(3)  // rows → sums
      if (threadIdx.x == 0) {  // lane 0 of EVERY workgroup advances its own copy of the loop (identical bits everywhere)
        if (blockIdx.x == 0 && cost_history != nullptr && executed < history_capacity)
          __hip_atomic_store(cost_history + executed, s_tot[kOut - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        lm_step_lane<kOut>(lds_ptr(s_tot), lds_ptr(&s_lmd));
        s_flag = s_state.done != 0 ? 1 : 0;
      }
    }
    ++executed;
    ++it;
    __syncthreads();
    NOS_RES_STAMP(4)  // LM step + barrier
  }
  NOS_PROBE(
  if (blockIdx.x == 0 && threadIdx.x == 0 && entry_host != nullptr) {
    for (int k = 0; k < 5; ++k)
      __hip_atomic_store(entry_host + 50 + k, double(tq[k]) / double(executed > 0 ? executed : 1), __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_SYSTEM);
    for (int k = 0; k < 3; ++k)
      __hip_atomic_store(entry_host + 56 + k, double(s_step_cycles[k]) / double(executed > 0 ? executed : 1), __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_SYSTEM);
  }
  )
  // workgroup 0 reports (on abort nobody does: the host sees the missing sequence word)
  if (s_flag == 1 && blockIdx.x == 0) {
    if (threadIdx.x < kOut && entry_host != nullptr && executed > 0)
      __hip_atomic_store(entry_host + kLogOut + threadIdx.x, s_tot[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (threadIdx.x == 0) {
      const nos_host::LmState st = s_state;
      lm->st = st;
      if (multi) *mail->round = s_round0 + (unsigned long long)executed;  // exchange rounds this launch went through
      if (entry_host != nullptr) {
#pragma unroll
        for (int k = 0; k < 9; ++k)
          __hip_atomic_store(entry_host + kLogR + k, st.R[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
#pragma unroll
        for (int k = 0; k < 3; ++k)
          __hip_atomic_store(entry_host + kLogT + k, st.t[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(entry_host + kLogLambda, st.lambda, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(entry_host + kLogPrevCost, st.previous_cost, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(entry_host + kLogCost, st.cost, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(entry_host + kLogIteration, double(st.iteration), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(entry_host + kLogDone, double(st.done), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(entry_host + kLogOk, double(st.ok), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(entry_host + kLogExecuted, double(executed), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
    if (threadIdx.x < kWave) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (threadIdx.x == 0 && seq_host != nullptr) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_store(seq_host, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
  }
}

}  // namespace nos
