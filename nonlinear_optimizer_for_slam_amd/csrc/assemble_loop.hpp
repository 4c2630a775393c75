// assemble_loop.hpp — device-resident LM loop state, the in-launch mailbox exchange, the loop body on one lane, and the in-launch final reduce of a launch-per-pass kernel.
// Part of the hand-written gfx950 kernels of the Gauss-Newton normal-equation assembly path; see assemble_kernels.hpp
// (the umbrella header every translation unit includes) for the overview and the reference citations.
#pragma once

#include "assemble_reduce.hpp"

namespace nos {

// ---------------------------------------------------------------- in-launch final reduce

// When `counter` is set the grid finishes its own reduction: every block publishes its row,
// takes a ticket, and the block that draws the last ticket sums all rows in fixed order and
// writes the result (device pointer and/or host-mapped pinned pointer), then bumps a host
// visible sequence word.  This removes the dependent 1-block kernel and the D2H memcpy from
// the per-iteration critical path.  Hand-off protocol = /opt/skills/guides
// cdna_hip_programming.md Guideline 16: storing wave drains (vmcnt(0)) → one lane
// agent-scope release → asm vmcnt(0) → relaxed agent atomic ticket;  last block: ticket
// value is the "poll", one lane agent-scope acquire → vmcnt(0) → barrier → plain loads.
// Device-resident Levenberg-Marquardt loop (nos_*_solve): the pose lives in device memory, every launch reads it
// from there instead of from its kernel arguments, and the workgroup that finishes the reduction also runs the
// loop body of the reference (damped 6x6 solve, pose update, convergence tests, λ schedule — the same
// nos_host::LmAdvance6 / LmAdvance3 the host loop calls) and leaves the new pose for the next launch.  The host
// only keeps a few launches in flight and watches a log in pinned memory, so consecutive iterations run
// back-to-back on the GPU without a host round trip in between.
struct LmDevice {
  nos_host::LmState st;
  nos_host::LmSettings settings;
};

// Layout (in doubles) of one entry of the pinned host log the loop writes per iteration.
constexpr int kLogOut = 0;        // [0..27] the sums of this iteration
constexpr int kLogR = 32;         // [32..40] pose after the update
constexpr int kLogT = 41;         // [41..43]
constexpr int kLogLambda = 44;
constexpr int kLogPrevCost = 45;
constexpr int kLogCost = 46;
constexpr int kLogIteration = 47;
constexpr int kLogDone = 48;
constexpr int kLogOk = 49;
constexpr int kLogExecuted = 62;  // single-workgroup solve: iterations executed inside the launch
constexpr int kLogEntryDoubles = 64;

// In-kernel all-reduce of the per-GPU sums for one-process-per-GPU runs on one node (nos_ctx_comm_init_shm): a mailbox
// in host memory shared by the ranks (POSIX shm, mapped into every rank's GPU address space).  The workgroup that
// finished its GPU's sums stores them into its own slot followed by a round number (system-scope release), polls the
// round numbers of all ranks (one lane per rank) and adds the slots in rank order — every rank gets identical bits,
// with no extra kernel launch, no RCCL call and no host step in the iteration.  Slots are double buffered by round
// parity: a rank can be at most one round ahead of the slowest reader.  The wait is bounded (kMailboxTimeoutTicks = 8 s of
// the 100 MHz wall clock): on a time-out the launch flags an error instead of spinning for ever.
constexpr int kMailSlotDoubles = 64;                         // one slot: [0..27] sums, [32] round number; 512 bytes
constexpr unsigned long long kMailboxTimeoutTicks = 800000000ull;  // 8 s
struct Mailbox {
  double* base;                 // device address of the shared mailbox: [n_ranks][2][kMailSlotDoubles]; null = no exchange
  double* const* peers;         // device-memory form: peers[r] = rank r's [n_ranks][2][kMailSlotDoubles] buffer (fine-grained
                                // device memory, peers[rank] is local); null = the slots behind `base` (host memory)
  unsigned long long* round;    // device words: [0] rounds completed by this rank (all ranks run the same sequence),
                                // [1] rounds up to this one wait four times as long (set after an abandoned one-launch solve)
  unsigned int* error_host;     // host-mapped word set to 1 when a peer did not arrive in time
  int n_ranks;
  int rank;
};

struct FusedFinal {
  unsigned int* counter;           // device words (top counter at [0], 8 group counters at [32 * (1 + g)]), all 0
                                   // before the launch and reset to 0 by the blocks that complete them
  double* out_dev;                 // device result (may be null)
  double* out_host;                // host-mapped pinned result (may be null)
  unsigned long long* seq_host;    // host-mapped pinned sequence word (may be null)
  unsigned long long seq;          // value stored to *seq_host when the result is complete
  int write_through;               // 1: rows travel as sc1 stores / sc1 loads instead of release / acquire fences
  LmDevice* lm;                    // device-resident loop state: pose source of this launch (null = pose from arguments)
  int lm_step;                     // 1: the finishing workgroup also advances the loop; 0: a separate kernel does
  const Mailbox* mail;             // cross-rank exchange of the sums inside the launch: descriptor in device memory,
                                   // read by the finishing workgroup only (null: none) — kept out of the kernel
                                   // arguments proper because every argument stays in scalar registers through the loop
};

// The exchange itself; called by the first NOUT threads of one workgroup (wave 0 included: NOUT <= 64 and
// n_ranks <= 64) with `tot` = this GPU's sum number threadIdx.x.  Contains block-wide barriers: every thread of the
// block must call it.  Returns the sum over ranks (valid in threads < NOUT).
template <int NOUT>
__device__ __forceinline__ double mailbox_allreduce(const Mailbox& mb, double tot, bool* failed = nullptr) {
  __shared__ unsigned long long s_round, s_patient_until;
  __shared__ int s_failed;
  if (threadIdx.x == 0) {
    s_round = mb.round[0] + 1ull;
    s_patient_until = mb.round[1];
    s_failed = 0;
  }
  __syncthreads();
  const unsigned long long round = s_round;
  const unsigned long long patience = round <= s_patient_until ? 4ull * kMailboxTimeoutTicks : kMailboxTimeoutTicks;
  const size_t parity = size_t(round & 1ull);
  NOS_PROBE(unsigned long long tm0 = wall_clock64(), tm1 = 0, tm2 = 0;)
  // Host-memory form: every rank stores into ITS slot of the one shared segment and polls the others' slots there.
  // Device-memory form: every rank PUSHES its slot into every peer's buffer (remote stores over the fabric; its own buffer
  // included) and polls only its own, local memory — the same slots, the same round numbers, the same rank-order sum.
  const size_t my_slot = (size_t(mb.rank) * 2 + parity) * kMailSlotDoubles;
  const bool pushed = mb.peers != nullptr;
  double* const local = pushed ? mb.peers[mb.rank] : mb.base;  // where this rank polls and sums
  if (threadIdx.x < NOUT) {
    if (pushed) {
      for (int p = 0; p < mb.n_ranks; ++p)
        __hip_atomic_store(mb.peers[p] + my_slot + threadIdx.x, tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    } else {
      __hip_atomic_store(mb.base + my_slot + threadIdx.x, tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
  if (threadIdx.x < kWave) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the sums left through lanes of wave 0
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (pushed) {
      if (int(threadIdx.x) < mb.n_ranks)  // lane p raises this rank's flag in peer p's buffer
        __hip_atomic_store(reinterpret_cast<unsigned long long*>(mb.peers[threadIdx.x] + my_slot + 32), round, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_SYSTEM);
    } else if (threadIdx.x == 0) {
      __hip_atomic_store(reinterpret_cast<unsigned long long*>(mb.base + my_slot + 32), round, __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_SYSTEM);
    }
    NOS_PROBE(tm1 = wall_clock64();)
    if (int(threadIdx.x) < mb.n_ranks) {
      const unsigned long long* flag = reinterpret_cast<const unsigned long long*>(
          local + (size_t(threadIdx.x) * 2 + parity) * kMailSlotDoubles + 32);
      const unsigned long long deadline = wall_clock64() + patience;
      while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != round) {
        if (wall_clock64() > deadline) {  // a peer is missing: report, do not hang
          __hip_atomic_store(mb.error_host, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          s_failed = 1;
          break;
        }
        __builtin_amdgcn_s_sleep(2);
      }
      // No system-scope acquire here: on this part it invalidates the whole L2 (measured 45-110 µs per call); every
      // load of the exchanged values below is itself a system-scope (cache-bypassing) load issued after the barrier.
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    NOS_PROBE(tm2 = wall_clock64();)
  }
  __syncthreads();
  double sum = 0.0;
  if (threadIdx.x < NOUT) {
    for (int r = 0; r < mb.n_ranks; ++r)  // rank order: the same additions on every rank
      sum += __hip_atomic_load(local + (size_t(r) * 2 + parity) * kMailSlotDoubles + threadIdx.x, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_SYSTEM);
  }
  if (threadIdx.x == 0) *mb.round = round;
  NOS_PROBE(
  if (threadIdx.x == 0) {
    const unsigned long long tm3 = wall_clock64() + (unsigned long long)(sum * 0.0);
    mb.base[(size_t(mb.rank) * 2) * kMailSlotDoubles + 40] = double(tm1 - tm0);
    mb.base[(size_t(mb.rank) * 2) * kMailSlotDoubles + 41] = double(tm2 - tm1);
    mb.base[(size_t(mb.rank) * 2) * kMailSlotDoubles + 42] = double(tm3 - tm2);
  }
  )
  if (failed != nullptr) *failed = s_failed != 0;
  return sum;
}

__device__ __forceinline__ double uniform_load(const double* p) {
  // the address is the same for every lane of the grid: keep the value in scalar registers
  const double v = *p;
  const unsigned long long u = __double_as_longlong(v);
  const unsigned int lo = __builtin_amdgcn_readfirstlane((unsigned int)(u & 0xFFFFFFFFull));
  const unsigned int hi = __builtin_amdgcn_readfirstlane((unsigned int)(u >> 32));
  return __longlong_as_double(((unsigned long long)hi << 32) | lo);
}

template <typename T>
__device__ __forceinline__ void set_pose(Ndt6Params<T>& P, const LmDevice* lm) {
#pragma unroll
  for (int k = 0; k < 9; ++k) P.R[k] = T(uniform_load(&lm->st.R[k]));
#pragma unroll
  for (int k = 0; k < 3; ++k) P.t[k] = T(uniform_load(&lm->st.t[k]));
}
template <typename T>
__device__ __forceinline__ void set_pose(ReprojParams<T>& P, const LmDevice* lm) {
#pragma unroll
  for (int k = 0; k < 9; ++k) P.R[k] = T(uniform_load(&lm->st.R[k]));
#pragma unroll
  for (int k = 0; k < 3; ++k) P.t[k] = T(uniform_load(&lm->st.t[k]));
}
template <typename T>
__device__ __forceinline__ void set_pose(Ndt3Params<T>& P, const LmDevice* lm) {
#pragma unroll
  for (int k = 0; k < 4; ++k) P.R2[k] = T(uniform_load(&lm->st.R[k]));
#pragma unroll
  for (int k = 0; k < 2; ++k) P.t2[k] = T(uniform_load(&lm->st.t[k]));
}

// Launch prologue of the device-resident loop.  Returns true if this launch has nothing to do (the loop already
// finished): block 0 then only forwards the sequence word so the host's wait completes.
template <typename Params>
__device__ __forceinline__ bool lm_prologue(const FusedFinal& fin, Params& P) {
  if (fin.lm == nullptr) return false;
  // pose and the done flag are fetched together (one memory round trip at the head of the launch)
  Params Q = P;
  set_pose(Q, fin.lm);
  const int done = __builtin_amdgcn_readfirstlane(*reinterpret_cast<const int*>(&fin.lm->st.done));
  if (done != 0) {
    if (blockIdx.x == 0 && threadIdx.x == 0 && fin.seq_host != nullptr)
      __hip_atomic_store(fin.seq_host, fin.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    return true;
  }
  P = Q;
  return false;
}

// ---------------------------------------------------------------- the loop body on the device: one lane, two real functions
//
// nos_host::LmAdvance6 / LmAdvance3 (csrc/host/nos_lm.hpp; the reference's loop body, MDM/..._analytic_simd.cc:78-102) as
// every device form of the loop runs it (launch per iteration, stand-alone step kernel, single workgroup, one-launch
// resident / streamed).  Round 2 had that function inlined into the kernels; unrolled for instruction-level parallelism it
// wanted ≈ 230 VGPRs (a 6x6 system, its factor, the sums, the state), which pinned every kernel that contained it at the
// 256-register ceiling and made the streaming kernels spill around it.  Now it is ONE NOINLINE function called by lane 0 —
// the damped solve (nos_host::DampedStep itself), a scheduling barrier, then the O(1) rest (pose update, convergence tests,
// λ schedule): 117 VGPRs — so a kernel's own allocation is set by its hot loop and what it keeps alive across the call
// (the streaming kernels: the prefetched first chunk of the next iteration).  A wave-parallel elimination (one matrix
// element per lane, pivots by v_readlane, operands by ds_bpermute) was built and measured first: it needs only ≈ 40
// registers but turns the step into ONE dependent chain — 2.45 µs against the 1.5 µs of the single lane's interleaved
// chains (profiles/r03_lm_step_forms.txt) — so the single lane stayed.
// `tot` (the NOUT sums) and `lmd` (loop state and settings) are LDS.
using LdsDouble = __attribute__((address_space(3))) double;
using LdsLmDevice = __attribute__((address_space(3))) LmDevice;
constexpr int kLmTotDoubles(int n_out) { return n_out; }

__device__ __forceinline__ void wave_sync_lds() {
  // LDS instructions of one wave execute in issue order; this only keeps the compiler from moving accesses across
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

NOS_PROBE(__shared__ unsigned long long s_step_cycles[4];)  // shader-clock cycles of the two halves of the step
#define NOS_STEP_TICK(slot_) \
  NOS_PROBE({ const unsigned long long now_ = clock64(); s_step_cycles[slot_] += now_ - tick_; tick_ = now_; })

// cos(x) and sin(x) / x as power series in v = x^2, for v < 1/256 (|x| < 1/16): six terms each, first omitted term < 1e-23
__device__ __forceinline__ void series_cos_sinc(double v, double* c_out, double* sinc_out) {
  double c = -1.0 / 3628800.0, sc = -1.0 / 39916800.0;
  c = __builtin_fma(c, v, 1.0 / 40320.0), sc = __builtin_fma(sc, v, 1.0 / 362880.0);
  c = __builtin_fma(c, v, -1.0 / 720.0), sc = __builtin_fma(sc, v, -1.0 / 5040.0);
  c = __builtin_fma(c, v, 1.0 / 24.0), sc = __builtin_fma(sc, v, 1.0 / 120.0);
  c = __builtin_fma(c, v, -0.5), sc = __builtin_fma(sc, v, -1.0 / 6.0);
  *c_out = __builtin_fma(c, v, 1.0);
  *sinc_out = __builtin_fma(sc, v, 1.0);
}

// First half: δ = -(H with its diagonal scaled by 1 + λ)^-1 g — nos_host::DampedStep, the host loop's own function
// (right-looking LDLT with reciprocal pivots; on the device the reciprocal is the hardware seed + two Newton steps).
template <int NOUT, int N>
__device__ __forceinline__ bool lm_solve_lane(const LdsDouble* tot, double lambda, double (&step)[N]) {
  double out[NOUT - 1];
#pragma unroll
  for (int k = 0; k < NOUT - 1; ++k) out[k] = tot[k];
  return nos_host::DampedStep<N>(out, lambda, step);
}

// Second half: pose update, the two convergence tests (after the update, as in the reference), λ schedule — the rest of
// nos_host::LmAdvance6 / LmAdvance3 with the transcendental part written for a lone GPU lane, where every fp64 instruction
// costs 8 cycles whatever it computes: the exponential map's two factors are even in θ and are summed as power series for
// θ < 1/8 (no square root, no argument reduction, no division; sincos beyond), normalisation by reciprocal square root (seed
// + two Newton steps), the tests on squared norms.  Within an ulp or two of the host loop's libm calls per operation.
template <int NOUT, int N>
__device__ __forceinline__ void lm_finish_lane(const LdsDouble* tot, LdsLmDevice* lmd, const double (&step)[N], bool solved) {
  NOS_PROBE(unsigned long long tick_ = clock64();)
  const double lambda = lmd->st.lambda, previous_cost = lmd->st.previous_cost, cost = tot[NOUT - 1];
  const int iteration = lmd->st.iteration;
  const int max_iterations = lmd->settings.max_iterations, float_schedule = lmd->settings.float_schedule;
  const double gtol = lmd->settings.gradient_tolerance, ptol = lmd->settings.parameter_tolerance;
  double g2 = 0.0, s2 = 0.0;
#pragma unroll
  for (int r = 0; r < N; ++r) {
    const double gr = tot[N * (N + 1) / 2 + r];
    g2 = __builtin_fma(gr, gr, g2);
    s2 = __builtin_fma(step[r], step[r], s2);
  }
  lmd->st.cost = cost;
  if (!solved) {
    lmd->st.ok = 0;
    lmd->st.done = 1;
    NOS_STEP_TICK(1)
    return;
  }
  if constexpr (N == 6) {
#pragma unroll
    for (int r = 0; r < 3; ++r) lmd->st.t[r] += step[r];
    // ExpQuat (MahalanobisDistanceMinimizer::ComputeQuaternion, MDM/mahalanobis_distance_minimizer.cc:20-33):
    //   theta < 1e-6: (1, w / 2);  else (cos(theta / 2), sin(theta / 2) / theta * w)
    const double wx = step[3], wy = step[4], wz = step[5];
    const double th2 = __builtin_fma(wx, wx, __builtin_fma(wy, wy, wz * wz));
    double dw, kk;
    if (th2 < 1.0 / 64.0) {
      double c, sc;
      series_cos_sinc(0.25 * th2, &c, &sc);
      const bool tiny = !(th2 >= 1e-12);  // theta < 1e-6: the reference's un-normalised small-angle form
      dw = tiny ? 1.0 : c;
      kk = tiny ? 0.5 : 0.5 * sc;
    } else {
      const double inv_th = fast_rsqrt<double>(th2);  // 1 / theta
      double sn, cs;
      sincos(0.5 * (th2 * inv_th), &sn, &cs);
      dw = cs;
      kk = sn * inv_th;
    }
    const double dx = kk * wx, dy = kk * wy, dz = kk * wz;
    // q <- normalize(q (x) dq)
    const double aw = lmd->st.q.w, ax = lmd->st.q.x, ay = lmd->st.q.y, az = lmd->st.q.z;
    const double rw = aw * dw - ax * dx - ay * dy - az * dz;
    const double rx = aw * dx + ax * dw + ay * dz - az * dy;
    const double ry = aw * dy + ay * dw + az * dx - ax * dz;
    const double rz = aw * dz + az * dw + ax * dy - ay * dx;
    const double inv_n = fast_rsqrt<double>((rx * rx + ry * ry) + (rz * rz + rw * rw));
    nos_host::Quat q;
    q.w = rw * inv_n, q.x = rx * inv_n, q.y = ry * inv_n, q.z = rz * inv_n;
    lmd->st.q.w = q.w, lmd->st.q.x = q.x, lmd->st.q.y = q.y, lmd->st.q.z = q.z;
    double R[9];
    nos_host::QuatToMatrix(q, R);
#pragma unroll
    for (int r = 0; r < 9; ++r) lmd->st.R[r] = R[r];
  } else {
    lmd->st.t[0] += step[0];
    lmd->st.t[1] += step[1];
    double c, sn;
    if (step[2] * step[2] < 1.0 / 256.0) {  // |dtheta| < 1/16
      double sc;
      series_cos_sinc(step[2] * step[2], &c, &sc);
      sn = step[2] * sc;
    } else {
      sincos(step[2], &sn, &c);
    }
    const double a = lmd->st.R[0], b = lmd->st.R[1], dd = lmd->st.R[2], e = lmd->st.R[3];
    lmd->st.R[0] = a * c + b * sn;  // linear <- linear * Rot2(dtheta)   (Isometry2d::rotate)
    lmd->st.R[1] = b * c - a * sn;
    lmd->st.R[2] = dd * c + e * sn;
    lmd->st.R[3] = e * c - dd * sn;
  }
  if ((ptol > 0.0 && s2 < ptol * ptol) || (gtol > 0.0 && g2 < gtol * gtol)) {  // |step| < ptol || |g| < gtol
    lmd->st.done = 1;
  } else {
    if (float_schedule) {
      lmd->st.lambda = nos_host::NextLambdaFloat(lambda, cost, previous_cost);
      lmd->st.previous_cost = double(float(cost));
    } else {
      lmd->st.lambda = nos_host::NextLambda(lambda, cost, previous_cost);
      lmd->st.previous_cost = cost;
    }
    lmd->st.iteration = iteration + 1;
    if (iteration + 1 >= max_iterations) lmd->st.done = 1;
  }
  NOS_STEP_TICK(1)
}

// The loop body; call with ONE lane.
template <int NOUT>
__device__ __attribute__((noinline)) void lm_step_lane(const LdsDouble* tot, LdsLmDevice* lmd) {
  constexpr int N = NOUT == 28 ? 6 : 3;
  NOS_PROBE(unsigned long long tick_ = clock64();)
  double step[N];
  const bool solved = lm_solve_lane<NOUT, N>(tot, lmd->st.lambda, step);
  NOS_STEP_TICK(0)
  // the scheduler must not weave the two halves into each other: together they would want ≈ 230 registers again
  __builtin_amdgcn_sched_barrier(0);
  lm_finish_lane<NOUT, N>(tot, lmd, step, solved);
}

// LDS address of a __shared__ object (the generic pointer HIP hands out, narrowed back to its address space)
template <typename T>
__device__ __forceinline__ __attribute__((address_space(3))) T* lds_ptr(T* p) {
  return (__attribute__((address_space(3))) T*)p;
}

#define NOS_LM_STAMP(slot) \
  NOS_PROBE(if (entry_host != nullptr) __hip_atomic_store(entry_host + 50 + (slot), double(wall_clock64()), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM))

// One lane: the state after a step → device memory for the next launch and, if `entry_host` is set, the pinned log entry
// the host is waiting for.
__device__ __forceinline__ void lm_publish(const nos_host::LmState& st, LmDevice* lm, double* entry_host) {
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
  }
}

template <int NOUT, int BLOCK>
__device__ __forceinline__ void finish_in_last_block(const double* partials, const FusedFinal& fin,
                                                     unsigned long long t_start = 0) {
  (void)t_start;
  __shared__ unsigned int s_last;
  constexpr int kCols = 32;
  constexpr int kSlices = BLOCK / kCols;
  __shared__ double red[kSlices][kCols];
  // the row was stored by lanes 0..NOUT-1 of wave 0; thread 0 is in that wave
  if (threadIdx.x < kWave) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  // Two forms of the same hand-off (both listed as valid in the guide):
  //   fences:        plain row stores → drain → agent release → ticket;  last block: agent acquire → plain loads
  //   write-through: sc1 row stores → drain → ticket;                     last block: sc1 loads (bypass L1), no fences
  // The second is used for the one-workgroup-per-CU geometry it was measured for (MI355X_MICROARCH.md,
  // "Hand-offs measured with sc1 loads in place of the acquire", row 1) and saves both fences (≈3 µs).
  const bool wt = fin.write_through != 0;
  if (threadIdx.x == 0) {
    if (!wt) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    // Two-level ticket: one device-scope counter saturates at ≈ 88 arrivals/µs (guide, "dequeue" / "fanin"
    // rows: 256 arrivals ≈ 2.9 µs), so blocks first arrive on one of 8 group counters (group = blockIdx mod 8,
    // i.e. blocks that share an XCD under round-robin placement — used for speed only, any grouping is correct);
    // the last arriver of a group resets it and arrives on the top counter; the last of those finishes.  Every
    // block has released (or written through and drained) its row BEFORE its first arrival, the atomics execute
    // in arrival order at the memory side and each later arrival is issued only after the earlier one returned
    // (data dependence), so when the top ticket reads "last" every row is already out of the writers' L2s.
    unsigned int last = 0u;
    const unsigned int group = blockIdx.x & 7u;
    const unsigned int group_size = (gridDim.x - group + 7u) >> 3;
    const unsigned int n_groups = gridDim.x < 8u ? gridDim.x : 8u;
    unsigned int* group_counter = fin.counter + 32u * (1u + group);  // 128 bytes apart
    const unsigned int t1 = __hip_atomic_fetch_add(group_counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (t1 == group_size - 1u) {
      __hip_atomic_store(group_counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // ready for the next launch
      const unsigned int t2 = __hip_atomic_fetch_add(fin.counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      last = (t2 == n_groups - 1u) ? 1u : 0u;
    }
    if (last && !wt) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    s_last = last;
  }
  __syncthreads();
  if (s_last == 0u) return;  // block-uniform
  NOS_PROBE(
  if (threadIdx.x == 0 && fin.out_host != nullptr && fin.lm != nullptr) {
    __hip_atomic_store(fin.out_host + 50, double(t_start), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(fin.out_host + 51, double(wall_clock64()), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  )
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");  // no instruction: keeps the loads below the ticket
  const int col = threadIdx.x % kCols;
  const int slice = threadIdx.x / kCols;
  // the loop state is requested now so that its latency hides behind the row sums
  const bool step_here = fin.lm != nullptr && fin.lm_step != 0;  // grid-uniform
  __shared__ double s_lmd_raw[(sizeof(LmDevice) + 7) / 8];       // the loop state while wave 0 advances it
  LmDevice& s_lmd = *reinterpret_cast<LmDevice*>(s_lmd_raw);
  constexpr int kLmdWords = int(sizeof(LmDevice) / sizeof(double));
  static_assert(sizeof(LmDevice) % sizeof(double) == 0, "LmDevice must be a whole number of doubles");
  double lmd_pre[kLmdWords];  // thread 0: loop state + settings, in flight while the rows are summed
  if (step_here && threadIdx.x == 0) {
    const double* src = reinterpret_cast<const double*>(fin.lm);
#pragma unroll
    for (int k = 0; k < kLmdWords; ++k) lmd_pre[k] = src[k];
  }
  // Thread (slice, col) adds rows slice, slice + S, slice + 2S, … in that order.  Sixteen row loads are put in flight
  // before the first add: the loop is latency bound (each row comes from another XCD's L2 / memory).
  constexpr int kUnroll = 16;
  double s = 0.0;
  if (col < NOUT) {
    const double* p = partials + col;
    auto sum_rows = [&](auto load) {
      for (uint32_t r = slice; r < gridDim.x; r += kUnroll * kSlices) {
        double v[kUnroll];
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) {
          const uint32_t rr = r + u * kSlices;
          const double x = load(p + size_t(rr < gridDim.x ? rr : r) * NOUT);  // clamped address, value masked below
          v[u] = rr < gridDim.x ? x : 0.0;
        }
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) s += v[u];
      }
    };
    if (wt)
      sum_rows([](const double* q) { return __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); });
    else
      sum_rows([](const double* q) { return *q; });
  }
  red[slice][col] = s;
  __syncthreads();
  __shared__ double s_tot[kLmTotDoubles(NOUT)];
  double tot = 0.0;
  if (threadIdx.x < NOUT) {
#pragma unroll
    for (int sl = 0; sl < kSlices; ++sl) tot += red[sl][threadIdx.x];
  }
  bool exchange_failed = false;
  if (fin.mail != nullptr) {  // grid-uniform branch
    const Mailbox mb = *fin.mail;
    tot = mailbox_allreduce<NOUT>(mb, tot, &exchange_failed);
  }
  if (threadIdx.x < NOUT) {
    if (fin.out_dev != nullptr) fin.out_dev[threadIdx.x] = tot;
    if (fin.out_host != nullptr)
      __hip_atomic_store(fin.out_host + threadIdx.x, tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (step_here) s_tot[threadIdx.x] = tot;
  }
  if (step_here) {
    if (threadIdx.x == 0) {
#pragma unroll
      for (int k = 0; k < kLmdWords; ++k) s_lmd_raw[k] = lmd_pre[k];
      if (exchange_failed) {  // a peer never arrived: stop the loop here, the host reports the error
        s_lmd.st.ok = 0;
        s_lmd.st.done = 1;
        fin.lm->st = s_lmd.st;
        if (fin.out_host != nullptr) {
          __hip_atomic_store(fin.out_host + kLogDone, 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          __hip_atomic_store(fin.out_host + kLogOk, 0.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
      }
    }
    __syncthreads();
    if (threadIdx.x == 0 && !exchange_failed) {
      double* entry_host = fin.out_host;
      (void)entry_host;
      NOS_LM_STAMP(2);
      lm_step_lane<NOUT>(lds_ptr(s_tot), lds_ptr(&s_lmd));
      NOS_LM_STAMP(3);
      lm_publish(s_lmd.st, fin.lm, fin.out_host);
      NOS_LM_STAMP(4);
    }
  }
  if (threadIdx.x < kWave) {
    // results leave through lanes 0..NOUT-1 of wave 0: drain them, then one lane publishes
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (threadIdx.x == 0) {
      __hip_atomic_store(fin.counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // ready for the next launch
      if (fin.seq_host != nullptr) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");  // system scope: results before the sequence word
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_store(fin.seq_host, fin.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
  }
}

}  // namespace nos
