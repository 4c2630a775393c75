// assemble_reduce.hpp — vector loads, the wave / workgroup reductions (reduce-scatter butterfly, gfx950 half exchanges) and the tagged 16-byte units of the in-launch all-reduce.
// Part of the hand-written gfx950 kernels of the Gauss-Newton normal-equation assembly path; see assemble_kernels.hpp
// (the umbrella header every translation unit includes) for the overview and the reference citations.
#pragma once

#include "assemble_items.hpp"

namespace nos {

// ---------------------------------------------------------------- loads

template <typename T, int N>
struct VecOf;
template <>
struct VecOf<double, 1> { using type = double; };
template <>
struct VecOf<double, 2> { using type = double __attribute__((ext_vector_type(2))); };
template <>
struct VecOf<double, 4> { using type = double __attribute__((ext_vector_type(4))); };  // two 16-byte loads
template <>
struct VecOf<double, 8> { using type = double __attribute__((ext_vector_type(8))); };
template <>
struct VecOf<float, 1> { using type = float; };
template <>
struct VecOf<float, 2> { using type = float __attribute__((ext_vector_type(2))); };
template <>
struct VecOf<float, 4> { using type = float __attribute__((ext_vector_type(4))); };
template <>
struct VecOf<float, 8> { using type = float __attribute__((ext_vector_type(8))); };

template <typename T, int N, bool NT>
__device__ __forceinline__ void load_items(const T* p, T (&dst)[N]) {
  using V = typename VecOf<T, N>::type;
  const V* vp = reinterpret_cast<const V*>(p);
  V v;
  if constexpr (NT)
    v = __builtin_nontemporal_load(vp);
  else
    v = *vp;
  if constexpr (N == 1) {
    dst[0] = v;
  } else {
#pragma unroll
    for (int k = 0; k < N; ++k) dst[k] = v[k];
  }
}

// ---------------------------------------------------------------- reduction

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = kWave / 2; off > 0; off >>= 1) v += __shfl_xor(v, off, kWave);
  return v;
}

// Sums NOUT values per lane over the 64 lanes of a wave with a reduce-scatter butterfly: at every step a lane
// keeps one half of its values and trades the other half with its partner (lane ^ 32, ^ 16, …), so the number of
// values halves each time — P + log2(64 / P) cross-lane exchanges in total (P = NOUT rounded up to a power of two)
// instead of 6·NOUT for NOUT independent butterflies.  The cross-lane exchanges (ds_bpermute) are what bounds this
// phase: at 28 values and 8 waves per CU the independent form kept the LDS crossbar busy for ≈ 6-12 µs at the end of
// every launch.  On return lane L holds the wave total of value number  L >> (6 - log2 P)  (lanes that share a value
// number hold the same total).  Fixed order of additions → bit-identical results run to run.
// v_permlane32_swap (rows16 = false): lanes 32-63 of `a` trade places with lanes 0-31 of `b`;
// v_permlane16_swap (rows16 = true): the odd 16-lane rows of `a` trade places with the even rows of `b`.
__device__ __forceinline__ void swap_lane_halves(double& a, double& b, bool rows16) {
  const unsigned long long ab = __double_as_longlong(a), bb = __double_as_longlong(b);
  unsigned int a0 = (unsigned int)ab, a1 = (unsigned int)(ab >> 32), b0 = (unsigned int)bb, b1 = (unsigned int)(bb >> 32);
  if (rows16) {
    const auto r0 = __builtin_amdgcn_permlane16_swap(a0, b0, false, false);
    const auto r1 = __builtin_amdgcn_permlane16_swap(a1, b1, false, false);
    a0 = r0[0], b0 = r0[1], a1 = r1[0], b1 = r1[1];
  } else {
    const auto r0 = __builtin_amdgcn_permlane32_swap(a0, b0, false, false);
    const auto r1 = __builtin_amdgcn_permlane32_swap(a1, b1, false, false);
    a0 = r0[0], b0 = r0[1], a1 = r1[0], b1 = r1[1];
  }
  a = __longlong_as_double(((unsigned long long)a1 << 32) | a0);
  b = __longlong_as_double(((unsigned long long)b1 << 32) | b0);
}

template <int NOUT>
struct WaveScatter {
  static constexpr int kP = NOUT > 16 ? 32 : (NOUT > 8 ? 16 : 8);
  static constexpr int kLog2P = kP == 32 ? 5 : (kP == 16 ? 4 : 3);
  static constexpr int kShift = 6 - kLog2P;  // value number of lane L is L >> kShift
  __device__ static __forceinline__ double run(const double (&acc)[NOUT]) {
    double v[kP];
#pragma unroll
    for (int k = 0; k < kP; ++k) v[k] = k < NOUT ? acc[k] : 0.0;
    const int lane = threadIdx.x & (kWave - 1);
#pragma unroll
    for (int s = 0; s < kLog2P; ++s) {
      const int mask = 32 >> s;
      const int half = kP >> (s + 1);
      const bool upper = (lane & mask) != 0;
#pragma unroll
      for (int j = 0; j < half; ++j) {
        // gfx950 half exchanges: after the swap the two registers hold, in every lane, this lane's kept value and its
        // partner's copy of the same value — 2 swaps + 1 add per exchange instead of 2 ds_bpermute + 4 selects + 1 add,
        // same operands, same bits (the reduce was VALU-issue bound: ≈ 1.8 µs of every resident LM iteration)
        if (mask >= 16) {
          double a = v[j], b = v[j + half];
          swap_lane_halves(a, b, mask == 16);
          v[j] = a + b;
          continue;
        }
        const double send = upper ? v[j] : v[j + half];
        const double keep = upper ? v[j + half] : v[j];
        v[j] = keep + __shfl_xor(send, mask, kWave);
      }
    }
#pragma unroll
    for (int mask = (32 >> kLog2P); mask > 0; mask >>= 1) v[0] += __shfl_xor(v[0], mask, kWave);
    return v[0];
  }
};

// Sums acc[] over the block and writes one row of kOut doubles.  Fixed order:
// reduce-scatter butterfly inside a wave, then waves 0..W-1.
template <int NOUT, int BLOCK>
__device__ __forceinline__ void block_reduce_store(const double (&acc)[NOUT], double* row, bool write_through) {
  constexpr int kWaves = BLOCK / kWave;
  __shared__ double lds[kWaves][NOUT];
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = threadIdx.x / kWave;
  {
    const double s = WaveScatter<NOUT>::run(acc);
    constexpr int kShift = WaveScatter<NOUT>::kShift;
    const int k = lane >> kShift;
    if ((lane & ((1 << kShift) - 1)) == 0 && k < NOUT) lds[wave][k] = s;
  }
  __syncthreads();
  if (threadIdx.x < NOUT) {
    double s = 0.0;
#pragma unroll
    for (int wv = 0; wv < kWaves; ++wv) s += lds[wv][threadIdx.x];
    if (write_through)  // sc1 store: leaves the XCD's L2 at once (hand-off without a release fence)
      __hip_atomic_store(row + threadIdx.x, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else
      row[threadIdx.x] = s;
  }
}

// Same reduction, result returned instead of stored: thread k < NOUT of the block gets block total number k.
template <int NOUT, int BLOCK>
__device__ __forceinline__ double block_reduce_value(const double (&acc)[NOUT]) {
  constexpr int kWaves = BLOCK / kWave;
  __shared__ double lds_v[kWaves][NOUT];
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = threadIdx.x / kWave;
  {
    const double s = WaveScatter<NOUT>::run(acc);
    constexpr int kShift = WaveScatter<NOUT>::kShift;
    const int k = lane >> kShift;
    if ((lane & ((1 << kShift) - 1)) == 0 && k < NOUT) lds_v[wave][k] = s;
  }
  __syncthreads();
  double s = 0.0;
  if (threadIdx.x < NOUT) {
#pragma unroll
    for (int wv = 0; wv < kWaves; ++wv) s += lds_v[wv][threadIdx.x];
  }
  return s;
}

// A value and the sequence number it belongs to in ONE naturally aligned 16-byte unit, written and read with single
// 16-byte cache-bypassing accesses (global_store / global_load_dwordx4 sc1): the reader sees either the old pair or the
// new pair, so "has it arrived" and "what is it" are one memory round trip, and the writer needs no drain between data
// and flag (MI355X_MICROARCH.md lists 16-byte sc1 flag stores / polls among the measured-valid hand-off forms).
struct alignas(16) TaggedUnit {
  double value;
  unsigned long long seq;
};
__device__ __forceinline__ void tagged_store(TaggedUnit* p, double value, unsigned long long seq) {
  using V4 = unsigned int __attribute__((ext_vector_type(4)));
  const unsigned long long bits = __double_as_longlong(value);
  V4 w;
  w[0] = (unsigned int)(bits & 0xFFFFFFFFull);
  w[1] = (unsigned int)(bits >> 32);
  w[2] = (unsigned int)(seq & 0xFFFFFFFFull);
  w[3] = (unsigned int)(seq >> 32);
  // (s_nop 1 inside the string: a 16-byte store reads its data registers up to two states after issue and hipcc pads
  //  nothing around inline asm — without it the next instruction may overwrite them; cdna_hip_programming.md §5.7 item 1)
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(w) : "memory");
}
// The same unit with a PLAIN store: the line stays in the storing CU's XCD L2, where a reader on the SAME XCD finds it with its
// sc1 load (L1-bypassing, L2-served) without the trip through the fabric; a reader on another XCD never sees it.
__device__ __forceinline__ void tagged_store_plain(TaggedUnit* p, double value, unsigned long long seq) {
  using V4 = unsigned int __attribute__((ext_vector_type(4)));
  const unsigned long long bits = __double_as_longlong(value);
  V4 w;
  w[0] = (unsigned int)(bits & 0xFFFFFFFFull);
  w[1] = (unsigned int)(bits >> 32);
  w[2] = (unsigned int)(seq & 0xFFFFFFFFull);
  w[3] = (unsigned int)(seq >> 32);
  asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" ::"v"(p), "v"(w) : "memory");
}
// XCD (XCC) this wave runs on, 0…7
__device__ __forceinline__ unsigned int xcc_id() {
  return __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 0xFu;  // hwreg(HW_REG_XCC_ID, 0, 4)
}
__device__ __forceinline__ TaggedUnit tagged_load(const TaggedUnit* p) {
  using V4 = unsigned int __attribute__((ext_vector_type(4)));
  V4 w;
  asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(w) : "v"(p) : "memory");
  TaggedUnit u;
  u.value = __longlong_as_double(((unsigned long long)w[1] << 32) | w[0]);
  u.seq = ((unsigned long long)w[3] << 32) | w[2];
  return u;
}

}  // namespace nos
