// assemble_pass.hpp — the launch-per-pass assemble kernel (nos_*_accumulate, host loop, launch-per-iteration device loop).
// Part of the hand-written gfx950 kernels of the Gauss-Newton normal-equation assembly path; see assemble_kernels.hpp
// (the umbrella header every translation unit includes) for the overview and the reference citations.
#pragma once

#include "assemble_loop.hpp"

namespace nos {

// ---------------------------------------------------------------- the assemble kernel

// Grid-stride over chunks of BLOCK*ITEMS correspondences.  `n_chunks * BLOCK * ITEMS`
// must equal L.n_padded and the tile size must be a multiple of BLOCK*ITEMS (checked on
// the host before launch).
// PREFETCH: 0 = the loads of a chunk, then its math; 3 = the ping-pong form (two named buffers, the loop unrolled twice).
// (Prefetching through register copies — one or two chunks ahead, PREFETCH 1 / 2 of rounds 1-3 — measured slower once the LM
//  step had left the kernel: tools/exp/assemble_variants_r03.hpp keeps those loop bodies, profiles/r03_tune_*.txt the numbers.)
template <typename Problem, typename T, int ITEMS, int BLOCK, int MINW, bool NT, int PREFETCH = 0>
__global__ __launch_bounds__(BLOCK, MINW) void assemble_kernel(TiledLayout L,
                                                              typename Problem::Params P,
                                                              uint32_t n_chunks,
                                                              double* __restrict__ partials,
                                                              FusedFinal fin) {
  constexpr int kF = Problem::kFields;
  constexpr int kOut = Problem::kOut;
  constexpr uint32_t kChunk = BLOCK * ITEMS;
  const T* __restrict__ base = static_cast<const T*>(L.base);

  unsigned long long t_start = 0;
  NOS_PROBE(t_start = wall_clock64();)
  // The pose of this launch (device-resident loop) is awaited only AFTER the loads of the first chunk have been issued:
  // they do not depend on it, and its memory round trip (≈ 1.5 µs at the head of every launch) hides behind them.
  NOS_PROBE(unsigned long long t_prologue = t_start;)

  T acc[kOut];
#pragma unroll
  for (int k = 0; k < kOut; ++k) acc[k] = T(0);
  auto chunk_offset = [&](uint32_t c, uint64_t& i0) {
    i0 = uint64_t(c) * kChunk + uint64_t(threadIdx.x) * ITEMS;
    return (i0 >> L.tile_shift) * L.tile_stride + (i0 & L.tile_mask);
  };
  if constexpr (PREFETCH == 3) {
    // ping-pong: two named buffers, the loop unrolled twice — while buffer A is evaluated the loads into B are in flight and
    // vice versa.  No register copies and NO branch around a load in the steady state (the tail is peeled), so the wait
    // before an evaluation is a counted one for the OLDER group of loads only: a wave always has a chunk in flight.
    T xa[kF][ITEMS], xb[kF][ITEMS];
    uint32_t c = blockIdx.x;
    const uint32_t G = gridDim.x;
    uint64_t ia = 0, ib = 0;
    auto issue = [&](uint32_t cc, T (&dst)[kF][ITEMS], uint64_t& i0) {
      const uint64_t off = chunk_offset(cc, i0);
#pragma unroll
      for (int f = 0; f < kF; ++f) load_items<T, ITEMS, NT>(base + off + uint64_t(f) * L.field_stride, dst[f]);
    };
    auto evaluate = [&](const T (&src)[kF][ITEMS], uint64_t i0) {
#pragma unroll
      for (int it = 0; it < ITEMS; ++it) {
        T xi[kF];
#pragma unroll
        for (int f = 0; f < kF; ++f) xi[f] = src[f][it];
        Problem::item(xi, P, (i0 + it) < L.n, acc);
      }
    };
    const bool any = c < n_chunks;  // block-uniform
    if (any) issue(c, xa, ia);
    if (lm_prologue(fin, P)) return;  // grid-uniform
    if (any) {
      while (uint64_t(c) + 2ull * G < n_chunks) {
        issue(c + G, xb, ib);
        __builtin_amdgcn_sched_barrier(0);
        evaluate(xa, ia);
        issue(c + 2 * G, xa, ia);
        __builtin_amdgcn_sched_barrier(0);
        evaluate(xb, ib);
        c += 2 * G;
      }
      const bool has_b = uint64_t(c) + G < n_chunks;  // block-uniform
      if (has_b) issue(c + G, xb, ib);
      evaluate(xa, ia);
      if (has_b) evaluate(xb, ib);
    }
  } else {
    bool first = true;
    for (uint32_t c = blockIdx.x; c < n_chunks || first; c += gridDim.x) {
      uint64_t i0 = 0;
      T x[kF][ITEMS];
      const bool live = c < n_chunks;  // false only for a block without any chunk, which still has to pass the prologue
      if (live) {
        const uint64_t off = chunk_offset(c, i0);
#pragma unroll
        for (int f = 0; f < kF; ++f) load_items<T, ITEMS, NT>(base + off + uint64_t(f) * L.field_stride, x[f]);
      }
      // All loads of the chunk go out before any of the item math: the machine scheduler otherwise interleaves them
      // with their uses in groups of 4-6 (seen in the ISA), which cuts the bytes a wave keeps in flight and costs ≈ 7 %
      // of the streaming rate.
      __builtin_amdgcn_sched_barrier(0);
      if (first) {  // block-uniform
        first = false;
        if (lm_prologue(fin, P)) return;  // grid-uniform
        NOS_PROBE(t_prologue = wall_clock64() + (unsigned long long)(*reinterpret_cast<const T*>(&P) * T(0));)  // after the pose arrived
        if (!live) break;
      }
#pragma unroll
      for (int it = 0; it < ITEMS; ++it) {
        T xi[kF];
#pragma unroll
        for (int f = 0; f < kF; ++f) xi[f] = x[f][it];
        Problem::item(xi, P, (i0 + it) < L.n, acc);
      }
    }
  }

  double dacc[kOut];
#pragma unroll
  for (int k = 0; k < kOut; ++k) dacc[k] = double(acc[k]);
  NOS_PROBE(const unsigned long long t_loop = wall_clock64() + (unsigned long long)(dacc[0] * 0.0);)  // after the item math
  block_reduce_store<kOut, BLOCK>(dacc, partials + size_t(blockIdx.x) * kOut, fin.write_through != 0);
  NOS_PROBE(
  if (threadIdx.x == 0 && fin.out_host != nullptr && fin.lm != nullptr) {
    // overwritten by every block; the last writer is (almost always) the finishing block
    __hip_atomic_store(fin.out_host + 56, double(t_prologue - t_start), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(fin.out_host + 57, double(t_loop - t_prologue), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(fin.out_host + 58, double(wall_clock64() - t_loop), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  )
  if (fin.counter != nullptr) finish_in_last_block<kOut, BLOCK>(partials, fin, t_start);
}

}  // namespace nos
