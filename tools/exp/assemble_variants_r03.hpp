// assemble_variants_r03.hpp — loop forms and reduction variants that were measured in rounds 1-3 and LOST, moved out of
// csrc/assemble_kernels.hpp in round 4 so that the product kernels read as what runs.  NOT compiled and not included by
// anything: a record of what was tried, next to the stand-alone sweep tools/exp/tune_stream.hip.  The numbers are in
// DESIGN.md's appendix and in profiles/r02_tune_*.txt, profiles/r03_tune_*.txt, profiles/r03_ab_*.txt.
//
// To revive one: paste the block back into the function named in its heading (round 3's tree, commit 7050eb9, has them in
// place and compiles them with -DNOS_ALL_VARIANTS / the macro in the heading).

// --------------------------------------------------------------------------------------------------------------
// PREFETCH == 2 and PREFETCH == 1 loop bodies of assemble_kernel (prefetch through register copies, two / one chunk ahead)
// --------------------------------------------------------------------------------------------------------------
#if 0
  } else if constexpr (PREFETCH == 2) {
    // two chunks ahead: while chunk c is evaluated the loads of c + G and c + 2G are in flight
    T xa[kF][ITEMS], xb[kF][ITEMS];
    uint32_t c = blockIdx.x;
    uint64_t i0 = 0, i1 = 0;
    if (c < n_chunks) {
      const uint64_t off = chunk_offset(c, i0);
#pragma unroll
      for (int f = 0; f < kF; ++f) load_items<T, ITEMS, NT>(base + off + uint64_t(f) * L.field_stride, xa[f]);
    }
    if (c + gridDim.x < n_chunks) {
      const uint64_t off = chunk_offset(c + gridDim.x, i1);
#pragma unroll
      for (int f = 0; f < kF; ++f) load_items<T, ITEMS, NT>(base + off + uint64_t(f) * L.field_stride, xb[f]);
    }
    if (lm_prologue(fin, P)) return;  // grid-uniform
    for (; c < n_chunks; c += gridDim.x) {
      T xc[kF][ITEMS];
      uint64_t i2 = 0;
      const uint32_t cn = c + 2 * gridDim.x;
      if (cn < n_chunks) {
        const uint64_t off = chunk_offset(cn, i2);
#pragma unroll
        for (int f = 0; f < kF; ++f) load_items<T, ITEMS, NT>(base + off + uint64_t(f) * L.field_stride, xc[f]);
      }
#pragma unroll
      for (int it = 0; it < ITEMS; ++it) {
        T xi[kF];
#pragma unroll
        for (int f = 0; f < kF; ++f) xi[f] = xa[f][it];
        Problem::item(xi, P, (i0 + it) < L.n, acc);
      }
#pragma unroll
      for (int f = 0; f < kF; ++f)
#pragma unroll
        for (int it = 0; it < ITEMS; ++it) {
          xa[f][it] = xb[f][it];
          xb[f][it] = xc[f][it];
        }
      i0 = i1;
      i1 = i2;
    }
  } else if constexpr (PREFETCH == 1) {
    // software pipelined: the 15 loads of the NEXT chunk are issued before the current chunk is
    // evaluated, so a wave always has a chunk in flight while it computes
    T xa[kF][ITEMS];
    uint32_t c = blockIdx.x;
    uint64_t i0 = 0;
    if (c < n_chunks) {
      const uint64_t off = chunk_offset(c, i0);
#pragma unroll
      for (int f = 0; f < kF; ++f) load_items<T, ITEMS, NT>(base + off + uint64_t(f) * L.field_stride, xa[f]);
    }
    if (lm_prologue(fin, P)) return;  // grid-uniform
#ifdef NOS_LM_TIMING
    t_prologue = wall_clock64() + (unsigned long long)(*reinterpret_cast<const T*>(&P) * T(0));  // after the pose arrived
#endif
    for (; c < n_chunks; c += gridDim.x) {
      T xb[kF][ITEMS];
      uint64_t i1 = 0;
      const uint32_t cn = c + gridDim.x;
      if (cn < n_chunks) {
        const uint64_t off = chunk_offset(cn, i1);
#pragma unroll
        for (int f = 0; f < kF; ++f) load_items<T, ITEMS, NT>(base + off + uint64_t(f) * L.field_stride, xb[f]);
      }
#pragma unroll
      for (int it = 0; it < ITEMS; ++it) {
        T xi[kF];
#pragma unroll
        for (int f = 0; f < kF; ++f) xi[f] = xa[f][it];
        Problem::item(xi, P, (i0 + it) < L.n, acc);
      }
#pragma unroll
      for (int f = 0; f < kF; ++f)
#pragma unroll
        for (int it = 0; it < ITEMS; ++it) xa[f][it] = xb[f][it];
      i0 = i1;
    }

#endif

// --------------------------------------------------------------------------------------------------------------
// NOS_PACKED_F32: item math on pairs of correspondences (v_pk_* instructions), fp32 only
// --------------------------------------------------------------------------------------------------------------
#if 0
  // fp32 with an even number of correspondences per lane CAN run the item math on pairs (packed v_pk_* instructions) — and
  // is SLOWER that way on gfx950: 0.0942 → 0.1068 ms per launch at 10 M (profiles/r02_tune_f32_packed.txt; the guide's
  // constants table prices one v_pk_fma_f32 above two v_fma_f32).  Kept as a compile-time experiment (-DNOS_PACKED_F32).
#ifdef NOS_PACKED_F32
  constexpr bool kPacked = sizeof(T) == 4 && (ITEMS % 2 == 0) && PREFETCH == 0;
#else
  constexpr bool kPacked = false;
#endif
  [[maybe_unused]] float2_t acc2[kPacked ? kOut : 1];
  if constexpr (kPacked) {
#pragma unroll
    for (int k = 0; k < kOut; ++k) acc2[k] = float2_t{0.0f, 0.0f};
  }

      if constexpr (kPacked) {
#pragma unroll
        for (int it = 0; it < ITEMS; it += 2) {
          float2_t xi[kF];
#pragma unroll
          for (int f = 0; f < kF; ++f) xi[f] = float2_t{x[f][it], x[f][it + 1]};
          const bool valid2[2] = {(i0 + it) < L.n, (i0 + it + 1) < L.n};
          Problem::template item<float2_t>(xi, P, valid2, acc2);
        }
  if constexpr (kPacked) {
#pragma unroll
    for (int k = 0; k < kOut; ++k) acc[k] = acc2[k][0] + acc2[k][1];
  }


#endif

// --------------------------------------------------------------------------------------------------------------
// NOS_SCATTER_BPERMUTE: every exchange of the reduce-scatter butterfly through ds_bpermute (the form before the gfx950 half exchanges)
// --------------------------------------------------------------------------------------------------------------
#if 0
        const double send = upper ? v[j] : v[j + half];
        const double keep = upper ? v[j + half] : v[j];
        v[j] = keep + __shfl_xor(send, mask, kWave);
      }

#endif

// --------------------------------------------------------------------------------------------------------------
// lm_cluster = 3 (PROTO == 0): the counter all-reduce of the one-launch loop — 8 arrival counters, every workgroup reads every row
// --------------------------------------------------------------------------------------------------------------
#if 0
    double* rows = partials + size_t(it & 1u) * size_t(kClusterMaxBlocks) * kOut;  // this iteration's buffer
    block_reduce_store<kOut, BLOCK>(dacc, rows + size_t(blockIdx.x) * kOut, true);  // sc1 row
    NOS_RES_STAMP(1)  // block reduce + row store issued
    if (threadIdx.x < kWave) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the row left through lanes of wave 0
      if (threadIdx.x == 0)  // arrive (no value returned: nothing waits for this add)
        (void)__hip_atomic_fetch_add(&ctl->arrival[group].count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      // wait for everybody's arrival: lane g watches counter g
      const unsigned int g = threadIdx.x & 7u;
      const unsigned int g_size = (gridDim.x - g + 7u) >> 3;
      const unsigned int target = (it + 1u) * g_size;
      const unsigned long long deadline = wall_clock64() + kClusterTimeoutTicks;
      int flag = 0;
      unsigned int polls = 0;
      for (;;) {
        const unsigned int seen = (g < n_groups && threadIdx.x < 8u)
                                      ? __hip_atomic_load(&ctl->arrival[g].count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                      : target;
        if (__ballot(seen < target) == 0ull) break;  // wave-uniform
        // the abort word and the clock are looked at on the first and then every 16th poll
        if ((polls++ & 15u) == 0u &&
            (__hip_atomic_load(&ctl->abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u || wall_clock64() > deadline)) {
          flag = 2;
          break;
        }
      }
      if (threadIdx.x == 0 && flag == 2) {
        __hip_atomic_store(&ctl->abort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_flag = 2;
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");  // no instruction: keeps the row loads below the poll
    }
    __syncthreads();
    NOS_RES_STAMP(2)  // drain + arrive + everybody arrived
    if (s_flag == 2) break;  // block-uniform
    {
      // every workgroup adds the rows of this iteration in the same fixed order
      const int col = threadIdx.x % kCols;
      const int slice = threadIdx.x / kCols;
      constexpr int kUnroll = 16;
      double sum = 0.0;
      if (col < kOut) {
        const double* p = rows + col;
        for (uint32_t r = slice; r < gridDim.x; r += kUnroll * kSlices) {
          double v[kUnroll];
#pragma unroll
          for (int u = 0; u < kUnroll; ++u) {
            const uint32_t rr = r + u * kSlices;
            const double q = sc1_load(p + size_t(rr < gridDim.x ? rr : r) * kOut);
            v[u] = rr < gridDim.x ? q : 0.0;
          }
#pragma unroll
          for (int u = 0; u < kUnroll; ++u) sum += v[u];
        }
      }
      red[slice][col] = sum;
      __syncthreads();
      if (threadIdx.x < kOut) {
        double tot = 0.0;
#pragma unroll
        for (int sl = 0; sl < kSlices; ++sl) tot += red[sl][threadIdx.x];
        s_tot[threadIdx.x] = tot;
      }
      __syncthreads();
    }

#endif

// --------------------------------------------------------------------------------------------------------------
// NOS_NDT6_SFORM_F32: with this macro the fp32 6-DoF item used the literal r = S e, J = S [I | M] form of the fp64 item instead of A = S^T S first
// --------------------------------------------------------------------------------------------------------------
#if 0
// (no code of its own: the macro only disabled the A-form branch of Ndt6Problem::item)

#endif

