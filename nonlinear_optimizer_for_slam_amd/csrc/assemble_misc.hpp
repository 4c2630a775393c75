// assemble_misc.hpp — loop initialisation, the stand-alone LM step and final-reduce kernels (RCCL path, NOS_FUSED=0), ingestion kernels.
// Part of the hand-written gfx950 kernels of the Gauss-Newton normal-equation assembly path; see assemble_kernels.hpp
// (the umbrella header every translation unit includes) for the overview and the reference citations.
#pragma once

#include "assemble_indexed.hpp"

namespace nos {

// Device-resident loop: initial state (one lane; the arguments travel by value, no copy is needed).
struct LmInitArgs {
  double R[9];
  double t[3];
  nos_host::LmSettings settings;
  int dof;  // 6 or 3
};
__attribute__((unused)) static __global__ void lm_init_kernel(LmDevice* lm, LmInitArgs a) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  nos_host::LmState st;
  if (a.dof == 6)
    nos_host::LmInit6(&st, a.R, a.t, a.settings.max_iterations, a.settings.float_schedule);
  else
    nos_host::LmInit3(&st, a.R, a.t, a.settings.max_iterations, a.settings.float_schedule);
  lm->st = st;
  lm->settings = a.settings;
}

// Device-resident loop, stand-alone step (one wave): used when the sums come out of an RCCL all-reduce (or when
// the in-launch step is switched off).  Reads the sums from `sums`, publishes them and the new state to the pinned
// log entry, then the sequence word.
template <int NOUT>
__global__ __launch_bounds__(64) void lm_step_kernel(const double* __restrict__ sums, LmDevice* lm, double* entry_host,
                                                     unsigned long long* seq_host, unsigned long long seq) {
  __shared__ double s_tot[kLmTotDoubles(NOUT)];
  __shared__ double s_lmd_raw[(sizeof(LmDevice) + 7) / 8];
  LmDevice& s_lmd = *reinterpret_cast<LmDevice*>(s_lmd_raw);
  const int done = *reinterpret_cast<const volatile int*>(&lm->st.done);  // loop finished earlier: forward seq only
  if (done == 0) {
    if (threadIdx.x < NOUT) {
      const double v = sums[threadIdx.x];
      s_tot[threadIdx.x] = v;
      if (entry_host != nullptr)
        __hip_atomic_store(entry_host + kLogOut + threadIdx.x, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    if (threadIdx.x == 0) {
      s_lmd.st = lm->st;
      s_lmd.settings = lm->settings;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      lm_step_lane<NOUT>(lds_ptr(s_tot), lds_ptr(&s_lmd));
      lm_publish(s_lmd.st, lm, entry_host);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (threadIdx.x == 0 && seq_host != nullptr) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __hip_atomic_store(seq_host, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// nos_ctx_comm_allreduce over the mailbox: values[0..count) (device) → sums over ranks, in place.  One workgroup.
__attribute__((unused)) static __global__ __launch_bounds__(64) void mailbox_allreduce_kernel(Mailbox mb, double* values,
                                                                                              int count) {
  const double mine = int(threadIdx.x) < count ? values[threadIdx.x] : 0.0;
  const double sum = mailbox_allreduce<28>(mb, mine);
  if (int(threadIdx.x) < count) values[threadIdx.x] = sum;
}

// Fixed-order sum of the block rows: thread (slice, col) adds rows slice, slice+S, …;
// then the S slice sums are added in slice order.  One block, 1024 threads.
template <int NOUT>
__global__ __launch_bounds__(1024) void final_reduce_kernel(const double* __restrict__ partials,
                                                            uint32_t n_rows,
                                                            double* __restrict__ out) {
  constexpr int kCols = 32;
  constexpr int kSlices = 1024 / kCols;
  __shared__ double lds[kSlices][kCols];
  const int col = threadIdx.x % kCols;
  const int slice = threadIdx.x / kCols;
  double s = 0.0;
  if (col < NOUT)
    for (uint32_t r = slice; r < n_rows; r += kSlices) s += partials[size_t(r) * NOUT + col];
  lds[slice][col] = s;
  __syncthreads();
  if (threadIdx.x < NOUT) {
    double tot = 0.0;
#pragma unroll
    for (int sl = 0; sl < kSlices; ++sl) tot += lds[sl][threadIdx.x];
    out[threadIdx.x] = tot;
  }
}

// ---------------------------------------------------------------- ingestion kernels

// planar source planes (15 or 5 pointers, element type SRC) → tiled layout of DST, zero pads.
struct PlanePtrs {
  const void* p[15];
};

template <typename SRC, typename DST>
__global__ __launch_bounds__(256) void retile_kernel(PlanePtrs src, int n_fields, TiledLayout L,
                                                     DST* __restrict__ dst) {
  const uint64_t i = uint64_t(blockIdx.x) * 256 + threadIdx.x;
  const int f = blockIdx.y;
  if (i >= L.n_padded || f >= n_fields) return;
  const uint64_t off = (i >> L.tile_shift) * L.tile_stride + uint64_t(f) * L.field_stride + (i & L.tile_mask);
  DST v = DST(0);
  if (i < L.n) v = DST(static_cast<const SRC*>(src.p[f])[i]);
  dst[off] = v;
}

// array-of-structures records (double fields at byte offsets) → tiled layout.
// `first` is the index of records[0] inside the dataset; count records are unpacked.
struct FieldOffsets {
  uint32_t off[15];
};

template <typename DST>
__global__ __launch_bounds__(256) void unpack_records_kernel(const unsigned char* __restrict__ records,
                                                             uint64_t stride_bytes, FieldOffsets fo,
                                                             int n_fields, uint64_t first,
                                                             uint64_t count, TiledLayout L,
                                                             DST* __restrict__ dst) {
  const uint64_t j = uint64_t(blockIdx.x) * 256 + threadIdx.x;
  if (j >= count) return;
  const unsigned char* rec = records + j * stride_bytes;
  const uint64_t i = first + j;
  const uint64_t o = (i >> L.tile_shift) * L.tile_stride + (i & L.tile_mask);
  for (int f = 0; f < n_fields; ++f) {
    const double v = *reinterpret_cast<const double*>(rec + fo.off[f]);
    dst[o + uint64_t(f) * L.field_stride] = DST(v);
  }
}

template <typename DST>
__global__ __launch_bounds__(256) void zero_pad_kernel(int n_fields, TiledLayout L, DST* __restrict__ dst) {
  const uint64_t i = L.n + uint64_t(blockIdx.x) * 256 + threadIdx.x;
  if (i >= L.n_padded) return;
  const uint64_t o = (i >> L.tile_shift) * L.tile_stride + (i & L.tile_mask);
  for (int f = 0; f < n_fields; ++f) dst[o + uint64_t(f) * L.field_stride] = DST(0);
}

}  // namespace nos
