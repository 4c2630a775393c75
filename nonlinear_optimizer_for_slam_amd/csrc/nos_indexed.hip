// nos_indexed.hip — voxel-indexed NDT datasets.  Kernel: assemble_indexed_kernel in assemble_kernels.hpp.
#include "nos_internal.hpp"

#include <rocprim/rocprim.hpp>

using namespace nosd;

namespace {

template <typename Problem, typename T, int K>
int launch_indexed_kernel(const nos::IndexedLayout& L, const typename Problem::Params& P, int grid_cap, int num_cus,
                          double* partials, const nos::FusedFinal& fin_in, hipStream_t stream, int* rows_out) {
  // one large workgroup per CU, sized so that the hand-pipelined loop (two chunks of loads in flight next to
  // the chunk being evaluated) stays in registers: fp64 8 waves (<= 256 VGPRs), fp32 12 waves (<= 168)
  constexpr int kBlock = sizeof(T) == 8 ? 512 : 768;
  constexpr int kMinWaves = sizeof(T) == 8 ? 2 : 3;
  const uint64_t n_chunks64 = L.n_padded / kBlock;
  if (n_chunks64 > 0xFFFFFFFFull) return fail(NOS_ERR_UNSUPPORTED, "dataset too large for one shard");
  int grid = int(std::min<uint64_t>(std::max<uint64_t>(n_chunks64, 1), uint64_t(grid_cap)));
  if (grid > kMaxPartialRows) grid = kMaxPartialRows;
  nos::FusedFinal fin = fin_in;
  fin.write_through = (fin.counter != nullptr && grid <= num_cus && fin_in.write_through != 0) ? 1 : 0;  // in: allowed
  hipLaunchKernelGGL((nos::assemble_indexed_kernel<Problem, T, K, kBlock, kMinWaves>), dim3(grid), dim3(kBlock), 0, stream, L,
                     P, uint32_t(n_chunks64), partials, fin);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(NOS_ERR_HIP, "indexed assemble launch failed: %s", hipGetErrorString(e));
  *rows_out = grid;
  return NOS_OK;
}

template <template <typename, int> class ProblemT, typename T, typename ParamsT>
int launch_indexed_by_loss(int loss_kind, int n_slots, const nos::IndexedLayout& L, const ParamsT& P, int grid_cap,
                           int num_cus, double* partials, const nos::FusedFinal& fin, hipStream_t stream, int* rows_out) {
#define NOS_IDX(LOSS_)                                                                                               \
  return n_slots == 1 ? launch_indexed_kernel<ProblemT<T, LOSS_>, T, 1>(L, P, grid_cap, num_cus, partials, fin, stream, \
                                                                       rows_out)                                      \
                      : launch_indexed_kernel<ProblemT<T, LOSS_>, T, 2>(L, P, grid_cap, num_cus, partials, fin, stream, \
                                                                       rows_out)
  switch (loss_kind) {
    case NOS_LOSS_NONE: NOS_IDX(nos::kLossNone);
    case NOS_LOSS_EXPONENTIAL: NOS_IDX(nos::kLossExponential);
    case NOS_LOSS_HUBER: NOS_IDX(nos::kLossHuber);
  }
#undef NOS_IDX
  return fail(NOS_ERR_INVALID_ARGUMENT, "unknown loss kind %d", loss_kind);
}

}  // namespace

int nosd::launch_indexed(const nos_dataset* ds, const Shard& sh, const Request& rq, double* partials,
                         const nos::FusedFinal& fin, hipStream_t stream, int* rows_out) {
  const nos_ctx* ctx = ds->ctx;
  const DeviceSlot& slot = ctx->slots[sh.slot];
  nos::IndexedLayout L{};
  L.points = sh.data;
  L.index = sh.index;
  L.table = sh.table;
  L.n_padded = sh.layout.n_padded;
  const int bpc = ctx->settings.indexed_bpc;
  const int cap = bpc * slot.num_cus;
  if (rq.problem == 6) {
    if (ds->dtype == NOS_F64) {
      nos::Ndt6Params<double> P;
      for (int k = 0; k < 9; ++k) P.R[k] = rq.R[k];
      for (int k = 0; k < 3; ++k) P.t[k] = rq.t[k];
      fill_loss(&rq.loss, P.la, P.lb, P.lc);
      return launch_indexed_by_loss<nos::Ndt6Problem, double>(rq.loss_kind, sh.n_slots, L, P, cap, slot.num_cus, partials, fin,
                                                              stream, rows_out);
    }
    nos::Ndt6Params<float> P;
    for (int k = 0; k < 9; ++k) P.R[k] = float(rq.R[k]);
    for (int k = 0; k < 3; ++k) P.t[k] = float(rq.t[k]);
    fill_loss(&rq.loss, P.la, P.lb, P.lc);
    return launch_indexed_by_loss<nos::Ndt6Problem, float>(rq.loss_kind, sh.n_slots, L, P, cap, slot.num_cus, partials, fin,
                                                           stream, rows_out);
  }
  if (rq.problem == 3) {
    if (ds->dtype == NOS_F64) {
      nos::Ndt3Params<double> P;
      for (int k = 0; k < 4; ++k) P.R2[k] = rq.R[k];
      for (int k = 0; k < 2; ++k) P.t2[k] = rq.t[k];
      fill_loss(&rq.loss, P.la, P.lb, P.lc);
      return launch_indexed_by_loss<nos::Ndt3Problem, double>(rq.loss_kind, sh.n_slots, L, P, cap, slot.num_cus, partials, fin,
                                                              stream, rows_out);
    }
    nos::Ndt3Params<float> P;
    for (int k = 0; k < 4; ++k) P.R2[k] = float(rq.R[k]);
    for (int k = 0; k < 2; ++k) P.t2[k] = float(rq.t[k]);
    fill_loss(&rq.loss, P.la, P.lb, P.lc);
    return launch_indexed_by_loss<nos::Ndt3Problem, float>(rq.loss_kind, sh.n_slots, L, P, cap, slot.num_cus, partials, fin,
                                                           stream, rows_out);
  }
  return fail(NOS_ERR_WRONG_KIND, "voxel-indexed datasets serve the NDT entry points only");
}

namespace {

// Builds the dataset from device-resident inputs: point planes [3][n] (double), index planes [K][n] (int32),
// voxel arrays (double).  Sorts by slot-0 voxel id when asked.  All on the context's stream.
int indexed_from_device(nos_ctx* ctx, size_t n, const double* d_points, int n_slots, const int32_t* d_index,
                        size_t n_voxels, const double* d_means, const double* d_sqrt_infos, int dtype, int sort_by_voxel,
                        nos_dataset** out_ds) {
  if (dtype != NOS_F64 && dtype != NOS_F32) return fail(NOS_ERR_INVALID_ARGUMENT, "unknown dtype %d", dtype);
  if (n >= 0xFFFFFFFFull) return fail(NOS_ERR_UNSUPPORTED, "too many points for one indexed dataset");
  nos_dataset* ds = new (std::nothrow) nos_dataset();
  if (!ds) return fail(NOS_ERR_OUT_OF_MEMORY, "host allocation failed");
  ds->ctx = ctx;
  ds->kind = kKindNdtIndexed;
  ds->dtype = dtype;
  ds->n_fields = 3;
  ds->n = n;
  ds->shards.resize(1);
  Shard& sh = ds->shards[0];
  sh.slot = 0;
  sh.n_slots = n_slots;
  sh.n_voxels = n_voxels;
  const size_t pad = 3072;  // common multiple of the kernel's workgroup sizes (768, 1024)
  const size_t n_padded = std::max<size_t>(((n + pad - 1) / pad) * pad, pad);
  sh.layout.n = n;
  sh.layout.n_padded = n_padded;
  sh.layout.tile_stride = 0;  // the three point planes are plain planar (nos_dataset_download relies on this)
  sh.layout.field_stride = n_padded;
  sh.layout.tile_shift = 40;
  sh.layout.tile_mask = 0xFFFFFFFFu;
  const size_t es = elem_size(dtype);
  sh.bytes = n_padded * (3 * es + sizeof(int32_t) * size_t(n_slots));
  DeviceSlot& slot = ctx->slots[0];
  hipStream_t st = slot.stream;
  DeviceBuffers tmp(&slot);  // arena (pooled slabs) for the temporaries
  uint32_t *keys = nullptr, *keys_sorted = nullptr, *ids = nullptr, *perm = nullptr;
  hipError_t e = hipSetDevice(slot.device);
  // ONE pooled allocation for the point planes, the id planes and the voxel table: scan-to-map builds such a dataset every
  // round, and three hipMalloc / hipFree pairs per round (a hipFree waits for the device) were a tenth of the round
  if (e == hipSuccess) {
    auto up = [](size_t b) { return (b + 255) & ~size_t(255); };
    const size_t b_data = up(n_padded * 3 * es), b_index = up(n_padded * sizeof(int32_t) * size_t(n_slots));
    const size_t b_table = up(std::max<size_t>(n_voxels, 1) * 16 * es);
    void* block = nullptr;
    size_t cap = 0;
    if (pool_alloc(slot, b_data + b_index + b_table, &block, &cap) != NOS_OK) {
      e = hipErrorOutOfMemory;
    } else {
      sh.data = block;
      sh.capacity = cap;
      sh.pooled = true;
      sh.one_block = true;
      sh.index = reinterpret_cast<int32_t*>(static_cast<char*>(block) + b_data);
      sh.table = static_cast<char*>(block) + b_data + b_index;
    }
  }
  if (e == hipSuccess && sort_by_voxel && n > 0) {
    e = tmp.alloc(&keys, n);
    if (e == hipSuccess) e = tmp.alloc(&keys_sorted, n);
    if (e == hipSuccess) e = tmp.alloc(&ids, n);
    if (e == hipSuccess) e = tmp.alloc(&perm, n);
    if (e == hipSuccess) {
      hipLaunchKernelGGL(nos::index_sort_key_kernel, dim3(unsigned((n + 255) / 256)), dim3(256), 0, st, d_index, uint64_t(n),
                         keys, ids);
      e = hipGetLastError();
    }
    size_t tb = 0;
    void* scratch = nullptr;
    if (e == hipSuccess) e = rocprim::radix_sort_pairs(nullptr, tb, keys, keys_sorted, ids, perm, n, 0, 32, st);
    if (e == hipSuccess) e = tmp.alloc_bytes(&scratch, std::max<size_t>(tb, 16));
    if (e == hipSuccess) e = rocprim::radix_sort_pairs(scratch, tb, keys, keys_sorted, ids, perm, n, 0, 32, st);
  }
  if (e == hipSuccess) {
    const dim3 grid(unsigned((n_padded + 255) / 256));
    for (int f = 0; f < 3 && e == hipSuccess; ++f) {
      if (dtype == NOS_F64)
        hipLaunchKernelGGL((nos::gather_plane_kernel<double, double>), grid, dim3(256), 0, st, d_points + size_t(f) * n, perm,
                           uint64_t(n), uint64_t(n_padded), 0.0, static_cast<double*>(sh.data) + size_t(f) * n_padded);
      else
        hipLaunchKernelGGL((nos::gather_plane_kernel<double, float>), grid, dim3(256), 0, st, d_points + size_t(f) * n, perm,
                           uint64_t(n), uint64_t(n_padded), 0.0f, static_cast<float*>(sh.data) + size_t(f) * n_padded);
      e = hipGetLastError();
    }
    for (int k = 0; k < n_slots && e == hipSuccess; ++k) {
      hipLaunchKernelGGL((nos::gather_plane_kernel<int32_t, int32_t>), grid, dim3(256), 0, st, d_index + size_t(k) * n, perm,
                         uint64_t(n), uint64_t(n_padded), int32_t(-1), sh.index + size_t(k) * n_padded);
      e = hipGetLastError();
    }
  }
  if (e == hipSuccess && n_voxels > 0) {
    const dim3 grid(unsigned((n_voxels * 16 + 255) / 256));
    if (dtype == NOS_F64)
      hipLaunchKernelGGL((nos::build_voxel_table_kernel<double>), grid, dim3(256), 0, st, d_means, d_sqrt_infos,
                         uint64_t(n_voxels), static_cast<double*>(sh.table));
    else
      hipLaunchKernelGGL((nos::build_voxel_table_kernel<float>), grid, dim3(256), 0, st, d_means, d_sqrt_infos,
                         uint64_t(n_voxels), static_cast<float*>(sh.table));
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  if (e != hipSuccess) {
    nos_dataset_destroy(ds);
    return fail(e == hipErrorOutOfMemory ? NOS_ERR_OUT_OF_MEMORY : NOS_ERR_HIP, "indexed dataset build failed: %s",
                hipGetErrorString(e));
  }
  *out_ds = ds;
  return NOS_OK;
}

// like match_kernel, but emits voxel ids (positions in the map's cell-ordered arrays) instead of records
__global__ __launch_bounds__(256) void match_index_kernel(nos::MapView map, const double* __restrict__ px,
                                                          const double* __restrict__ py, const double* __restrict__ pz,
                                                          uint64_t n_points, nos::PosePod pose, int max_neighbors,
                                                          int32_t* __restrict__ idx0, int32_t* __restrict__ idx1,
                                                          unsigned long long* __restrict__ n_matches) {
  const uint64_t i = uint64_t(blockIdx.x) * 256 + threadIdx.x;
  int found = 0;
  if (i < n_points) {
    const double x = px[i], y = py[i], z = pz[i];
    const double qx = pose.R[0] * x + pose.R[1] * y + pose.R[2] * z + pose.t[0];
    const double qy = pose.R[3] * x + pose.R[4] * y + pose.R[5] * z + pose.t[1];
    const double qz = pose.R[6] * x + pose.R[7] * y + pose.R[8] * z + pose.t[2];
    nos::TwoNearest best;
    nos::find_two_nearest(map, qx, qy, qz, best);
    const uint32_t (&best_j)[2] = best.j;
    const bool ok0 = best_j[0] != 0xFFFFFFFFu, ok1 = best_j[1] != 0xFFFFFFFFu && max_neighbors > 1;
    idx0[i] = ok0 ? int32_t(best_j[0]) : -1;
    idx1[i] = ok1 ? int32_t(best_j[1]) : -1;
    found = int(ok0) + int(ok1);
  }
  int s = found;
#pragma unroll
  for (int o = nos::kWave / 2; o > 0; o >>= 1) s += __shfl_xor(s, o, nos::kWave);
  if ((threadIdx.x & (nos::kWave - 1)) == 0 && s > 0) atomicAdd(n_matches, (unsigned long long)s);
}

}  // namespace

extern "C" {

int nos_ndt_indexed_dataset_create(nos_ctx* ctx, size_t n_points, const double* const point_planes[3], int n_slots,
                                   const int32_t* const index_planes[], size_t n_voxels, const double* means_xyz,
                                   const double* sqrt_infos, int dtype, int sort_by_voxel, nos_dataset** out_ds) {
  nosd::CtxGuard guard_(ctx);  // one solve / accumulate / create at a time per context
  if (!ctx || !out_ds) return fail(NOS_ERR_INVALID_ARGUMENT, "ctx / out_ds is NULL");
  *out_ds = nullptr;
  if (ctx->slots.size() != 1) return fail(NOS_ERR_UNSUPPORTED, "voxel-indexed datasets need a single-device context");
  if (n_slots < 1 || n_slots > 2) return fail(NOS_ERR_INVALID_ARGUMENT, "n_slots must be 1 or 2");
  if (n_points > 0 && (!point_planes || !index_planes)) return fail(NOS_ERR_INVALID_ARGUMENT, "input planes are NULL");
  if (n_voxels > 0x7FFFFFFFull || (n_voxels > 0 && (!means_xyz || !sqrt_infos))) return fail(NOS_ERR_INVALID_ARGUMENT, "bad voxel table");
  for (int k = 0; k < n_slots && n_points > 0; ++k) {
    if (!index_planes[k]) return fail(NOS_ERR_INVALID_ARGUMENT, "index plane %d is NULL", k);
    for (size_t i = 0; i < n_points; ++i)
      if (index_planes[k][i] >= int64_t(n_voxels)) return fail(NOS_ERR_INVALID_ARGUMENT, "voxel id out of range at point %zu", i);
  }
  DeviceSlot& slot = ctx->slots[0];
  DeviceBuffers buf(&slot);
  double *d_pts = nullptr, *d_means = nullptr, *d_S = nullptr;
  int32_t* d_idx = nullptr;
  hipError_t e = hipSetDevice(slot.device);
  if (e == hipSuccess) e = buf.alloc(&d_pts, n_points * 3);
  if (e == hipSuccess) e = buf.alloc(&d_idx, n_points * size_t(n_slots));
  if (e == hipSuccess) e = buf.alloc(&d_means, n_voxels * 3);
  if (e == hipSuccess) e = buf.alloc(&d_S, n_voxels * 9);
  for (int f = 0; f < 3 && e == hipSuccess && n_points > 0; ++f) {
    if (!point_planes[f]) return fail(NOS_ERR_INVALID_ARGUMENT, "point plane %d is NULL", f);
    e = hipMemcpyAsync(d_pts + size_t(f) * n_points, point_planes[f], n_points * sizeof(double), hipMemcpyHostToDevice, slot.stream);
  }
  for (int k = 0; k < n_slots && e == hipSuccess && n_points > 0; ++k)
    e = hipMemcpyAsync(d_idx + size_t(k) * n_points, index_planes[k], n_points * sizeof(int32_t), hipMemcpyHostToDevice, slot.stream);
  if (e == hipSuccess && n_voxels > 0) e = hipMemcpyAsync(d_means, means_xyz, n_voxels * 3 * sizeof(double), hipMemcpyHostToDevice, slot.stream);
  if (e == hipSuccess && n_voxels > 0) e = hipMemcpyAsync(d_S, sqrt_infos, n_voxels * 9 * sizeof(double), hipMemcpyHostToDevice, slot.stream);
  if (e != hipSuccess) return fail(e == hipErrorOutOfMemory ? NOS_ERR_OUT_OF_MEMORY : NOS_ERR_HIP, "indexed upload failed: %s", hipGetErrorString(e));
  return indexed_from_device(ctx, n_points, d_pts, n_slots, d_idx, n_voxels, d_means, d_S, dtype, sort_by_voxel, out_ds);
}

int nos_ndt_match_indexed(nos_ndt_map* map, nos_scan* scan, const double R[9], const double t[3], int max_neighbors,
                          int dtype, int sort_by_voxel, nos_dataset** out_ds, size_t* n_matches) {
  nosd::CtxGuard guard_(map ? map->ctx : nullptr);  // one solve / accumulate / create at a time per context
  if (!map || !scan || !R || !t || !out_ds) return fail(NOS_ERR_INVALID_ARGUMENT, "NULL argument");
  *out_ds = nullptr;
  if (map->ctx != scan->ctx) return fail(NOS_ERR_INVALID_ARGUMENT, "map and scan belong to different contexts");
  if (max_neighbors < 1 || max_neighbors > 2) return fail(NOS_ERR_UNSUPPORTED, "max_neighbors must be 1 or 2");
  nos_ctx* ctx = map->ctx;
  DeviceSlot& slot = ctx->slots[0];
  DeviceBuffers buf(&slot);
  int32_t* d_idx = nullptr;
  const size_t n = scan->n;
  hipError_t e = hipSetDevice(slot.device);
  if (e == hipSuccess) e = buf.alloc(&d_idx, 2 * n);
  if (e == hipSuccess) e = hipMemsetAsync(map->d_n_matches, 0, sizeof(unsigned long long), slot.stream);
  nos::PosePod pose;
  for (int k = 0; k < 9; ++k) pose.R[k] = R[k];
  for (int k = 0; k < 3; ++k) pose.t[k] = t[k];
  if (e == hipSuccess && n > 0) {
    hipLaunchKernelGGL(match_index_kernel, dim3(unsigned((n + 255) / 256)), dim3(256), 0, slot.stream, map->view, scan->d_planes,
                       scan->d_planes + n, scan->d_planes + 2 * n, uint64_t(n), pose, max_neighbors, d_idx, d_idx + n,
                       map->d_n_matches);
    e = hipGetLastError();
  }
  unsigned long long count = 0;
  if (e == hipSuccess) e = hipMemcpyAsync(&count, map->d_n_matches, sizeof count, hipMemcpyDeviceToHost, slot.stream);
  if (e == hipSuccess) e = hipStreamSynchronize(slot.stream);
  if (e != hipSuccess) return fail(NOS_ERR_HIP, "indexed matching failed: %s", hipGetErrorString(e));
  const int rc = indexed_from_device(ctx, n, scan->d_planes, max_neighbors, d_idx, map->n_voxels, map->d_mean, map->d_sqrt_info,
                                     dtype, sort_by_voxel, out_ds);
  if (rc != NOS_OK) return rc;
  if (n_matches) *n_matches = size_t(count);
  return NOS_OK;
}

}  // extern "C"
