// nos_match.hip — device correspondence matcher, scans, dataset download.
#include "nos_internal.hpp"

#include <rocprim/rocprim.hpp>

using namespace nosd;

namespace {

// ---- matcher tables built on the device (round 4; the host used to sort (cell, voxel) pairs, walk the runs and fill the
// hash table and the dense grid: 50-70 ms at 8·10^5 voxels, behind a download of the voxel statistics and in front of an
// upload of the re-ordered copies).  Same layout as before: records in ascending (cell key, voxel index) order — a stable
// radix sort of (key, index) pairs —, dense offsets = exclusive prefix sums of the per-cell counts, and an open-addressing
// table cell key → (first record, count).  The dense form and the record order are unique, hence bit-identical to the host
// construction; the hash table's slot assignment depends on the order in which colliding cells arrive (atomic compare-and-
// swap), which no lookup can observe: a probe sequence ends at the key or at a free slot either way.

constexpr int64_t kCellLimit = (1 << 20) - 2;

// flags[0] != 0: a valid voxel with a non-finite mean (flags[1] = its index + 1); flags[2] != 0: outside the addressable grid
__global__ __launch_bounds__(256) void map_cell_key_kernel(const double* __restrict__ means, const unsigned char* __restrict__ valid,
                                                           uint32_t n_voxels, double inv_cell, uint64_t* __restrict__ keys,
                                                           uint32_t* __restrict__ idx, unsigned int* __restrict__ flags) {
  const uint32_t v = blockIdx.x * 256 + threadIdx.x;
  if (v >= n_voxels) return;
  idx[v] = v;
  if (valid != nullptr && !valid[v]) {  // `if (!ndt.is_valid) continue;` of the reference's matcher
    keys[v] = nos::kEmptyCell;          // sorts behind every cell
    return;
  }
  const double x = means[3 * size_t(v)], y = means[3 * size_t(v) + 1], z = means[3 * size_t(v) + 2];
  if (!(isfinite(x) && isfinite(y) && isfinite(z))) {
    if (atomicCAS(&flags[0], 0u, 1u) == 0u) flags[1] = v + 1u;
    keys[v] = nos::kEmptyCell;
    return;
  }
  const int64_t ix = int64_t(floor(x * inv_cell)), iy = int64_t(floor(y * inv_cell)), iz = int64_t(floor(z * inv_cell));
  if (llabs(ix) > kCellLimit || llabs(iy) > kCellLimit || llabs(iz) > kCellLimit) {
    if (atomicCAS(&flags[2], 0u, 1u) == 0u) flags[3] = v + 1u;
    keys[v] = nos::kEmptyCell;
    return;
  }
  keys[v] = nos::pack_cell(ix, iy, iz);
}

// records in cell order: mean [V][3], sqrt-information [V][9], candidate records [V][4] = {mean, original id bits}
__global__ __launch_bounds__(256) void map_gather_kernel(const double* __restrict__ means, const double* __restrict__ sqrt_infos,
                                                         const uint32_t* __restrict__ orig, uint32_t n_valid,
                                                         double* __restrict__ mean_sorted, double* __restrict__ s_sorted,
                                                         double* __restrict__ records) {
  const uint32_t j = blockIdx.x * 256 + threadIdx.x;
  if (j >= n_valid) return;
  const uint32_t v = orig[j];
  double m[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    m[k] = means[3 * size_t(v) + k];
    mean_sorted[3 * size_t(j) + k] = m[k];
  }
#pragma unroll
  for (int k = 0; k < 9; ++k) s_sorted[9 * size_t(j) + k] = sqrt_infos[9 * size_t(v) + k];
  if (records != nullptr) {
    records[4 * size_t(j)] = m[0];
    records[4 * size_t(j) + 1] = m[1];
    records[4 * size_t(j) + 2] = m[2];
    records[4 * size_t(j) + 3] = __longlong_as_double((long long)(unsigned long long)v);
  }
}

// bounding box of the occupied cells: box[0..2] = min, box[3..5] = max (biased coordinates, 21 bits each)
__global__ __launch_bounds__(256) void map_cell_box_kernel(const uint64_t* __restrict__ cell_keys, uint32_t n_cells,
                                                           unsigned int* __restrict__ box) {
  const uint32_t c = blockIdx.x * 256 + threadIdx.x;
  unsigned int lo[3] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}, hi[3] = {0u, 0u, 0u};
  if (c < n_cells) {
    const uint64_t key = cell_keys[c];
    lo[0] = hi[0] = unsigned((key >> 42) & 0x1FFFFFull);
    lo[1] = hi[1] = unsigned((key >> 21) & 0x1FFFFFull);
    lo[2] = hi[2] = unsigned(key & 0x1FFFFFull);
  }
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    unsigned a = lo[k], b = hi[k];
#pragma unroll
    for (int o = nos::kWave / 2; o > 0; o >>= 1) {
      a = min(a, unsigned(__shfl_xor(int(a), o, nos::kWave)));
      b = max(b, unsigned(__shfl_xor(int(b), o, nos::kWave)));
    }
    if ((threadIdx.x & (nos::kWave - 1)) == 0 && a != 0xFFFFFFFFu) {
      atomicMin(&box[k], a);
      atomicMax(&box[3 + k], b);
    }
  }
}

// hash table cell key → (first record, count), and the dense grid's per-cell counts (dense_begin[idx + 1] = count)
__global__ __launch_bounds__(256) void map_cell_tables_kernel(const uint64_t* __restrict__ cell_keys,
                                                              const uint32_t* __restrict__ cell_counts,
                                                              const uint32_t* __restrict__ cell_starts, uint32_t n_cells,
                                                              uint32_t table_mask, unsigned long long* __restrict__ tab_key,
                                                              uint32_t* __restrict__ tab_start, uint32_t* __restrict__ tab_count,
                                                              uint32_t* __restrict__ dense_begin, int64_t ox, int64_t oy, int64_t oz,
                                                              int64_t ny, int64_t nz) {
  const uint32_t c = blockIdx.x * 256 + threadIdx.x;
  if (c >= n_cells) return;
  const uint64_t key = cell_keys[c];
  uint32_t h = nos::hash_cell(key) & table_mask;
  for (;;) {  // the table is at most half full: a free slot always turns up
    const unsigned long long seen = atomicCAS(&tab_key[h], (unsigned long long)nos::kEmptyCell, (unsigned long long)key);
    if (seen == nos::kEmptyCell) break;
    h = (h + 1) & table_mask;
  }
  tab_start[h] = cell_starts[c];
  tab_count[h] = cell_counts[c];
  if (dense_begin != nullptr) {
    const int64_t bias = int64_t(1) << 20;
    const int64_t cx = int64_t((key >> 42) & 0x1FFFFFull) - bias, cy = int64_t((key >> 21) & 0x1FFFFFull) - bias,
                  cz = int64_t(key & 0x1FFFFFull) - bias;
    const size_t idx = (size_t(cx - ox) * size_t(ny) + size_t(cy - oy)) * size_t(nz) + size_t(cz - oz);
    dense_begin[idx + 1] = cell_counts[c];
  }
}

// Builds the map object from DEVICE arrays of voxel statistics (means [V][3], sqrt-informations [V][9], valid [V] or null).
int map_create_from_device(nos_ctx* ctx, size_t n_voxels, const double* d_means, const double* d_S, const unsigned char* d_valid,
                           double search_radius_sq, nos_ndt_map** out_map) {
  const double inv_cell = 1.0 / std::sqrt(search_radius_sq);
  DeviceSlot& slot = ctx->slots[0];
  hipStream_t st = slot.stream;
  const uint32_t V = uint32_t(n_voxels);
  nos_ndt_map* map = new (std::nothrow) nos_ndt_map();
  if (!map) return fail(NOS_ERR_OUT_OF_MEMORY, "host allocation failed");
  map->ctx = ctx;
  DeviceBuffers buf(&slot);  // arena (pooled slabs) for the temporaries
  buf.reserve(std::max<size_t>(V, 1) * (3 * sizeof(uint64_t) + 3 * sizeof(uint32_t)) + (size_t(16) << 20));
  uint64_t *keys = nullptr, *keys_sorted = nullptr, *uniq = nullptr;
  uint32_t *idx = nullptr, *run_count = nullptr, *run_start = nullptr, *n_runs = nullptr;
  unsigned int *flags = nullptr, *box = nullptr;
  hipError_t e = hipSetDevice(slot.device);
  const size_t cap = std::max<size_t>(V, 1);
  if (e == hipSuccess) e = buf.alloc(&keys, cap);
  if (e == hipSuccess) e = buf.alloc(&keys_sorted, cap);
  if (e == hipSuccess) e = buf.alloc(&uniq, cap);
  if (e == hipSuccess) e = buf.alloc(&idx, cap);
  if (e == hipSuccess) e = buf.alloc(&run_count, cap);
  if (e == hipSuccess) e = buf.alloc(&run_start, cap);
  if (e == hipSuccess) e = buf.alloc(&n_runs, 1);
  if (e == hipSuccess) e = buf.alloc(&flags, 4);
  if (e == hipSuccess) e = buf.alloc(&box, 6);
  if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&map->d_orig_id), cap * sizeof(uint32_t));
  if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&map->d_n_matches), sizeof(unsigned long long));
  uint32_t runs = 0, n_cells = 0, n_valid = 0;
  unsigned int h_flags[4] = {0, 0, 0, 0}, h_box[6] = {0, 0, 0, 0, 0, 0};
  size_t t_sort = 0, t_rle = 0, t_scan = 0;
  void* tmp = nullptr;
  if (e == hipSuccess && V > 0) {
    const unsigned int box_init[6] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0u, 0u, 0u};
    e = hipMemsetAsync(flags, 0, 4 * sizeof(unsigned int), st);
    if (e == hipSuccess) e = hipMemcpyAsync(box, box_init, sizeof box_init, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) {
      hipLaunchKernelGGL(map_cell_key_kernel, dim3((V + 255) / 256), dim3(256), 0, st, d_means, d_valid, V, inv_cell, keys, idx,
                         flags);
      e = hipGetLastError();
    }
    if (e == hipSuccess) e = rocprim::radix_sort_pairs(nullptr, t_sort, keys, keys_sorted, idx, map->d_orig_id, size_t(V), 0, 64, st);
    if (e == hipSuccess) e = rocprim::run_length_encode(nullptr, t_rle, keys_sorted, size_t(V), uniq, run_count, n_runs, st);
    if (e == hipSuccess) e = rocprim::exclusive_scan(nullptr, t_scan, run_count, run_start, 0u, size_t(V), rocprim::plus<uint32_t>(), st);
    if (e == hipSuccess) e = buf.alloc_bytes(&tmp, std::max(std::max(t_sort, t_rle), std::max(t_scan, size_t(16))));
    // stable: the voxels of a cell stay in index order, as std::sort on (key, index) pairs left them
    if (e == hipSuccess) e = rocprim::radix_sort_pairs(tmp, t_sort, keys, keys_sorted, idx, map->d_orig_id, size_t(V), 0, 64, st);
    if (e == hipSuccess) e = rocprim::run_length_encode(tmp, t_rle, keys_sorted, size_t(V), uniq, run_count, n_runs, st);
    if (e == hipSuccess) e = hipMemcpyAsync(&runs, n_runs, sizeof runs, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipMemcpyAsync(h_flags, flags, sizeof h_flags, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e == hipSuccess && h_flags[0] != 0) {
      nos_ndt_map_destroy(map);
      return fail(NOS_ERR_INVALID_ARGUMENT, "voxel %u has a non-finite mean", h_flags[1] - 1u);
    }
    if (e == hipSuccess && h_flags[2] != 0) {
      nos_ndt_map_destroy(map);
      return fail(NOS_ERR_UNSUPPORTED, "voxel %u lies outside the addressable grid", h_flags[3] - 1u);
    }
    if (e == hipSuccess && runs > 0) {
      // the invalid voxels, if any, form the last run (key = all ones)
      uint64_t last_key = 0;
      uint32_t last_count = 0;
      e = hipMemcpyAsync(&last_key, uniq + (runs - 1), sizeof last_key, hipMemcpyDeviceToHost, st);
      if (e == hipSuccess) e = hipMemcpyAsync(&last_count, run_count + (runs - 1), sizeof last_count, hipMemcpyDeviceToHost, st);
      if (e == hipSuccess) e = rocprim::exclusive_scan(tmp, t_scan, run_count, run_start, 0u, size_t(runs), rocprim::plus<uint32_t>(), st);
      if (e == hipSuccess) e = hipStreamSynchronize(st);
      const bool has_invalid = last_key == nos::kEmptyCell;
      n_cells = has_invalid ? runs - 1 : runs;
      n_valid = has_invalid ? V - last_count : V;
    }
    if (e == hipSuccess && n_cells > 0) {
      hipLaunchKernelGGL(map_cell_box_kernel, dim3((n_cells + 255) / 256), dim3(256), 0, st, uniq, n_cells, box);
      e = hipGetLastError();
      if (e == hipSuccess) e = hipMemcpyAsync(h_box, box, sizeof h_box, hipMemcpyDeviceToHost, st);
      if (e == hipSuccess) e = hipStreamSynchronize(st);
    }
  }
  map->n_voxels = n_valid;
  // dense form of the grid (see MapView): bounding box of the occupied cells plus a one-cell border
  int64_t lo[3] = {0, 0, 0}, dim[3] = {0, 0, 0};
  size_t n_dense = 0;
  if (e == hipSuccess && n_cells > 0 && ctx->settings.match_dense != 0) {
    const int64_t bias = int64_t(1) << 20;
    double total = 1.0;
    for (int k = 0; k < 3; ++k) {
      lo[k] = int64_t(h_box[k]) - bias - 1;
      dim[k] = int64_t(h_box[3 + k]) - int64_t(h_box[k]) + 3;
      total *= double(dim[k]);
    }
    if (total <= double(size_t(1) << 26))  // ≤ 64 M cells = 256 MB of offsets; beyond that the hash table serves
      n_dense = size_t(dim[0]) * size_t(dim[1]) * size_t(dim[2]);
  }
  size_t table_size = 16;
  while (table_size < 2 * size_t(n_cells) + 1) table_size <<= 1;
  const size_t nv = std::max<size_t>(n_valid, 1);
  // one allocation for everything the map keeps (seven arrays), one memset for the part that starts cleared
  {
    auto up = [](size_t b) { return (b + 255) & ~size_t(255); };
    const size_t b_mean = up(nv * 3 * sizeof(double)), b_S = up(nv * 9 * sizeof(double));
    const size_t b_rec = n_dense > 0 ? up(nv * 4 * sizeof(double)) : 0;
    const size_t b_start = up(table_size * sizeof(uint32_t)), b_count = up(table_size * sizeof(uint32_t));
    const size_t b_dense = n_dense > 0 ? up((n_dense + 1) * sizeof(uint32_t)) : 0, b_key = up(table_size * sizeof(uint64_t));
    char* base = nullptr;
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&base), b_mean + b_S + b_rec + b_start + b_count + b_dense + b_key);
    if (e == hipSuccess) {
      map->d_block = base;
      map->d_mean = reinterpret_cast<double*>(base);
      map->d_sqrt_info = reinterpret_cast<double*>(base + b_mean);
      if (n_dense > 0) map->d_record = reinterpret_cast<double*>(base + b_mean + b_S);
      char* cleared = base + b_mean + b_S + b_rec;  // start, count, dense offsets: zero; keys behind them: kEmptyCell
      map->d_cell_start = reinterpret_cast<uint32_t*>(cleared);
      map->d_cell_count = reinterpret_cast<uint32_t*>(cleared + b_start);
      if (n_dense > 0) map->d_dense_begin = reinterpret_cast<uint32_t*>(cleared + b_start + b_count);
      map->d_cell_key = reinterpret_cast<uint64_t*>(cleared + b_start + b_count + b_dense);
      e = hipMemsetAsync(cleared, 0, b_start + b_count + b_dense, st);
      if (e == hipSuccess) e = hipMemsetAsync(map->d_cell_key, 0xFF, table_size * sizeof(uint64_t), st);
    }
  }
  if (e == hipSuccess && n_valid > 0) {
    hipLaunchKernelGGL(map_gather_kernel, dim3((n_valid + 255) / 256), dim3(256), 0, st, d_means, d_S, map->d_orig_id, n_valid,
                       map->d_mean, map->d_sqrt_info, map->d_record);
    hipLaunchKernelGGL(map_cell_tables_kernel, dim3((n_cells + 255) / 256), dim3(256), 0, st, uniq, run_count, run_start, n_cells,
                       uint32_t(table_size - 1), reinterpret_cast<unsigned long long*>(map->d_cell_key), map->d_cell_start,
                       map->d_cell_count, map->d_dense_begin, lo[0], lo[1], lo[2], dim[1], dim[2]);
    e = hipGetLastError();
    if (e == hipSuccess && n_dense > 0) {
      size_t t_inc = 0;
      void* tmp2 = nullptr;
      e = rocprim::inclusive_scan(nullptr, t_inc, map->d_dense_begin, map->d_dense_begin, n_dense + 1, rocprim::plus<uint32_t>(), st);
      if (e == hipSuccess) e = buf.alloc_bytes(&tmp2, std::max<size_t>(t_inc, 16));
      if (e == hipSuccess)
        e = rocprim::inclusive_scan(tmp2, t_inc, map->d_dense_begin, map->d_dense_begin, n_dense + 1, rocprim::plus<uint32_t>(), st);
    }
  }
  if (e == hipSuccess) e = hipStreamSynchronize(st);  // the temporaries go out of scope
  if (e != hipSuccess) {
    nos_ndt_map_destroy(map);
    return fail(e == hipErrorOutOfMemory ? NOS_ERR_OUT_OF_MEMORY : NOS_ERR_HIP, "map tables failed: %s", hipGetErrorString(e));
  }
  map->view.mean = map->d_mean;
  map->view.sqrt_info = map->d_sqrt_info;
  map->view.orig_id = map->d_orig_id;
  map->view.cell_key = map->d_cell_key;
  map->view.cell_start = map->d_cell_start;
  map->view.cell_count = map->d_cell_count;
  map->view.table_mask = uint32_t(table_size - 1);
  map->view.inv_cell = inv_cell;
  map->view.radius_sq = search_radius_sq;
  map->view.dense_begin = map->d_dense_begin;
  map->view.record = map->d_record;
  map->view.ox = lo[0];
  map->view.oy = lo[1];
  map->view.oz = lo[2];
  map->view.nx = int32_t(dim[0]);
  map->view.ny = int32_t(dim[1]);
  map->view.nz = int32_t(dim[2]);
  *out_map = map;
  return NOS_OK;
}

}  // namespace

// nos_mapbuild.hip: the voxel statistics are already on the device
int nosd::map_create_device(nos_ctx* ctx, size_t n_voxels, const double* d_means, const double* d_S, const unsigned char* d_valid,
                            double search_radius_sq, nos_ndt_map** out_map) {
  return map_create_from_device(ctx, n_voxels, d_means, d_S, d_valid, search_radius_sq, out_map);
}

extern "C" {

int nos_ndt_map_create(nos_ctx* ctx, size_t n_voxels, const double* means_xyz, const double* sqrt_infos,
                       const unsigned char* valid, double search_radius_sq, nos_ndt_map** out_map) {
  nosd::CtxGuard guard_(ctx);  // one solve / accumulate / create at a time per context
  if (!ctx || !out_map) return fail(NOS_ERR_INVALID_ARGUMENT, "ctx / out_map is NULL");
  *out_map = nullptr;
  if (ctx->slots.size() != 1) return fail(NOS_ERR_UNSUPPORTED, "the matcher needs a single-device context");
  if ((!means_xyz || !sqrt_infos) && n_voxels > 0) return fail(NOS_ERR_INVALID_ARGUMENT, "map arrays are NULL");
  if (!(search_radius_sq > 0.0) || !std::isfinite(search_radius_sq)) return fail(NOS_ERR_INVALID_ARGUMENT, "bad search radius");
  if (n_voxels >= 0xFFFFFFFFull) return fail(NOS_ERR_UNSUPPORTED, "too many voxels");
  // the voxel statistics go to the device as they are; bucketing by matcher cell, the hash table and the dense grid are
  // built there (map_create_from_device)
  DeviceSlot& slot = ctx->slots[0];
  DeviceBuffers buf(&slot);
  double *d_means = nullptr, *d_S = nullptr;
  unsigned char* d_valid = nullptr;
  hipError_t e = hipSetDevice(slot.device);
  if (e == hipSuccess) e = buf.alloc(&d_means, n_voxels * 3);
  if (e == hipSuccess) e = buf.alloc(&d_S, n_voxels * 9);
  if (e == hipSuccess && valid) e = buf.alloc(&d_valid, n_voxels);
  if (e == hipSuccess && n_voxels > 0) {
    e = hipMemcpyAsync(d_means, means_xyz, n_voxels * 3 * sizeof(double), hipMemcpyHostToDevice, slot.stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_S, sqrt_infos, n_voxels * 9 * sizeof(double), hipMemcpyHostToDevice, slot.stream);
    if (e == hipSuccess && valid) e = hipMemcpyAsync(d_valid, valid, n_voxels, hipMemcpyHostToDevice, slot.stream);
  }
  if (e != hipSuccess)
    return fail(e == hipErrorOutOfMemory ? NOS_ERR_OUT_OF_MEMORY : NOS_ERR_HIP, "map upload failed: %s", hipGetErrorString(e));
  return map_create_from_device(ctx, n_voxels, d_means, d_S, d_valid, search_radius_sq, out_map);
}

int nos_ndt_map_destroy(nos_ndt_map* map) {
  nosd::CtxGuard guard_(map ? map->ctx : nullptr);  // one solve / accumulate / create at a time per context
  if (!map) return NOS_OK;
  (void)hipSetDevice(map->ctx->slots[0].device);
  if (map->d_block) (void)hipFree(map->d_block);  // d_mean, d_sqrt_info, the hash table, d_dense_begin, d_record
  if (map->d_orig_id) (void)hipFree(map->d_orig_id);
  if (map->d_n_matches) (void)hipFree(map->d_n_matches);
  delete map;
  return NOS_OK;
}

size_t nos_ndt_map_size(const nos_ndt_map* map) { return map ? map->n_voxels : 0; }

int nos_scan_create(nos_ctx* ctx, size_t n_points, const double* points_xyz, nos_scan** out_scan) {
  nosd::CtxGuard guard_(ctx);  // one solve / accumulate / create at a time per context
  if (!ctx || !out_scan) return fail(NOS_ERR_INVALID_ARGUMENT, "ctx / out_scan is NULL");
  *out_scan = nullptr;
  if (ctx->slots.size() != 1) return fail(NOS_ERR_UNSUPPORTED, "the matcher needs a single-device context");
  if (!points_xyz && n_points > 0) return fail(NOS_ERR_INVALID_ARGUMENT, "points is NULL");
  nos_scan* scan = new (std::nothrow) nos_scan();
  if (!scan) return fail(NOS_ERR_OUT_OF_MEMORY, "host allocation failed");
  scan->ctx = ctx;
  scan->n = n_points;
  DeviceSlot& slot = ctx->slots[0];
  // [n][3] records (std::vector<Vec3>) → 3 planes, through the same record-unpack kernel as datasets
  hipError_t e = hipSetDevice(slot.device);
  void* staging = nullptr;
  const size_t bytes = std::max<size_t>(n_points, 1) * 3 * sizeof(double);
  if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&scan->d_planes), bytes);
  if (e == hipSuccess) e = hipMalloc(&staging, bytes);
  int rc = NOS_OK;
  if (e == hipSuccess && n_points > 0) {
    e = hipMemcpyAsync(staging, points_xyz, n_points * 3 * sizeof(double), hipMemcpyHostToDevice, slot.stream);
    nos::TiledLayout L{};
    L.n = n_points;
    L.n_padded = n_points;
    L.tile_stride = 0;
    L.field_stride = n_points;
    L.tile_shift = 40;
    L.tile_mask = 0xFFFFFFFFu;
    nos::FieldOffsets fo{};
    fo.off[0] = 0;
    fo.off[1] = 8;
    fo.off[2] = 16;
    if (e == hipSuccess)
      rc = unpack_records(NOS_F64, static_cast<unsigned char*>(staging), 24, fo, 3, 0, n_points, L, scan->d_planes, slot.stream);
    if (e == hipSuccess && rc == NOS_OK) e = hipStreamSynchronize(slot.stream);
  }
  if (staging) (void)hipFree(staging);
  if (e != hipSuccess || rc != NOS_OK) {
    nos_scan_destroy(scan);
    if (rc != NOS_OK) return rc;
    return fail(e == hipErrorOutOfMemory ? NOS_ERR_OUT_OF_MEMORY : NOS_ERR_HIP, "scan upload failed: %s", hipGetErrorString(e));
  }
  *out_scan = scan;
  return NOS_OK;
}

int nos_scan_destroy(nos_scan* scan) {
  nosd::CtxGuard guard_(scan ? scan->ctx : nullptr);  // one solve / accumulate / create at a time per context
  if (!scan) return NOS_OK;
  (void)hipSetDevice(scan->ctx->slots[0].device);
  if (scan->d_planes) (void)hipFree(scan->d_planes);
  if (scan->d_order) (void)hipFree(scan->d_order);
  delete scan;
  return NOS_OK;
}

size_t nos_scan_size(const nos_scan* scan) { return scan ? scan->n : 0; }

int nos_ndt_match(nos_ndt_map* map, nos_scan* scan, const double R[9], const double t[3], int max_neighbors,
                  int dtype, nos_dataset** out_ds, size_t* n_matches) {
  nosd::CtxGuard guard_(map ? map->ctx : nullptr);  // one solve / accumulate / create at a time per context
  if (!map || !scan || !R || !t || !out_ds) return fail(NOS_ERR_INVALID_ARGUMENT, "NULL argument");
  if (map->ctx != scan->ctx) return fail(NOS_ERR_INVALID_ARGUMENT, "map and scan belong to different contexts");
  if (max_neighbors < 1 || max_neighbors > 2) return fail(NOS_ERR_UNSUPPORTED, "max_neighbors must be 1 or 2");
  nos_ctx* ctx = map->ctx;
  nos_dataset* ds = nullptr;
  int rc = dataset_new(ctx, kKindNdt, 2 * scan->n, dtype, out_ds, &ds);
  if (rc != NOS_OK) return rc;
  Shard& sh = ds->shards[0];
  DeviceSlot& slot = ctx->slots[0];
  nos::PosePod pose;
  for (int k = 0; k < 9; ++k) pose.R[k] = R[k];
  for (int k = 0; k < 3; ++k) pose.t[k] = t[k];
  hipError_t e = hipSetDevice(slot.device);
  if (e == hipSuccess) e = hipMemsetAsync(map->d_n_matches, 0, sizeof(unsigned long long), slot.stream);
  if (e == hipSuccess && scan->n > 0) {
    const dim3 grid(unsigned((scan->n + 255) / 256));
    const double* px = scan->d_planes;
    const double* py = scan->d_planes + scan->n;
    const double* pz = scan->d_planes + 2 * scan->n;
    if (dtype == NOS_F64)
      hipLaunchKernelGGL((nos::match_kernel<double>), grid, dim3(256), 0, slot.stream, map->view, px, py, pz,
                         uint64_t(scan->n), pose, max_neighbors, sh.layout, static_cast<double*>(sh.data),
                         map->d_n_matches);
    else
      hipLaunchKernelGGL((nos::match_kernel<float>), grid, dim3(256), 0, slot.stream, map->view, px, py, pz,
                         uint64_t(scan->n), pose, max_neighbors, sh.layout, static_cast<float*>(sh.data),
                         map->d_n_matches);
    e = hipGetLastError();
  }
  if (e == hipSuccess)
    rc = zero_pad(dtype, ds->n_fields, sh.layout, sh.data, slot.stream);
  unsigned long long count = 0;
  if (e == hipSuccess && rc == NOS_OK)
    e = hipMemcpyAsync(&count, map->d_n_matches, sizeof count, hipMemcpyDeviceToHost, slot.stream);
  if (e == hipSuccess && rc == NOS_OK) e = hipStreamSynchronize(slot.stream);
  if (e != hipSuccess || rc != NOS_OK) {
    nos_dataset_destroy(ds);
    if (rc != NOS_OK) return rc;
    return fail(NOS_ERR_HIP, "matching failed: %s", hipGetErrorString(e));
  }
  if (n_matches) *n_matches = size_t(count);
  *out_ds = ds;
  return NOS_OK;
}

int nos_dataset_drop_last_matches(nos_dataset* ds, size_t n_drop) {
  nosd::CtxGuard guard_(ds ? ds->ctx : nullptr);  // one solve / accumulate / create at a time per context
  if (!ds) return fail(NOS_ERR_INVALID_ARGUMENT, "dataset is NULL");
  if (ds->kind != kKindNdt) return fail(NOS_ERR_WRONG_KIND, "flat NDT datasets only");
  if (ds->shards.size() != 1) return fail(NOS_ERR_UNSUPPORTED, "single-device datasets only");
  if (n_drop == 0 || ds->n == 0) return NOS_OK;
  Shard& sh = ds->shards[0];
  DeviceSlot& slot = ds->ctx->slots[sh.slot];
  NOS_HIP_CHECK(hipSetDevice(slot.device));
  if (ds->dtype == NOS_F64)
    hipLaunchKernelGGL((nos::drop_last_matches_kernel<double>), dim3(1), dim3(64), 0, slot.stream,
                       static_cast<double*>(sh.data), sh.layout, uint64_t(n_drop));
  else
    hipLaunchKernelGGL((nos::drop_last_matches_kernel<float>), dim3(1), dim3(64), 0, slot.stream,
                       static_cast<float*>(sh.data), sh.layout, uint64_t(n_drop));
  NOS_HIP_CHECK(hipGetLastError());
  return NOS_OK;  // stream order: every later launch on this context sees the cleared records
}

int nos_dataset_download(nos_dataset* ds, double* const planes[]) {
  nosd::CtxGuard guard_(ds ? ds->ctx : nullptr);  // one solve / accumulate / create at a time per context
  if (!ds || !planes) return fail(NOS_ERR_INVALID_ARGUMENT, "NULL argument");
  size_t begin = 0;
  for (const Shard& sh : ds->shards) {
    DeviceSlot& slot = ds->ctx->slots[sh.slot];
    const size_t cnt = sh.layout.n;
    if (cnt == 0) continue;
    NOS_HIP_CHECK(hipSetDevice(slot.device));
    double* tmp = nullptr;
    NOS_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&tmp), cnt * size_t(ds->n_fields) * sizeof(double)));
    const dim3 grid(unsigned((cnt + 255) / 256), unsigned(ds->n_fields));
    if (ds->dtype == NOS_F64)
      hipLaunchKernelGGL((nos::untile_kernel<double>), grid, dim3(256), 0, slot.stream,
                         static_cast<const double*>(sh.data), ds->n_fields, sh.layout, tmp);
    else
      hipLaunchKernelGGL((nos::untile_kernel<float>), grid, dim3(256), 0, slot.stream,
                         static_cast<const float*>(sh.data), ds->n_fields, sh.layout, tmp);
    hipError_t e = hipGetLastError();
    for (int f = 0; f < ds->n_fields && e == hipSuccess; ++f)
      e = hipMemcpyAsync(planes[f] + begin, tmp + size_t(f) * cnt, cnt * sizeof(double), hipMemcpyDeviceToHost, slot.stream);
    if (e == hipSuccess) e = hipStreamSynchronize(slot.stream);
    (void)hipFree(tmp);
    if (e != hipSuccess) return fail(NOS_ERR_HIP, "dataset download failed: %s", hipGetErrorString(e));
    begin += cnt;
  }
  return NOS_OK;
}

}  // extern "C"

