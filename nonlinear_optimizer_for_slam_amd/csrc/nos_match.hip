// nos_match.hip — device correspondence matcher, scans, dataset download.
#include "nos_internal.hpp"

using namespace nosd;

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
  const double cell = std::sqrt(search_radius_sq);
  const double inv_cell = 1.0 / cell;
  // bucket the valid voxels by grid cell: sort (cell key, voxel index) pairs — cells in key order, voxels of a cell in
  // index order (deterministic layout; a sorted vector instead of a std::map of vectors: 5-10x less host time at 10^5+ voxels)
  std::vector<std::pair<uint64_t, uint32_t>> keyed;
  keyed.reserve(n_voxels);
  for (size_t v = 0; v < n_voxels; ++v) {
    if (valid && !valid[v]) continue;  // `if (!ndt.is_valid) continue;` of the reference's matcher
    const double* m = means_xyz + 3 * v;
    if (!std::isfinite(m[0]) || !std::isfinite(m[1]) || !std::isfinite(m[2]))
      return fail(NOS_ERR_INVALID_ARGUMENT, "voxel %zu has a non-finite mean", v);
    const int64_t ix = int64_t(std::floor(m[0] * inv_cell)), iy = int64_t(std::floor(m[1] * inv_cell)),
                  iz = int64_t(std::floor(m[2] * inv_cell));
    const int64_t lim = (1 << 20) - 2;
    if (std::llabs(ix) > lim || std::llabs(iy) > lim || std::llabs(iz) > lim)
      return fail(NOS_ERR_UNSUPPORTED, "voxel %zu lies outside the addressable grid", v);
    keyed.emplace_back(nos::pack_cell(ix, iy, iz), uint32_t(v));
  }
  std::sort(keyed.begin(), keyed.end());
  struct CellRun {
    uint64_t key;
    uint32_t first, count;  // range of `keyed`
  };
  std::vector<CellRun> cells;
  for (size_t i = 0; i < keyed.size();) {
    size_t j = i;
    while (j < keyed.size() && keyed[j].first == keyed[i].first) ++j;
    cells.push_back({keyed[i].first, uint32_t(i), uint32_t(j - i)});
    i = j;
  }
  size_t table_size = 16;
  while (table_size < 2 * cells.size() + 1) table_size <<= 1;
  std::vector<uint64_t> keys(table_size, nos::kEmptyCell);
  std::vector<uint32_t> starts(table_size, 0), counts(table_size, 0), orig;
  std::vector<double> mean_sorted, s_sorted;
  orig.reserve(keyed.size());
  mean_sorted.reserve(keyed.size() * 3);
  s_sorted.reserve(keyed.size() * 9);
  for (const CellRun& cr : cells) {
    uint32_t h = nos::hash_cell(cr.key) & uint32_t(table_size - 1);
    while (keys[h] != nos::kEmptyCell) h = (h + 1) & uint32_t(table_size - 1);
    keys[h] = cr.key;
    starts[h] = uint32_t(orig.size());
    counts[h] = cr.count;
    for (uint32_t q = cr.first; q < cr.first + cr.count; ++q) {
      const uint32_t v = keyed[q].second;
      orig.push_back(v);
      for (int k = 0; k < 3; ++k) mean_sorted.push_back(means_xyz[3 * size_t(v) + k]);
      for (int k = 0; k < 9; ++k) s_sorted.push_back(sqrt_infos[9 * size_t(v) + k]);
    }
  }
  // dense form of the grid (see MapView): bounding box of the occupied cells plus a one-cell border
  std::vector<uint32_t> dense_begin;
  std::vector<double> records;
  int64_t lo[3] = {0, 0, 0}, dim[3] = {0, 0, 0};
  if (!cells.empty() && ctx->settings.match_dense != 0) {
    const int64_t bias = int64_t(1) << 20;
    int64_t mn[3] = {INT64_MAX, INT64_MAX, INT64_MAX}, mx[3] = {INT64_MIN, INT64_MIN, INT64_MIN};
    auto unpack = [&](uint64_t key, int64_t c[3]) {
      c[0] = int64_t((key >> 42) & 0x1FFFFFull) - bias;
      c[1] = int64_t((key >> 21) & 0x1FFFFFull) - bias;
      c[2] = int64_t(key & 0x1FFFFFull) - bias;
    };
    for (const CellRun& cr : cells) {
      int64_t c[3];
      unpack(cr.key, c);
      for (int k = 0; k < 3; ++k) {
        mn[k] = std::min(mn[k], c[k]);
        mx[k] = std::max(mx[k], c[k]);
      }
    }
    double total = 1.0;
    for (int k = 0; k < 3; ++k) {
      lo[k] = mn[k] - 1;
      dim[k] = mx[k] - mn[k] + 3;
      total *= double(dim[k]);
    }
    if (total <= double(size_t(1) << 26)) {  // ≤ 64 M cells = 256 MB of offsets; beyond that the hash table serves
      const size_t n_cells = size_t(dim[0]) * size_t(dim[1]) * size_t(dim[2]);
      dense_begin.assign(n_cells + 1, 0);
      // ascending packed keys = (x, y, z) lexicographic order = dense index order = record order
      for (const CellRun& cr : cells) {
        int64_t c[3];
        unpack(cr.key, c);
        const size_t idx = (size_t(c[0] - lo[0]) * size_t(dim[1]) + size_t(c[1] - lo[1])) * size_t(dim[2]) + size_t(c[2] - lo[2]);
        dense_begin[idx + 1] = cr.count;
      }
      for (size_t c = 0; c < n_cells; ++c) dense_begin[c + 1] += dense_begin[c];
      records.resize(orig.size() * 4);
      for (size_t j = 0; j < orig.size(); ++j) {
        for (int k = 0; k < 3; ++k) records[4 * j + k] = mean_sorted[3 * j + k];
        const uint64_t bits = orig[j];
        memcpy(&records[4 * j + 3], &bits, sizeof(double));
      }
    }
  }
  nos_ndt_map* map = new (std::nothrow) nos_ndt_map();
  if (!map) return fail(NOS_ERR_OUT_OF_MEMORY, "host allocation failed");
  map->ctx = ctx;
  map->n_voxels = orig.size();
  hipError_t e = hipSetDevice(ctx->slots[0].device);
  if (e == hipSuccess) e = upload(&map->d_mean, mean_sorted);
  if (e == hipSuccess) e = upload(&map->d_sqrt_info, s_sorted);
  if (e == hipSuccess) e = upload(&map->d_orig_id, orig);
  if (e == hipSuccess) e = upload(&map->d_cell_key, keys);
  if (e == hipSuccess) e = upload(&map->d_cell_start, starts);
  if (e == hipSuccess) e = upload(&map->d_cell_count, counts);
  if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&map->d_n_matches), sizeof(unsigned long long));
  if (e == hipSuccess && !dense_begin.empty()) {
    e = upload(&map->d_dense_begin, dense_begin);
    if (e == hipSuccess) e = upload(&map->d_record, records);
  }
  if (e != hipSuccess) {
    nos_ndt_map_destroy(map);
    return fail(e == hipErrorOutOfMemory ? NOS_ERR_OUT_OF_MEMORY : NOS_ERR_HIP, "map upload failed: %s", hipGetErrorString(e));
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

int nos_ndt_map_destroy(nos_ndt_map* map) {
  nosd::CtxGuard guard_(map ? map->ctx : nullptr);  // one solve / accumulate / create at a time per context
  if (!map) return NOS_OK;
  (void)hipSetDevice(map->ctx->slots[0].device);
  if (map->d_mean) (void)hipFree(map->d_mean);
  if (map->d_sqrt_info) (void)hipFree(map->d_sqrt_info);
  if (map->d_orig_id) (void)hipFree(map->d_orig_id);
  if (map->d_cell_key) (void)hipFree(map->d_cell_key);
  if (map->d_cell_start) (void)hipFree(map->d_cell_start);
  if (map->d_cell_count) (void)hipFree(map->d_cell_count);
  if (map->d_n_matches) (void)hipFree(map->d_n_matches);
  if (map->d_dense_begin) (void)hipFree(map->d_dense_begin);
  if (map->d_record) (void)hipFree(map->d_record);
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

