// nos_mapbuild.hip — NDT map construction on the device.
#include "nos_internal.hpp"

#include <rocprim/rocprim.hpp>

#include "mapbuild_kernels.hpp"

using namespace nosd;

struct nos_map_stats {
  std::vector<double> means;            // [V][3] voxel order = ascending packed (ix, iy, iz); reference-exact mode: first seen
  std::vector<double> sqrt_infos;       // [V][9]
  std::vector<unsigned char> valid;     // [V]
  std::vector<uint32_t> counts;         // [V]
  std::vector<int64_t> cells;           // [V][3] integer voxel coordinates
  std::vector<double> evals;            // [V][3] un-floored eigenvalues           (reference-exact mode only)
  std::vector<double> evecs;            // [V][9] row-major eigenvector matrix V   (reference-exact mode only)
};

namespace nosd {
// nos_mapexact.hip (compiled with -ffp-contract=off)
hipError_t launch_map_exact(const double* px, const double* py, const double* pz, const uint32_t* sorted_idx,
                            const uint32_t* seg_offset, const uint32_t* seg_count, uint32_t n_voxels, int fma_mask,
                            int eigen_version, double* acc, double* mean, double* sqrt_info, unsigned char* valid,
                            double* evals, double* evecs, uint32_t* first_idx, hipStream_t stream);
}  // namespace nosd

extern "C" {

// Reorders the points of a scan by grid cell (lexicographic in the scan's own frame).  A rigid pose keeps
// neighbours neighbours, so afterwards the 64 points of a wave of the matcher walk the same few cells of the map:
// their candidate loads fall into the same cache lines and their loops have similar trip counts.  One radix sort
// per scan (not per outer iteration).  The order of the matcher's output slots follows the new point order;
// nos_scan_order returns the permutation.
int nos_scan_sort_by_cell(nos_scan* scan, double cell_edge) {
  nosd::CtxGuard guard_(scan ? scan->ctx : nullptr);  // one solve / accumulate / create at a time per context
  if (!scan) return fail(NOS_ERR_INVALID_ARGUMENT, "scan is NULL");
  if (!(cell_edge > 0.0) || !std::isfinite(cell_edge)) return fail(NOS_ERR_INVALID_ARGUMENT, "bad cell edge");
  const size_t n = scan->n;
  if (n == 0) return NOS_OK;
  if (n >= 0xFFFFFFFFull) return fail(NOS_ERR_UNSUPPORTED, "too many points");
  DeviceSlot& slot = scan->ctx->slots[0];
  hipStream_t st = slot.stream;
  DeviceBuffers buf(&slot);  // arena (pooled slabs) for the temporaries
  uint64_t *keys = nullptr, *keys_sorted = nullptr;
  uint32_t *idx = nullptr, *order = nullptr;
  double* sorted = nullptr;
  hipError_t e = hipSetDevice(slot.device);
  if (e == hipSuccess) e = buf.alloc(&keys, n);
  if (e == hipSuccess) e = buf.alloc(&keys_sorted, n);
  if (e == hipSuccess) e = buf.alloc(&idx, n);
  if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&order), n * sizeof(uint32_t));
  if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&sorted), n * 3 * sizeof(double));
  // compact keys (see voxel_compact_key_kernel): the cell's index inside the scan's bounding box, same order, fewer bits
  long long h_box[6] = {0x7FFFFFFFFFFFFFFFll, 0x7FFFFFFFFFFFFFFFll, 0x7FFFFFFFFFFFFFFFll,
                        -0x7FFFFFFFFFFFFFFFll - 1, -0x7FFFFFFFFFFFFFFFll - 1, -0x7FFFFFFFFFFFFFFFll - 1};
  long long dims[3] = {1, 1, 1};
  bool compact = false;
  unsigned key_bits = 64;
  if (e == hipSuccess) {
    long long* d_box = nullptr;
    e = buf.alloc(&d_box, 6);
    if (e == hipSuccess) e = hipMemcpyAsync(d_box, h_box, sizeof h_box, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) {
      hipLaunchKernelGGL(nos::voxel_box_kernel, dim3(unsigned(std::min<size_t>((n + 255) / 256, 1024))), dim3(256), 0, st,
                         scan->d_planes, scan->d_planes + n, scan->d_planes + 2 * n, uint64_t(n), 1.0 / cell_edge, d_box);
      e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(h_box, d_box, sizeof h_box, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e == hipSuccess && scan->ctx->settings.map_compact_keys != 0 && h_box[0] <= h_box[3] && h_box[1] <= h_box[4] && h_box[2] <= h_box[5]) {
      double total = 1.0;
      for (int k = 0; k < 3; ++k) total *= double(h_box[3 + k]) - double(h_box[k]) + 1.0;
      if (total < 4.0e18) {
        compact = true;
        for (int k = 0; k < 3; ++k) dims[k] = h_box[3 + k] - h_box[k] + 1;
        key_bits = 1;
        while (key_bits < 64 && double(1ull << key_bits) < total) ++key_bits;
      }
    }
  }
  if (e == hipSuccess) {
    const dim3 grid(unsigned((n + 255) / 256));
    if (compact)
      hipLaunchKernelGGL(nos::voxel_compact_key_kernel, grid, dim3(256), 0, st, scan->d_planes, scan->d_planes + n,
                         scan->d_planes + 2 * n, uint64_t(n), 1.0 / cell_edge, h_box[0], h_box[1], h_box[2], dims[0], dims[1],
                         dims[2], keys, idx);
    else
      hipLaunchKernelGGL(nos::voxel_key_kernel, grid, dim3(256), 0, st, scan->d_planes, scan->d_planes + n,
                         scan->d_planes + 2 * n, uint64_t(n), 1.0 / cell_edge, keys, idx);
    e = hipGetLastError();
    size_t tmp_bytes = 0;
    void* tmp = nullptr;
    if (e == hipSuccess) e = rocprim::radix_sort_pairs(nullptr, tmp_bytes, keys, keys_sorted, idx, order, n, 0, key_bits, st);
    if (e == hipSuccess) e = buf.alloc_bytes(&tmp, std::max(tmp_bytes, size_t(16)));
    if (e == hipSuccess) e = rocprim::radix_sort_pairs(tmp, tmp_bytes, keys, keys_sorted, idx, order, n, 0, key_bits, st);
    for (int f = 0; f < 3 && e == hipSuccess; ++f) {
      hipLaunchKernelGGL((nos::gather_plane_kernel<double, double>), grid, dim3(256), 0, st, scan->d_planes + size_t(f) * n,
                         order, uint64_t(n), uint64_t(n), 0.0, sorted + size_t(f) * n);
      e = hipGetLastError();
    }
    if (e == hipSuccess && scan->d_order != nullptr) {
      // already sorted once: compose the permutations so that d_order still refers to the ORIGINAL indices
      uint32_t* composed = nullptr;
      e = buf.alloc(&composed, n);
      if (e == hipSuccess) {
        hipLaunchKernelGGL((nos::gather_plane_kernel<uint32_t, uint32_t>), grid, dim3(256), 0, st, scan->d_order, order,
                           uint64_t(n), uint64_t(n), 0u, composed);
        e = hipGetLastError();
      }
      if (e == hipSuccess) e = hipMemcpyAsync(order, composed, n * sizeof(uint32_t), hipMemcpyDeviceToDevice, st);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(st);
  }
  if (e != hipSuccess) {
    if (order) (void)hipFree(order);
    if (sorted) (void)hipFree(sorted);
    return fail(e == hipErrorOutOfMemory ? NOS_ERR_OUT_OF_MEMORY : NOS_ERR_HIP, "scan sort failed: %s", hipGetErrorString(e));
  }
  (void)hipFree(scan->d_planes);
  if (scan->d_order) (void)hipFree(scan->d_order);
  scan->d_planes = sorted;
  scan->d_order = order;
  return NOS_OK;
}

int nos_scan_order(const nos_scan* scan, uint32_t* order_out) {
  nosd::CtxGuard guard_(scan ? scan->ctx : nullptr);  // one solve / accumulate / create at a time per context
  if (!scan || (!order_out && scan->n > 0)) return fail(NOS_ERR_INVALID_ARGUMENT, "NULL argument");
  if (scan->d_order == nullptr) {
    for (size_t i = 0; i < scan->n; ++i) order_out[i] = uint32_t(i);
    return NOS_OK;
  }
  NOS_HIP_CHECK(hipSetDevice(scan->ctx->slots[0].device));
  NOS_HIP_CHECK(hipMemcpy(order_out, scan->d_order, scan->n * sizeof(uint32_t), hipMemcpyDeviceToHost));
  return NOS_OK;
}


int nos_map_stats_destroy(nos_map_stats* stats) {
  delete stats;
  return NOS_OK;
}

size_t nos_map_stats_size(const nos_map_stats* stats) { return stats ? stats->counts.size() : 0; }

int nos_map_stats_get(const nos_map_stats* stats, double* means_xyz, double* sqrt_infos, unsigned char* valid,
                      uint32_t* counts, int64_t* cells_xyz) {
  if (!stats) return fail(NOS_ERR_INVALID_ARGUMENT, "stats is NULL");
  const size_t V = stats->counts.size();
  if (means_xyz) memcpy(means_xyz, stats->means.data(), V * 3 * sizeof(double));
  if (sqrt_infos) memcpy(sqrt_infos, stats->sqrt_infos.data(), V * 9 * sizeof(double));
  if (valid) memcpy(valid, stats->valid.data(), V);
  if (counts) memcpy(counts, stats->counts.data(), V * sizeof(uint32_t));
  if (cells_xyz) memcpy(cells_xyz, stats->cells.data(), V * 3 * sizeof(int64_t));
  return NOS_OK;
}

int nos_map_stats_get_eigen(const nos_map_stats* stats, double* eigenvalues, double* eigenvectors) {
  if (!stats) return fail(NOS_ERR_INVALID_ARGUMENT, "stats is NULL");
  if (stats->evals.empty() && !stats->counts.empty())
    return fail(NOS_ERR_UNSUPPORTED, "eigen-decompositions are kept by NOS_MAP_REFERENCE_EXACT builds only");
  const size_t V = stats->counts.size();
  if (eigenvalues) memcpy(eigenvalues, stats->evals.data(), V * 3 * sizeof(double));
  if (eigenvectors) memcpy(eigenvectors, stats->evecs.data(), V * 9 * sizeof(double));
  return NOS_OK;
}

int nos_ndt_map_build(nos_ctx* ctx, size_t n_points, const double* points_xyz, double voxel_resolution,
                      double search_radius_sq, int flags, nos_ndt_map** out_map, nos_map_stats** out_stats) {
  nosd::CtxGuard guard_(ctx);  // one solve / accumulate / create at a time per context
  if (!ctx || !out_map) return fail(NOS_ERR_INVALID_ARGUMENT, "ctx / out_map is NULL");
  *out_map = nullptr;
  if (out_stats) *out_stats = nullptr;
  if (ctx->slots.size() != 1) return fail(NOS_ERR_UNSUPPORTED, "map build needs a single-device context");
  if (!(voxel_resolution > 0.0) || !std::isfinite(voxel_resolution)) return fail(NOS_ERR_INVALID_ARGUMENT, "bad voxel resolution");
  if (n_points >= 0xFFFFFFFFull) return fail(NOS_ERR_UNSUPPORTED, "too many points for one build");
  const bool exact = (flags & NOS_MAP_REFERENCE_EXACT) != 0;
  if (exact && (flags & NOS_MAP_PROPER_SQRT_INFORMATION))
    return fail(NOS_ERR_INVALID_ARGUMENT, "NOS_MAP_REFERENCE_EXACT reproduces the harness formula D^-1/2 V; it cannot be combined with NOS_MAP_PROPER_SQRT_INFORMATION");
  nos_scan* scan = nullptr;
  int rc = nos_scan_create(ctx, n_points, points_xyz, &scan);  // [n][3] → 3 planes on the device
  if (rc != NOS_OK) return rc;
  DeviceSlot& slot = ctx->slots[0];
  hipStream_t st = slot.stream;
  DeviceBuffers buf(&slot);  // arena: slabs from the slot's buffer pool instead of ≈ 20 hipMalloc / hipFree pairs per build
  uint64_t *keys = nullptr, *keys_sorted = nullptr, *uniq = nullptr;
  uint32_t *idx = nullptr, *idx_sorted = nullptr, *counts = nullptr, *offsets = nullptr, *n_runs = nullptr;
  hipError_t e = hipSetDevice(slot.device);
  const size_t n = n_points;
  buf.reserve(n * (3 * sizeof(uint64_t) + 4 * sizeof(uint32_t)) + (size_t(64) << 20));  // the seven n-sized arrays + sort temporaries
  if (e == hipSuccess) e = buf.alloc(&keys, n);
  if (e == hipSuccess) e = buf.alloc(&keys_sorted, n);
  if (e == hipSuccess) e = buf.alloc(&idx, n);
  if (e == hipSuccess) e = buf.alloc(&idx_sorted, n);
  if (e == hipSuccess) e = buf.alloc(&uniq, n);
  if (e == hipSuccess) e = buf.alloc(&counts, n);
  if (e == hipSuccess) e = buf.alloc(&offsets, n);
  if (e == hipSuccess) e = buf.alloc(&n_runs, 1);
  const double* px = scan->d_planes;
  const double* py = scan->d_planes + n;
  const double* pz = scan->d_planes + 2 * n;
  uint32_t V = 0;
  // compact keys: cell index inside the points' bounding box (voxel_compact_key_kernel) — same order, a third of the bits
  long long h_box[6] = {0x7FFFFFFFFFFFFFFFll, 0x7FFFFFFFFFFFFFFFll, 0x7FFFFFFFFFFFFFFFll,
                        -0x7FFFFFFFFFFFFFFFll - 1, -0x7FFFFFFFFFFFFFFFll - 1, -0x7FFFFFFFFFFFFFFFll - 1};
  long long dims[3] = {1, 1, 1};
  bool compact = false;
  unsigned key_bits = 64;
  if (e == hipSuccess && n > 0) {
    long long* d_box = nullptr;
    e = buf.alloc(&d_box, 6);
    if (e == hipSuccess) e = hipMemcpyAsync(d_box, h_box, sizeof h_box, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) {
      const unsigned blocks = unsigned(std::min<size_t>((n + 255) / 256, 1024));
      hipLaunchKernelGGL(nos::voxel_box_kernel, dim3(blocks), dim3(256), 0, st, px, py, pz, uint64_t(n), 1.0 / voxel_resolution, d_box);
      e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(h_box, d_box, sizeof h_box, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e == hipSuccess && ctx->settings.map_compact_keys != 0 && h_box[0] <= h_box[3] && h_box[1] <= h_box[4] && h_box[2] <= h_box[5]) {
      double total = 1.0;
      for (int k = 0; k < 3; ++k) {
        const double d = double(h_box[3 + k]) - double(h_box[k]) + 1.0;
        total *= d;
      }
      if (total < 4.0e18) {  // the cell index fits 62 bits
        compact = true;
        for (int k = 0; k < 3; ++k) dims[k] = h_box[3 + k] - h_box[k] + 1;
        key_bits = 1;
        while (key_bits < 64 && double(1ull << key_bits) < total) ++key_bits;
      }
    }
  }
  if (e == hipSuccess && n > 0) {
    if (compact)
      hipLaunchKernelGGL(nos::voxel_compact_key_kernel, dim3(unsigned((n + 255) / 256)), dim3(256), 0, st, px, py, pz, uint64_t(n),
                         1.0 / voxel_resolution, h_box[0], h_box[1], h_box[2], dims[0], dims[1], dims[2], keys, idx);
    else
      hipLaunchKernelGGL(nos::voxel_key_kernel, dim3(unsigned((n + 255) / 256)), dim3(256), 0, st, px, py, pz, uint64_t(n),
                         1.0 / voxel_resolution, keys, idx);
    e = hipGetLastError();
    size_t t1 = 0, t2 = 0, t3 = 0;
    void* tmp = nullptr;
    if (e == hipSuccess) e = rocprim::radix_sort_pairs(nullptr, t1, keys, keys_sorted, idx, idx_sorted, n, 0, key_bits, st);
    if (e == hipSuccess) e = rocprim::run_length_encode(nullptr, t2, keys_sorted, n, uniq, counts, n_runs, st);
    if (e == hipSuccess) e = rocprim::exclusive_scan(nullptr, t3, counts, offsets, 0u, n, rocprim::plus<uint32_t>(), st);
    if (e == hipSuccess) e = buf.alloc_bytes(&tmp, std::max(std::max(t1, t2), std::max(t3, size_t(16))));
    if (e == hipSuccess) e = rocprim::radix_sort_pairs(tmp, t1, keys, keys_sorted, idx, idx_sorted, n, 0, key_bits, st);
    if (e == hipSuccess) e = rocprim::run_length_encode(tmp, t2, keys_sorted, n, uniq, counts, n_runs, st);
    if (e == hipSuccess) e = hipMemcpyAsync(&V, n_runs, sizeof V, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e == hipSuccess && V > 0) e = rocprim::exclusive_scan(tmp, t3, counts, offsets, 0u, size_t(V), rocprim::plus<uint32_t>(), st);
  }
  std::unique_ptr<nos_map_stats> stats(new (std::nothrow) nos_map_stats());
  if (!stats) e = hipErrorOutOfMemory;
  double *d_mean = nullptr, *d_S = nullptr, *d_acc = nullptr, *d_evals = nullptr, *d_evecs = nullptr;
  unsigned char* d_valid = nullptr;
  uint32_t* d_first = nullptr;
  buf.reserve(size_t(V) * (3 + 9 + 12 + 3 + 9) * sizeof(double) + size_t(V) * 8 + (size_t(1) << 20));  // the V-sized arrays
  if (e == hipSuccess) e = buf.alloc(&d_mean, size_t(V) * 3);
  if (e == hipSuccess) e = buf.alloc(&d_S, size_t(V) * 9);
  if (e == hipSuccess) e = buf.alloc(&d_valid, size_t(V));
  if (e == hipSuccess && exact) {
    e = buf.alloc(&d_acc, size_t(V) * 12);
    if (e == hipSuccess) e = buf.alloc(&d_evals, size_t(V) * 3);
    if (e == hipSuccess) e = buf.alloc(&d_evecs, size_t(V) * 9);
    if (e == hipSuccess) e = buf.alloc(&d_first, size_t(V));
    if (e == hipSuccess)
      e = launch_map_exact(px, py, pz, idx_sorted, offsets, counts, V, ctx->settings.map_fma_mask,
                           ctx->settings.map_eigen_version, d_acc, d_mean, d_S, d_valid, d_evals, d_evecs, d_first, st);
  } else if (e == hipSuccess && V > 0) {
    const nos::MapBuildParams prm{5, 0.01, 0.01, (flags & NOS_MAP_PROPER_SQRT_INFORMATION) ? 1 : 0};
    e = buf.alloc(&d_acc, size_t(V) * 9);
    if (e == hipSuccess) {
      // the points as 32-byte records for the gather (one sector per point instead of three); without room for them: planes
      double* d_rec = nullptr;
      if (n >= (size_t(1) << 16) && buf.alloc(&d_rec, n * 4) == hipSuccess) {
        hipLaunchKernelGGL(nos::points_to_records_kernel, dim3(unsigned((n + 255) / 256)), dim3(256), 0, st, px, py, pz, uint64_t(n),
                           d_rec);
      } else {
        d_rec = nullptr;
      }
      const unsigned blocks = unsigned((size_t(V) * nos::kWave + 255) / 256);
      hipLaunchKernelGGL(nos::voxel_sums_kernel, dim3(blocks), dim3(256), 0, st, px, py, pz, d_rec, idx_sorted, offsets, counts, V,
                         d_acc);
      hipLaunchKernelGGL(nos::voxel_eigen_kernel, dim3(unsigned((size_t(V) + 255) / 256)), dim3(256), 0, st, d_acc, counts, V,
                         prm, d_mean, d_S, d_valid);
      e = hipGetLastError();
    }
  }
  // The statistics stay on the device for the matcher's tables (map_create_device: bucketing by matcher cell, dense grid
  // and hash table are built there); they travel to the host only for the caller's nos_map_stats and for the reference-exact
  // mode, whose voxel list is re-ordered to first-seen order before the tables are built.
  const bool want_stats = out_stats != nullptr || exact;
  std::vector<uint64_t> h_keys(want_stats ? V : 0);
  std::vector<uint32_t> h_first;
  if (e == hipSuccess && exact) {
    stats->evals.resize(size_t(V) * 3);
    stats->evecs.resize(size_t(V) * 9);
    h_first.resize(V);
  }
  if (e == hipSuccess && want_stats) {
    stats->means.resize(size_t(V) * 3);
    stats->sqrt_infos.resize(size_t(V) * 9);
    stats->valid.resize(V);
    stats->counts.resize(V);
    stats->cells.resize(size_t(V) * 3);
  }
  if (e == hipSuccess && V > 0 && want_stats) {
    e = hipMemcpyAsync(stats->means.data(), d_mean, size_t(V) * 3 * sizeof(double), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipMemcpyAsync(stats->sqrt_infos.data(), d_S, size_t(V) * 9 * sizeof(double), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipMemcpyAsync(stats->valid.data(), d_valid, V, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipMemcpyAsync(stats->counts.data(), counts, size_t(V) * sizeof(uint32_t), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipMemcpyAsync(h_keys.data(), uniq, size_t(V) * sizeof(uint64_t), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess && exact) {
      e = hipMemcpyAsync(stats->evals.data(), d_evals, size_t(V) * 3 * sizeof(double), hipMemcpyDeviceToHost, st);
      if (e == hipSuccess) e = hipMemcpyAsync(stats->evecs.data(), d_evecs, size_t(V) * 9 * sizeof(double), hipMemcpyDeviceToHost, st);
      if (e == hipSuccess) e = hipMemcpyAsync(h_first.data(), d_first, size_t(V) * sizeof(uint32_t), hipMemcpyDeviceToHost, st);
    }
  }
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  if (e == hipSuccess && !exact) {  // tables straight from the device-resident statistics
    rc = map_create_device(ctx, V, d_mean, d_S, d_valid, search_radius_sq, out_map);
    if (rc != NOS_OK) {
      nos_scan_destroy(scan);
      return rc;
    }
  }
  nos_scan_destroy(scan);
  if (e != hipSuccess)
    return fail(e == hipErrorOutOfMemory ? NOS_ERR_OUT_OF_MEMORY : NOS_ERR_HIP, "map build failed: %s", hipGetErrorString(e));
  const int64_t bias = int64_t(1) << 20;
  for (uint32_t v = 0; v < V && want_stats; ++v) {
    if (compact) {
      const uint64_t k = h_keys[v], nyz = uint64_t(dims[1]) * uint64_t(dims[2]);
      stats->cells[3 * size_t(v) + 0] = int64_t(k / nyz) + h_box[0];
      stats->cells[3 * size_t(v) + 1] = int64_t((k % nyz) / uint64_t(dims[2])) + h_box[1];
      stats->cells[3 * size_t(v) + 2] = int64_t(k % uint64_t(dims[2])) + h_box[2];
    } else {
      stats->cells[3 * size_t(v) + 0] = int64_t((h_keys[v] >> 42) & 0x1FFFFFull) - bias;
      stats->cells[3 * size_t(v) + 1] = int64_t((h_keys[v] >> 21) & 0x1FFFFFull) - bias;
      stats->cells[3 * size_t(v) + 2] = int64_t(h_keys[v] & 0x1FFFFFull) - bias;
    }
  }
  if (exact && V > 1) {
    // the reference's map lists its voxels as they were first seen (our restatement of its unordered_map walk): voxel
    // ids — the matcher's tie-break — then agree with the reference-exact CPU restatement's
    std::vector<uint32_t> perm(V);
    for (uint32_t v = 0; v < V; ++v) perm[v] = v;
    std::sort(perm.begin(), perm.end(), [&](uint32_t a, uint32_t b) { return h_first[a] < h_first[b]; });
    auto permute = [&](auto& vec, size_t width) {
      auto old = vec;
      for (uint32_t v = 0; v < V; ++v)
        for (size_t k = 0; k < width; ++k) vec[size_t(v) * width + k] = old[size_t(perm[v]) * width + k];
    };
    permute(stats->means, 3);
    permute(stats->sqrt_infos, 9);
    permute(stats->valid, 1);
    permute(stats->counts, 1);
    permute(stats->cells, 3);
    permute(stats->evals, 3);
    permute(stats->evecs, 9);
  }
  if (exact) {
    rc = nos_ndt_map_create(ctx, V, stats->means.data(), stats->sqrt_infos.data(), stats->valid.data(), search_radius_sq,
                            out_map);
    if (rc != NOS_OK) return rc;
  }
  if (out_stats) *out_stats = stats.release();
  return NOS_OK;
}

}  // extern "C"

