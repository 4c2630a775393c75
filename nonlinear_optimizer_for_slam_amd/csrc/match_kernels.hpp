// match_kernels.hpp — GPU correspondence matcher for NDT scan-to-map.
//
// Restates MatchPointCloud of the reference's test harness
// (nonlinear_optimizer/mahalanobis_distance_minimizer/tests/simple_optimization_test.cc:296-342):
// every scan point is warped by the current pose and matched to its (up to) two nearest valid NDT
// voxel means within the search radius (FLANN radiusSearch on L2_Simple, i.e. SQUARED distance
// < radius, max_neighbors = 2, sorted); each match becomes one correspondence
// {local point, mean, sqrt-information}.  The k-d tree is replaced by a uniform grid with cell
// edge = sqrt(radius): all candidates of a point lie in its 27-cell neighbourhood.  The voxel
// records are stored in cell order so one cell's candidates are contiguous.
//
// Output goes straight into the tiled-SoA dataset the assemble kernels read: two slots per point
// (slot 2i + k = k-th nearest), an absent neighbour is an all-zero record, which contributes
// exactly nothing to H, g and cost — so no compaction pass and no host round trip are needed
// between matching and solving.  HBM-bound integer/pointer work: coalesced point reads, 16-byte
// coalesced record writes, candidate reads served from L2 (the map is small).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "assemble_kernels.hpp"

namespace nos {

constexpr uint64_t kEmptyCell = ~0ull;

struct MapView {
  const double* mean;        // [V][3] in cell order
  const double* sqrt_info;   // [V][9] row-major, cell order
  const uint32_t* orig_id;   // [V] original voxel index (tie-break, diagnostics)
  const uint64_t* cell_key;  // open-addressing table, kEmptyCell = free
  const uint32_t* cell_start;
  const uint32_t* cell_count;
  uint32_t table_mask;       // table size - 1 (power of two)
  double inv_cell;           // 1 / cell edge
  double radius_sq;
  // Dense form of the same grid (set when the map's bounding box is small enough, which it is for every
  // scene the reference handles): cells in lexicographic (x, y, z) order — the order the voxel records are
  // stored in — with `dense_begin[c]` = number of records in cells before c.  The three z-neighbours of a
  // column are consecutive cells, so their records are ONE contiguous run [begin[c-1], begin[c+2]): a point
  // inspects 9 runs (2 loads each) instead of probing a hash table 27 times.
  const uint32_t* dense_begin;  // [nx*ny*nz + 1], null = hash table only
  const double* record;         // [V][4] = mean x, y, z, original id (bit pattern) — one 32-byte candidate record
  int64_t ox, oy, oz;           // cell coordinates of dense cell (0,0,0)
  int32_t nx, ny, nz;
};

__host__ __device__ __forceinline__ uint64_t pack_cell(int64_t ix, int64_t iy, int64_t iz) {
  const uint64_t bias = 1ull << 20;
  return ((uint64_t(ix + int64_t(bias)) & 0x1FFFFFull) << 42) | ((uint64_t(iy + int64_t(bias)) & 0x1FFFFFull) << 21) |
         (uint64_t(iz + int64_t(bias)) & 0x1FFFFFull);
}

__host__ __device__ __forceinline__ uint32_t hash_cell(uint64_t k) {
  k ^= k >> 33;
  k *= 0xff51afd7ed558ccdull;
  k ^= k >> 33;
  k *= 0xc4ceb9fe1a85ec53ull;
  k ^= k >> 33;
  return uint32_t(k);
}

struct PosePod {
  double R[9];
  double t[3];
};

struct TwoNearest {
  double d[2];
  uint32_t j[2];   // position in the map's cell-ordered arrays, 0xFFFFFFFF = none
  uint32_t id[2];  // original voxel id (tie-break)
  __device__ __forceinline__ void init() {
    d[0] = d[1] = 1e300;
    j[0] = j[1] = id[0] = id[1] = 0xFFFFFFFFu;
  }
  // keep the two smallest (distance, original id) pairs, nearest first
  __device__ __forceinline__ void offer(double dist, uint32_t pos, uint32_t orig) {
    if (dist < d[0] || (dist == d[0] && orig < id[0])) {
      d[1] = d[0];
      j[1] = j[0];
      id[1] = id[0];
      d[0] = dist;
      j[0] = pos;
      id[0] = orig;
    } else if (dist < d[1] || (dist == d[1] && orig < id[1])) {
      d[1] = dist;
      j[1] = pos;
      id[1] = orig;
    }
  }
};

// Squared distance, one fixed evaluation order for every form of the search (left to the compiler, "ex ex + ey ey + ez ez"
// fuses a different product in different surroundings: an ulp apart, enough to move a point across the radius test).
__device__ __forceinline__ double match_dist(double ex, double ey, double ez) {
  return __builtin_fma(ez, ez, __builtin_fma(ey, ey, ex * ex));
}

// The (up to) two nearest valid voxel means within the radius of the world point q — FLANN radiusSearch with
// max_neighbors = 2 on squared distances, ties broken by original voxel id.  Same result from both grid forms.
__device__ __forceinline__ void find_two_nearest(const MapView& map, double qx, double qy, double qz, TwoNearest& best) {
  best.init();
  const int64_t cx = int64_t(floor(qx * map.inv_cell));
  const int64_t cy = int64_t(floor(qy * map.inv_cell));
  const int64_t cz = int64_t(floor(qz * map.inv_cell));
  if (map.dense_begin != nullptr) {
    const int64_t rx = cx - map.ox, ry = cy - map.oy, rz = cz - map.oz;
    const int64_t z0 = rz - 1 < 0 ? 0 : rz - 1;
    const int64_t z1 = rz + 1 > map.nz - 1 ? map.nz - 1 : rz + 1;
    if (z0 > z1) return;
    using V2 = double __attribute__((ext_vector_type(2)));
    const V2* rec = reinterpret_cast<const V2*>(map.record);
#pragma unroll
    for (int dx = -1; dx <= 1; ++dx) {
      const int64_t xx = rx + dx;
      if (xx < 0 || xx >= map.nx) continue;
#pragma unroll
      for (int dy = -1; dy <= 1; ++dy) {
        const int64_t yy = ry + dy;
        if (yy < 0 || yy >= map.ny) continue;
        const int64_t col = (xx * map.ny + yy) * map.nz;
        const uint32_t b = map.dense_begin[col + z0];
        const uint32_t e = map.dense_begin[col + z1 + 1];
        for (uint32_t j = b; j < e; ++j) {
          const V2 m01 = rec[2 * size_t(j)];
          const V2 m23 = rec[2 * size_t(j) + 1];
          const double dist = match_dist(qx - m01[0], qy - m01[1], qz - m23[0]);
          if (dist < map.radius_sq) best.offer(dist, j, uint32_t(__double_as_longlong(m23[1])));
        }
      }
    }
    return;
  }
  for (int dz = -1; dz <= 1; ++dz)
    for (int dy = -1; dy <= 1; ++dy)
      for (int dx = -1; dx <= 1; ++dx) {
        const uint64_t key = pack_cell(cx + dx, cy + dy, cz + dz);
        uint32_t h = hash_cell(key) & map.table_mask;
        uint32_t start = 0, count = 0;
        for (uint32_t probe = 0; probe <= map.table_mask; ++probe) {  // bounded: table is never full
          const uint64_t k = map.cell_key[h];
          if (k == key) {
            start = map.cell_start[h];
            count = map.cell_count[h];
            break;
          }
          if (k == kEmptyCell) break;
          h = (h + 1) & map.table_mask;
        }
        for (uint32_t j = start; j < start + count; ++j) {
          const double ex = qx - map.mean[3 * size_t(j)];
          const double ey = qy - map.mean[3 * size_t(j) + 1];
          const double ez = qz - map.mean[3 * size_t(j) + 2];
          const double dist = match_dist(ex, ey, ez);
          if (dist < map.radius_sq) best.offer(dist, j, map.orig_id[j]);
        }
      }
}

// One thread per scan point.  points: 3 planes of n doubles (local frame).
template <typename DST>
__global__ __launch_bounds__(256) void match_kernel(MapView map, const double* __restrict__ px,
                                                    const double* __restrict__ py,
                                                    const double* __restrict__ pz, uint64_t n_points,
                                                    PosePod pose, int max_neighbors, TiledLayout L,
                                                    DST* __restrict__ dst,
                                                    unsigned long long* __restrict__ n_matches) {
  const uint64_t i = uint64_t(blockIdx.x) * 256 + threadIdx.x;
  int found = 0;
  if (i < n_points) {
    const double x = px[i], y = py[i], z = pz[i];
    const double qx = pose.R[0] * x + pose.R[1] * y + pose.R[2] * z + pose.t[0];
    const double qy = pose.R[3] * x + pose.R[4] * y + pose.R[5] * z + pose.t[1];
    const double qz = pose.R[6] * x + pose.R[7] * y + pose.R[8] * z + pose.t[2];
    TwoNearest best;
    find_two_nearest(map, qx, qy, qz, best);
    const uint32_t (&best_j)[2] = best.j;
    // two consecutive slots 2i, 2i+1 → one 2-wide store per field
    const uint64_t i0 = 2 * i;
    const uint64_t off = (i0 >> L.tile_shift) * L.tile_stride + (i0 & L.tile_mask);
    using V2 = DST __attribute__((ext_vector_type(2)));
    const bool ok0 = best_j[0] != 0xFFFFFFFFu;
    const bool ok1 = best_j[1] != 0xFFFFFFFFu && max_neighbors > 1;
    found = int(ok0) + int(ok1);
    const double pl[3] = {x, y, z};
#pragma unroll
    for (int f = 0; f < 3; ++f) {
      V2 v;
      v[0] = ok0 ? DST(pl[f]) : DST(0);
      v[1] = ok1 ? DST(pl[f]) : DST(0);
      *reinterpret_cast<V2*>(dst + off + uint64_t(f) * L.field_stride) = v;
    }
#pragma unroll
    for (int f = 0; f < 3; ++f) {
      V2 v;
      v[0] = ok0 ? DST(map.mean[3 * size_t(best_j[0]) + f]) : DST(0);
      v[1] = ok1 ? DST(map.mean[3 * size_t(best_j[1]) + f]) : DST(0);
      *reinterpret_cast<V2*>(dst + off + uint64_t(3 + f) * L.field_stride) = v;
    }
#pragma unroll
    for (int f = 0; f < 9; ++f) {
      V2 v;
      v[0] = ok0 ? DST(map.sqrt_info[9 * size_t(best_j[0]) + f]) : DST(0);
      v[1] = ok1 ? DST(map.sqrt_info[9 * size_t(best_j[1]) + f]) : DST(0);
      *reinterpret_cast<V2*>(dst + off + uint64_t(6 + f) * L.field_stride) = v;
    }
  }
  // match count: wave sum → one atomic per wave (integer, order independent)
  int s = found;
#pragma unroll
  for (int o = kWave / 2; o > 0; o >>= 1) s += __shfl_xor(s, o, kWave);
  if ((threadIdx.x & (kWave - 1)) == 0 && s > 0) atomicAdd(n_matches, (unsigned long long)s);
}

// Clears the last n_drop NON-EMPTY records of a matcher-written NDT dataset, in slot order: what dropping the tail of
// the reference's compacted correspondence vector does (floor(N/4)*4 of the scalar 3-DoF class,
// MDM/mahalanobis_distance_minimizer_analytic_3dof.cc:33-36).  A record is empty when its sqrt-information is all zero.
// One wave walks backwards from the end, 64 slots at a time; n_drop is small (< 8), so it touches the tail only.
template <typename T>
__global__ __launch_bounds__(64) void drop_last_matches_kernel(T* __restrict__ data, TiledLayout L, uint64_t n_drop) {
  const int lane = threadIdx.x;
  uint64_t remaining = n_drop;
  for (uint64_t pos = L.n; remaining > 0 && pos > 0; pos = pos > 64 ? pos - 64 : 0) {
    const bool in_range = pos > uint64_t(lane);
    const uint64_t i = in_range ? pos - 1 - uint64_t(lane) : 0;
    const uint64_t off = (i >> L.tile_shift) * L.tile_stride + (i & L.tile_mask);
    bool nonempty = false;
    if (in_range)
      for (int f = 6; f < 15; ++f) nonempty = nonempty || data[off + uint64_t(f) * L.field_stride] != T(0);
    const unsigned long long mask = __ballot(nonempty);
    const uint64_t before = uint64_t(__popcll(mask & ((1ull << lane) - 1ull)));  // non-empty slots nearer to the end
    if (nonempty && before < remaining)
      for (int f = 0; f < 15; ++f) data[off + uint64_t(f) * L.field_stride] = T(0);
    const uint64_t found = uint64_t(__popcll(mask));
    remaining -= found < remaining ? found : remaining;
  }
}

// tiled dataset → planar host-order planes (diagnostics / tests)
template <typename SRC>
__global__ __launch_bounds__(256) void untile_kernel(const SRC* __restrict__ src, int n_fields, TiledLayout L,
                                                     double* __restrict__ dst /* [n_fields][L.n] */) {
  const uint64_t i = uint64_t(blockIdx.x) * 256 + threadIdx.x;
  const int f = blockIdx.y;
  if (i >= L.n || f >= n_fields) return;
  const uint64_t off = (i >> L.tile_shift) * L.tile_stride + uint64_t(f) * L.field_stride + (i & L.tile_mask);
  dst[uint64_t(f) * L.n + i] = double(src[off]);
}

}  // namespace nos
