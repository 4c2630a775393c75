// tools/exp/match_wave_search_r04.hpp — a measured-and-lost form of the matcher's search (round 4); not compiled into the library.
//
// Idea: the 64 points of a wave of a cell-sorted scan share a few map cells, so the wave walks each distinct cell's nine
// candidate runs ONCE with wave-uniform indices (run bounds and the 32-byte candidate records through the scalar cache,
// s_load_dwordx8, the next record in flight while the lanes of that cell measure the current one) instead of every lane
// fetching every record for itself.  Results identical to find_two_nearest ((distance, id) is a total order; parity tests
// green with it).  Measured on MI355X, 10 M cell-sorted points x 200 k voxels: 2.01 ms per match against 2.01 ms for the
// lane-per-point search — the scan is sorted by cell in ITS OWN frame, the pose turns that order against the map's grid,
// neighbouring lanes alternate between ≈ 10 map cells, the per-wave test ("lanes whose cell differs from their left
// neighbour's" <= 6) sends nearly every wave down the lane-per-point path; and one cooperative round costs about as many
// instructions as a whole lane-per-point search (≈ 32 per candidate, scalar bookkeeping included), so it only pays with
// <= 2-3 distinct cells per wave, which a rigidly moved scan does not have.  Sorting by MAP cell would need a sort per
// pose (12 ms per 10 M points against a 2 ms match).
//
// The function below is the form that was measured (it replaced the call of find_two_nearest in match_kernel /
// match_index_kernel and must be called by every lane of the wave).
#pragma once

#include "../../nonlinear_optimizer_for_slam_amd/csrc/match_kernels.hpp"

namespace nos {

constexpr int kMatchWaveCells = 6;
using MatchConstU32 = const __attribute__((address_space(4))) uint32_t;
using MatchConstF64 = const __attribute__((address_space(4))) double;

__device__ __forceinline__ void find_two_nearest_wave(const MapView& map, bool active, double qx, double qy, double qz,
                                                      TwoNearest& best) {
  best.init();
  if (map.dense_begin == nullptr) {  // hash-table maps: lane per point
    if (active) find_two_nearest(map, qx, qy, qz, best);
    return;
  }
  const int64_t rx64 = int64_t(floor(qx * map.inv_cell)) - map.ox;
  const int64_t ry64 = int64_t(floor(qy * map.inv_cell)) - map.oy;
  const int64_t rz64 = int64_t(floor(qz * map.inv_cell)) - map.oz;
  // a point whose 27-cell neighbourhood misses the grid has no candidate at all
  const bool reach = active && rx64 >= -1 && rx64 <= map.nx && ry64 >= -1 && ry64 <= map.ny && rz64 >= -1 && rz64 <= map.nz;
  const int32_t rx = reach ? int32_t(rx64) : 0, ry = reach ? int32_t(ry64) : 0, rz = reach ? int32_t(rz64) : 0;
  // (nx + 2)(ny + 2)(nz + 2) <= 27 nx ny nz <= 27 x 2^26 < 2^31: the padded cell index fits 32 bits
  const uint32_t key = (uint32_t(rx + 1) * uint32_t(map.ny + 2) + uint32_t(ry + 1)) * uint32_t(map.nz + 2) + uint32_t(rz + 1);
  const int lane = int(threadIdx.x) & (kWave - 1);
  const uint32_t left_key = uint32_t(__shfl_up(int(key), 1, kWave));
  const bool left_reach = __shfl_up(int(reach), 1, kWave) != 0;
  const bool first_of_its_cell = reach && (lane == 0 || !left_reach || left_key != key);
  if (__popcll(__ballot(first_of_its_cell)) > kMatchWaveCells) {
    if (reach) find_two_nearest(map, qx, qy, qz, best);
    return;
  }
  MatchConstU32* cbegin = (MatchConstU32*)map.dense_begin;
  MatchConstF64* crec = (MatchConstF64*)map.record;
  unsigned long long todo = __ballot(reach);
  while (todo != 0ull) {
    const int leader = __ffsll((long long)todo) - 1;
    const uint32_t lkey = uint32_t(__builtin_amdgcn_readlane(int(key), leader));
    const int32_t lx = __builtin_amdgcn_readlane(rx, leader), ly = __builtin_amdgcn_readlane(ry, leader),
                  lz = __builtin_amdgcn_readlane(rz, leader);
    const bool member = reach && key == lkey;
    const int32_t z0 = lz - 1 < 0 ? 0 : lz - 1, z1 = lz + 1 > map.nz - 1 ? map.nz - 1 : lz + 1;
    // the nine runs' bounds, requested together; the empty asm statements pin every load where it stands (left alone, the
    // compiler moves each one behind the test that uses it and the eighteen scalar-cache round trips happen in sequence)
    uint32_t rb[9], re[9];
#pragma unroll
    for (int c = 0; c < 9; ++c) {
      const int32_t xx = lx + c / 3 - 1, yy = ly + c % 3 - 1;
      const bool ok = xx >= 0 && xx < map.nx && yy >= 0 && yy < map.ny;
      const int64_t col = ok ? (int64_t(xx) * map.ny + yy) * map.nz : 0;
      rb[c] = cbegin[col + z0];
      re[c] = cbegin[col + z1 + 1];
    }
#pragma unroll
    for (int c = 0; c < 9; ++c) asm volatile("" : "+s"(rb[c]), "+s"(re[c]));
#pragma unroll
    for (int c = 0; c < 9; ++c) {
      const int32_t xx = lx + c / 3 - 1, yy = ly + c % 3 - 1;
      const bool ok = xx >= 0 && xx < map.nx && yy >= 0 && yy < map.ny;
      re[c] = ok ? re[c] : rb[c];
    }
#pragma unroll
    for (int c = 0; c < 9; ++c) {
      if (rb[c] >= re[c]) continue;
      unsigned long long cur[4], nxt[4];
      auto fetch = [&](uint32_t j, unsigned long long (&r)[4]) {
        const __attribute__((address_space(4))) unsigned long long* q =
            (const __attribute__((address_space(4))) unsigned long long*)crec + 4 * size_t(j);
#pragma unroll
        for (int m = 0; m < 4; ++m) r[m] = q[m];
      };
      fetch(rb[c], cur);
      for (uint32_t j = rb[c]; j < re[c]; ++j) {
        fetch(j + 1 < re[c] ? j + 1 : j, nxt);
        asm volatile("" : "+s"(cur[0]), "+s"(cur[1]), "+s"(cur[2]), "+s"(cur[3]));
        if (member) {
          const double dist = match_dist(qx - __longlong_as_double((long long)cur[0]), qy - __longlong_as_double((long long)cur[1]),
                                         qz - __longlong_as_double((long long)cur[2]));
          if (dist < map.radius_sq) best.offer(dist, j, uint32_t(cur[3]));
        }
#pragma unroll
        for (int m = 0; m < 4; ++m) cur[m] = nxt[m];
      }
    }
    todo &= ~__ballot(member);
  }
}

}  // namespace nos
