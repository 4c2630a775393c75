// tools/exp/match_block_sort_r04.hpp — a second measured-and-lost form of the matcher's search (round 4); not compiled into the library.
//
// Idea: the vector L1 coalesces a 16-byte load per quarter wave only, and the 16 lanes of a quarter of a cell-sorted scan sit
// in ≈ 8 different MAP cells (the pose turns the scan's own cell order against the map's grid).  So the 256 points of a
// workgroup are sorted by map cell among themselves first (bitonic network in LDS, 36 stages), every lane runs the unchanged
// find_two_nearest for the point it was dealt and hands the result back through LDS.  Parity green (each point's search is
// the same sequence of operations).  Measured on MI355X, 10 M cell-sorted points x 200 k voxels: 2.29-2.34 ms per match
// against 2.01 ms without the sort; L1 line accesses per launch fell only from 1.16e9 to 0.99e9 (profiles/r04_match_summary.json
// against the r04f set of the same call) — the kernel does not wait for L1 THROUGHPUT but for the chain of dependent round
// trips of each lane's walk, which the sort does not shorten and the 72 barriers lengthen.  What replaced it: the run bounds
// requested together and the records fetched four at a time (match_kernels.hpp, find_two_nearest).
#pragma once

#include "../../nonlinear_optimizer_for_slam_amd/csrc/match_kernels.hpp"

namespace nos {

// The search of a whole 256-thread workgroup, with the points re-dealt to the lanes in MAP-cell order first (round 4).
// What bounds the lane-per-point search (profiles/r04_match_summary.json): not HBM and not arithmetic but the vector L1 — 66
// line accesses per point; a 16-byte load is coalesced per quarter wave only, and the 16 lanes of a quarter sit in ≈ 8
// different map cells, because the scan is sorted by cell in ITS OWN frame and the pose turns that order against the map's
// grid.  So the workgroup sorts its 256 points by map cell among themselves (bitonic network in LDS on {cell index relative to
// the workgroup's bounding box | lane}: 36 stages, ≈ 3 k cycles), every lane runs the unchanged find_two_nearest for the point
// it was dealt — the lanes of a quarter wave now share one or two cells, i.e. the same candidate records and the same trip
// counts — and hands the two positions back to the point's own lane through LDS.  Each point's search is the same sequence of
// operations as before, so the results are identical.  A workgroup whose bounding box has more than 65 536 cells (an unsorted
// scan) keeps its points where they are.  Must be called by all 256 threads (`active` = has a point).
__device__ __forceinline__ void find_two_nearest_block(const MapView& map, bool active, double qx, double qy, double qz,
                                                       uint32_t (&best_j)[2]) {
  __shared__ int box[6];          // min x, y, z | max x, y, z of the reachable points' cells
  __shared__ uint32_t skey[256];
  __shared__ double sq[3][256];
  __shared__ uint32_t sres[2][256];
  const uint32_t t = threadIdx.x;
  best_j[0] = best_j[1] = 0xFFFFFFFFu;
  if (map.dense_begin == nullptr) {  // hash-table maps: no grid to sort by
    if (active) {
      TwoNearest best;
      find_two_nearest(map, qx, qy, qz, best);
      best_j[0] = best.j[0];
      best_j[1] = best.j[1];
    }
    return;
  }
  const int64_t rx64 = int64_t(floor(qx * map.inv_cell)) - map.ox;
  const int64_t ry64 = int64_t(floor(qy * map.inv_cell)) - map.oy;
  const int64_t rz64 = int64_t(floor(qz * map.inv_cell)) - map.oz;
  // a point whose 27-cell neighbourhood misses the grid has no candidate at all: it takes no part
  const bool reach = active && rx64 >= -1 && rx64 <= map.nx && ry64 >= -1 && ry64 <= map.ny && rz64 >= -1 && rz64 <= map.nz;
  const int rx = int(rx64), ry = int(ry64), rz = int(rz64);
  if (t < 3) box[t] = 0x7FFFFFFF;
  if (t >= 3 && t < 6) box[t] = -0x7FFFFFFF;
  __syncthreads();
  if (reach) {
    atomicMin(&box[0], rx);
    atomicMin(&box[1], ry);
    atomicMin(&box[2], rz);
    atomicMax(&box[3], rx);
    atomicMax(&box[4], ry);
    atomicMax(&box[5], rz);
  }
  sq[0][t] = qx;
  sq[1][t] = qy;
  sq[2][t] = qz;
  __syncthreads();
  const int bx = box[0], by = box[1], bz = box[2];
  const int64_t dx = int64_t(box[3]) - bx + 1, dy = int64_t(box[4]) - by + 1, dz = int64_t(box[5]) - bz + 1;
  const bool any = box[3] >= bx;                                  // block-uniform: somebody reaches the grid
  const bool sorted = any && dx * dy * dz <= 65536;               // block-uniform
  uint32_t mine = t;  // the lane whose point this lane searches for
  bool search = reach;
  if (sorted) {
    const uint32_t rel = reach ? uint32_t((int64_t(rx - bx) * dy + (ry - by)) * dz + (rz - bz)) : 0xFFFFFFu;
    skey[t] = (rel << 8) | t;
    for (uint32_t k = 2; k <= 256; k <<= 1)
      for (uint32_t j = k >> 1; j > 0; j >>= 1) {
        __syncthreads();
        const uint32_t partner = t ^ j;
        const uint32_t a = skey[t], b = skey[partner];
        const bool ascending = (t & k) == 0;
        const uint32_t keep = ((t < partner) == ascending) ? (a < b ? a : b) : (a < b ? b : a);
        __syncthreads();
        skey[t] = keep;
      }
    __syncthreads();
    const uint32_t sk = skey[t];
    mine = sk & 0xFFu;
    search = (sk >> 8) != 0xFFFFFFu;
  }
  TwoNearest best;
  best.init();
  if (search) find_two_nearest(map, sq[0][mine], sq[1][mine], sq[2][mine], best);
  if (!sorted) {
    best_j[0] = best.j[0];
    best_j[1] = best.j[1];
    return;
  }
  sres[0][mine] = best.j[0];
  sres[1][mine] = best.j[1];
  __syncthreads();
  best_j[0] = sres[0][t];
  best_j[1] = sres[1][t];
}


}  // namespace nos

// A third form, also measured and dropped (round 4): inside find_two_nearest's dense path the eighteen run bounds requested
// together (no load behind a branch) and the candidate records fetched FOUR at a time with clamped indices — ≈ 1 + 18 memory
// round trips per point instead of ≈ 63.  Cell-sorted scan 1.96-2.02 ms (unchanged), unsorted scan 5.2 → 5.75 ms (the
// clamped extra loads cost cache lines it does not have to spare); profiles/r04_match_quad_walk.txt.  Three restructurings of
// the memory side that change nothing say the search is not waiting for memory alone: 2 265 vector instructions per wave are
// 0.57 ms of pure issue at 4 cycles each, the scalar bookkeeping and the divergent `offer` branches of 54 candidates about as
// much again.
