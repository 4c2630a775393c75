// nos_pgo.hip — host side of the pose-graph entry points.  Kernels: pgo_kernels.hpp.
#include "nos_internal.hpp"

#include "pgo_kernels.hpp"
#include "pgo_coarse_kernels.hpp"

using namespace nosd;

struct nos_pose_graph {
  nos_ctx* ctx = nullptr;
  uint32_t n_poses = 0, n_edges = 0;
  size_t n_unknowns = 0;  // 6 n_poses + n_edges
  uint32_t n_free_switches = 0;
  // graph (gathered data as 64-byte records, see pgo_kernels.hpp)
  double* d_pose = nullptr;       // [n_poses][8]: px py pz qw qx qy qz pad
  int32_t *d_ref = nullptr, *d_qry = nullptr;
  double* d_edge = nullptr;       // [n_edges][8]: t_m (3) q_m wxyz (4) switch (1)
  uint8_t *d_sw_free = nullptr, *d_fixed = nullptr;
  uint32_t *d_adj_off = nullptr, *d_adj = nullptr, *d_adj_nbr = nullptr;
  // linear system
  double *d_hdiag = nullptr, *d_minv = nullptr;  // 21 planes each
  double* d_hs = nullptr;                         // switch curvature
  double *d_grad = nullptr, *d_x = nullptr, *d_r = nullptr, *d_z = nullptr, *d_p = nullptr, *d_ap = nullptr;
  double* d_p2 = nullptr;  // second direction buffer: the block-local product writes p_new = z + beta p while it reads p
  double *d_partials = nullptr, *d_scalars = nullptr;
  unsigned int* d_tickets = nullptr;  // [2] arrival counters of the in-launch tails (product, preconditioner); zero between launches
  double* h_scalars = nullptr;  // pinned [4]
  uint32_t partial_blocks = 0;
  nos::PgoView view{};
  // block-local product (pgo_matvec_block_kernel): per-block entry lists; block_poses = 0 → owner-computes kernels
  double* d_bent = nullptr;        // [n_entries][10]
  uint32_t* d_bent_off = nullptr;  // [n_blocks + 1]
  uint32_t* d_bent_edge = nullptr; // [n_entries] constraint of every entry (switch refresh after a retract)
  uint32_t* d_bhalo = nullptr;     // halo pose ids, block after block
  uint32_t* d_bhalo_off = nullptr; // [n_blocks + 1]
  uint32_t block_poses = 0, n_blocks = 0, n_entries = 0;
  size_t n_halo = 0;               // halo poses over all blocks
  int product_grid = 0;            // workgroups of the block-local product (what is resident), decided at the first launch
  nos::PgoBlockView bview{};
  // coarse level of the two-level preconditioner (pgo_coarse_kernels.hpp), allocated on first use
  uint32_t agg = 0, n_agg = 0, pcr_levels = 0;
  size_t pcr_pitch = 0;  // elements between two planes of the PCR factors (see PcrLevelLayout)
  double* d_coarse = nullptr;  // one allocation: L/D/U x 2, Dinv, alpha/gamma per level, rhs x 2
  double *c_L[2] = {nullptr, nullptr}, *c_D[2] = {nullptr, nullptr}, *c_U[2] = {nullptr, nullptr};
  double *c_Dinv = nullptr, *c_alpha = nullptr, *c_gamma = nullptr, *c_b[2] = {nullptr, nullptr};
};

namespace {

constexpr uint32_t kPgoDotBlocks = 1024;

int pgo_read_scalars(nos_pose_graph* pg, int count, double* out) {
  DeviceSlot& slot = pg->ctx->slots[0];
  NOS_HIP_CHECK(hipMemcpyAsync(pg->h_scalars, pg->d_scalars, sizeof(double) * count, hipMemcpyDeviceToHost, slot.stream));
  NOS_HIP_CHECK(hipStreamSynchronize(slot.stream));
  for (int k = 0; k < count; ++k) out[k] = pg->h_scalars[k];
  return NOS_OK;
}

int pgo_dot(nos_pose_graph* pg, const double* a, const double* b, double* out) {
  DeviceSlot& slot = pg->ctx->slots[0];
  const uint32_t blocks = uint32_t(std::min<size_t>(kPgoDotBlocks, (pg->n_unknowns + 255) / 256));
  hipLaunchKernelGGL(nos::pgo_dot_kernel, dim3(blocks), dim3(256), 0, slot.stream, a, b, pg->n_unknowns, pg->d_partials);
  hipLaunchKernelGGL(nos::pgo_sum_partials_kernel, dim3(1), dim3(1024), 0, slot.stream, pg->d_partials, blocks, 1,
                     pg->d_scalars);
  NOS_HIP_CHECK(hipGetLastError());
  return pgo_read_scalars(pg, 1, out);
}

int pgo_launch_product(nos_pose_graph* pg, double lambda, const double* x, double* y, const double* z = nullptr,
                       double* x_new = nullptr);

// y = A x on the device and x.y to the host (host-read CG scalars, option pgo_host_scalars): the same product kernel and the
// same in-launch sum as the device-resident form, the total read back from the scalar block.
int pgo_matvec(nos_pose_graph* pg, double lambda, const double* x, double* y, double* x_dot_y) {
  const int rc = pgo_launch_product(pg, lambda, x, y);
  if (rc != NOS_OK || x_dot_y == nullptr) return rc;
  return pgo_read_scalars(pg, 1, x_dot_y);
}

// The product of a PCG iteration with its in-launch tail (p.Ap, alpha): block-local form when the graph has its entry lists.
// z / x_new (block-local form only): the vector multiplied is x_new = z + beta x, formed in the same launch (see the kernel).
int pgo_launch_product(nos_pose_graph* pg, double lambda, const double* x, double* y, const double* z, double* x_new) {
  DeviceSlot& slot = pg->ctx->slots[0];
  const nos::PgoTail tail{pg->d_partials, pg->d_tickets, pg->d_scalars};
  const int switch_rows = pg->n_free_switches > 0 ? 1 : 0;
  if (pg->block_poses != 0) {
    constexpr int kP = 128, kT = 256;
    const size_t lds = ((size_t(kP) + pg->bview.halo_cap) * 14 + size_t(pg->bview.slot_cap) * 6) * sizeof(double);
    auto* kernel = switch_rows ? nos::pgo_matvec_block_kernel<kP, kT, true> : nos::pgo_matvec_block_kernel<kP, kT, false>;
    if (lds > size_t(48) * 1024) {
      const hipError_t ea = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds));
      if (ea != hipSuccess) return fail(NOS_ERR_HIP, "pose-graph product: %zu bytes of LDS refused: %s", lds, hipGetErrorString(ea));
    }
    if (pg->product_grid == 0) {  // persistent workgroups: as many as are resident together
      int per_cu = 0;
      const hipError_t eo = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(kernel), kT, lds);
      if (eo != hipSuccess || per_cu < 1) per_cu = 1;
      pg->product_grid = int(std::min<size_t>(pg->n_blocks, size_t(per_cu) * size_t(slot.num_cus)));
    }
    hipLaunchKernelGGL(kernel, dim3(unsigned(pg->product_grid)), dim3(kT), lds, slot.stream, pg->view, pg->bview, pg->d_hdiag,
                       pg->d_hs, lambda, x, y, tail, z, x_new);
  } else {
    const uint32_t pose_blocks = (pg->n_poses + 255) / 256;
    const uint32_t mv_blocks = pose_blocks + (switch_rows ? (pg->n_edges + 255) / 256 : 0u);
    hipLaunchKernelGGL(nos::pgo_matvec_cg_kernel, dim3(mv_blocks), dim3(256), 0, slot.stream, pg->view, pg->d_hdiag, pg->d_hs,
                       lambda, x, y, pose_blocks, tail);
  }
  NOS_HIP_CHECK(hipGetLastError());
  return NOS_OK;
}

// ---- coarse level (pgo_coarse_kernels.hpp)

// The PCR levels run in spans of up to kPcrSpanLevels levels per launch (pgo_pcr_span_kernel); a span whose residue classes
// fit one workgroup takes all remaining levels.  → first level of the span that contains level l.
constexpr uint32_t kPcrSpanLevels = 4, kPcrSpanThreads = 128;
uint32_t pcr_span_start(uint32_t n_agg, uint32_t levels, uint32_t l) {
  uint32_t s = 0;
  for (;;) {
    const uint32_t rows = uint32_t((uint64_t(n_agg) + (1ull << s) - 1) >> s);
    if (rows <= kPcrSpanThreads) return s;  // the last span: levels [s, levels)
    const uint32_t k = std::min(kPcrSpanLevels, levels - s);
    if (l < s + k) return s;
    s += k;
  }
}
nos::PcrLevelLayout pcr_layout(uint32_t n_agg, uint32_t levels, uint32_t l) {
  const uint32_t shift = pcr_span_start(n_agg, levels, l);
  return nos::PcrLevelLayout{shift, uint32_t((uint64_t(n_agg) + (1ull << shift) - 1) >> shift)};
}

int pgo_coarse_alloc(nos_pose_graph* pg, uint32_t agg) {
  if (pg->d_coarse != nullptr && pg->agg == agg) return NOS_OK;
  if (pg->d_coarse != nullptr) {
    (void)hipFree(pg->d_coarse);
    pg->d_coarse = nullptr;
  }
  pg->agg = agg;
  pg->n_agg = (pg->n_poses + agg - 1) / agg;
  pg->pcr_levels = 0;
  while ((1u << pg->pcr_levels) < pg->n_agg) ++pg->pcr_levels;
  const size_t blk = size_t(36) * pg->n_agg, vec = size_t(6) * pg->n_agg;
  pg->pcr_pitch = pg->n_agg;
  for (uint32_t l = 0; l < pg->pcr_levels; ++l) {
    const nos::PcrLevelLayout lay = pcr_layout(pg->n_agg, pg->pcr_levels, l);
    pg->pcr_pitch = std::max(pg->pcr_pitch, (size_t(1) << lay.shift) * lay.rows);
  }
  const size_t fac = size_t(36) * pg->pcr_pitch;  // one level's alpha (or gamma) planes
  const size_t total = 6 * blk + blk + 2 * size_t(std::max<uint32_t>(pg->pcr_levels, 1)) * fac + 2 * vec;
  NOS_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&pg->d_coarse), total * sizeof(double)));
  double* p = pg->d_coarse;
  for (int k = 0; k < 2; ++k) {
    pg->c_L[k] = p;
    p += blk;
    pg->c_D[k] = p;
    p += blk;
    pg->c_U[k] = p;
    p += blk;
  }
  pg->c_Dinv = p;
  p += blk;
  pg->c_alpha = p;
  p += size_t(std::max<uint32_t>(pg->pcr_levels, 1)) * fac;
  pg->c_gamma = p;
  p += size_t(std::max<uint32_t>(pg->pcr_levels, 1)) * fac;
  pg->c_b[0] = p;
  p += vec;
  pg->c_b[1] = p;
  return NOS_OK;
}

// A_c = P^T H' P (assembled directly; option pgo_coarse_probe: probed with 18 masked products), then the PCR elimination.
// The probing form uses d_p / d_ap as scratch: call before the CG vectors are initialised.
int pgo_coarse_setup(nos_pose_graph* pg, double lambda, uint32_t agg) {
  int rc = pgo_coarse_alloc(pg, agg);
  if (rc != NOS_OK) return rc;
  DeviceSlot& slot = pg->ctx->slots[0];
  const uint32_t N = pg->n_poses, C = pg->n_agg;
  const uint32_t pose_blocks = (N + 255) / 256, cblocks = (C + 127) / 128;
  const size_t N6 = size_t(6) * N;
  if (pg->ctx->settings.pgo_coarse_probe == 0) {
    // the three block diagonals of A_c = P^T H' P, assembled directly: one sweep over the constraints each
    const dim3 grid((C + 3) / 4);
    hipLaunchKernelGGL(nos::pgo_coarse_assemble_kernel<0>, grid, dim3(256), 0, slot.stream, pg->view, pg->d_hdiag, lambda, agg, C,
                       pg->c_D[0]);
    hipLaunchKernelGGL(nos::pgo_coarse_assemble_kernel<1>, grid, dim3(256), 0, slot.stream, pg->view, pg->d_hdiag, lambda, agg, C,
                       pg->c_L[0]);
    hipLaunchKernelGGL(nos::pgo_coarse_assemble_kernel<2>, grid, dim3(256), 0, slot.stream, pg->view, pg->d_hdiag, lambda, agg, C,
                       pg->c_U[0]);
  } else {
    // round 2's way, kept for comparison (option pgo_coarse_probe): A_c probed with 18 masked matrix-free products
    NOS_HIP_CHECK(hipMemsetAsync(pg->c_L[0], 0, size_t(3) * 36 * C * sizeof(double), slot.stream));  // L, D, U of buffer 0
    NOS_HIP_CHECK(hipMemsetAsync(pg->d_p + N6, 0, size_t(pg->n_edges) * sizeof(double), slot.stream));  // switch part of the probe
    for (int colour = 0; colour < 3; ++colour)
      for (int dof = 0; dof < 6; ++dof) {
        hipLaunchKernelGGL(nos::pgo_coarse_probe_kernel, dim3(pose_blocks), dim3(256), 0, slot.stream, pg->view, agg, colour, dof,
                           pg->d_p);
        hipLaunchKernelGGL(nos::pgo_matvec_pose_kernel, dim3(pose_blocks), dim3(256), 0, slot.stream, pg->view, pg->d_hdiag,
                           lambda, pg->d_p, pg->d_p + N6, pg->d_ap, pg->d_partials, agg);
        hipLaunchKernelGGL(nos::pgo_coarse_restrict_kernel, dim3((C + 3) / 4), dim3(256), 0, slot.stream, pg->view, agg, C,
                           pg->d_ap, pg->c_b[0]);
        hipLaunchKernelGGL(nos::pgo_coarse_scatter_kernel, dim3(cblocks), dim3(128), 0, slot.stream, C, colour, dof, pg->c_b[0],
                           pg->c_L[0], pg->c_D[0], pg->c_U[0]);
      }
  }
  hipLaunchKernelGGL(nos::pgo_coarse_symmetrize_kernel, dim3(cblocks), dim3(128), 0, slot.stream, C, pg->c_L[0], pg->c_D[0],
                     pg->c_U[0]);
  int cur = 0;
  for (uint32_t lvl = 0; lvl < pg->pcr_levels; ++lvl) {
    hipLaunchKernelGGL(nos::pgo_pcr_setup_kernel, dim3(cblocks), dim3(128), 0, slot.stream, C, 1u << lvl, pg->c_L[cur],
                       pg->c_D[cur], pg->c_U[cur], pg->c_L[1 - cur], pg->c_D[1 - cur], pg->c_U[1 - cur],
                       pg->c_alpha + size_t(lvl) * 36 * pg->pcr_pitch, pg->c_gamma + size_t(lvl) * 36 * pg->pcr_pitch,
                       pcr_layout(C, pg->pcr_levels, lvl), pg->pcr_pitch);
    cur = 1 - cur;
  }
  hipLaunchKernelGGL(nos::pgo_pcr_finish_kernel, dim3(cblocks), dim3(128), 0, slot.stream, C, pg->c_D[cur], pg->c_Dinv);
  NOS_HIP_CHECK(hipGetLastError());
  return NOS_OK;
}

// Right-hand side in c_b[0] → xc = A_c^-1 rhs; returns the buffer holding xc.  The ⌈log2 n_agg⌉ PCR levels run in
// spans (pcr_span_start): at 1 M poses / 20 834 aggregates three launches — levels 0-3 and 4-7 with halos (213 and 224
// workgroups of 128 rows), then levels 8-14 and the block solves inside one workgroup per residue class (256) — instead of 16.
int pgo_pcr_rhs(nos_pose_graph* pg, const double** xc) {
  DeviceSlot& slot = pg->ctx->slots[0];
  const uint32_t C = pg->n_agg, L = pg->pcr_levels;
  constexpr uint32_t kT = kPcrSpanThreads;
  int cur = 0;
  uint32_t l = 0;
  for (;;) {
    const uint32_t rows = uint32_t((uint64_t(C) + (1ull << l) - 1) >> l);  // rows of the largest residue class
    const double* a = pg->c_alpha;
    const double* g = pg->c_gamma;
    const nos::PcrLevelLayout lay{l, rows};
    if (rows <= kT) {  // the rest of the levels and the block solves, one workgroup per class
      hipLaunchKernelGGL(nos::pgo_pcr_span_kernel<kT>, dim3(1u << l), dim3(kT), 0, slot.stream, C, l, L, 0u, a, g,
                         pg->pcr_pitch, lay, pg->c_Dinv, pg->c_b[cur], pg->c_b[1 - cur]);
      cur = 1 - cur;
      break;
    }
    const uint32_t k = std::min(kPcrSpanLevels, L - l), halo = (1u << k) - 1u, per = kT - 2 * halo;
    const uint32_t chunks = (rows + per - 1) / per;
    hipLaunchKernelGGL(nos::pgo_pcr_span_kernel<kT>, dim3(chunks << l), dim3(kT), 0, slot.stream, C, l, l + k, halo, a, g,
                       pg->pcr_pitch, lay, static_cast<const double*>(nullptr), pg->c_b[cur], pg->c_b[1 - cur]);
    cur = 1 - cur;
    l += k;
  }
  NOS_HIP_CHECK(hipGetLastError());
  *xc = pg->c_b[cur];
  return NOS_OK;
}

// xc = A_c^-1 P^T r  →  returns the buffer holding xc.
int pgo_coarse_apply(nos_pose_graph* pg, const double* r, const double** xc) {
  DeviceSlot& slot = pg->ctx->slots[0];
  const uint32_t C = pg->n_agg;
  hipLaunchKernelGGL(nos::pgo_coarse_restrict_kernel, dim3((C + 3) / 4), dim3(256), 0, slot.stream, pg->view, pg->agg, C, r,
                     pg->c_b[0]);
  return pgo_pcr_rhs(pg, xc);
}

}  // namespace

extern "C" {

int nos_pgo_destroy(nos_pose_graph* pg) {
  nosd::CtxGuard guard_(pg ? pg->ctx : nullptr);  // one solve / accumulate / create at a time per context
  if (!pg) return NOS_OK;
  void* bufs[] = {pg->d_pose, pg->d_ref, pg->d_qry, pg->d_edge, pg->d_adj_nbr, pg->d_sw_free, pg->d_fixed, pg->d_adj_off,
                  pg->d_adj, pg->d_hdiag, pg->d_minv, pg->d_hs, pg->d_grad, pg->d_x, pg->d_r, pg->d_z, pg->d_p,
                  pg->d_ap, pg->d_p2, pg->d_partials, pg->d_scalars, pg->d_coarse, pg->d_tickets, pg->d_bent, pg->d_bent_off, pg->d_bent_edge, pg->d_bhalo, pg->d_bhalo_off};
  for (void* b : bufs)
    if (b) (void)hipFree(b);
  if (pg->h_scalars) (void)hipHostFree(pg->h_scalars);
  delete pg;
  return NOS_OK;
}

// poses: [n_poses][7] = px py pz qw qx qy qz (PoseParameter, pose_graph_optimizer.h:16-19);
// meas:  [n_edges][7] = relative_pose_from_reference_to_query as t (3) + quaternion wxyz (types.h:13-19);
// switch_init / switch_free / fixed may be NULL (1.0 / none free / none fixed).
int nos_pgo_create(nos_ctx* ctx, size_t n_poses, const double* poses, size_t n_edges, const int32_t* ref,
                   const int32_t* qry, const double* meas, const double* switch_init,
                   const unsigned char* switch_free, const unsigned char* fixed, nos_pose_graph** out_pg) {
  nosd::CtxGuard guard_(ctx);  // one solve / accumulate / create at a time per context
  if (!ctx || !out_pg) return fail(NOS_ERR_INVALID_ARGUMENT, "ctx / out_pg is NULL");
  *out_pg = nullptr;
  if (ctx->slots.size() != 1) return fail(NOS_ERR_UNSUPPORTED, "pose-graph optimisation needs a single-device context");
  if (n_poses == 0 || n_poses > 0x7FFFFFFFull || n_edges > 0x7FFFFFFFull) return fail(NOS_ERR_INVALID_ARGUMENT, "bad graph size");
  if (!poses || (n_edges > 0 && (!ref || !qry || !meas))) return fail(NOS_ERR_INVALID_ARGUMENT, "graph arrays are NULL");
  for (size_t e = 0; e < n_edges; ++e)
    if (ref[e] < 0 || qry[e] < 0 || size_t(ref[e]) >= n_poses || size_t(qry[e]) >= n_poses || ref[e] == qry[e])
      return fail(NOS_ERR_INVALID_ARGUMENT, "constraint %zu has an invalid pose index", e);  // "Constraint is invalid."
  const uint32_t N = uint32_t(n_poses), M = uint32_t(n_edges);
  // host-side 64-byte records + adjacency (once per graph)
  std::vector<double> pose_rec(size_t(8) * N, 0.0), edge_rec(size_t(8) * std::max<uint32_t>(M, 1), 0.0);
  for (uint32_t i = 0; i < N; ++i) {
    for (int k = 0; k < 7; ++k) pose_rec[size_t(8) * i + k] = poses[size_t(7) * i + k];
    pose_rec[size_t(8) * i + 7] = (fixed && fixed[i]) ? 1.0 : 0.0;  // element 7 = "fixed" flag (block-local product)
  }
  std::vector<uint8_t> swf(std::max<uint32_t>(M, 1), 0), fx(N, 0);
  uint32_t n_free = 0;
  for (uint32_t e = 0; e < M; ++e) {
    for (int k = 0; k < 7; ++k) edge_rec[size_t(8) * e + k] = meas[size_t(7) * e + k];
    edge_rec[size_t(8) * e + 7] = switch_init ? switch_init[e] : 1.0;
    if (switch_free && switch_free[e]) {
      swf[e] = 1;
      ++n_free;
    }
  }
  if (fixed)
    for (uint32_t i = 0; i < N; ++i) fx[i] = fixed[i] ? 1 : 0;
  std::vector<uint32_t> adj_off(size_t(N) + 1, 0), adj(std::max<size_t>(size_t(2) * M, 1)),
      adj_nbr(std::max<size_t>(size_t(2) * M, 1));
  for (uint32_t e = 0; e < M; ++e) {
    ++adj_off[size_t(ref[e]) + 1];
    ++adj_off[size_t(qry[e]) + 1];
  }
  for (uint32_t i = 0; i < N; ++i) adj_off[i + 1] += adj_off[i];
  {
    std::vector<uint32_t> cursor(adj_off.begin(), adj_off.end() - 1);
    for (uint32_t e = 0; e < M; ++e) {  // constraint order → deterministic accumulation order per pose
      adj_nbr[cursor[ref[e]]] = uint32_t(qry[e]);
      adj[cursor[ref[e]]++] = 2 * e;
      adj_nbr[cursor[qry[e]]] = uint32_t(ref[e]);
      adj[cursor[qry[e]]++] = 2 * e + 1;
    }
  }
  // Block-local product: per block of kBlockPoses consecutive poses, the constraints that touch it, in constraint order,
  // each with the adjacency slots (positions relative to the block's first adjacency entry) of its ends inside the block and
  // with its two poses as indices LOCAL to the block: < kBlockPoses = own pose, else kBlockPoses + position in the block's
  // halo list (the other blocks' poses its constraints refer to, ascending).
  constexpr uint32_t kBlockPoses = 128, kSlotLimit = 2304, kHaloLimit = 512;  // LDS: (128 + 512) x 112 B + 2304 x 48 B = 182 KB
                                                                               // is the refusal line; a trajectory graph needs ≈ 70 KB
  const uint32_t n_blocks = (N + kBlockPoses - 1) / kBlockPoses;
  std::vector<uint32_t> bent_off(size_t(n_blocks) + 1, 0), bhalo_off(size_t(n_blocks) + 1, 0);
  struct Piece {  // what one host thread produced for its contiguous range of blocks [first_block, …)
    uint32_t first_block = 0;
    std::vector<double> ent;  // [entries][10]
    std::vector<uint32_t> edge, halo;
  };
  std::vector<Piece> pieces;
  uint32_t slot_cap = 0, halo_cap = 0;
  bool blocks_ok = ctx->settings.pgo_block != 0 && M > 0;
  if (blocks_ok) {
    for (uint32_t b = 0; b < n_blocks; ++b) {
      const uint32_t lo = b * kBlockPoses, hi = std::min(N, lo + kBlockPoses);
      slot_cap = std::max(slot_cap, adj_off[hi] - adj_off[lo]);
    }
    blocks_ok = slot_cap <= kSlotLimit;
  }
  if (blocks_ok) {
    // ONE pass over the blocks, spread over a few host threads (blocks are independent; a thread takes a contiguous range of
    // blocks, so what it produces is a contiguous piece of every list), then the pieces are laid end to end
    struct Ent {
      uint32_t e, slot_r, slot_q;
    };
    const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
    const unsigned n_threads = std::min<unsigned>(std::min(16u, hw), std::max(1u, n_blocks / 64u));
    const uint32_t per = (n_blocks + n_threads - 1) / n_threads;
    pieces.resize(n_threads);
    std::atomic<bool> too_large{false};
    {
      std::vector<std::thread> pool;
      for (unsigned w = 0; w < n_threads; ++w)
        pool.emplace_back([&, w]() {
          const uint32_t b0 = std::min(n_blocks, w * per), b1 = std::min(n_blocks, b0 + per);
          Piece& out = pieces[w];
          out.first_block = b0;
          std::vector<Ent> ents;
          std::vector<uint32_t> halo;
          for (uint32_t b = b0; b < b1 && !too_large.load(std::memory_order_relaxed); ++b) {
            const uint32_t lo = b * kBlockPoses, hi = std::min(N, lo + kBlockPoses), a0 = adj_off[lo];
            ents.clear();
            halo.clear();
            for (uint32_t a = a0; a < adj_off[hi]; ++a) {
              const uint32_t e = adj[a] >> 1, role = adj[a] & 1u;
              ents.push_back({e, role == 0 ? a - a0 : nos::kPgoNoSlot, role == 0 ? nos::kPgoNoSlot : a - a0});
              const uint32_t other = adj_nbr[a];
              if (other < lo || other >= hi) halo.push_back(other);
            }
            // constraint order; the two adjacency positions of a constraint with both ends in the block become one entry
            std::sort(ents.begin(), ents.end(), [](const Ent& x, const Ent& y) { return x.e < y.e; });
            size_t n_out = 0;
            for (size_t k = 0; k < ents.size(); ++k) {
              if (n_out > 0 && ents[n_out - 1].e == ents[k].e) {
                if (ents[k].slot_r != nos::kPgoNoSlot) ents[n_out - 1].slot_r = ents[k].slot_r;
                if (ents[k].slot_q != nos::kPgoNoSlot) ents[n_out - 1].slot_q = ents[k].slot_q;
              } else {
                ents[n_out++] = ents[k];
              }
            }
            ents.resize(n_out);
            std::sort(halo.begin(), halo.end());
            halo.erase(std::unique(halo.begin(), halo.end()), halo.end());
            bent_off[b + 1] = uint32_t(ents.size());
            bhalo_off[b + 1] = uint32_t(halo.size());
            if (halo.size() > kHaloLimit) too_large.store(true);
            auto local = [&](uint32_t pose) -> uint32_t {
              if (pose >= lo && pose < hi) return pose - lo;
              return kBlockPoses + uint32_t(std::lower_bound(halo.begin(), halo.end(), pose) - halo.begin());
            };
            size_t k = out.edge.size();
            out.ent.resize(size_t(10) * (k + ents.size()));
            out.edge.resize(k + ents.size());
            for (const Ent& en : ents) {
              double* rec = out.ent.data() + size_t(10) * k;
              for (int q = 0; q < 8; ++q) rec[q] = edge_rec[size_t(8) * en.e + q];
              const unsigned long long w8 = (unsigned long long)en.e | ((unsigned long long)local(uint32_t(ref[en.e])) << 32) |
                                            ((unsigned long long)local(uint32_t(qry[en.e])) << 48);
              const unsigned long long w9 = (unsigned long long)en.slot_r | ((unsigned long long)en.slot_q << 16);
              memcpy(&rec[8], &w8, sizeof w8);
              memcpy(&rec[9], &w9, sizeof w9);
              out.edge[k++] = en.e;
            }
            out.halo.insert(out.halo.end(), halo.begin(), halo.end());
          }
        });
      for (std::thread& th : pool) th.join();
    }
    blocks_ok = !too_large.load();
    if (blocks_ok) {
      for (uint32_t b = 0; b < n_blocks; ++b) {
        halo_cap = std::max(halo_cap, bhalo_off[b + 1]);
        bent_off[b + 1] += bent_off[b];
        bhalo_off[b + 1] += bhalo_off[b];
      }
    }
  }
  nos_pose_graph* pg = new (std::nothrow) nos_pose_graph();
  if (!pg) return fail(NOS_ERR_OUT_OF_MEMORY, "host allocation failed");
  pg->ctx = ctx;
  pg->n_poses = N;
  pg->n_edges = M;
  pg->n_unknowns = size_t(6) * N + M;
  pg->n_free_switches = n_free;
  pg->partial_blocks = std::max<uint32_t>(kPgoDotBlocks, (N + 255) / 256 + (M + 255) / 256 + 2);
  hipError_t e = hipSetDevice(ctx->slots[0].device);
  if (e == hipSuccess) e = upload(&pg->d_pose, pose_rec);
  if (e == hipSuccess) e = upload(&pg->d_edge, edge_rec);
  if (e == hipSuccess) e = upload(&pg->d_adj_nbr, adj_nbr);
  if (e == hipSuccess) e = upload(&pg->d_sw_free, swf);
  if (e == hipSuccess) e = upload(&pg->d_fixed, fx);
  if (e == hipSuccess) e = upload(&pg->d_adj_off, adj_off);
  if (e == hipSuccess) e = upload(&pg->d_adj, adj);
  if (blocks_ok) {
    // the pieces go to the device end to end (no host-side concatenation); one spare element behind the entry and halo
    // lists: the product kernel's unconditional (clamped) loads of a block without entries or without a halo then stay
    // inside the allocation
    const size_t n_ent_total = bent_off[n_blocks], n_halo_total = bhalo_off[n_blocks];
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&pg->d_bent), (n_ent_total + 1) * 10 * sizeof(double));
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&pg->d_bent_edge), std::max<size_t>(n_ent_total, 1) * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&pg->d_bhalo), (n_halo_total + 1) * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMemset(pg->d_bent + n_ent_total * 10, 0, 10 * sizeof(double));
    if (e == hipSuccess) e = hipMemset(pg->d_bhalo + n_halo_total, 0, sizeof(uint32_t));
    for (const Piece& pc : pieces) {
      const size_t eo = bent_off[pc.first_block], ho = bhalo_off[pc.first_block];
      if (e == hipSuccess && !pc.edge.empty()) {
        e = hipMemcpy(pg->d_bent + eo * 10, pc.ent.data(), pc.ent.size() * sizeof(double), hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(pg->d_bent_edge + eo, pc.edge.data(), pc.edge.size() * sizeof(uint32_t), hipMemcpyHostToDevice);
      }
      if (e == hipSuccess && !pc.halo.empty())
        e = hipMemcpy(pg->d_bhalo + ho, pc.halo.data(), pc.halo.size() * sizeof(uint32_t), hipMemcpyHostToDevice);
    }
    if (e == hipSuccess) e = upload(&pg->d_bent_off, bent_off);
    if (e == hipSuccess) e = upload(&pg->d_bhalo_off, bhalo_off);
    pg->block_poses = kBlockPoses;
    pg->n_blocks = n_blocks;
    pg->n_entries = uint32_t(n_ent_total);
    pg->n_halo = n_halo_total;
  }
  {
    std::vector<int32_t> r(ref, ref + M), q(qry, qry + M);
    if (e == hipSuccess) e = upload(&pg->d_ref, r);
    if (e == hipSuccess) e = upload(&pg->d_qry, q);
  }
  auto dalloc = [&](double** p, size_t count) {
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(p), std::max<size_t>(count, 1) * sizeof(double));
    if (e == hipSuccess) e = hipMemset(*p, 0, std::max<size_t>(count, 1) * sizeof(double));
  };
  dalloc(&pg->d_hdiag, size_t(21) * N);
  dalloc(&pg->d_minv, size_t(21) * N);
  dalloc(&pg->d_hs, M);
  dalloc(&pg->d_grad, pg->n_unknowns);
  dalloc(&pg->d_x, pg->n_unknowns);
  dalloc(&pg->d_r, pg->n_unknowns);
  dalloc(&pg->d_z, pg->n_unknowns);
  dalloc(&pg->d_p, pg->n_unknowns);
  dalloc(&pg->d_ap, pg->n_unknowns);
  dalloc(&pg->d_p2, pg->n_unknowns);
  dalloc(&pg->d_partials, size_t(2) * pg->partial_blocks);
  dalloc(&pg->d_scalars, 8);
  if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&pg->d_tickets), 16);
  if (e == hipSuccess) e = hipMemset(pg->d_tickets, 0, 16);
  if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void**>(&pg->h_scalars), sizeof(double) * 8, hipHostMallocDefault);
  if (e != hipSuccess) {
    nos_pgo_destroy(pg);
    return fail(e == hipErrorOutOfMemory ? NOS_ERR_OUT_OF_MEMORY : NOS_ERR_HIP, "pose-graph upload failed: %s", hipGetErrorString(e));
  }
  nos::PgoView& G = pg->view;
  G.pose = pg->d_pose;
  G.edge = pg->d_edge;
  G.ref = pg->d_ref;
  G.qry = pg->d_qry;
  G.sw_free = pg->d_sw_free;
  G.adj_off = pg->d_adj_off;
  G.adj = pg->d_adj;
  G.adj_nbr = pg->d_adj_nbr;
  G.fixed = pg->d_fixed;
  G.n_poses = N;
  G.n_edges = M;
  pg->bview.entries = pg->d_bent;
  pg->bview.entry_off = pg->d_bent_off;
  pg->bview.halo = pg->d_bhalo;
  pg->bview.halo_off = pg->d_bhalo_off;
  pg->bview.n_blocks = pg->n_blocks;
  pg->bview.halo_cap = (halo_cap + 7u) & ~7u;
  pg->bview.slot_cap = std::max<uint32_t>(slot_cap, 256u);  // the tail's sum re-uses the slots: >= 1024 doubles
  *out_pg = pg;
  return NOS_OK;
}

size_t nos_pgo_num_unknowns(const nos_pose_graph* pg) { return pg ? pg->n_unknowns : 0; }

// Linearise at the current estimate: diagonal blocks, gradient, cost = sum of squared residuals,
// |gradient|_2.  Must precede nos_pgo_solve / nos_pgo_matvec.
int nos_pgo_linearize(nos_pose_graph* pg, double* cost, double* gradient_norm) {
  nosd::CtxGuard guard_(pg ? pg->ctx : nullptr);  // one solve / accumulate / create at a time per context
  if (!pg) return fail(NOS_ERR_INVALID_ARGUMENT, "pg is NULL");
  DeviceSlot& slot = pg->ctx->slots[0];
  NOS_HIP_CHECK(hipSetDevice(slot.device));
  const uint32_t blocks = (pg->n_poses + 255) / 256;
  hipLaunchKernelGGL(nos::pgo_linearize_kernel, dim3(blocks), dim3(256), 0, slot.stream, pg->view, pg->d_hdiag, pg->d_grad,
                     pg->d_partials);
  if (pg->n_edges > 0)
    hipLaunchKernelGGL(nos::pgo_switch_linearize_kernel, dim3((pg->n_edges + 255) / 256), dim3(256), 0, slot.stream,
                       pg->view, pg->d_grad + size_t(6) * pg->n_poses, pg->d_hs);
  hipLaunchKernelGGL(nos::pgo_sum_partials_kernel, dim3(1), dim3(1024), 0, slot.stream, pg->d_partials, blocks, 1,
                     pg->d_scalars);
  NOS_HIP_CHECK(hipGetLastError());
  double c = 0.0, gg = 0.0;
  int rc = pgo_read_scalars(pg, 1, &c);
  if (rc != NOS_OK) return rc;
  rc = pgo_dot(pg, pg->d_grad, pg->d_grad, &gg);
  if (rc != NOS_OK) return rc;
  if (cost) *cost = c;
  if (gradient_norm) *gradient_norm = std::sqrt(gg);
  return NOS_OK;
}

// Solve (H with its diagonal scaled by 1 + lambda) step = -gradient by block-Jacobi preconditioned CG.
// Stops at |residual| <= rel_tolerance |gradient| or after max_iterations.
int nos_pgo_solve(nos_pose_graph* pg, double lambda, int max_iterations, double rel_tolerance, int* iterations,
                  double* rel_residual, double* step_norm) {
  nosd::CtxGuard guard_(pg ? pg->ctx : nullptr);  // one solve / accumulate / create at a time per context
  if (!pg) return fail(NOS_ERR_INVALID_ARGUMENT, "pg is NULL");
  DeviceSlot& slot = pg->ctx->slots[0];
  NOS_HIP_CHECK(hipSetDevice(slot.device));
  // without free switches the switch rows of every CG vector stay identically zero: leave them out
  const size_t N6 = size_t(6) * pg->n_poses;
  const uint32_t active_edges = pg->n_free_switches > 0 ? pg->n_edges : 0;
  const size_t n = N6 + active_edges;
  const unsigned vblocks = unsigned((n + 255) / 256);
  const uint32_t pblocks = (pg->n_poses + active_edges + 255) / 256;
  hipLaunchKernelGGL(nos::pgo_precond_kernel, dim3((pg->n_poses + 255) / 256), dim3(256), 0, slot.stream, pg->d_hdiag,
                     lambda, pg->n_poses, pg->d_minv);
  // two-level preconditioner: rigid-motion coarse space over aggregates of consecutive poses (pgo_coarse_kernels.hpp);
  // worth its set-up (18 products) from a few aggregates on
  const uint32_t agg = uint32_t(std::max(2, pg->ctx->settings.pgo_agg));
  const bool two_level = pg->ctx->settings.pgo_precond != 0 && pg->n_poses >= 4 * agg;
  if (two_level) {
    const int rcs = pgo_coarse_setup(pg, lambda, agg);
    if (rcs != NOS_OK) return rcs;
  }
  auto apply_precond = [&]() -> int {
    const double* xc = nullptr;
    if (two_level) {
      const int rca = pgo_coarse_apply(pg, pg->d_r, &xc);
      if (rca != NOS_OK) return rca;
    }
    hipLaunchKernelGGL(nos::pgo_apply_precond_kernel, dim3(pblocks), dim3(256), 0, slot.stream, pg->d_minv, pg->d_hs, lambda,
                       pg->n_poses, active_edges, pg->d_r, pg->d_r + N6, pg->d_z, pg->d_z + N6, pg->d_partials, pg->d_pose,
                       pg->d_fixed, xc, agg);
    return NOS_OK;
  };
  // x = 0, r = -g
  NOS_HIP_CHECK(hipMemsetAsync(pg->d_x, 0, n * sizeof(double), slot.stream));
  NOS_HIP_CHECK(hipMemsetAsync(pg->d_r, 0, n * sizeof(double), slot.stream));
  hipLaunchKernelGGL(nos::pgo_cg_update_kernel, dim3(vblocks), dim3(256), 0, slot.stream, n, 1.0, pg->d_x, pg->d_grad,
                     pg->d_p /*scratch x: untouched since alpha*p with p = d_x = 0*/, pg->d_r);
  auto precond = [&](double* rz, double* rr) -> int {
    const int rcp = apply_precond();
    if (rcp != NOS_OK) return rcp;
    hipLaunchKernelGGL(nos::pgo_sum_partials_kernel, dim3(1), dim3(1024), 0, slot.stream, pg->d_partials, pblocks, 2,
                       pg->d_scalars);
    NOS_HIP_CHECK(hipGetLastError());
    double s[2];
    int rc = pgo_read_scalars(pg, 2, s);
    *rz = s[0];
    *rr = s[1];
    return rc;
  };
  double rz = 0.0, rr = 0.0;
  int rc = precond(&rz, &rr);
  if (rc != NOS_OK) return rc;
  const double b_norm = std::sqrt(rr);
  NOS_HIP_CHECK(hipMemcpyAsync(pg->d_p, pg->d_z, n * sizeof(double), hipMemcpyDeviceToDevice, slot.stream));
  int it = 0;
  double res = b_norm;
  if (b_norm > 0.0 && pg->ctx->settings.pgo_host_scalars == 0) {
    // CG scalars stay on the device (pgo_cg_alpha / beta kernels); the host looks at |r| only every kCheck iterations,
    // so up to kCheck - 1 iterations more than strictly needed may run (they only improve the step).
    constexpr int kCheck = 8;
    bool dir_flip = false;
    if (pg->block_poses != 0)  // the first product forms p = z + 0 * p_old: p_old must be finite
      NOS_HIP_CHECK(hipMemsetAsync(pg->d_p, 0, pg->n_unknowns * sizeof(double), slot.stream));
    const double init[8] = {0.0, 0.0, 0.0, rr, rz, 0.0, 0.0, 0.0};
    NOS_HIP_CHECK(hipMemcpyAsync(pg->d_scalars, init, sizeof init, hipMemcpyHostToDevice, slot.stream));
    NOS_HIP_CHECK(hipStreamSynchronize(slot.stream));  // `init` is a stack array
    const double stop = rel_tolerance * b_norm;
    while (it < max_iterations) {
      const int batch = std::min(kCheck, max_iterations - it);
      // one PCG iteration = 6 launches (4 without the coarse level): product (+ p.Ap, alpha) · update + restriction ·
      // PCR spans (2 at 1 M poses) · preconditioner (+ r.z, r.r, beta) · direction
      const nos::PgoTail pc_tail{pg->d_partials, pg->d_tickets + 1, pg->d_scalars};
      // one PCG iteration = 6 launches with the block-local product (direction update inside the product: p ping-pongs
      // between two buffers), 7 with the owner-computes product, 4 / 5 without the coarse level
      const bool fuse_direction = pg->block_poses != 0;
      for (int k = 0; k < batch; ++k) {
        double* p_cur = pg->d_p;
        if (fuse_direction) {  // p_new = z + beta p_old (beta = 0 and p_old = 0 before the first product: p = z)
          double* p_old = dir_flip ? pg->d_p2 : pg->d_p;
          p_cur = dir_flip ? pg->d_p : pg->d_p2;
          rc = pgo_launch_product(pg, lambda, p_old, pg->d_ap, pg->d_z, p_cur);
          dir_flip = !dir_flip;
        } else {
          rc = pgo_launch_product(pg, lambda, pg->d_p, pg->d_ap);
        }
        if (rc != NOS_OK) return rc;
        const double* xc = nullptr;
        if (two_level) {
          const uint32_t ur_blocks = (pg->n_agg + 3) / 4 + (active_edges + 255) / 256;
          hipLaunchKernelGGL(nos::pgo_update_restrict_kernel, dim3(ur_blocks), dim3(256), 0, slot.stream, pg->view, agg,
                             pg->n_agg, n, pg->d_scalars, p_cur, pg->d_ap, pg->d_x, pg->d_r, pg->c_b[0]);
          rc = pgo_pcr_rhs(pg, &xc);
          if (rc != NOS_OK) return rc;
        } else {
          hipLaunchKernelGGL(nos::pgo_cg_update_dev_kernel, dim3(vblocks), dim3(256), 0, slot.stream, n, pg->d_scalars, p_cur,
                             pg->d_ap, pg->d_x, pg->d_r);
        }
        hipLaunchKernelGGL(nos::pgo_apply_precond_kernel, dim3(pblocks), dim3(256), 0, slot.stream, pg->d_minv, pg->d_hs,
                           lambda, pg->n_poses, active_edges, pg->d_r, pg->d_r + N6, pg->d_z, pg->d_z + N6, pg->d_partials,
                           pg->d_pose, pg->d_fixed, xc, agg, pc_tail);
        if (!fuse_direction)
          hipLaunchKernelGGL(nos::pgo_cg_direction_dev_kernel, dim3(vblocks), dim3(256), 0, slot.stream, n, pg->d_scalars,
                             pg->d_z, pg->d_p);
        NOS_HIP_CHECK(hipGetLastError());
      }
      it += batch;
      double sc[8];
      rc = pgo_read_scalars(pg, 8, sc);
      if (rc != NOS_OK) return rc;
      res = std::sqrt(sc[3]);
      if (sc[7] != 0.0 || !(res > stop)) break;  // breakdown, or converged (NaN also ends the loop)
    }
  } else if (b_norm > 0.0) {
    for (; it < max_iterations; ++it) {
      double pap = 0.0;
      rc = pgo_matvec(pg, lambda, pg->d_p, pg->d_ap, &pap);
      if (rc != NOS_OK) return rc;
      if (!(pap > 0.0) || !std::isfinite(pap)) break;  // breakdown: keep the current iterate
      const double alpha = rz / pap;
      hipLaunchKernelGGL(nos::pgo_cg_update_kernel, dim3(vblocks), dim3(256), 0, slot.stream, n, alpha, pg->d_p, pg->d_ap,
                         pg->d_x, pg->d_r);
      double rz_new = 0.0;
      rc = precond(&rz_new, &rr);
      if (rc != NOS_OK) return rc;
      res = std::sqrt(rr);
      if (res <= rel_tolerance * b_norm) {
        ++it;
        break;
      }
      const double beta = rz_new / rz;
      rz = rz_new;
      hipLaunchKernelGGL(nos::pgo_cg_direction_kernel, dim3(vblocks), dim3(256), 0, slot.stream, n, beta, pg->d_z, pg->d_p);
    }
  }
  double xx = 0.0;
  rc = pgo_dot(pg, pg->d_x, pg->d_x, &xx);
  if (rc != NOS_OK) return rc;
  if (iterations) *iterations = it;
  if (rel_residual) *rel_residual = b_norm > 0.0 ? res / b_norm : 0.0;
  if (step_norm) *step_norm = std::sqrt(xx);
  return NOS_OK;
}

// Apply the last computed step: p += dp, q = normalize(q (x) Exp(dw)), s += ds.
int nos_pgo_retract(nos_pose_graph* pg) {
  nosd::CtxGuard guard_(pg ? pg->ctx : nullptr);  // one solve / accumulate / create at a time per context
  if (!pg) return fail(NOS_ERR_INVALID_ARGUMENT, "pg is NULL");
  DeviceSlot& slot = pg->ctx->slots[0];
  NOS_HIP_CHECK(hipSetDevice(slot.device));
  const uint32_t N = pg->n_poses;
  hipLaunchKernelGGL(nos::pgo_retract_kernel, dim3((N + pg->n_edges + 255) / 256), dim3(256), 0, slot.stream, N, pg->n_edges,
                     pg->d_fixed, pg->d_sw_free, pg->d_x, pg->d_x + size_t(6) * N, pg->d_pose, pg->d_edge);
  if (pg->block_poses != 0 && pg->n_free_switches > 0)  // the block entries carry their own copy of the constraint records
    hipLaunchKernelGGL(nos::pgo_refresh_entries_kernel, dim3((pg->n_entries + 255) / 256), dim3(256), 0, slot.stream,
                       pg->n_entries, pg->d_bent_edge, pg->d_edge, pg->d_bent);
  NOS_HIP_CHECK(hipGetLastError());
  NOS_HIP_CHECK(hipStreamSynchronize(slot.stream));
  return NOS_OK;
}

int nos_pgo_get_state(nos_pose_graph* pg, double* poses, double* switches) {
  nosd::CtxGuard guard_(pg ? pg->ctx : nullptr);  // one solve / accumulate / create at a time per context
  if (!pg) return fail(NOS_ERR_INVALID_ARGUMENT, "pg is NULL");
  const uint32_t N = pg->n_poses, M = pg->n_edges;
  NOS_HIP_CHECK(hipSetDevice(pg->ctx->slots[0].device));
  NOS_HIP_CHECK(hipStreamSynchronize(pg->ctx->slots[0].stream));
  if (poses) {
    std::vector<double> rec(size_t(8) * N);
    NOS_HIP_CHECK(hipMemcpy(rec.data(), pg->d_pose, rec.size() * sizeof(double), hipMemcpyDeviceToHost));
    for (uint32_t i = 0; i < N; ++i)
      for (int k = 0; k < 7; ++k) poses[size_t(7) * i + k] = rec[size_t(8) * i + k];
  }
  if (switches && M > 0) {
    std::vector<double> rec(size_t(8) * M);
    NOS_HIP_CHECK(hipMemcpy(rec.data(), pg->d_edge, rec.size() * sizeof(double), hipMemcpyDeviceToHost));
    for (uint32_t e = 0; e < M; ++e) switches[e] = rec[size_t(8) * e + 7];
  }
  return NOS_OK;
}

// Diagnostics for the parity tests: which = 0 gradient, 1 last step, 2 diagonal blocks (21 planes).
int nos_pgo_get_vector(nos_pose_graph* pg, int which, double* out) {
  nosd::CtxGuard guard_(pg ? pg->ctx : nullptr);  // one solve / accumulate / create at a time per context
  if (!pg || !out) return fail(NOS_ERR_INVALID_ARGUMENT, "NULL argument");
  NOS_HIP_CHECK(hipSetDevice(pg->ctx->slots[0].device));
  NOS_HIP_CHECK(hipStreamSynchronize(pg->ctx->slots[0].stream));
  const double* src = which == 0 ? pg->d_grad : which == 1 ? pg->d_x : pg->d_hdiag;
  if (which == 2) {
    NOS_HIP_CHECK(hipMemcpy(out, src, size_t(21) * pg->n_poses * sizeof(double), hipMemcpyDeviceToHost));
    return NOS_OK;
  }
  // device vectors hold [n_poses][6] records; the documented host order is 6 planes of n_poses
  std::vector<double> rec(pg->n_unknowns);
  NOS_HIP_CHECK(hipMemcpy(rec.data(), src, rec.size() * sizeof(double), hipMemcpyDeviceToHost));
  const size_t N = pg->n_poses;
  for (size_t i = 0; i < N; ++i)
    for (int k = 0; k < 6; ++k) out[size_t(k) * N + i] = rec[6 * i + k];
  for (size_t e = 0; e < pg->n_edges; ++e) out[6 * N + e] = rec[6 * N + e];
  return NOS_OK;
}

int nos_pgo_layout_info(const nos_pose_graph* pg, unsigned long long info[8]) {
  if (!pg || !info) return fail(NOS_ERR_INVALID_ARGUMENT, "NULL argument");
  const bool block = pg->block_poses != 0;
  info[0] = pg->n_poses, info[1] = pg->n_edges, info[2] = block ? pg->n_entries : 0, info[3] = block ? pg->block_poses : 0;
  info[4] = block ? pg->n_blocks : 0, info[5] = block ? pg->n_halo : 0, info[6] = pg->n_agg, info[7] = pg->pcr_levels;
  return NOS_OK;
}

// Timing aid: `repeats` sweeps back to back between one pair of events (bench.py; nothing else uses it).
int nos_pgo_time_sweep(nos_pose_graph* pg, int which, double lambda, int repeats, double* ms_per_sweep) {
  nosd::CtxGuard guard_(pg ? pg->ctx : nullptr);  // one solve / accumulate / create at a time per context
  if (!pg || !ms_per_sweep || repeats < 1 || which < 0 || which > 1) return fail(NOS_ERR_INVALID_ARGUMENT, "bad argument");
  DeviceSlot& slot = pg->ctx->slots[0];
  NOS_HIP_CHECK(hipSetDevice(slot.device));
  const uint32_t pose_blocks = (pg->n_poses + 255) / 256;
  if (which == 0) {  // x = gradient (its switch rows included); CG scalars zeroed: beta = 0, no breakdown flag
    NOS_HIP_CHECK(hipMemcpyAsync(pg->d_p, pg->d_grad, pg->n_unknowns * sizeof(double), hipMemcpyDeviceToDevice, slot.stream));
    NOS_HIP_CHECK(hipMemsetAsync(pg->d_scalars, 0, 8 * sizeof(double), slot.stream));
  }
  auto sweep = [&]() {
    if (which == 0) {  // as the PCG iteration launches it: with the direction update in its staging where that form exists
      if (pg->block_poses != 0)
        (void)pgo_launch_product(pg, lambda, pg->d_p, pg->d_ap, pg->d_grad, pg->d_p2);
      else
        (void)pgo_launch_product(pg, lambda, pg->d_p, pg->d_ap);
    } else {
      hipLaunchKernelGGL(nos::pgo_linearize_kernel, dim3(pose_blocks), dim3(256), 0, slot.stream, pg->view, pg->d_hdiag,
                         pg->d_grad, pg->d_partials);
      if (pg->n_edges > 0)
        hipLaunchKernelGGL(nos::pgo_switch_linearize_kernel, dim3((pg->n_edges + 255) / 256), dim3(256), 0, slot.stream,
                           pg->view, pg->d_grad + size_t(6) * pg->n_poses, pg->d_hs);
    }
  };
  sweep();  // warm
  NOS_HIP_CHECK(hipEventRecord(slot.ev0, slot.stream));
  for (int k = 0; k < repeats; ++k) sweep();
  NOS_HIP_CHECK(hipEventRecord(slot.ev1, slot.stream));
  NOS_HIP_CHECK(hipGetLastError());
  NOS_HIP_CHECK(hipEventSynchronize(slot.ev1));
  float ms = 0.f;
  NOS_HIP_CHECK(hipEventElapsedTime(&ms, slot.ev0, slot.ev1));
  *ms_per_sweep = double(ms) / repeats;
  return NOS_OK;
}

// y = (H with damped diagonal) x for a host vector (tests).
int nos_pgo_matvec(nos_pose_graph* pg, double lambda, const double* x, double* y) {
  nosd::CtxGuard guard_(pg ? pg->ctx : nullptr);  // one solve / accumulate / create at a time per context
  if (!pg || !x || !y) return fail(NOS_ERR_INVALID_ARGUMENT, "NULL argument");
  DeviceSlot& slot = pg->ctx->slots[0];
  NOS_HIP_CHECK(hipSetDevice(slot.device));
  const size_t N = pg->n_poses, n = pg->n_unknowns;
  std::vector<double> rec(n);
  for (size_t i = 0; i < N; ++i)
    for (int k = 0; k < 6; ++k) rec[6 * i + k] = x[size_t(k) * N + i];
  for (size_t e = 0; e < pg->n_edges; ++e) rec[6 * N + e] = x[6 * N + e];
  NOS_HIP_CHECK(hipMemcpyAsync(pg->d_p, rec.data(), n * sizeof(double), hipMemcpyHostToDevice, slot.stream));
  NOS_HIP_CHECK(hipMemsetAsync(pg->d_ap, 0, n * sizeof(double), slot.stream));
  int rc = pgo_launch_product(pg, lambda, pg->d_p, pg->d_ap);  // the product the PCG iteration makes
  if (rc != NOS_OK) return rc;
  NOS_HIP_CHECK(hipMemcpyAsync(rec.data(), pg->d_ap, n * sizeof(double), hipMemcpyDeviceToHost, slot.stream));
  NOS_HIP_CHECK(hipStreamSynchronize(slot.stream));
  for (size_t i = 0; i < N; ++i)
    for (int k = 0; k < 6; ++k) y[size_t(k) * N + i] = rec[6 * i + k];
  for (size_t e = 0; e < pg->n_edges; ++e) y[6 * N + e] = (pg->n_free_switches > 0) ? rec[6 * N + e] : (1.0 + lambda) * x[6 * N + e];
  return NOS_OK;
}

}  // extern "C"
