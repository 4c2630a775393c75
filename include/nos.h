/*
 * nos.h — C ABI of the MI355X (gfx950) Gauss-Newton normal-equation assembly path.
 *
 * This is the drop-in boundary for the one data-parallel hot path of
 * ChanghyeonKim93/nonlinear_optimizer_for_slam: per-correspondence residual +
 * analytic Jacobian + robust weight, reduced into the upper triangle of JᵀJ, Jᵀr and
 * the robust cost.  Every entry point replaces one piece of the reference's CPU path;
 * the reference location is cited next to each declaration (paths are relative to the
 * reference checkout, `NO/` = nonlinear_optimizer/, `MDM/` =
 * NO/mahalanobis_distance_minimizer/, `REM/` = NO/reprojection_error_minimizer/).
 *
 * Conventions
 *   - plain C, no exceptions cross this boundary; every function returns a nos_status
 *     (0 = NOS_OK) and never a torch / Eigen / STL type.
 *   - rotation matrices are row-major double[9]; translations double[3].
 *   - parameter order of gradient / Hessian is (t_x, t_y, t_z, w_x, w_y, w_z), the
 *     order the reference's update step uses (MDM/..._analytic_simd.cc:86-87).
 *   - NDT planes are ordered  px py pz  mx my mz  s00 s01 s02 s10 s11 s12 s20 s21 s22
 *     (point, NDT mean, row-major sqrt-information) — the 15 scalars the analytic
 *     solvers read from each 304-byte Correspondence (MDM/types.h:11-26,
 *     MDM/..._analytic.cc:164-168); reprojection planes are  X Y Z  u v
 *     (REM/types.h:25-28).
 *   - results:  out28 = { H upper triangle row-major (21) | g (6) | cost (1) },
 *               out10 = { H upper triangle row-major (6)  | g (3) | cost (1) }.
 *     The same 28 / 10 numbers the reference returns in PartialResult
 *     (MDM/mahalanobis_distance_minimizer.h:14-18).
 *   - all N correspondences are processed (the single-thread scalar class's behaviour,
 *     MDM/..._analytic.cc:98-100), no SIMD tail is dropped.
 */
#ifndef NOS_H_
#define NOS_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NOS_NDT_PLANES 15
#define NOS_REPROJ_PLANES 5
#define NOS_NDT6_OUT 28
#define NOS_NDT3_OUT 10
#define NOS_REPROJ_OUT 28

typedef enum nos_status {
  NOS_OK = 0,
  NOS_ERR_INVALID_ARGUMENT = 1,
  NOS_ERR_NO_DEVICE = 2,     /* no HIP device / runtime not usable: there is NO CPU fallback */
  NOS_ERR_HIP = 3,           /* a HIP runtime call failed; see nos_last_error() */
  NOS_ERR_OUT_OF_MEMORY = 4,
  NOS_ERR_WRONG_KIND = 5,    /* e.g. an NDT entry point given a reprojection dataset */
  NOS_ERR_UNSUPPORTED = 6
} nos_status;

/* Storage / arithmetic type of a device-resident dataset.  NOS_F64 mirrors the scalar
 * fp64 classes (MDM/..._analytic.cc), NOS_F32 mirrors the SIMD classes which cast to
 * float at pack time (MDM/..._analytic_simd.cc:25-27).  Reductions are always fp64. */
typedef enum nos_dtype { NOS_F64 = 0, NOS_F32 = 1 } nos_dtype;

/* Device-side restatement of NO/loss_function.h (a host virtual cannot cross to the
 * GPU).  kind NONE: rho = s, w = 1 (loss_function_ == nullptr branch,
 * MDM/..._analytic.cc:44-48).  EXPONENTIAL(a = c1, b = c2): NO/loss_function.h:28-33.
 * HUBER(a = threshold): NO/loss_function.h:57-66. */
typedef enum nos_loss_kind {
  NOS_LOSS_NONE = 0,
  NOS_LOSS_EXPONENTIAL = 1,
  NOS_LOSS_HUBER = 2
} nos_loss_kind;

typedef struct nos_loss {
  int32_t kind; /* nos_loss_kind */
  int32_t reserved;
  double a; /* c1 (exponential) | threshold (huber) */
  double b; /* c2 (exponential) */
} nos_loss;

typedef struct nos_ctx nos_ctx;
typedef struct nos_dataset nos_dataset;

/* ---- context ------------------------------------------------------------------
 * Replaces MultiThreadExecutor (NO/multi_thread_executor.h:51-73): instead of a pool
 * of pinned threads the context owns, per listed device, one HIP stream, a block-partial
 * workspace and a pinned result slot.  One process per GPU passes n_devices = 1 and
 * all-reduces the 28 scalars with RCCL itself (see nos_*_accumulate_async); listing
 * several devices gives the single-process fan-out the reference's thread pool had
 * (correspondences are split into contiguous ranges, partials are summed on the
 * caller in device order — MDM/..._analytic_simd.cc:55-76).
 * The same device may be listed more than once (useful for testing the sharded path
 * on a one-GPU box). */
int nos_ctx_create(const int* device_ids, int n_devices, nos_ctx** out_ctx);
int nos_ctx_destroy(nos_ctx* ctx);
int nos_ctx_num_devices(const nos_ctx* ctx);
/* Make shard `shard`'s launches go to an externally owned hipStream_t (e.g. the
 * stream torch uses), so that collectives enqueued by the caller order after them.
 * Passing NULL restores the context's own stream. */
int nos_ctx_set_stream(nos_ctx* ctx, int shard, void* hip_stream);
/* Block until all work enqueued on the context's streams has finished. */
int nos_ctx_synchronize(nos_ctx* ctx);

/* ---- inter-process reduction (one process per GPU) ------------------------------
 * The reference sums per-thread partials on the caller (MDM/..._analytic_simd.cc:70-75); with
 * one process per GPU that sum is a single RCCL all-reduce of the 28 (10) doubles over xGMI.
 * Rank 0 calls nos_comm_get_unique_id and hands the 128 bytes to every rank out of band
 * (torch.distributed / MPI / a file); every rank then calls nos_ctx_comm_init (collective).
 * From then on every *_accumulate / *_accumulate_async on that context returns the sum over
 * all ranks (identical bits on every rank), so the unchanged host LM loop runs in lock-step.
 * RCCL is loaded with dlopen on first use. */
#define NOS_COMM_ID_BYTES 128
int nos_comm_get_unique_id(unsigned char id[NOS_COMM_ID_BYTES]);
int nos_ctx_comm_init(nos_ctx* ctx, int n_ranks, int rank, const unsigned char id[NOS_COMM_ID_BYTES]);
int nos_ctx_comm_size(const nos_ctx* ctx); /* 0 if no communicator */
/* Ranks RCCL itself reports for the context's communicator (ncclCommCount); *count = 0 without an RCCL communicator. */
int nos_ctx_comm_rccl_count(const nos_ctx* ctx, int* count);
/* One line of JSON: HIP version the library was built with, HIP runtime / driver version mapped into this process,
 * the files the runtime and librccl were loaded from, RCCL's version, and two verdicts — "same_rocm_tree" (runtime and
 * librccl come from one directory) and "runtime_matches_build" (major.minor).  librccl is bound on first use: from
 * $NOS_RCCL_PATH, else from the directory of the HIP runtime already in the process, else the loader's default. */
int nos_runtime_info(char* buf, size_t capacity);
/* Alternative communicator for ranks on ONE node: a mailbox in POSIX shared memory (name, starting with '/',
 * chosen by the caller and identical on every rank).  Collective: rank 0 unlinks any segment of that name, creates a
 * fresh one exclusively and acknowledges every other rank; the other ranks open the name (retrying) and accept a
 * mapping only once rank 0 has acknowledged THEIR random hello word — a segment left behind by a crashed run is never
 * joined (bounded by NOS_SHM_ATTACH_TIMEOUT_MS, default 30 s).  Rank 0 may call nos_comm_shm_unlink once all ranks have returned.  The sums are then exchanged INSIDE the launch: the workgroup
 * that completes a GPU's sums stores them in its mailbox slot, waits (bounded) for the other ranks' slots and adds
 * them in rank order, so nos_*_accumulate and nos_*_solve return identical bits on every rank with no extra
 * kernel, no RCCL call and no host step per iteration.  This is the reference's "sum the per-thread partials"
 * (MDM/..._analytic_simd.cc:70-75) across processes.  At most 64 ranks; all ranks must issue the same sequence of
 * calls; a rank that never arrives makes the others return NOS_ERR_HIP after 8 s instead of hanging. */
int nos_ctx_comm_init_shm(nos_ctx* ctx, int n_ranks, int rank, const char* shm_name);
/* The same communicator with its SLOTS in device memory: every rank allocates its slot buffer as fine-grained device
 * memory, exports it with hipIpcGetMemHandle through the shared-memory segment (which keeps the handshake and control
 * words only) and opens the buffers of its peers; inside the launch a rank then writes its 28 sums and its round number
 * straight into every peer's buffer — device to device (between GPUs of a node presumably over xGMI; not verified on more than
 * one GPU) — and polls only its own memory.  Same
 * slots, same round numbers, same rank-order sum: bit-identical to the host-memory form.  NOS_ERR_UNSUPPORTED when the
 * platform refuses the fine-grained allocation or the IPC export / import.
 * With this communicator nos_*_solve keeps the whole LM loop in ONE launch on every rank: the exchange is a third stage of the
 * in-launch all-reduce (8-byte {tag | half} granules pushed into the peers' buffers, bounded wait of 8 s).  If a rank has to
 * give up (GPU shared with other processes), all ranks abandon that launch together, redo the solve with one launch per
 * iteration (nos_lm_report.fallback = 1) and pause the one-launch form for the next 64 solves, doubling on repeats — counted
 * in solves so that all ranks switch back in the same call. */
int nos_ctx_comm_init_shm_device(nos_ctx* ctx, int n_ranks, int rank, const char* shm_name);
int nos_comm_shm_unlink(const char* shm_name);
/* Leaves whichever communicator the context has (collective for RCCL); nos_ctx_destroy does it implicitly. */
int nos_ctx_comm_destroy(nos_ctx* ctx);
/* Sum `count` (<= 28) host doubles over the ranks, in place (diagnostic / self-test). */
int nos_ctx_comm_allreduce(nos_ctx* ctx, double* values, int count);

/* ---- datasets ------------------------------------------------------------------
 * Replaces the per-Solve AoS→SoA pack (SOAData, MDM/..._analytic_simd.h:15-19 and
 * .cc:19-28; AlignedBufferVarious, MDM/..._analytic_simd_various.h:14-44; AlignedBuffer,
 * REM/..._analytic_simd.h:15-35).  The handle owns device memory in the library's
 * tiled SoA layout; the caller keeps ownership of the host arrays (they are copied). */
int nos_ndt_dataset_create(nos_ctx* ctx, size_t n, const double* const planes[NOS_NDT_PLANES],
                           int dtype, nos_dataset** out_ds);
int nos_reproj_dataset_create(nos_ctx* ctx, size_t n,
                              const double* const planes[NOS_REPROJ_PLANES], int dtype,
                              nos_dataset** out_ds);
/* Same, from planes that already live in device memory of the context's FIRST device
 * (element type given by src_dtype, planes contiguous, length n each).  The data is
 * re-tiled on the device; the source planes are not referenced afterwards.
 * Only valid for single-device contexts. */
int nos_ndt_dataset_create_from_device(nos_ctx* ctx, size_t n,
                                       const void* const d_planes[NOS_NDT_PLANES],
                                       int src_dtype, int dtype, nos_dataset** out_ds);
int nos_reproj_dataset_create_from_device(nos_ctx* ctx, size_t n,
                                          const void* const d_planes[NOS_REPROJ_PLANES],
                                          int src_dtype, int dtype, nos_dataset** out_ds);
/* Ingest the reference's array-of-structures records directly: `records` points at n
 * host records of `stride_bytes` each; the 15 (or 5) doubles are found at the given
 * byte offsets inside a record (for the reference's 304-byte Correspondence:
 * point @0, ndt.mean @128, ndt.sqrt_information @224 column-major — Eigen default —
 * see INTEGRATION.md).  Records are staged through pinned memory and unpacked on the
 * device (K6 in SURVEY.md §2.2). */
int nos_ndt_dataset_create_from_records(nos_ctx* ctx, size_t n, const void* records,
                                        size_t stride_bytes,
                                        const size_t field_offsets[NOS_NDT_PLANES], int dtype,
                                        nos_dataset** out_ds);
int nos_reproj_dataset_create_from_records(nos_ctx* ctx, size_t n, const void* records,
                                           size_t stride_bytes,
                                           const size_t field_offsets[NOS_REPROJ_PLANES],
                                           int dtype, nos_dataset** out_ds);
int nos_dataset_destroy(nos_dataset* ds);
size_t nos_dataset_size(const nos_dataset* ds);
int nos_dataset_dtype(const nos_dataset* ds);
/* Algorithmic bytes one accumulate pass streams from HBM for this dataset
 * (n × planes × sizeof(element)); the figure roofline numbers are quoted against. */
size_t nos_dataset_stream_bytes(const nos_dataset* ds);

/* Semantics of the reference's fp32 ("SIMD") solver classes for solves and accumulates on this dataset (0 = off, the
 * default: the scalar classes' semantics at whatever element type the dataset has):
 *   NDT (6- and 3-DoF): the LM loop keeps lambda and previous_cost in float
 *     (MDM/mahalanobis_distance_minimizer_analytic_simd.cc:38-39,99-101; ..._3dof_simd.cc:73-74,201-207);
 *   reprojection: a correspondence counts when its depth is > 0 (instead of >= min_depth), the mask multiplies the
 *     WEIGHT only — the robust loss of a masked correspondence is still added to the cost
 *     (REM/reprojection_error_minimizer_analytic_simd.cc:66,92,134).
 * The classes' tail drop (only floor(N/8)*8 correspondences are used) and their float 1/fx are the caller's side: create
 * the dataset from the first floor(N/8)*8 records (the drop-in classes do, HipOptions::simd_class).  The lane arithmetic of
 * the un-vendored simd_helper library is NOT reproduced digit for digit (parity unpinned for it). */
int nos_dataset_set_simd_class(nos_dataset* ds, int on);

/* Copy a dataset back to host planes (n_fields arrays of nos_dataset_size doubles, plane order
 * as above).  Diagnostics / tests only. */
int nos_dataset_download(nos_dataset* ds, double* const planes[]);

/* ---- correspondence matching on the device -----------------
 * Replaces MatchPointCloud of the reference's test harness
 * (MDM/tests/simple_optimization_test.cc:296-342): FLANN KDTreeSingleIndex over the valid
 * voxel means + radiusSearch(radius = 1.0 on L2_Simple, i.e. squared distance, max_neighbors
 * = 2) becomes a uniform-grid lookup on the GPU.  nos_ndt_match writes {local point, mean,
 * sqrt-information} for the (up to) two nearest voxels of every scan point straight into a
 * device dataset: slot 2*i + k holds the k-th nearest voxel of point i, an absent neighbour is
 * an all-zero record (contributes nothing).  The dataset is then used with nos_ndt6_accumulate /
 * nos_ndt3_accumulate like any other; nothing returns to the host between matching and solving.
 * means_xyz: [n_voxels][3], sqrt_infos: [n_voxels][9] row-major, valid: optional [n_voxels]
 * (NDT::is_valid, MDM/types.h:21; NULL = all valid), points_xyz: [n_points][3] in the scan's
 * local frame.  Single-device contexts only. */
typedef struct nos_ndt_map nos_ndt_map;
typedef struct nos_scan nos_scan;
int nos_ndt_map_create(nos_ctx* ctx, size_t n_voxels, const double* means_xyz,
                       const double* sqrt_infos, const unsigned char* valid,
                       double search_radius_sq, nos_ndt_map** out_map);
int nos_ndt_map_destroy(nos_ndt_map* map);
size_t nos_ndt_map_size(const nos_ndt_map* map); /* valid voxels */
int nos_scan_create(nos_ctx* ctx, size_t n_points, const double* points_xyz, nos_scan** out_scan);
int nos_scan_destroy(nos_scan* scan);
size_t nos_scan_size(const nos_scan* scan);
/* Optional, once per scan: reorder the points by grid cell of edge cell_edge (in the scan's own frame) so that
 * the points one wavefront of the matcher handles are spatial neighbours (a rigid pose keeps them so).  The
 * matcher's output slots then follow the new order; nos_scan_order gives order[j] = index, in the array handed
 * to nos_scan_create, of the point now stored at position j (identity if the scan was never sorted).  The
 * solver does not care about correspondence order (sums), the reference's harness pushes them in scan order
 * (MDM/tests/simple_optimization_test.cc:296-342). */
int nos_scan_sort_by_cell(nos_scan* scan, double cell_edge);
int nos_scan_order(const nos_scan* scan, uint32_t* order_out);
int nos_ndt_match(nos_ndt_map* map, nos_scan* scan, const double R[9], const double t[3],
                  int max_neighbors, int dtype, nos_dataset** out_ds, size_t* n_matches);
/* Tail drop of the reference's solver classes for a matcher-written dataset.  The scalar 3-DoF class uses only the
 * first floor(N/4)*4 entries of the correspondence vector (MDM/..._analytic_3dof.cc:33-36; the 6-DoF lines of
 * the reference's results directory were captured from a revision that did the same), the SIMD classes floor(N/8)*8
 * (MDM/..._analytic_simd.cc:46-51).  The reference's vector holds matches only, in scan order, nearest first
 * (MDM/tests/simple_optimization_test.cc:320-340); here absent neighbours are zero records, so "drop the last k
 * entries" is: clear the last n_drop NON-EMPTY records in slot order (on the device, asynchronously on the context's
 * stream).  "Empty" is decided by the sqrt-information: a record whose nine S entries are all zero counts as an absent
 * neighbour (what the matcher writes for one) — a real match whose S is identically zero would be taken for empty, but it
 * contributes nothing to the sums either way.  Flat NDT datasets on single-device contexts. */
int nos_dataset_drop_last_matches(nos_dataset* ds, size_t n_drop);

/* ---- voxel-indexed NDT datasets (additive; SURVEY.md §8d "voxel-indexed layout") ------
 * The reference copies the full NDT into every correspondence (MDM/types.h:23-26, :336 of the
 * test harness), hence the flat 120-byte layout.  When many points share a voxel the same sums
 * can be formed from {point, voxel id(s)} plus a table of voxel records: 24 B + 4 B per slot
 * instead of 120 B per correspondence; the table stays in L2 / Infinity Cache and the kernel
 * becomes fp64-ALU bound.  Such a dataset is used with nos_ndt6_accumulate / nos_ndt3_accumulate
 * (and their _async forms) exactly like a flat one and gives the same sums (to rounding: the
 * summation order differs).  index_planes[k][i] = voxel id of point i's k-th correspondence, or
 * -1 for none (n_slots = 1 or 2).  sort_by_voxel != 0 reorders the points by slot-0 voxel id on
 * the device so that a wave's table reads hit a few cache lines (order does not affect the sums).
 * nos_ndt_match_indexed is nos_ndt_match producing this form directly.
 * Roofline accounting: nos_dataset_stream_bytes() = n * (3 * sizeof(elem) + 4 * n_slots); never
 * compare it with the 120-byte figure of the flat layout. */
int nos_ndt_indexed_dataset_create(nos_ctx* ctx, size_t n_points,
                                   const double* const point_planes[3], int n_slots,
                                   const int32_t* const index_planes[], size_t n_voxels,
                                   const double* means_xyz, const double* sqrt_infos, int dtype,
                                   int sort_by_voxel, nos_dataset** out_ds);
int nos_ndt_match_indexed(nos_ndt_map* map, nos_scan* scan, const double R[9], const double t[3],
                          int max_neighbors, int dtype, int sort_by_voxel, nos_dataset** out_ds,
                          size_t* n_matches);

/* ---- NDT map construction on the device --------------------
 * Replaces UpdateNdtMap of the reference's test harness
 * (MDM/tests/simple_optimization_test.cc:236-281): voxelise points_xyz ([n][3], map frame) at
 * voxel_resolution, accumulate count / sum / moment per voxel (moment starts at identity,
 * MDM/types.h:14), mean, covariance, symmetric 3x3 eigen-decomposition, eigenvalue flooring and
 * sqrt_information = diag(eigvals^-1/2) * eigenvectors; a voxel is valid with >= 5 points and a
 * largest eigenvalue >= 0.01.  The result is a ready-to-match nos_ndt_map; *out_stats (optional)
 * gives the per-voxel numbers back (voxels ordered by ascending integer cell coordinates).
 * Voxel coordinates are limited to +-2^20 cells per axis. */
typedef struct nos_map_stats nos_map_stats;
/* flags: 0 reproduces the harness formula sqrt_information = D^-1/2 * V (:275-276), whose result
 * depends on the eigenvector sign convention (here: the first near-largest component of every
 * eigenvector is positive; Eigen's own convention is not reproducible without Eigen);
 * NOS_MAP_PROPER_SQRT_INFORMATION uses D^-1/2 * V^T, the true square root of the inverse
 * covariance, which is sign- and degenerate-subspace-invariant. */
#define NOS_MAP_PROPER_SQRT_INFORMATION 1
/* NOS_MAP_REFERENCE_EXACT: the harness formula with the reference BINARY's rounding — count / sum / moment accumulated per
 * voxel sequentially in point order, Eigen::SelfAdjointEigenSolver<Matrix3d> restated step by step, multiply-adds fused
 * exactly where the reference's -O2 -march=native x86-64 build fuses them (options "map_fma_mask", "map_eigen_version").
 * The map then equals the one the reference's test drivers build bit for bit (tests/golden/ndt_reference_map.npz), and
 * map build -> nos_ndt_match -> nos_ndt6_solve / nos_ndt3_solve reproduce the captured COST / iter lines of the reference's results directory.
 * QUALIFICATION: identical when no voxel with >= 5 points is rejected by the eigenvalue test.  The reference's UpdateNdtMap
 * leaves the function (`return`, not `continue`: .../tests/simple_optimization_test.cc:263-266) at the first such voxel, so
 * every voxel its unordered_map walk would have visited later stays invalid; that order-dependent quirk is deliberately NOT
 * reproduced — voxels are treated independently here (the reference's own room scene has no such voxel).  The defaults of
 * "map_fma_mask" (which multiply-adds are fused, element by element) and the N/4 tail drop the captured 6-DoF runs need are
 * CALIBRATED to the captured x86-64 runs, not derived from today's sources: parity unpinned for other builds
 * of the reference (its aarch64 captures are followed with map_fma_mask = 0, as a band).
 * Voxels are listed in first-seen order (stats and voxel ids).  Not combinable with NOS_MAP_PROPER_SQRT_INFORMATION. */
#define NOS_MAP_REFERENCE_EXACT 2
int nos_ndt_map_build(nos_ctx* ctx, size_t n_points, const double* points_xyz,
                      double voxel_resolution, double search_radius_sq, int flags,
                      nos_ndt_map** out_map, nos_map_stats** out_stats);
size_t nos_map_stats_size(const nos_map_stats* stats);
/* Any output pointer may be NULL.  means [V][3], sqrt_infos [V][9] row-major, valid [V],
 * counts [V], cells [V][3]. */
int nos_map_stats_get(const nos_map_stats* stats, double* means_xyz, double* sqrt_infos,
                      unsigned char* valid, uint32_t* counts, int64_t* cells_xyz);
/* NOS_MAP_REFERENCE_EXACT builds only: eigenvalues [V][3] ascending, before flooring; eigenvectors [V][9] row-major V
 * (column k = eigenvector k) as Eigen returns them. */
int nos_map_stats_get_eigen(const nos_map_stats* stats, double* eigenvalues, double* eigenvectors);
int nos_map_stats_destroy(nos_map_stats* stats);

/* ---- the hot path -------------------------------------------------------------
 * nos_ndt6_accumulate replaces
 *   MahalanobisDistanceMinimizerAnalyticSIMD::ComputeCostAndDerivatives
 *     (MDM/..._analytic_simd.cc:113-177) and its scalar twin
 *   MahalanobisDistanceMinimizerAnalytic::ComputeCostAndDerivatives
 *     (MDM/..._analytic.cc:12-52, 159-218)
 * plus the thread fan-out / partial sum around them (MDM/..._analytic_simd.cc:55-76).
 * Blocking; out28 is host memory. */
int nos_ndt6_accumulate(nos_dataset* ds, const double R[9], const double t[3],
                        const nos_loss* loss, double out28[NOS_NDT6_OUT]);
/* nos_ndt3_accumulate replaces the planar (x, y, yaw) loops
 *   MDM/..._analytic_3dof.cc:36-69,110-139 and MDM/..._analytic_3dof_simd.cc:85-158.
 * R2 is the row-major 2×2 rotation, t2 the planar translation. */
int nos_ndt3_accumulate(nos_dataset* ds, const double R2[4], const double t2[2],
                        const nos_loss* loss, double out10[NOS_NDT3_OUT]);
/* nos_reproj_accumulate replaces REM/..._analytic.cc:31-64,107-162 and the SIMD loop
 * REM/..._analytic_simd.cc:55-138.  intr = {inv_fx, inv_fy, cx, cy}; points whose
 * transformed depth is < min_depth contribute nothing (scalar class: 0.03,
 * REM/..._analytic.cc:111,119-123). */
int nos_reproj_accumulate(nos_dataset* ds, const double R[9], const double t[3],
                          const double intr[4], const nos_loss* loss, double min_depth,
                          double out28[NOS_REPROJ_OUT]);

/* Asynchronous forms for one-process-per-GPU use: enqueue kernel + final reduce on
 * the context's stream (single-device contexts only) and leave the 28 / 10 doubles
 * in DEVICE memory at d_out, e.g. a torch tensor that the caller then hands to
 * ncclAllReduce / torch.distributed.all_reduce on the same stream.  This is the
 * reference's "sum the per-thread partials" step (MDM/..._analytic_simd.cc:70-75)
 * moved onto RCCL.  No host synchronisation happens inside. */
int nos_ndt6_accumulate_async(nos_dataset* ds, const double R[9], const double t[3],
                              const nos_loss* loss, double* d_out28);
int nos_ndt3_accumulate_async(nos_dataset* ds, const double R2[4], const double t2[2],
                              const nos_loss* loss, double* d_out10);
int nos_reproj_accumulate_async(nos_dataset* ds, const double R[9], const double t[3],
                                const double intr[4], const nos_loss* loss, double min_depth,
                                double* d_out28);

/* ---- the whole Levenberg-Marquardt loop, device resident ---------------------------------
 * nos_ndt6_solve / nos_ndt3_solve / nos_reproj_solve replace the body of the reference's Solve():
 *   MDM/..._analytic_simd.cc:30-108 (= ..._analytic.cc:57-157), MDM/..._analytic_3dof.cc:17-108,
 *   REM/reprojection_error_minimizer_analytic.cc:15-105
 * i.e. per iteration: ComputeCostAndDerivatives, H_kk *= 1 + lambda, ldlt().solve(-g), right-multiplicative
 * pose update, the two convergence tests after the update, lambda *= 2 / 0.6 clamped to [1e-6, 1e-2].
 * The loop state stays in device memory: the workgroup that completes the sums of a launch runs that
 * loop body and leaves the new pose for the next launch, so consecutive iterations run back-to-back on
 * the GPU with no host step in between (the host keeps `launches_in_flight` launches queued and reads
 * one pinned log entry per iteration).  Semantics are those of the host loop around nos_*_accumulate
 * (same source for the loop body); only floating-point contraction may differ in the last bits.
 * Small and mid-size problems run the whole loop in ONE launch: below 1 024 NDT / 3 072 reprojection
 * correspondences inside a single workgroup; up to 131 072 correspondences with one 512-correspondence chunk per
 * workgroup held in registers and a bounded epoch hand-off between iterations (if the grid cannot become resident
 * in time the launch gives up and the solve is redone with one launch per iteration) — report->launches tells
 * which form ran.
 * With a communicator (nos_ctx_comm_init) each launch is followed by the all-reduce and a one-wave step
 * kernel; every rank ends with identical bits.  Single-device contexts only (NOS_ERR_UNSUPPORTED
 * otherwise: use the host loop).  R / t are in-out. */
typedef struct nos_lm_options {
  int32_t max_iterations;      /* Options::max_iterations (options.h) */
  int32_t launches_in_flight;  /* 0 = default (3) */
  double gradient_tolerance;   /* Options::convergence_handle.gradient_tolerance */
  double parameter_tolerance;  /* Options::convergence_handle.parameter_tolerance */
  double* cost_history;        /* NULL, or room for max_iterations costs (one per executed iteration) */
} nos_lm_options;

typedef struct nos_lm_report {
  int32_t iterations;   /* loop index at exit: the "iter:" of the reference's stderr line */
  int32_t ok;           /* 0 if the damped solve met a non-positive pivot */
  int32_t launches;     /* kernels enqueued (>= iterations executed; the surplus exits at once) */
  int32_t fallback;     /* 1: the one-launch form of the loop gave up (GPU shared with another process, or the LDS it needs
                           refused) and the solve was re-run with one launch per iteration — same result, slower */
  double printed_cost;  /* previous_cost at exit: the "COST:" of that line */
  double last_cost;
  double final_lambda;
} nos_lm_report;

int nos_ndt6_solve(nos_dataset* ds, double R[9], double t[3], const nos_loss* loss,
                   const nos_lm_options* options, nos_lm_report* report);
int nos_ndt3_solve(nos_dataset* ds, double R2[4], double t2[2], const nos_loss* loss,
                   const nos_lm_options* options, nos_lm_report* report);
int nos_reproj_solve(nos_dataset* ds, double R[9], double t[3], const double intr[4],
                     const nos_loss* loss, double min_depth, const nos_lm_options* options,
                     nos_lm_report* report);

/* Test hook: ONE step of the device-resident loop on given sums and a given loop state — the stand-alone step kernel
 * (the same single-lane function every device loop form calls).  dof 6: sums[28], dof 3: sums[10].
 * state[22] = R (9, row-major; planar: R[0..3] = the 2x2 rotation) | t (3) | q w x y z (4) | lambda | previous_cost | cost |
 * iteration | done | ok; settings[4] = max_iterations | gradient_tolerance | parameter_tolerance | float_schedule.
 * state is updated in place.  The host's own step for the same arguments: nos_host_lm_advance in libnos_host.so. */
int nos_debug_lm_step(nos_ctx* ctx, int dof, const double* sums, const double settings[4], double state[22]);

/* ---- pose-graph optimisation (BASELINE.json configs[4]) --------
 * The reference's PoseGraphOptimizerAnalytic::Solve is an empty loop
 * (NO/pose_graph_optimizer/pose_graph_optimizer_analytic.cc:12-51); only the Ceres path is real.
 * These entry points provide what its TODO comments ask for (:36-42 "Make sparse Hessian / Solve
 * normal equation / Update poses / Check convergence") for the residual Ceres minimises
 * (NO/pose_graph_optimizer/ceres_cost_functor.h:17-53, switchable :55-98):
 *   r_t = (p_q - p_r) - q_r (x) t_m,  r_R = 2 vec(q_q^* q_r q_m);  loop constraints: r <- s r and a
 *   seventh residual (1 - s) * 1e-9 with a free switch s.
 * poses [n][7] = px py pz qw qx qy qz (PoseParameter, pose_graph_optimizer.h:16-19);
 * meas [m][7] = relative_pose_from_reference_to_query, same packing (types.h:13-19);
 * switch_init / switch_free / fixed may be NULL.  One Gauss-Newton / LM iteration is
 *   nos_pgo_linearize → nos_pgo_solve (block-Jacobi PCG on the damped normal equations, matrix
 *   free) → nos_pgo_retract (p += dp, q = normalize(q (x) Exp(dw)), s += ds).
 * The normal matrix is never stored: every sweep re-derives the per-constraint Jacobian blocks
 * from the poses ("owner computes", no atomics, bit-reproducible).  Single-device contexts. */
typedef struct nos_pose_graph nos_pose_graph;
int nos_pgo_create(nos_ctx* ctx, size_t n_poses, const double* poses, size_t n_edges,
                   const int32_t* ref, const int32_t* qry, const double* meas,
                   const double* switch_init, const unsigned char* switch_free,
                   const unsigned char* fixed, nos_pose_graph** out_pg);
int nos_pgo_destroy(nos_pose_graph* pg);
size_t nos_pgo_num_unknowns(const nos_pose_graph* pg); /* 6 n_poses + n_edges */
int nos_pgo_linearize(nos_pose_graph* pg, double* cost, double* gradient_norm);
int nos_pgo_solve(nos_pose_graph* pg, double lambda, int max_iterations, double rel_tolerance,
                  int* iterations, double* rel_residual, double* step_norm);
int nos_pgo_retract(nos_pose_graph* pg);
int nos_pgo_get_state(nos_pose_graph* pg, double* poses, double* switches);
/* which: 0 gradient, 1 last step (both 6 planes of n_poses then n_edges switch entries),
 * 2 diagonal blocks (21 planes of n_poses, upper triangle row-major).  Diagnostics. */
int nos_pgo_get_vector(nos_pose_graph* pg, int which, double* out);
/* y = (J^T J with its diagonal scaled by 1 + lambda) x for host vectors.  Diagnostics. */
int nos_pgo_matvec(nos_pose_graph* pg, double lambda, const double* x, double* y);
/* What the sweeps have to touch (bench.py's byte models): info[0] poses, [1] constraints, [2] entries of the block-local
 * product (one per constraint and block it touches; 0 = the owner-computes product is in use), [3] poses per block,
 * [4] blocks, [5] halo poses over all blocks, [6] aggregates of the coarse level, [7] PCR levels (6, 7: 0 before the
 * first two-level solve). */
int nos_pgo_layout_info(const nos_pose_graph* pg, unsigned long long info[8]);
/* Timing aid (bench.py): `repeats` device-resident products of the kind a PCG iteration makes (x = the gradient of the
 * last nos_pgo_linearize), back to back on the context's stream between one pair of HIP events → milliseconds per
 * product.  which: 0 the product of a PCG iteration (with its in-launch p.Ap sum), 1 the linearisation sweeps. */
int nos_pgo_time_sweep(nos_pose_graph* pg, int which, double lambda, int repeats, double* ms_per_sweep);

/* ---- thread safety ----------------------------------------------------------------
 * Every entry point that takes a context, or an object created on one (dataset, map, scan, pose graph), holds that
 * context's lock for its whole duration: calls on ONE context are serialised, calls on different contexts run
 * concurrently.  The drop-in solver classes share one context per device list, so two threads that each own a solver
 * object are safe (their solves take turns) — the guarantee the reference's independent solver objects give.
 * nos_ctx_destroy must not race with any other call on that context.
 *
 * ---- experiment knobs ----------------------------------------------------------------
 * Read from the environment once, in nos_ctx_create (NOS_SC1, NOS_NT, NOS_FUSED, NOS_LM_FUSED, NOS_LM_WINDOW,
 * NOS_LM_SINGLE, NOS_LM_CLUSTER, NOS_POOL, NOS_TILE_LOG2, NOS_PLANE_SKEW, NOS_INGEST, NOS_INGEST_THREADS,
 * NOS_INDEXED_BPC, NOS_MATCH_DENSE, NOS_PGO_HOST_SCALARS, NOS_PGO_PRECOND, NOS_PGO_AGG; option only: "map_fma_mask",
 * "map_eigen_version"); afterwards only through these setters (keys =
 * the names in lower case without the prefix, e.g. "lm_cluster"; "ingest": 0 auto, 1 pack, 2 unpack;
 * "debug_cluster_abort": test hook, makes the next one-launch solve give up and fall back).  Nothing on the solve /
 * accumulate path reads the environment.
 *   "lm_cluster"  1 (default) the device-resident loop runs in ONE launch wherever it can: correspondences resident on chip
 *                 up to the register + LDS capacity, streamed from HBM every iteration beyond it; 4 = one launch only for
 *                 resident data (one launch per iteration above), 5 = as 1 with the all-reduce's first stage always through
 *                 sc1 stores, 3 = the arrival-counter all-reduce, 2 = at most one correspondence per lane, 0 = off.
 *   "tile_log2"   layout of datasets created afterwards: -1 (default) by element type — fp64 planar planes, fp32 tiles of
 *                 1024 correspondences; 0 planar; 10…24 tiles of 2^k. */
int nos_ctx_set_option(nos_ctx* ctx, const char* key, int value);
int nos_ctx_get_option(const nos_ctx* ctx, const char* key, int* value);

/* ---- measurement / diagnostics ------------------------------------------------- */
/* Launch geometry override (0 = library default): blocks per CU of the assemble grid and
 * the index of the compiled kernel geometry.  Tuning knob, not needed for normal use. */
int nos_ctx_set_launch(nos_ctx* ctx, int blocks_per_cu, int variant);
/* Time `repeats` back-to-back assemble launches of shard 0 with HIP events recorded on
 * the stream the kernels are launched on; returns the mean per-launch duration of the
 * assemble kernel alone (kernel_ms) and of assemble + final reduce (total_ms). */
int nos_ndt6_time_kernel(nos_dataset* ds, const double R[9], const double t[3],
                         const nos_loss* loss, int repeats, double* kernel_ms,
                         double* total_ms);
int nos_reproj_time_kernel(nos_dataset* ds, const double R[9], const double t[3],
                           const double intr[4], const nos_loss* loss, double min_depth,
                           int repeats, double* kernel_ms, double* total_ms);
int nos_ndt3_time_kernel(nos_dataset* ds, const double R2[4], const double t2[2],
                         const nos_loss* loss, int repeats, double* kernel_ms,
                         double* total_ms);
/* Per-launch device timing inside a running loop: between _begin and _end every assemble
 * kernel launched through this context is bracketed by a HIP event pair recorded on the
 * stream it is launched on (every sample_every-th launch, up to max_launches timed launches per
 * device; the two event records cost about a microsecond of host time each).  _end synchronises
 * and returns the number of timed launches and their mean / min / max duration.
 * sample_every == 0 selects the bracket form for back-to-back launch trains (the device-resident
 * loop): ONE event is recorded at _begin and one at _end on the launch stream, nothing in between
 * (an event between two queued kernels would serialise their dispatch and show up in what it
 * measures); _end returns the launches in between and (t_end - t_begin) / launches as mean = min =
 * max — an upper bound of the kernel duration that includes whatever else ran in the train. */
int nos_ctx_profile_begin(nos_ctx* ctx, int max_launches, int sample_every);
int nos_ctx_profile_end(nos_ctx* ctx, int* n_launches, double* mean_ms, double* min_ms,
                        double* max_ms);
/* Demangled symbol of the hot-path kernel launched last on `shard` of this context ("" if none yet): the template
 * instantiation the library chose — problem, element type, loss, launch geometry or loop form — exactly as
 * rocprofv3 --kernel-trace lists it. */
int nos_ctx_last_kernel(const nos_ctx* ctx, int shard, char* buf, size_t capacity);
const char* nos_status_string(int status);
/* Thread-local description of the last failure in this thread ("" if none). */
const char* nos_last_error(void);
const char* nos_version(void);

#ifdef __cplusplus
} /* extern "C" */
#endif

#endif /* NOS_H_ */
