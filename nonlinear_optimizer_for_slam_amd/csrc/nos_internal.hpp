// nos_internal.hpp — declarations shared by the translation units of libnos_hip.so (not part of the ABI).
#pragma once

#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <random>
#include <climits>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <new>
#include <string>
#include <thread>
#include <mutex>
#include <vector>

#include "../../include/nos.h"
#include "assemble_kernels.hpp"
#include "match_kernels.hpp"

namespace nosd {

// thread-local text of the last failure + status pass-through (defined in nos_core.hip)
int fail(int status, const char* fmt, ...);
void clear_last_error();  // a failure that was handled (a fallback that went on to succeed) leaves no text behind

#define NOS_HIP_CHECK(expr)                                                               \
  do {                                                                                    \
    hipError_t e_ = (expr);                                                               \
    if (e_ != hipSuccess)                                                                 \
      return ::nosd::fail(e_ == hipErrorOutOfMemory ? NOS_ERR_OUT_OF_MEMORY : NOS_ERR_HIP,        \
                  "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
  } while (0)

// RCCL is bound at run time (dlopen) so that single-GPU use never needs it and so that a process
// that already carries torch's copy of librccl shares that copy instead of loading a second one.
struct RcclApi {
  void* handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
  ncclResult_t (*GetVersion)(int*) = nullptr;
  std::string path;  // file the library was loaded from
  bool ok = false;
};

RcclApi* Rccl();

#define NOS_RCCL_CHECK(expr)                                                                     \
  do {                                                                                           \
    ncclResult_t r_ = (expr);                                                                    \
    if (r_ != ncclSuccess) return ::nosd::fail(NOS_ERR_HIP, "%s failed: %s", #expr, Rccl()->GetErrorString(r_)); \
  } while (0)

enum DatasetKind { kKindNdt = 1, kKindReproj = 2, kKindNdtIndexed = 3 };

constexpr int kMaxPartialRows = 8192;  // upper bound on grid size of the assemble kernel
constexpr int kMaxOut = 28;
constexpr int kHistCapacity = 4096;     // iterations whose cost the single-workgroup solve can report
constexpr int kLogSlots = 64;          // ring of loop log entries; bounds the number of launches in flight

struct DeviceSlot {
  int device = 0;
  int num_cus = 256;
  hipStream_t own_stream = nullptr;
  hipStream_t stream = nullptr;    // own_stream or an external one
  double* partials = nullptr;      // [kMaxPartialRows][kMaxOut] device
  double* d_out = nullptr;         // [kMaxOut] device
  double* h_out = nullptr;         // pinned, device-mapped host block: [0..27] result, [32] sequence word
  double* h_out_dev = nullptr;     // device-side address of h_out
  unsigned int* counter = nullptr; // device ticket word of the in-launch final reduce (kept at 0 between launches)
  unsigned long long seq = 0;      // last sequence value handed to a fused launch
  nos::LmDevice* d_lm = nullptr;   // device-resident loop state (nos_*_solve)
  double* h_log = nullptr;         // pinned, device-mapped ring of per-iteration log entries [kLogSlots][kLogEntryDoubles]
  double* h_log_dev = nullptr;
  nos::ClusterCtl* d_cluster = nullptr;  // abort word + arrival counters of the resident one-launch solve
  double* h_hist = nullptr;        // pinned, device-mapped cost history of the single-workgroup solve [kHistCapacity]
  double* h_hist_dev = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr, ev2 = nullptr;
  // per-launch kernel timing (nos_ctx_profile_begin/_end): event pairs recorded on the
  // launch stream around every assemble kernel while profiling is on
  // Ingestion resources, created on first use and kept (a cold Solve() used to spend ≈ 3 ms creating and freeing
  // them): copy stream, events, two staging buffers that only grow.
  hipStream_t copy_stream = nullptr;
  hipEvent_t ing_done[2] = {nullptr, nullptr};
  hipEvent_t ing_copied = nullptr;
  void* stage[2] = {nullptr, nullptr};
  size_t stage_bytes = 0;
  void* pack_pinned[2] = {nullptr, nullptr};  // pinned host staging of the host-pack ingestion: [n_fields][chunk] elements
  size_t pack_bytes = 0;
  hipEvent_t pack_done[2] = {nullptr, nullptr};
  // Device-buffer pool for dataset storage: nos_dataset_destroy parks the buffer here, the next dataset of a similar
  // size takes it over (the reference's outer loop re-Solves up to 10 times with correspondences of similar count).
  struct PoolEntry {
    void* ptr;
    size_t bytes;
  };
  std::vector<PoolEntry> pool;
  size_t pool_bytes = 0;
  bool pool_enabled = true;  // Settings::pool
  const void* last_kernel = nullptr;  // host function of the hot-path kernel launched last on this device (nos_ctx_last_kernel)
  bool cluster_gave_up = false;  // the last one-launch solve on this device timed out waiting for its grid (shared GPU)
  std::chrono::steady_clock::time_point cluster_gave_up_at{};
  // Ranks of a device-memory mailbox communicator must agree on the loop form of every solve (the one-launch loop and the
  // launch-per-iteration loop exchange through different protocols), so there the pause after a give-up is counted in
  // SOLVES — every rank gives up in the same solve and counts the same calls — not in wall time.
  int cluster_paused_solves = 0;
  int cluster_next_pause = 64;  // doubles with every give-up in a row (up to 65 536 solves), back to 64 after a one-launch solve that finished
  std::vector<hipEvent_t> prof_events;
  size_t prof_used = 0;
  bool prof_on = false;
  int prof_every = 1;       // time every prof_every-th launch (event records cost ≈ 1 µs of host time each)
  long prof_launches = 0;
};


}  // namespace nosd

namespace nosd {
constexpr int kMapReferenceFmaMask = (4 | 8 | 16 | 32) | ((1 | 4 | 8 | 32 | 256) << 8) | (0x1ff << 17);
// Experiment / test knobs.  Read from the environment ONCE, when the context is created (nos_ctx_create), and
// changed afterwards only through nos_ctx_set_option: nothing on the solve / accumulate path calls getenv().
struct Settings {
  int plane_skew = 1088;     // NOS_PLANE_SKEW      elements between consecutive planes beyond n_padded
  int sc1 = 1;               // NOS_SC1             write-through rows instead of release / acquire fences
  int nt = -1;               // NOS_NT              -1 auto, 0 / 1 force non-temporal loads off / on
  int fused = 1;             // NOS_FUSED           in-launch final reduce
  int lm_fused = 1;          // NOS_LM_FUSED        LM step in the finishing workgroup
  int lm_window = 3;         // NOS_LM_WINDOW       launches kept in flight by the device loop
  int lm_single = 1;         // NOS_LM_SINGLE       whole solve in one workgroup for tiny problems
  int lm_cluster_retry_ms = 5000;  // after a one-launch solve gave up (GPU shared): how long the context goes straight to the launch-per-iteration loop
  int lm_cluster = 1;        // NOS_LM_CLUSTER      whole solve in one launch (resident on chip, or streamed per iteration beyond that); 5 = all-reduce stage 1 always through sc1 stores, 4 = resident form only, 3 = (round 2's counter all-reduce: removed in round 4, behaves like 1), 2 = one item per lane, 0 = off
  int pool = 1;              // NOS_POOL            device-buffer pool
  int tile_log2 = -1;        // NOS_TILE_LOG2       -1 = by element type (fp64 planar, fp32 1024-item tiles), 0 = planar
  int ingest = 0;            // NOS_INGEST          0 auto, 1 pack (host gather), 2 unpack (device)
  int ingest_threads = 0;    // NOS_INGEST_THREADS  0 = min(16, hw / 2)
  int indexed_bpc = 1;       // NOS_INDEXED_BPC
  int match_dense = 1;       // NOS_MATCH_DENSE
  int map_compact_keys = 1;  // NOS_MAP_COMPACT_KEYS  sort voxels / scan cells by their index inside the bounding box (0: 63-bit packed keys)
  int pgo_host_scalars = 0;  // NOS_PGO_HOST_SCALARS
  int pgo_precond = 1;       // NOS_PGO_PRECOND     0 block-Jacobi only, 1 two-level (rigid-motion coarse space)
  int pgo_agg = 48;          // NOS_PGO_AGG         poses per aggregate of the coarse level
  int pgo_coarse_probe = 0;  // NOS_PGO_COARSE_PROBE 1 = coarse operator probed with 18 masked products (round 2) instead of assembled
  int pgo_block = 1;         // NOS_PGO_BLOCK       block-local product (entry lists built by nos_pgo_create); 0 = owner-computes sweeps
  // NOS_MAP_REFERENCE_EXACT builds (mapexact_kernels.hpp): which multiply-adds are fused and which Eigen release's
  // deflation test / shift guard is followed.  Defaults = what reproduces the reference's captured x86-64 runs;
  // map_fma_mask = 0 follows the aarch64 captures (tests/test_reference_ndt_runs.py).
  int map_fma_mask = kMapReferenceFmaMask;
  int map_eigen_version = 34;
  int debug_cluster_abort = 0;  // test hook (no environment name): the next one-launch solve finds `abort` raised; 2 = and the
                                // give-up is remembered like a real one (the lm_cluster_retry_ms latch)
  int lm_cluster_max_blocks = 256;  // NOS_LM_CLUSTER_MAX_BLOCKS  workgroups of the one-launch loop (rehearsals: ranks sharing a GPU)
};
}  // namespace nosd

struct nos_ctx {
  std::recursive_mutex mu;  // serialises every entry point that touches per-context state (see nosd::CtxGuard)
  nosd::Settings settings;
  std::vector<nosd::DeviceSlot> slots;
  int blocks_per_cu = 0;  // 0 = default
  int variant = 0;        // 0 = default; tuning knob (see pick_variant)
  int tile_log2 = -1;     // -1 = default; 0 = planar
  ncclComm_t comm = nullptr;  // set by nos_ctx_comm_init: accumulate results are summed over its ranks
  int comm_ranks = 1;
  int comm_rank = 0;
  // shared-memory mailbox communicator (nos_ctx_comm_init_shm): the sums are exchanged inside the launch
  void* shm_host = nullptr;            // mmap of the POSIX shm segment
  size_t shm_bytes = 0;
  double* shm_dev = nullptr;           // its device address (hipHostRegister, mapped)
  unsigned long long* d_round = nullptr;  // device word: exchange rounds completed
  nos::Mailbox* d_mail = nullptr;         // device copy of the descriptor the kernels read
  // device-memory mailbox (nos_ctx_comm_init_shm_device): every rank's slot buffer, fine-grained device memory
  double* ipc_own = nullptr;              // this rank's buffer
  std::vector<double*> ipc_peers;         // [rank] → that rank's buffer as mapped here (own buffer at comm_rank)
  double** d_peers = nullptr;             // device copy of ipc_peers
};

namespace nosd {

// One solve / accumulate / dataset create / destroy at a time per context: the per-slot state (sequence words,
// partial rows, tickets, loop state, pinned result block, staging buffers, buffer pool) is shared by every object
// created on the context — e.g. by every drop-in solver object with the same device list (AcquireRuntime).  The
// reference's solver objects are independent per instance; this lock gives the same guarantee to threads that each own
// their solver.  Recursive: entry points call each other (match → dataset create).
struct CtxGuard {
  std::unique_lock<std::recursive_mutex> lock;
  explicit CtxGuard(const nos_ctx* ctx) {
    if (ctx != nullptr) lock = std::unique_lock<std::recursive_mutex>(const_cast<nos_ctx*>(ctx)->mu);
  }
};

struct Shard {
  int slot = 0;
  nos::TiledLayout layout{};
  void* data = nullptr;
  size_t bytes = 0;
  size_t capacity = 0;        // size of the allocation behind `data` (>= bytes when it came from the pool)
  bool pooled = false;        // data goes back to the slot's pool on destroy
  // voxel-indexed datasets (kKindNdtIndexed): data = 3 point planes; plus
  int32_t* index = nullptr;   // n_slots planes of n_padded voxel ids
  void* table = nullptr;      // [n_voxels][16] voxel records
  bool one_block = false;     // index and table live inside the (pooled) allocation behind `data`: nothing else to free
  int n_slots = 0;
  size_t n_voxels = 0;
};

}  // namespace nosd

struct nos_dataset {
  nos_ctx* ctx = nullptr;
  int kind = 0;
  int dtype = NOS_F64;
  int n_fields = 0;
  size_t n = 0;
  size_t tile = 0;
  int simd_class = 0;  // nos_dataset_set_simd_class: the semantics of the reference's fp32 (SIMD) solver classes
  std::vector<nosd::Shard> shards;
};

struct nos_ndt_map {
  nos_ctx* ctx = nullptr;
  size_t n_voxels = 0;  // valid voxels only
  double* d_mean = nullptr;
  double* d_sqrt_info = nullptr;
  uint32_t* d_orig_id = nullptr;
  uint64_t* d_cell_key = nullptr;
  uint32_t* d_cell_start = nullptr;
  uint32_t* d_cell_count = nullptr;
  unsigned long long* d_n_matches = nullptr;
  uint32_t* d_dense_begin = nullptr;  // dense grid offsets (null when the bounding box is too large)
  double* d_record = nullptr;         // [V][4] candidate records of the dense path
  void* d_block = nullptr;            // the ONE allocation behind d_mean, d_sqrt_info, the hash table, d_dense_begin, d_record
  nos::MapView view{};
};

struct nos_scan {
  nos_ctx* ctx = nullptr;
  size_t n = 0;
  double* d_planes = nullptr;  // [3][n]
  uint32_t* d_order = nullptr; // after nos_scan_sort_by_cell: original index of the point stored at each position
};

namespace nosd {

constexpr int kSeqSlot = 32;  // index (in doubles) of the sequence word inside the pinned block
constexpr int kCommErrorSlot = 40;  // index (in doubles) of the mailbox time-out flag (an unsigned int) in that block

template <typename T>
hipError_t upload(T** dptr, const std::vector<T>& host) {
  const size_t bytes = std::max<size_t>(host.size(), 1) * sizeof(T);
  hipError_t e = hipMalloc(reinterpret_cast<void**>(dptr), bytes);
  if (e == hipSuccess && !host.empty()) e = hipMemcpy(*dptr, host.data(), host.size() * sizeof(T), hipMemcpyHostToDevice);
  return e;
}

int pool_alloc(DeviceSlot& slot, size_t bytes, void** ptr, size_t* capacity);
void pool_release(DeviceSlot& slot, void* ptr, size_t capacity);

// Scratch buffers of one call, freed when it returns.  Without a slot: one hipMalloc / hipFree per buffer (round 1).  With a
// slot (round 4): an ARENA — buffers are carved out of a few large slabs that come from the slot's buffer pool and go back
// to it, so a map build no longer pays ≈ 20 hipMalloc + hipFree pairs (each hipFree also waits for the device) per call,
// and repeated builds of similar size pay none.  reserve() sizes the next slab when the caller knows what is coming.
struct DeviceBuffers {
  std::vector<void*> ptrs;  // buffers allocated one by one (and rocPRIM temporaries the callers push here)
  struct Slab {
    void* ptr;
    size_t capacity, used;
  };
  std::vector<Slab> slabs;
  DeviceSlot* slot = nullptr;
  size_t next_slab = 0;
  DeviceBuffers() = default;
  explicit DeviceBuffers(DeviceSlot* s) : slot(s) {}
  DeviceBuffers(const DeviceBuffers&) = delete;
  DeviceBuffers& operator=(const DeviceBuffers&) = delete;
  ~DeviceBuffers() {
    for (void* p : ptrs)
      if (p) (void)hipFree(p);
    if (!slabs.empty()) (void)hipStreamSynchronize(slot->stream);  // nothing in flight may still use a slab that goes back
    for (Slab& sl : slabs) pool_release(*slot, sl.ptr, sl.capacity);
  }
  void reserve(size_t bytes) { next_slab = bytes; }
  hipError_t alloc_bytes(void** out, size_t bytes) {
    bytes = (std::max<size_t>(bytes, 1) + 255) & ~size_t(255);
    *out = nullptr;
    if (slot == nullptr) {
      hipError_t e = hipMalloc(out, bytes);
      if (e == hipSuccess) ptrs.push_back(*out);
      return e;
    }
    if (slabs.empty() || slabs.back().capacity - slabs.back().used < bytes) {
      const size_t grown = slabs.empty() ? size_t(8) << 20 : std::min<size_t>(2 * slabs.back().capacity, size_t(1) << 30);
      const size_t want = std::max(std::max(bytes, next_slab), grown);
      next_slab = 0;
      void* p = nullptr;
      size_t cap = 0;
      if (pool_alloc(*slot, want, &p, &cap) != 0) return hipErrorOutOfMemory;
      slabs.push_back(Slab{p, cap, 0});
    }
    Slab& sl = slabs.back();
    *out = static_cast<char*>(sl.ptr) + sl.used;
    sl.used += bytes;
    return hipSuccess;
  }
  template <typename T>
  hipError_t alloc(T** out, size_t count) {
    void* p = nullptr;
    const hipError_t e = alloc_bytes(&p, std::max<size_t>(count, 1) * sizeof(T));
    *out = static_cast<T*>(p);
    return e;
  }
};


// What one accumulate call computes; POD so the same code path serves all three problems.
struct Request {
  int problem;  // 6, 3, 2 (reprojection)
  double R[9];
  double t[3];
  double intr[4];
  double min_depth;
  nos_loss loss;
  int loss_kind;
  int n_out;
};

template <typename T>
inline void fill_loss(const nos_loss* loss, T& la, T& lb, T& lc) {
  la = lb = lc = T(0);
  if (!loss) return;
  if (loss->kind == NOS_LOSS_EXPONENTIAL) {
    la = T(loss->a);
    lb = T(loss->b);
    lc = T(2.0 * loss->a * loss->b);
  } else if (loss->kind == NOS_LOSS_HUBER) {
    la = T(loss->a);
    lb = T(loss->a * loss->a);
    lc = T(2.0 * loss->a);
  }
}

// nos_core.hip
int pool_alloc(DeviceSlot& slot, size_t bytes, void** ptr, size_t* capacity);
void pool_release(DeviceSlot& slot, void* ptr, size_t capacity);
int env_int(const char* name, int dflt);
size_t elem_size(int dtype);
int dataset_new(nos_ctx* ctx, int kind, size_t n, int dtype, nos_dataset** out, nos_dataset** made);
int zero_pad(int dtype, int n_fields, const nos::TiledLayout& L, void* dst, hipStream_t stream);
int unpack_records(int dtype, const unsigned char* d_rec, size_t stride, const nos::FieldOffsets& fo, int n_fields,
                   size_t first, size_t count, const nos::TiledLayout& L, void* dst, hipStream_t stream);
// nos_match.hip: matcher tables from device-resident voxel statistics (valid may be null = all valid)
int map_create_device(nos_ctx* ctx, size_t n_voxels, const double* d_means, const double* d_S, const unsigned char* d_valid,
                      double search_radius_sq, nos_ndt_map** out_map);
// nos_indexed.hip
int launch_indexed(const nos_dataset* ds, const Shard& sh, const Request& rq, double* partials,
                   const nos::FusedFinal& fin, hipStream_t stream, int* rows_out);

}  // namespace nosd
