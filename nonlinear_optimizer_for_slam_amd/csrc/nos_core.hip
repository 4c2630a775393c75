// nos_core.hip — contexts, flat datasets, the accumulate entry points, RCCL hook (C ABI of include/nos.h).
//
// Owns contexts (per-device stream + workspaces), device-resident tiled-SoA datasets and the launch logic around
// the kernels in assemble_kernels.hpp.  There is no CPU fallback: without a usable HIP device every entry point
// fails with NOS_ERR_NO_DEVICE / NOS_ERR_HIP.
#include "nos_internal.hpp"
#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
#include <emmintrin.h>  // full-line non-temporal stores of the host-pack ingestion
#endif

#include <cxxabi.h>

namespace nosd {

namespace {
thread_local std::string g_last_error;
}

int fail(int status, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_last_error = buf;
  return status;
}

const char* last_error_text() { return g_last_error.c_str(); }
void clear_last_error() { g_last_error.clear(); }

// Directory of the shared object that provides `symbol` in this process ("" if unknown).
static std::string dir_of_symbol(const void* symbol, std::string* file_out = nullptr) {
  Dl_info info{};
  if (symbol == nullptr || dladdr(symbol, &info) == 0 || info.dli_fname == nullptr) return std::string();
  char resolved[PATH_MAX];
  std::string file = realpath(info.dli_fname, resolved) ? std::string(resolved) : std::string(info.dli_fname);
  if (file_out) *file_out = file;
  const size_t slash = file.rfind('/');
  return slash == std::string::npos ? std::string() : file.substr(0, slash);
}

// librccl is bound with dlopen on first use.  Search order: $NOS_RCCL_PATH, then the librccl that sits NEXT TO the HIP
// runtime already mapped into the process (so runtime and collectives always come from one ROCm tree — a process that
// imported torch first runs on torch's bundled runtime and gets torch's bundled librccl; one that did not gets the
// system pair), then the loader's default search.
namespace {
void LoadRccl(RcclApi& api) {
  std::vector<std::string> names;
  if (const char* forced = getenv("NOS_RCCL_PATH")) names.push_back(forced);
  const std::string hip_dir = dir_of_symbol(reinterpret_cast<const void*>(&hipGetDeviceCount));
  if (!hip_dir.empty()) {
    names.push_back(hip_dir + "/librccl.so.1");
    names.push_back(hip_dir + "/librccl.so");
  }
  names.push_back("librccl.so.1");
  names.push_back("librccl.so");
  names.push_back("/opt/rocm/lib/librccl.so.1");
  for (const std::string& n : names) {
    api.handle = dlopen(n.c_str(), RTLD_NOW | RTLD_LOCAL);
    if (api.handle) break;
  }
  if (!api.handle) return;
  api.GetUniqueId = reinterpret_cast<decltype(api.GetUniqueId)>(dlsym(api.handle, "ncclGetUniqueId"));
  api.CommInitRank = reinterpret_cast<decltype(api.CommInitRank)>(dlsym(api.handle, "ncclCommInitRank"));
  api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(dlsym(api.handle, "ncclCommDestroy"));
  api.AllReduce = reinterpret_cast<decltype(api.AllReduce)>(dlsym(api.handle, "ncclAllReduce"));
  api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(dlsym(api.handle, "ncclGetErrorString"));
  api.CommCount = reinterpret_cast<decltype(api.CommCount)>(dlsym(api.handle, "ncclCommCount"));
  api.GetVersion = reinterpret_cast<decltype(api.GetVersion)>(dlsym(api.handle, "ncclGetVersion"));
  api.ok = api.GetUniqueId && api.CommInitRank && api.CommDestroy && api.AllReduce && api.GetErrorString;
  if (api.ok) dir_of_symbol(reinterpret_cast<const void*>(api.AllReduce), &api.path);
}
}  // namespace

// Process-wide, bound once: two threads that each own a context may get here at the same time (std::call_once).
RcclApi* Rccl() {
  static RcclApi api;
  static std::once_flag once;
  std::call_once(once, [] { LoadRccl(api); });
  return &api;
}


// Default layout by element type (-1 = this rule; NOS_TILE_LOG2 / the "tile_log2" option / nos_ctx_set_layout override it):
//   fp64: planar planes with a skew (measured best and robust across sizes);
//   fp32: tiles of 1024 correspondences — one kernel chunk (512 lanes x 2) is one contiguous 60 KB block of memory;
//         measured at 10 M: planar 16-byte loads 6.37 TB/s, tiled 8-byte loads with the next chunk prefetched 6.94 TB/s
//         (profiles/r02_tune_f32_layout.txt).
constexpr int kDefaultTileLog2 = -1;
constexpr int kDefaultTileLog2F32 = 10;

size_t elem_size(int dtype) { return dtype == NOS_F32 ? sizeof(float) : sizeof(double); }

int env_int(const char* name, int dflt) {
  const char* v = getenv(name);
  if (!v || !*v) return dflt;
  return atoi(v);
}

// ------------------------------------------------------------------ layout

nos::TiledLayout make_layout(size_t n, int n_fields, int tile_log2, int plane_skew) {
  nos::TiledLayout L{};
  L.n = n;
  if (tile_log2 <= 0) {
    // planar: pad to the largest chunk any kernel variant uses
    const size_t pad = 4096;
    L.n_padded = ((n + pad - 1) / pad) * pad;
    if (L.n_padded == 0) L.n_padded = pad;
    L.tile_stride = 0;
    // planes are skewed against each other so that the 15 concurrent streams of a block never start at the same
    // offset modulo a large power of two (n_padded itself often is one)
    L.field_stride = L.n_padded + size_t(plane_skew);
    L.tile_shift = 40;
    L.tile_mask = 0xFFFFFFFFu;
  } else {
    const size_t tile = size_t(1) << tile_log2;
    L.n_padded = ((n + tile - 1) / tile) * tile;
    if (L.n_padded == 0) L.n_padded = tile;
    L.tile_stride = tile * size_t(n_fields);
    L.field_stride = tile;
    L.tile_shift = uint32_t(tile_log2);
    L.tile_mask = uint32_t(tile - 1);
  }
  return L;
}

size_t layout_elems(const nos::TiledLayout& L, int n_fields) {
  return (L.tile_stride == 0 ? L.field_stride : L.n_padded) * size_t(n_fields);
}

// ------------------------------------------------------------------ launch variants

// Number of compiled geometry variants per dtype (see the NOS_CASE tables below; index 0
// is the default).
constexpr int kNumVariants = 14;

// Host function of the hot-path kernel the current thread launched last (launch_variant / launch_single); copied into
// the device slot by launch_assemble_raw so that nos_ctx_last_kernel can name the instantiation that actually ran.
thread_local const void* t_last_kernel = nullptr;

template <typename Problem, typename T, int ITEMS, int BLOCK, int MINW, int PREFETCH = 0>
int launch_variant(const nos::TiledLayout& L, const typename Problem::Params& P, int grid_cap, int num_cus_hint,
                   bool nt, double* partials, const nos::FusedFinal& fin_in, hipStream_t stream, int* rows_out) {
  constexpr uint32_t kChunk = BLOCK * ITEMS;
  if (L.n_padded % kChunk != 0) return fail(NOS_ERR_INVALID_ARGUMENT, "n_padded %% chunk != 0");
  if (L.tile_stride != 0 && ((size_t(L.tile_mask) + 1) % kChunk) != 0)
    return fail(NOS_ERR_INVALID_ARGUMENT, "tile not a multiple of the kernel chunk");
  const uint64_t n_chunks64 = L.n_padded / kChunk;
  if (n_chunks64 > 0xFFFFFFFFull) return fail(NOS_ERR_UNSUPPORTED, "dataset too large for one shard");
  const uint32_t n_chunks = uint32_t(n_chunks64);
  int grid = int(std::min<uint64_t>(n_chunks, uint64_t(grid_cap)));
  if (grid < 1) grid = 1;
  if (grid > kMaxPartialRows) grid = kMaxPartialRows;
  nos::FusedFinal fin = fin_in;
  // write-through hand-off only in the geometry it is documented valid for: at most one workgroup per CU
  fin.write_through = (fin.counter != nullptr && grid <= num_cus_hint && fin_in.write_through != 0) ? 1 : 0;  // in: allowed (settings.sc1)
  const auto kernel = nt ? nos::assemble_kernel<Problem, T, ITEMS, BLOCK, MINW, true, PREFETCH>
                         : nos::assemble_kernel<Problem, T, ITEMS, BLOCK, MINW, false, PREFETCH>;
  t_last_kernel = reinterpret_cast<const void*>(kernel);
  hipLaunchKernelGGL(kernel, dim3(grid), dim3(BLOCK), 0, stream, L, P, n_chunks, partials, fin);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(NOS_ERR_HIP, "assemble launch failed: %s", hipGetErrorString(e));
  *rows_out = grid;
  return NOS_OK;
}

template <typename Problem, typename T>
int launch_by_variant(int variant, int blocks_per_cu, int num_cus, const nos::TiledLayout& L,
                      const typename Problem::Params& P, bool nt, double* partials, const nos::FusedFinal& fin,
                      hipStream_t stream, int* rows_out) {
  if (variant < 0 || variant >= kNumVariants) variant = 0;
#define NOS_CASE(idx, ITEMS_, BLOCK_, MINW_, BPC_)                                                  \
  case idx: {                                                                                       \
    const int bpc = blocks_per_cu > 0 ? blocks_per_cu : BPC_;                                       \
    return launch_variant<Problem, T, ITEMS_, BLOCK_, MINW_>(L, P, bpc * num_cus, num_cus, nt, partials, fin, stream, \
                                                             rows_out);                             \
  }
#define NOS_CASE_PP(idx, ITEMS_, BLOCK_, MINW_, BPC_)                                               \
  case idx: {                                                                                       \
    const int bpc = blocks_per_cu > 0 ? blocks_per_cu : BPC_;                                       \
    return launch_variant<Problem, T, ITEMS_, BLOCK_, MINW_, 3>(L, P, bpc * num_cus, num_cus, nt, partials, fin, \
                                                                stream, rows_out);                  \
  }
  // variant 0 = the library's choice for this problem / element type / layout (round 3, tools/exp/tune_stream.hip on
  // MI355X, profiles/r03_tune_*.txt):
  //   reprojection (5 planes: a chunk is only 40 B / 20 B per lane, so what a wave keeps in flight decides) — the ping-pong
  //     form, two named buffers and counted waits: fp64 21.3 -> 17.5 us per launch at 2 M, fp32 16.7 -> 14.5 us;
  //   NDT fp32 — 8-byte loads of two items, two waves per SIMD, no software prefetch: 89.0 us per launch at 10 M against
  //     100.5 us of round 2's prefetched form (its register copies forced full waits) and a loads-only floor of 89.8 us;
  //   NDT fp64 — one item per lane, 512-thread blocks (unchanged).
  // The default build carries the geometries something selects by default or a test drives; `make ALL_VARIANTS=1`
  // (-DNOS_ALL_VARIANTS) compiles every geometry ever tried for tools/tune_*.py and the geometry sweep test.
  constexpr bool kReproj = Problem::kFields == 5;
  if constexpr (sizeof(T) == 8) {
    if (variant == 0 && kReproj) variant = 7;
    switch (variant) {
      NOS_CASE(0, 1, 512, 3, 1)
      NOS_CASE(1, 1, 256, 3, 2)
      NOS_CASE(3, 2, 256, 2, 2)
      NOS_CASE_PP(7, 1, 512, kReproj ? 3 : 2, 1)   // ping-pong, one item per lane (two 15-plane buffers need the 256-register budget)
#ifdef NOS_ALL_VARIANTS
      NOS_CASE(2, 1, 256, 2, 2)
      NOS_CASE(4, 2, 512, 2, 1)
      NOS_CASE_PP(8, 2, 512, 2, 1)
#endif
    }
  } else {
    if (variant == 0) variant = kReproj ? 11 : 1;
    switch (variant) {
      NOS_CASE(1, 2, 512, 2, 1)      // 8-byte loads of two items, two waves per SIMD
      NOS_CASE_PP(11, 2, 512, 2, 1)  // ping-pong
#ifdef NOS_ALL_VARIANTS
      NOS_CASE(2, 1, 256, 4, 2)
      NOS_CASE(3, 2, 256, 5, 2)
      NOS_CASE(4, 2, 256, 4, 2)
      NOS_CASE(5, 1, 1024, 4, 1)  // four waves per SIMD, 4-byte loads
      NOS_CASE(6, 2, 1024, 4, 1)  // four waves per SIMD, 8-byte loads
      NOS_CASE(9, 2, 256, 3, 3)      // three waves per SIMD from three small workgroups per CU
      NOS_CASE_PP(12, 4, 512, 2, 1)
      NOS_CASE(13, 4, 256, 2, 1)     // round 1's default on planar planes: 16-byte loads, one wave per SIMD
#endif
    }
  }
#undef NOS_CASE
#undef NOS_CASE_PP
  return fail(NOS_ERR_UNSUPPORTED, "launch geometry %d is not compiled into this build for this element type "
              "(default build: fp64 0, 1, 3, 7; fp32 1, 11; `make ALL_VARIANTS=1` builds the rest)", variant);
}

// Correspondences a lane of the resident one-launch solve can hold (registers + LDS), by plane count and element type.
size_t resident_items_per_lane(int n_fields, int dtype) {
  if (n_fields == 15)
    return dtype == NOS_F64 ? size_t(nos::ResidentShape<15, 8>::RI + nos::ResidentShape<15, 8>::LI)
                            : size_t(nos::ResidentShape<15, 4>::RI + nos::ResidentShape<15, 4>::LI);
  if (n_fields == 5)
    return dtype == NOS_F64 ? size_t(nos::ResidentShape<5, 8>::RI + nos::ResidentShape<5, 8>::LI)
                            : size_t(nos::ResidentShape<5, 4>::RI + nos::ResidentShape<5, 4>::LI);
  return 1;
}

// Arguments of the single-workgroup whole-solve kernel (small problems, see nos::solve_single_block_kernel).
struct SingleBlockArgs {
  // cluster form (one chunk per workgroup, whole loop in one launch) when cluster_blocks > 0
  int cluster_blocks = 0;
  int items_per_lane = 1;  // correspondences every lane keeps resident (registers + LDS, nos::ResidentShape)
  int stream_chunks = 0;   // > 0: the streaming form (nothing resident; this many chunks of 512 x SI per iteration)
  bool nt = false;         // streaming form: non-temporal loads
  bool stage1_sc1 = false; // keep stage 1 of the tagged all-reduce on sc1 stores even where a group sits on one XCD (lm_cluster 5)
  const nos::Mailbox* mail = nullptr;  // device-memory mailbox communicator: the cross-rank exchange runs inside the launch
  double* partials = nullptr;
  nos::ClusterCtl* ctl = nullptr;
  nos::LmDevice* lm;
  double* history;  // device address of the pinned cost history (may be null)
  int history_capacity;
  double* entry;    // device address of the pinned log entry
  unsigned long long* seq_host;
  unsigned long long seq;
};

template <typename Problem, typename T>
int launch_single(const nos::TiledLayout& L, const typename Problem::Params& P, const SingleBlockArgs& a, hipStream_t stream) {
  constexpr int kBlock = 512;
  if (L.n_padded % kBlock != 0) return fail(NOS_ERR_INVALID_ARGUMENT, "n_padded %% 512 != 0");
  if (a.cluster_blocks > 0 && a.stream_chunks > 0) {
    // the whole loop in one launch, the data streamed from HBM every iteration (solve_cluster_kernel, SI > 0)
    constexpr int kSI = sizeof(T) == 8 ? 1 : 2;        // fp64: 8-byte loads of one item; fp32: 8-byte loads of two
    // fp32 prefetched the next chunk through register copies in round 2 (the geometry of launch variant 8); the copies
    // force full waits, and with the LM step out of the kernel the plain form is the faster one (tools/exp/tune_stream.hip:
    // 89.0 against 100.5 us per pass at 10 M; loads-only floor 89.8)
    constexpr bool kSPF = false;
    constexpr size_t kChunk = size_t(kBlock) * kSI;
    if (L.n_padded % kChunk != 0 || (L.tile_stride != 0 && ((size_t(L.tile_mask) + 1) % kChunk) != 0))
      return fail(NOS_ERR_INVALID_ARGUMENT, "streaming solve: layout not a multiple of the %zu-item chunk", kChunk);
    if (size_t(a.stream_chunks) * kChunk != L.n_padded) return fail(NOS_ERR_INVALID_ARGUMENT, "streaming solve: chunk count does not match the layout");
    const auto kernel = a.nt ? nos::solve_cluster_kernel<Problem, T, kBlock, 0, 0, kSI, kSPF, true>
                             : nos::solve_cluster_kernel<Problem, T, kBlock, 0, 0, kSI, kSPF, false>;
    t_last_kernel = reinterpret_cast<const void*>(kernel);
    hipLaunchKernelGGL(kernel, dim3(a.cluster_blocks), dim3(kBlock), 0, stream, L, P, a.partials, a.lm, a.ctl, a.history,
                       a.history_capacity, a.entry, a.seq_host, a.seq, uint32_t(a.stream_chunks) | (a.stage1_sc1 ? 0x80000000u : 0u),
                       a.mail);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(NOS_ERR_HIP, "streaming solve launch failed: %s", hipGetErrorString(e));
    return NOS_OK;
  }
  if (a.cluster_blocks > 0) {
    using Shape = nos::ResidentShape<Problem::kFields, int(sizeof(T))>;
    if (a.items_per_lane < 1 || a.items_per_lane > Shape::RI + Shape::LI)
      return fail(NOS_ERR_INVALID_ARGUMENT, "resident solve: %d items per lane do not fit (%d + %d)", a.items_per_lane,
                  Shape::RI, Shape::LI);
    const auto kernel = nos::solve_cluster_kernel<Problem, T, kBlock, Shape::RI, Shape::LI>;
    const size_t lds_items = a.items_per_lane > Shape::RI ? size_t(a.items_per_lane - Shape::RI) : 0;
    const size_t dyn_bytes = lds_items * size_t(nos::resident_fields<Problem::kFields, sizeof(T)>()) * kBlock * sizeof(T);
    // Dynamic LDS beyond the default limit has to be granted per kernel AND per device (the attribute belongs to the
    // function on the current device): asked for on every launch that needs it — a host-side call of about a microsecond,
    // once per solve — instead of remembered in a process-wide static that a second device or thread would trip over.
    if (dyn_bytes > size_t(48) * 1024) {
      const hipError_t ea = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, int(dyn_bytes));
      // NOS_ERR_UNSUPPORTED: the one status for which lm_solve falls back to the launch-per-iteration loop
      if (ea != hipSuccess) return fail(NOS_ERR_UNSUPPORTED, "resident solve: %zu bytes of LDS refused: %s", dyn_bytes, hipGetErrorString(ea));
    }
    t_last_kernel = reinterpret_cast<const void*>(kernel);
    hipLaunchKernelGGL(kernel, dim3(a.cluster_blocks), dim3(kBlock), dyn_bytes, stream, L, P, a.partials, a.lm, a.ctl,
                       a.history, a.history_capacity, a.entry, a.seq_host, a.seq,
                       uint32_t(a.items_per_lane) | (a.stage1_sc1 ? 0x80000000u : 0u), a.mail);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(NOS_ERR_HIP, "cluster solve launch failed: %s", hipGetErrorString(e));
    return NOS_OK;
  }
  const uint32_t n_chunks = uint32_t((std::max<uint64_t>(L.n, 1) + kBlock - 1) / kBlock);  // pads beyond are never read
  t_last_kernel = reinterpret_cast<const void*>(&nos::solve_single_block_kernel<Problem, T, kBlock>);
  hipLaunchKernelGGL((nos::solve_single_block_kernel<Problem, T, kBlock>), dim3(1), dim3(kBlock), 0, stream, L, P, n_chunks, a.lm,
                     a.history, a.history_capacity, a.entry, a.seq_host, a.seq);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(NOS_ERR_HIP, "single-workgroup solve launch failed: %s", hipGetErrorString(e));
  return NOS_OK;
}

template <template <typename, int> class ProblemT, typename T, typename ParamsT>
int launch_by_loss(int loss_kind, int variant, int blocks_per_cu, int num_cus,
                   const nos::TiledLayout& L, const ParamsT& P, bool nt, double* partials,
                   const nos::FusedFinal& fin, hipStream_t stream, int* rows_out, const SingleBlockArgs* single = nullptr) {
  switch (loss_kind) {
    case NOS_LOSS_NONE:
      if (single) return launch_single<ProblemT<T, nos::kLossNone>, T>(L, P, *single, stream);
      return launch_by_variant<ProblemT<T, nos::kLossNone>, T>(variant, blocks_per_cu, num_cus, L, P, nt, partials,
                                                               fin, stream, rows_out);
    case NOS_LOSS_EXPONENTIAL:
      if (single) return launch_single<ProblemT<T, nos::kLossExponential>, T>(L, P, *single, stream);
      return launch_by_variant<ProblemT<T, nos::kLossExponential>, T>(variant, blocks_per_cu, num_cus, L, P, nt,
                                                                      partials, fin, stream, rows_out);
    case NOS_LOSS_HUBER:
      if (single) return launch_single<ProblemT<T, nos::kLossHuber>, T>(L, P, *single, stream);
      return launch_by_variant<ProblemT<T, nos::kLossHuber>, T>(variant, blocks_per_cu, num_cus, L, P, nt, partials,
                                                                fin, stream, rows_out);
  }
  return fail(NOS_ERR_INVALID_ARGUMENT, "unknown loss kind %d", loss_kind);
}

int check_loss(const nos_loss* loss, int* kind_out) {
  int kind = loss ? loss->kind : NOS_LOSS_NONE;
  if (kind < NOS_LOSS_NONE || kind > NOS_LOSS_HUBER) return fail(NOS_ERR_INVALID_ARGUMENT, "unknown loss kind %d", kind);
  // same argument checks as the reference constructors (loss_function.h:24-25, 53-54)
  if (kind == NOS_LOSS_EXPONENTIAL && (loss->a < 0.0 || loss->b < 0.0))
    return fail(NOS_ERR_INVALID_ARGUMENT, "exponential loss needs c1 >= 0 and c2 >= 0");
  if (kind == NOS_LOSS_HUBER && !(loss->a > 0.0)) return fail(NOS_ERR_INVALID_ARGUMENT, "huber loss needs threshold > 0");
  *kind_out = kind;
  return NOS_OK;
}

// Streaming (non-temporal) loads when the shard cannot stay resident in the 256 MiB
// Infinity Cache between iterations; default-policy loads when it can.
bool use_nontemporal(const nos_dataset* ds, const Shard& sh) {
  const int force = ds->ctx->settings.nt;
  if (force >= 0) return force != 0;
  return sh.bytes > (size_t(192) << 20);
}

int launch_assemble_raw(const nos_dataset* ds, const Shard& sh, const Request& rq, double* partials,
                        const nos::FusedFinal& fin, hipStream_t stream, int* rows_out,
                        const SingleBlockArgs* single = nullptr);

// Launches the assemble kernel; with profiling on, brackets it with an event pair on the
// same stream so its device duration can be read back later without perturbing the loop.
int launch_assemble(const nos_dataset* ds, const Shard& sh, const Request& rq, double* partials,
                    const nos::FusedFinal& fin, hipStream_t stream, int* rows_out) {
  DeviceSlot& slot = ds->ctx->slots[sh.slot];
  if (slot.prof_on && slot.prof_every == 0) ++slot.prof_launches;  // bracket form: count only
  const bool prof = slot.prof_on && slot.prof_every > 0 && (slot.prof_launches++ % slot.prof_every) == 0 &&
                    slot.prof_used + 2 <= slot.prof_events.size();
  if (prof) NOS_HIP_CHECK(hipEventRecord(slot.prof_events[slot.prof_used], stream));
  const int rc = launch_assemble_raw(ds, sh, rq, partials, fin, stream, rows_out);
  if (rc != NOS_OK) return rc;
  if (prof) {
    NOS_HIP_CHECK(hipEventRecord(slot.prof_events[slot.prof_used + 1], stream));
    slot.prof_used += 2;
  }
  return NOS_OK;
}

int launch_assemble_inner(const nos_dataset* ds, const Shard& sh, const Request& rq, double* partials,
                          const nos::FusedFinal& fin_in, hipStream_t stream, int* rows_out, const SingleBlockArgs* single);

int launch_assemble_raw(const nos_dataset* ds, const Shard& sh, const Request& rq, double* partials,
                        const nos::FusedFinal& fin_in, hipStream_t stream, int* rows_out, const SingleBlockArgs* single) {
  t_last_kernel = nullptr;
  const int rc = launch_assemble_inner(ds, sh, rq, partials, fin_in, stream, rows_out, single);
  if (rc == NOS_OK && t_last_kernel != nullptr) ds->ctx->slots[sh.slot].last_kernel = t_last_kernel;
  return rc;
}

int launch_assemble_inner(const nos_dataset* ds, const Shard& sh, const Request& rq, double* partials,
                          const nos::FusedFinal& fin_in, hipStream_t stream, int* rows_out, const SingleBlockArgs* single) {
  const nos_ctx* ctx = ds->ctx;
  nos::FusedFinal fin = fin_in;
  fin.write_through = ctx->settings.sc1;  // "allowed"; the launcher keeps it only for the geometry it is valid for
  if (ds->kind == kKindNdtIndexed) return launch_indexed(ds, sh, rq, partials, fin, stream, rows_out);
  const DeviceSlot& slot = ctx->slots[sh.slot];
  const bool nt = use_nontemporal(ds, sh);
  const int variant = ctx->variant;
  const int bpc = ctx->blocks_per_cu;
  if (rq.problem == 6) {
    if (ds->dtype == NOS_F64) {
      nos::Ndt6Params<double> P;
      for (int k = 0; k < 9; ++k) P.R[k] = rq.R[k];
      for (int k = 0; k < 3; ++k) P.t[k] = rq.t[k];
      fill_loss(&rq.loss, P.la, P.lb, P.lc);
      return launch_by_loss<nos::Ndt6Problem, double>(rq.loss_kind, variant, bpc, slot.num_cus, sh.layout, P, nt,
                                                      partials, fin, stream, rows_out, single);
    }
    nos::Ndt6Params<float> P;
    for (int k = 0; k < 9; ++k) P.R[k] = float(rq.R[k]);
    for (int k = 0; k < 3; ++k) P.t[k] = float(rq.t[k]);
    fill_loss(&rq.loss, P.la, P.lb, P.lc);
    return launch_by_loss<nos::Ndt6Problem, float>(rq.loss_kind, variant, bpc, slot.num_cus, sh.layout, P, nt,
                                                   partials, fin, stream, rows_out, single);
  }
  if (rq.problem == 3) {
    if (ds->dtype == NOS_F64) {
      nos::Ndt3Params<double> P;
      for (int k = 0; k < 4; ++k) P.R2[k] = rq.R[k];
      for (int k = 0; k < 2; ++k) P.t2[k] = rq.t[k];
      fill_loss(&rq.loss, P.la, P.lb, P.lc);
      return launch_by_loss<nos::Ndt3Problem, double>(rq.loss_kind, variant, bpc, slot.num_cus, sh.layout, P, nt,
                                                      partials, fin, stream, rows_out, single);
    }
    nos::Ndt3Params<float> P;
    for (int k = 0; k < 4; ++k) P.R2[k] = float(rq.R[k]);
    for (int k = 0; k < 2; ++k) P.t2[k] = float(rq.t[k]);
    fill_loss(&rq.loss, P.la, P.lb, P.lc);
    return launch_by_loss<nos::Ndt3Problem, float>(rq.loss_kind, variant, bpc, slot.num_cus, sh.layout, P, nt,
                                                   partials, fin, stream, rows_out, single);
  }
  if (ds->dtype == NOS_F64) {
    nos::ReprojParams<double> P;
    for (int k = 0; k < 9; ++k) P.R[k] = rq.R[k];
    for (int k = 0; k < 3; ++k) P.t[k] = rq.t[k];
    P.inv_fx = rq.intr[0];
    P.inv_fy = rq.intr[1];
    P.cx = rq.intr[2];
    P.cy = rq.intr[3];
    P.min_depth = rq.min_depth;
    nos::set_reproj_rules(P, ds->simd_class != 0);
    fill_loss(&rq.loss, P.la, P.lb, P.lc);
    return launch_by_loss<nos::ReprojProblem, double>(rq.loss_kind, variant, bpc, slot.num_cus, sh.layout, P, nt,
                                                      partials, fin, stream, rows_out, single);
  }
  nos::ReprojParams<float> P;
  for (int k = 0; k < 9; ++k) P.R[k] = float(rq.R[k]);
  for (int k = 0; k < 3; ++k) P.t[k] = float(rq.t[k]);
  P.inv_fx = float(rq.intr[0]);
  P.inv_fy = float(rq.intr[1]);
  P.cx = float(rq.intr[2]);
  P.cy = float(rq.intr[3]);
  P.min_depth = float(rq.min_depth);
  nos::set_reproj_rules(P, ds->simd_class != 0);
  fill_loss(&rq.loss, P.la, P.lb, P.lc);
  return launch_by_loss<nos::ReprojProblem, float>(rq.loss_kind, variant, bpc, slot.num_cus, sh.layout, P, nt,
                                                   partials, fin, stream, rows_out, single);
}

int launch_final(int n_out, const double* partials, int rows, double* out, hipStream_t stream) {
  if (n_out == 28)
    hipLaunchKernelGGL((nos::final_reduce_kernel<28>), dim3(1), dim3(1024), 0, stream, partials, uint32_t(rows), out);
  else
    hipLaunchKernelGGL((nos::final_reduce_kernel<10>), dim3(1), dim3(1024), 0, stream, partials, uint32_t(rows), out);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(NOS_ERR_HIP, "final reduce launch failed: %s", hipGetErrorString(e));
  return NOS_OK;
}

int build_request(int problem, const nos_dataset* ds, const double* R, int nR, const double* t, int nt,
                  const double* intr, double min_depth, const nos_loss* loss, Request* rq) {
  if (!ds) return fail(NOS_ERR_INVALID_ARGUMENT, "dataset is NULL");
  if (!R || !t) return fail(NOS_ERR_INVALID_ARGUMENT, "pose pointer is NULL");
  const int want_kind = (problem == 2) ? kKindReproj : kKindNdt;
  if (ds->kind != want_kind && !(want_kind == kKindNdt && ds->kind == kKindNdtIndexed))
    return fail(NOS_ERR_WRONG_KIND, "dataset kind does not match the entry point");
  memset(rq, 0, sizeof *rq);
  rq->problem = problem;
  for (int k = 0; k < nR; ++k) rq->R[k] = R[k];
  for (int k = 0; k < nt; ++k) rq->t[k] = t[k];
  if (problem == 2) {
    if (!intr) return fail(NOS_ERR_INVALID_ARGUMENT, "intrinsics pointer is NULL");
    for (int k = 0; k < 4; ++k) rq->intr[k] = intr[k];
    rq->min_depth = min_depth;
  }
  int kind = 0;
  int rc = check_loss(loss, &kind);
  if (rc != NOS_OK) return rc;
  rq->loss_kind = kind;
  if (loss) rq->loss = *loss;
  rq->n_out = (problem == 3) ? 10 : 28;
  return NOS_OK;
}


// Device result → pinned host block + sequence word (used after an RCCL all-reduce, where the
// in-launch final reduce cannot write to the host itself).
__global__ void publish_kernel(const double* __restrict__ src, int n, double* dst_host,
                               unsigned long long* seq_host, unsigned long long seq) {
  if (int(threadIdx.x) < n)
    __hip_atomic_store(dst_host + threadIdx.x, src[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __hip_atomic_store(seq_host, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// Mailbox descriptor for the in-launch cross-rank exchange (base == null when the context has no shm communicator).
nos::Mailbox mailbox_of(const nos_ctx* ctx, const DeviceSlot& slot) {
  nos::Mailbox mb{};
  if (ctx->shm_dev == nullptr) return mb;
  mb.base = ctx->shm_dev;
  mb.peers = ctx->d_peers;  // null: the slots are the host-memory ones behind `base`
  mb.round = ctx->d_round;
  mb.error_host = reinterpret_cast<unsigned int*>(slot.h_out_dev + kCommErrorSlot);
  mb.n_ranks = ctx->comm_ranks;
  mb.rank = ctx->comm_rank;
  return mb;
}

int check_mailbox_error(const nos_ctx* ctx, DeviceSlot& slot) {
  if (ctx->shm_dev == nullptr) return NOS_OK;
  volatile unsigned int* err = reinterpret_cast<volatile unsigned int*>(slot.h_out + kCommErrorSlot);
  if (*err != 0u) {
    *err = 0u;
    return fail(NOS_ERR_HIP, "mailbox all-reduce timed out: a peer rank did not arrive (ranks must run the same sequence of calls)");
  }
  return NOS_OK;
}

// Spin on the host-mapped sequence word the last block stores after the result; falls back
// to a stream synchronise if the word has not arrived after a generous bound, so a protocol
// error can never hang the caller.
int wait_for_sequence(DeviceSlot& slot, unsigned long long want) {
  volatile unsigned long long* seq = reinterpret_cast<volatile unsigned long long*>(slot.h_out + kSeqSlot);
  for (long spins = 0; *seq < want; ++spins) {
    if (spins > 2000000) {
      NOS_HIP_CHECK(hipSetDevice(slot.device));
      NOS_HIP_CHECK(hipStreamSynchronize(slot.stream));
      if (*seq < want) return fail(NOS_ERR_HIP, "fused final reduce did not publish its sequence word");
      break;
    }
#if defined(__x86_64__)
    __builtin_ia32_pause();
#endif
  }
  __atomic_thread_fence(__ATOMIC_ACQUIRE);
  return NOS_OK;
}
int wait_for_sequence(DeviceSlot& slot) { return wait_for_sequence(slot, slot.seq); }

// Blocking accumulate over every shard; shard sums are added on the host in shard order
// (the reference sums its per-thread partials the same way).
int accumulate_sync(nos_dataset* ds, const Request& rq, double* out) {
  nos_ctx* ctx = ds->ctx;
  const bool fused = ctx->settings.fused != 0 || ctx->shm_dev != nullptr;  // the mailbox exchange lives in the fused tail
  if (ctx->comm != nullptr) {
    // one process per GPU: local sums → RCCL all-reduce of the n_out doubles (in place, on the
    // same stream) → publish to pinned host memory.  Every rank receives identical bits.
    const Shard& sh = ds->shards[0];
    DeviceSlot& slot = ctx->slots[sh.slot];
    NOS_HIP_CHECK(hipSetDevice(slot.device));
    int rows = 0;
    nos::FusedFinal fin{slot.counter, slot.d_out, nullptr, nullptr, 0};
    int rc = launch_assemble(ds, sh, rq, slot.partials, fin, slot.stream, &rows);
    if (rc != NOS_OK) return rc;
    NOS_RCCL_CHECK(Rccl()->AllReduce(slot.d_out, slot.d_out, size_t(rq.n_out), ncclDouble, ncclSum, ctx->comm,
                                     slot.stream));
    hipLaunchKernelGGL(publish_kernel, dim3(1), dim3(64), 0, slot.stream, slot.d_out, rq.n_out, slot.h_out_dev,
                       reinterpret_cast<unsigned long long*>(slot.h_out_dev + kSeqSlot), ++slot.seq);
    NOS_HIP_CHECK(hipGetLastError());
    rc = wait_for_sequence(slot);
    if (rc != NOS_OK) return rc;
    for (int k = 0; k < rq.n_out; ++k) out[k] = slot.h_out[k];
    return NOS_OK;
  }
  for (const Shard& sh : ds->shards) {
    DeviceSlot& slot = ctx->slots[sh.slot];
    NOS_HIP_CHECK(hipSetDevice(slot.device));
    int rows = 0;
    if (fused) {
      // one launch: the last block to finish reduces all rows and writes the result plus a
      // sequence word straight into pinned host memory (no second kernel, no memcpy)
      nos::FusedFinal fin{slot.counter, nullptr, slot.h_out_dev,
                          reinterpret_cast<unsigned long long*>(slot.h_out_dev + kSeqSlot), ++slot.seq};
      fin.mail = ctx->d_mail;
      int rc = launch_assemble(ds, sh, rq, slot.partials, fin, slot.stream, &rows);
      if (rc != NOS_OK) return rc;
    } else {
      int rc = launch_assemble(ds, sh, rq, slot.partials, nos::FusedFinal{}, slot.stream, &rows);
      if (rc != NOS_OK) return rc;
      rc = launch_final(rq.n_out, slot.partials, rows, slot.d_out, slot.stream);
      if (rc != NOS_OK) return rc;
      NOS_HIP_CHECK(hipMemcpyAsync(slot.h_out, slot.d_out, sizeof(double) * rq.n_out, hipMemcpyDeviceToHost, slot.stream));
    }
  }
  for (int k = 0; k < rq.n_out; ++k) out[k] = 0.0;
  for (const Shard& sh : ds->shards) {
    DeviceSlot& slot = ctx->slots[sh.slot];
    if (fused) {
      int rc = wait_for_sequence(slot);
      if (rc == NOS_OK) rc = check_mailbox_error(ctx, slot);
      if (rc != NOS_OK) return rc;
    } else {
      NOS_HIP_CHECK(hipSetDevice(slot.device));
      NOS_HIP_CHECK(hipStreamSynchronize(slot.stream));
    }
    for (int k = 0; k < rq.n_out; ++k) out[k] += slot.h_out[k];
  }
  return NOS_OK;
}

int accumulate_async(nos_dataset* ds, const Request& rq, double* d_out) {
  if (!d_out) return fail(NOS_ERR_INVALID_ARGUMENT, "d_out is NULL");
  if (ds->shards.size() != 1) return fail(NOS_ERR_UNSUPPORTED, "async accumulate needs a single-device context");
  nos_ctx* ctx = ds->ctx;
  const Shard& sh = ds->shards[0];
  DeviceSlot& slot = ctx->slots[sh.slot];
  NOS_HIP_CHECK(hipSetDevice(slot.device));
  int rows = 0;
  if (ctx->comm != nullptr) {
    nos::FusedFinal fin{slot.counter, d_out, nullptr, nullptr, 0};
    int rc = launch_assemble(ds, sh, rq, slot.partials, fin, slot.stream, &rows);
    if (rc != NOS_OK) return rc;
    NOS_RCCL_CHECK(Rccl()->AllReduce(d_out, d_out, size_t(rq.n_out), ncclDouble, ncclSum, ctx->comm, slot.stream));
    return NOS_OK;
  }
  if (ctx->settings.fused != 0 || ctx->shm_dev != nullptr) {
    nos::FusedFinal fin{slot.counter, d_out, nullptr, nullptr, 0};
    fin.mail = ctx->d_mail;
    return launch_assemble(ds, sh, rq, slot.partials, fin, slot.stream, &rows);
  }
  int rc = launch_assemble(ds, sh, rq, slot.partials, nos::FusedFinal{}, slot.stream, &rows);
  if (rc != NOS_OK) return rc;
  return launch_final(rq.n_out, slot.partials, rows, d_out, slot.stream);
}

int time_kernel(nos_dataset* ds, const Request& rq, int repeats, double* kernel_ms, double* total_ms) {
  if (repeats < 1) repeats = 1;
  nos_ctx* ctx = ds->ctx;
  const Shard& sh = ds->shards[0];
  DeviceSlot& slot = ctx->slots[sh.slot];
  NOS_HIP_CHECK(hipSetDevice(slot.device));
  int rows = 0;
  // warm-up
  const bool fused = ctx->settings.fused != 0;
  for (int i = 0; i < 2; ++i) {
    int rc = launch_assemble(ds, sh, rq, slot.partials, nos::FusedFinal{}, slot.stream, &rows);
    if (rc != NOS_OK) return rc;
  }
  NOS_HIP_CHECK(hipEventRecord(slot.ev0, slot.stream));
  for (int i = 0; i < repeats; ++i) {
    int rc = launch_assemble(ds, sh, rq, slot.partials, nos::FusedFinal{}, slot.stream, &rows);
    if (rc != NOS_OK) return rc;
  }
  NOS_HIP_CHECK(hipEventRecord(slot.ev1, slot.stream));
  for (int i = 0; i < repeats; ++i) {
    if (fused) {
      nos::FusedFinal fin{slot.counter, slot.d_out, nullptr, nullptr, 0};
      int rc = launch_assemble(ds, sh, rq, slot.partials, fin, slot.stream, &rows);
      if (rc != NOS_OK) return rc;
    } else {
      int rc = launch_assemble(ds, sh, rq, slot.partials, nos::FusedFinal{}, slot.stream, &rows);
      if (rc != NOS_OK) return rc;
      rc = launch_final(rq.n_out, slot.partials, rows, slot.d_out, slot.stream);
      if (rc != NOS_OK) return rc;
    }
  }
  NOS_HIP_CHECK(hipEventRecord(slot.ev2, slot.stream));
  NOS_HIP_CHECK(hipEventSynchronize(slot.ev2));
  float ms01 = 0.f, ms12 = 0.f;
  NOS_HIP_CHECK(hipEventElapsedTime(&ms01, slot.ev0, slot.ev1));
  NOS_HIP_CHECK(hipEventElapsedTime(&ms12, slot.ev1, slot.ev2));
  if (kernel_ms) *kernel_ms = double(ms01) / repeats;
  if (total_ms) *total_ms = double(ms12) / repeats;
  return NOS_OK;
}

// Device-resident Levenberg-Marquardt loop (see nos::LmDevice).  The host keeps `window` launches in flight and
// reads one pinned log entry per finished iteration; nothing on the host sits between two consecutive kernels.
// With an RCCL communicator every launch is followed by the all-reduce of its sums and a one-wave step kernel, so
// all ranks advance identical states in lock-step without host synchronisation either.
int lm_solve(nos_dataset* ds, const Request& rq, const nos_lm_options* opt, double* R, int nR, double* t, int nt,
             nos_lm_report* report) {
  if (!opt) return fail(NOS_ERR_INVALID_ARGUMENT, "options pointer is NULL");
  if (ds->shards.size() != 1)
    return fail(NOS_ERR_UNSUPPORTED, "the device-resident loop needs a single-device context (use the host loop)");
  if (opt->max_iterations < 0) return fail(NOS_ERR_INVALID_ARGUMENT, "max_iterations < 0");
  nos_ctx* ctx = ds->ctx;
  const Shard& sh = ds->shards[0];
  DeviceSlot& slot = ctx->slots[sh.slot];
  NOS_HIP_CHECK(hipSetDevice(slot.device));
  const bool with_comm = ctx->comm != nullptr;
  const bool step_in_launch = !with_comm && ctx->settings.lm_fused != 0;
  int window = opt->launches_in_flight > 0 ? opt->launches_in_flight : ctx->settings.lm_window;
  window = std::max(1, std::min(window, kLogSlots - 2));

  nos::LmInitArgs init{};
  for (int k = 0; k < nR; ++k) init.R[k] = R[k];
  for (int k = 0; k < nt; ++k) init.t[k] = t[k];
  init.settings.max_iterations = opt->max_iterations;
  init.settings.gradient_tolerance = opt->gradient_tolerance;
  init.settings.parameter_tolerance = opt->parameter_tolerance;
  init.dof = rq.n_out == 28 ? 6 : 3;
  init.settings.float_schedule = (ds->simd_class != 0 && ds->kind != kKindReproj) ? 1 : 0;
  nos_host::LmState st;  // host mirror: what the log says after the last finished iteration
  if (init.dof == 6)
    nos_host::LmInit6(&st, init.R, init.t, opt->max_iterations, init.settings.float_schedule);
  else
    nos_host::LmInit3(&st, init.R, init.t, opt->max_iterations, init.settings.float_schedule);
  hipLaunchKernelGGL(nos::lm_init_kernel, dim3(1), dim3(1), 0, slot.stream, slot.d_lm, init);
  NOS_HIP_CHECK(hipGetLastError());

  unsigned long long* seq_dev = reinterpret_cast<unsigned long long*>(slot.h_out_dev + kSeqSlot);
  // Small problems: the whole loop in one workgroup and one launch (see nos::solve_single_block_kernel)
  if (ds->kind != kKindNdtIndexed && !with_comm && ctx->shm_dev == nullptr && opt->max_iterations > 0 &&
      sh.layout.n * size_t(ds->n_fields) <= nos::kSingleBlockMaxElements && ctx->settings.lm_single != 0 &&
      (opt->cost_history == nullptr || opt->max_iterations <= kHistCapacity)) {
    SingleBlockArgs single{};
    single.lm = slot.d_lm;
    single.history = opt->cost_history ? slot.h_hist_dev : nullptr;
    single.history_capacity = kHistCapacity;
    single.entry = slot.h_log_dev;
    single.seq_host = seq_dev;
    single.seq = ++slot.seq;
    int rows = 0;
    int rc = launch_assemble_raw(ds, sh, rq, slot.partials, nos::FusedFinal{}, slot.stream, &rows, &single);
    if (rc != NOS_OK) return rc;
    rc = wait_for_sequence(slot, single.seq);
    if (rc != NOS_OK) return rc;
    const double* e = slot.h_log;
    for (int k = 0; k < 9; ++k) st.R[k] = e[nos::kLogR + k];
    for (int k = 0; k < 3; ++k) st.t[k] = e[nos::kLogT + k];
    st.lambda = e[nos::kLogLambda];
    st.previous_cost = e[nos::kLogPrevCost];
    st.cost = e[nos::kLogCost];
    st.iteration = int(e[nos::kLogIteration]);
    st.done = int(e[nos::kLogDone]);
    st.ok = int(e[nos::kLogOk]);
    const int executed = int(e[nos::kLogExecuted]);
    if (slot.prof_on && slot.prof_every == 0) slot.prof_launches += executed;  // bracket profiling counts passes over the data
    if (opt->cost_history != nullptr)
      for (int k = 0; k < executed && k < opt->max_iterations; ++k) opt->cost_history[k] = slot.h_hist[k];
    for (int k = 0; k < nR; ++k) R[k] = st.R[k];
    for (int k = 0; k < nt; ++k) t[k] = st.t[k];
    if (report) {
      report->iterations = st.iteration;
      report->ok = st.ok;
      report->launches = 1;
      report->fallback = 0;
      report->printed_cost = st.previous_cost;
      report->last_cost = st.cost;
      report->final_lambda = st.lambda;
    }
    return NOS_OK;
  }
  // Mid-size problems: one chunk per workgroup, every workgroup resident, the whole loop in one launch
  // (nos::solve_cluster_kernel).  If a wait inside times out (grid not fully resident, e.g. the GPU is shared) the
  // launch gives up and the code below runs the loop with one launch per iteration instead.
  // One 512-thread workgroup per CU at most (all of them must be resident at once); every lane keeps items_per_lane
  // correspondences in registers + LDS.  lm_cluster: 0 off, 1 on, 2 = only the one-item-per-lane form of round 1.
  // lm_cluster_max_blocks: rehearsals of several ranks on ONE GPU give every rank its share of the CUs (all workgroups of
  // all ranks have to be resident together)
  const size_t max_blocks = std::min<size_t>(std::min<size_t>(nos::kClusterMaxBlocks, size_t(slot.num_cus)),
                                             size_t(std::max(1, ctx->settings.lm_cluster_max_blocks)));
  // A device-memory mailbox communicator keeps the one-launch loop: its exchange is a third stage inside the launch
  // (solve_cluster_kernel).  RCCL and the host-memory mailbox run one launch per iteration, as before.
  const bool mailbox_in_launch = ctx->shm_dev != nullptr && ctx->d_peers != nullptr && ctx->d_mail != nullptr &&
                                 ds->kind != kKindNdtIndexed;
  int fell_back = 0;
  const size_t cluster_blocks = std::min<size_t>(max_blocks, (sh.layout.n + 511) / 512);
  const size_t items_per_lane = cluster_blocks > 0 ? (sh.layout.n + cluster_blocks * 512 - 1) / (cluster_blocks * 512) : 0;
  const size_t resident_capacity = ctx->settings.lm_cluster == 2 ? 1 : resident_items_per_lane(ds->n_fields, ds->dtype);
  // Beyond what the chip can keep resident the same one-launch loop STREAMS the data every iteration (lm_cluster 1 only;
  // 4 = resident form only, as before).  The chunk count must fit the kernel's 32-bit counter.
  const size_t stream_chunk = size_t(512) * (ds->dtype == NOS_F64 ? 1 : 2);
  const bool resident_fits = items_per_lane >= 1 && items_per_lane <= resident_capacity;
  const bool stream_form = !resident_fits && (ctx->settings.lm_cluster == 1 || ctx->settings.lm_cluster == 5) && items_per_lane >= 1 &&
                           sh.layout.n_padded % stream_chunk == 0 && sh.layout.n_padded / stream_chunk < (size_t(1) << 31) &&
                           (sh.layout.tile_stride == 0 || (size_t(sh.layout.tile_mask) + 1) % stream_chunk == 0);
  bool paused = false;
  if (mailbox_in_launch) {
    paused = slot.cluster_paused_solves > 0;  // the same count on every rank (see DeviceSlot)
    if (paused) --slot.cluster_paused_solves;
  } else {
    if (slot.cluster_gave_up &&
        std::chrono::steady_clock::now() - slot.cluster_gave_up_at > std::chrono::milliseconds(ctx->settings.lm_cluster_retry_ms))
      slot.cluster_gave_up = false;  // try the one-launch form again
    paused = slot.cluster_gave_up;
  }
  if (ds->kind != kKindNdtIndexed && !with_comm && (ctx->shm_dev == nullptr || mailbox_in_launch) && opt->max_iterations > 0 &&
      cluster_blocks >= 1 && (resident_fits || stream_form) && !paused &&
      ctx->settings.lm_cluster != 0 && (opt->cost_history == nullptr || opt->max_iterations <= kHistCapacity)) {
    SingleBlockArgs cl{};
    cl.cluster_blocks = int(cluster_blocks);
    cl.items_per_lane = int(items_per_lane);
    if (stream_form) {
      cl.items_per_lane = 0;
      cl.stream_chunks = int(sh.layout.n_padded / stream_chunk);
      cl.nt = use_nontemporal(ds, sh);
    }
    cl.stage1_sc1 = ctx->settings.lm_cluster == 5;
    cl.mail = mailbox_in_launch ? ctx->d_mail : nullptr;
    cl.partials = slot.partials;
    cl.ctl = slot.d_cluster;
    cl.lm = slot.d_lm;
    cl.history = opt->cost_history ? slot.h_hist_dev : nullptr;
    cl.history_capacity = kHistCapacity;
    cl.entry = slot.h_log_dev;
    cl.seq_host = seq_dev;
    cl.seq = ++slot.seq;
    // arrival counters and the abort word start every launch at zero
    NOS_HIP_CHECK(hipMemsetAsync(slot.d_cluster, 0, sizeof(nos::ClusterCtl), slot.stream));
    if (ctx->settings.debug_cluster_abort != 0) {  // test hook (nos_ctx_set_option): the launch finds `abort` already raised and gives up
      const unsigned int raised = 1u;
      NOS_HIP_CHECK(hipMemcpyAsync(&slot.d_cluster->abort, &raised, sizeof raised, hipMemcpyHostToDevice, slot.stream));
      NOS_HIP_CHECK(hipStreamSynchronize(slot.stream));
    }
    int rows = 0;
    // A launch the device cannot take (the LDS grant refused: NOS_ERR_UNSUPPORTED) is replaced by the loop below; any other
    // failure — a layout / argument error, a sticky HIP error — is the caller's to see, not a reason to run slower.
    const int rc_launch = launch_assemble_raw(ds, sh, rq, slot.partials, nos::FusedFinal{}, slot.stream, &rows, &cl);
    if (rc_launch != NOS_OK && rc_launch != NOS_ERR_UNSUPPORTED) return rc_launch;
    const bool launched_ok = rc_launch == NOS_OK;
    // spin on the sequence word; a launch that gave up never writes it
    volatile unsigned long long* seqw = reinterpret_cast<volatile unsigned long long*>(slot.h_out + kSeqSlot);
    bool finished = false;
    for (long spins = 0; launched_ok && spins < 4000000; ++spins) {
      if (*seqw >= cl.seq) {
        finished = true;
        break;
      }
#if defined(__x86_64__)
      __builtin_ia32_pause();
#endif
    }
    if (!finished) {
      NOS_HIP_CHECK(hipStreamSynchronize(slot.stream));
      finished = *seqw >= cl.seq;
    }
    if (finished) {
      __atomic_thread_fence(__ATOMIC_ACQUIRE);
      slot.cluster_next_pause = 64;
      const double* e = slot.h_log;
      for (int k = 0; k < 9; ++k) st.R[k] = e[nos::kLogR + k];
      for (int k = 0; k < 3; ++k) st.t[k] = e[nos::kLogT + k];
      st.lambda = e[nos::kLogLambda];
      st.previous_cost = e[nos::kLogPrevCost];
      st.cost = e[nos::kLogCost];
      st.iteration = int(e[nos::kLogIteration]);
      st.done = int(e[nos::kLogDone]);
      st.ok = int(e[nos::kLogOk]);
      const int executed = int(e[nos::kLogExecuted]);
#ifdef NOS_LM_TIMING
      fprintf(stderr, "[resident-timing] %d blocks x %d items/lane (0 = streamed), %d iterations; workgroup 0, us per iteration: item math %.2f, "
              "block reduce %.2f, drain+arrive+wait %.2f, rows->sums %.2f, LM step+barrier %.2f\n", cl.cluster_blocks,
              cl.items_per_lane, executed, e[50] * 0.01, e[51] * 0.01, e[52] * 0.01, e[53] * 0.01, e[54] * 0.01);
      fprintf(stderr, "[resident-timing]    inside the step, shader-clock cycles per iteration: elimination %.0f, back substitution %.0f, "
              "lane 0 (pose update, tests, schedule) %.0f\n", e[56], e[57], e[58]);
#endif
      if (slot.prof_on && slot.prof_every == 0) slot.prof_launches += executed;  // bracket profiling counts passes over the data
      if (opt->cost_history != nullptr)
        for (int k = 0; k < executed && k < opt->max_iterations; ++k) opt->cost_history[k] = slot.h_hist[k];
      for (int k = 0; k < nR; ++k) R[k] = st.R[k];
      for (int k = 0; k < nt; ++k) t[k] = st.t[k];
      if (report) {
        report->iterations = st.iteration;
        report->ok = st.ok;
        report->launches = 1;
        report->fallback = 0;
        report->printed_cost = st.previous_cost;
        report->last_cost = st.cost;
        report->final_lambda = st.lambda;
      }
      return NOS_OK;
    }
    // gave up: put the shared words back in order and fall through to the launch-per-iteration loop from the start
    // (reported as nos_lm_report::fallback).  A launch that ran and timed out means the GPU is shared: remember it for a
    // while, so the next solves on this device do not each pay the bounded wait before falling back ("lm_cluster_retry_ms";
    // debug_cluster_abort = 2 is the test hook that leaves this latch active).  A launch the device refused sets no latch,
    // and the text of its refusal is dropped: the call goes on to succeed.
    fell_back = 1;
    if (launched_ok && ctx->settings.debug_cluster_abort != 1) {
      if (mailbox_in_launch) {
        slot.cluster_paused_solves = slot.cluster_next_pause;
        slot.cluster_next_pause = std::min(slot.cluster_next_pause * 2, 65536);
      } else {
        slot.cluster_gave_up = true;
        slot.cluster_gave_up_at = std::chrono::steady_clock::now();
      }
    }
    if (!launched_ok) clear_last_error();
    if (mailbox_in_launch) {
      // every rank gives up together (a rank that cannot go on tells its peers); the exchange rounds the abandoned launch
      // may have used are skipped on every rank, so that no later round finds their granules
      hipLaunchKernelGGL(nos::mailbox_skip_rounds_kernel, dim3(1), dim3(1), 0, slot.stream, ctx->d_round,
                         (unsigned long long)opt->max_iterations + 1ull);
      NOS_HIP_CHECK(hipGetLastError());
    }
    NOS_HIP_CHECK(hipMemsetAsync(slot.d_cluster, 0, sizeof(nos::ClusterCtl), slot.stream));
    hipLaunchKernelGGL(nos::lm_init_kernel, dim3(1), dim3(1), 0, slot.stream, slot.d_lm, init);
    NOS_HIP_CHECK(hipGetLastError());
    if (init.dof == 6)
      nos_host::LmInit6(&st, init.R, init.t, opt->max_iterations, init.settings.float_schedule);
    else
      nos_host::LmInit3(&st, init.R, init.t, opt->max_iterations, init.settings.float_schedule);
  }
  const unsigned long long base_seq2 = slot.seq;
  int launched = 0, completed = 0;
  auto launch_one = [&]() -> int {
    double* entry = slot.h_log_dev + size_t(launched % kLogSlots) * nos::kLogEntryDoubles;
    int rows = 0;
    nos::FusedFinal fin{};
    fin.counter = slot.counter;
    fin.lm = slot.d_lm;
    fin.seq = ++slot.seq;
    fin.mail = ctx->d_mail;
    if (step_in_launch) {
      fin.out_host = entry;
      fin.seq_host = seq_dev;
      fin.lm_step = 1;
      int rc = launch_assemble(ds, sh, rq, slot.partials, fin, slot.stream, &rows);
      if (rc != NOS_OK) return rc;
    } else {
      fin.out_dev = slot.d_out;
      int rc = launch_assemble(ds, sh, rq, slot.partials, fin, slot.stream, &rows);
      if (rc != NOS_OK) return rc;
      if (with_comm)
        NOS_RCCL_CHECK(Rccl()->AllReduce(slot.d_out, slot.d_out, size_t(rq.n_out), ncclDouble, ncclSum, ctx->comm,
                                         slot.stream));
      if (rq.n_out == 28)
        hipLaunchKernelGGL((nos::lm_step_kernel<28>), dim3(1), dim3(64), 0, slot.stream, slot.d_out, slot.d_lm, entry,
                           seq_dev, fin.seq);
      else
        hipLaunchKernelGGL((nos::lm_step_kernel<10>), dim3(1), dim3(64), 0, slot.stream, slot.d_out, slot.d_lm, entry,
                           seq_dev, fin.seq);
      NOS_HIP_CHECK(hipGetLastError());
    }
    ++launched;
    return NOS_OK;
  };
  int rc = NOS_OK;
  while (rc == NOS_OK && launched < std::min(window, opt->max_iterations)) rc = launch_one();
  while (rc == NOS_OK && completed < launched) {
    rc = wait_for_sequence(slot, base_seq2 + completed + 1);
    if (rc == NOS_OK) rc = check_mailbox_error(ctx, slot);
    if (rc != NOS_OK) break;
    if (!st.done) {
      const double* e = slot.h_log + size_t(completed % kLogSlots) * nos::kLogEntryDoubles;
      if (opt->cost_history != nullptr && completed < opt->max_iterations) opt->cost_history[completed] = e[rq.n_out - 1];
      for (int k = 0; k < 9; ++k) st.R[k] = e[nos::kLogR + k];
      for (int k = 0; k < 3; ++k) st.t[k] = e[nos::kLogT + k];
      st.lambda = e[nos::kLogLambda];
      st.previous_cost = e[nos::kLogPrevCost];
      st.cost = e[nos::kLogCost];
      st.iteration = int(e[nos::kLogIteration]);
      st.done = int(e[nos::kLogDone]);
      st.ok = int(e[nos::kLogOk]);
#ifdef NOS_LM_TIMING
      // wall_clock64 ticks (10 ns): start of the finishing workgroup, its ticket, step begin, step done, log written
      fprintf(stderr, "[lm-timing] it %d: loop+ticket %.2f us, rows->sums %.2f us, step %.2f us, log %.2f us\n", completed,
              (e[51] - e[50]) * 0.01, (e[52] - e[51]) * 0.01, (e[53] - e[52]) * 0.01, (e[54] - e[53]) * 0.01);
      if (ctx->shm_host != nullptr) {
        const double* mbx = static_cast<const double*>(ctx->shm_host) + size_t(ctx->comm_rank) * 2 * nos::kMailSlotDoubles;
        fprintf(stderr, "[lm-timing]        mailbox: store+flag %.2f us, poll %.2f us, gather %.2f us\n", mbx[40] * 0.01,
                mbx[41] * 0.01, mbx[42] * 0.01);
      }
      fprintf(stderr, "[lm-timing]        a block: prologue %.2f us, loads+math %.2f us, block reduce+store %.2f us\n",
              e[56] * 0.01, e[57] * 0.01, e[58] * 0.01);
#endif
    }
    ++completed;
    if (!st.done && launched < opt->max_iterations) rc = launch_one();
  }
  if (rc != NOS_OK) {
    (void)hipStreamSynchronize(slot.stream);  // leave nothing in flight behind an error
    return rc;
  }
  for (int k = 0; k < nR; ++k) R[k] = st.R[k];
  for (int k = 0; k < nt; ++k) t[k] = st.t[k];
  if (report) {
    report->iterations = st.iteration;
    report->ok = st.ok;
    report->launches = launched;
    report->fallback = fell_back;
    report->printed_cost = st.previous_cost;
    report->last_cost = st.cost;
    report->final_lambda = st.lambda;
  }
  return NOS_OK;
}

// ------------------------------------------------------------------ dataset construction

// Pool limits: at most 8 parked buffers and 16 GiB per device; a parked buffer serves a request if it is large
// enough and not more than twice (+1 MiB) the size asked for.  The caller has selected the slot's device.
constexpr size_t kPoolMaxEntries = 8;
constexpr size_t kPoolMaxBytes = size_t(16) << 30;

int pool_alloc(DeviceSlot& slot, size_t bytes, void** ptr, size_t* capacity) {
  if (bytes == 0) bytes = 8;
  int best = -1;
  for (int i = 0; i < int(slot.pool.size()); ++i) {
    const size_t have = slot.pool[i].bytes;
    if (have >= bytes && have <= 2 * bytes + (size_t(1) << 20) && (best < 0 || have < slot.pool[best].bytes)) best = i;
  }
  if (best >= 0) {
    *ptr = slot.pool[best].ptr;
    *capacity = slot.pool[best].bytes;
    slot.pool_bytes -= slot.pool[best].bytes;
    slot.pool.erase(slot.pool.begin() + best);
    return NOS_OK;
  }
  hipError_t e = hipMalloc(ptr, bytes);
  if (e == hipErrorOutOfMemory && !slot.pool.empty()) {  // give the parked buffers back and try once more
    for (auto& pe : slot.pool) (void)hipFree(pe.ptr);
    slot.pool.clear();
    slot.pool_bytes = 0;
    e = hipMalloc(ptr, bytes);
  }
  if (e != hipSuccess)
    return fail(e == hipErrorOutOfMemory ? NOS_ERR_OUT_OF_MEMORY : NOS_ERR_HIP, "hipMalloc(%zu bytes) failed: %s", bytes,
                hipGetErrorString(e));
  *capacity = bytes;
  return NOS_OK;
}

void pool_release(DeviceSlot& slot, void* ptr, size_t capacity) {
  if (!ptr) return;
  if (!slot.pool_enabled || capacity > kPoolMaxBytes) {
    (void)hipFree(ptr);
    return;
  }
  slot.pool.push_back({ptr, capacity});
  slot.pool_bytes += capacity;
  while (slot.pool.size() > kPoolMaxEntries || slot.pool_bytes > kPoolMaxBytes) {  // oldest first
    (void)hipFree(slot.pool.front().ptr);
    slot.pool_bytes -= slot.pool.front().bytes;
    slot.pool.erase(slot.pool.begin());
  }
}

int alloc_shards(nos_ctx* ctx, nos_dataset* ds) {
  const int n_shards = int(ctx->slots.size());
  const size_t n = ds->n;
  const size_t per = (n + n_shards - 1) / size_t(n_shards);  // contiguous equal ranges
  int tile_log2 = ctx->tile_log2 >= 0 ? ctx->tile_log2 : ctx->settings.tile_log2;
  if (tile_log2 < 0) tile_log2 = ds->dtype == NOS_F32 ? kDefaultTileLog2F32 : 0;
  if (tile_log2 != 0 && (tile_log2 < 10 || tile_log2 > 24)) return fail(NOS_ERR_INVALID_ARGUMENT, "tile_log2 out of range");
  ds->tile = tile_log2 > 0 ? (size_t(1) << tile_log2) : 0;
  ds->shards.resize(n_shards);
  size_t begin = 0;
  for (int s = 0; s < n_shards; ++s) {
    const size_t cnt = begin < n ? std::min(per, n - begin) : 0;
    Shard& sh = ds->shards[s];
    sh.slot = s;
    sh.layout = make_layout(cnt, ds->n_fields, tile_log2, ctx->settings.plane_skew);
    sh.bytes = layout_elems(sh.layout, ds->n_fields) * elem_size(ds->dtype);
    NOS_HIP_CHECK(hipSetDevice(ctx->slots[s].device));
    int prc = pool_alloc(ctx->slots[s], sh.bytes, &sh.data, &sh.capacity);
    if (prc != NOS_OK) return prc;
    sh.pooled = true;
    sh.layout.base = sh.data;
    begin += cnt;
  }
  return NOS_OK;
}

template <typename SRC>
int retile_dispatch(const nos::PlanePtrs& src, int n_fields, const nos::TiledLayout& L, void* dst, int dtype,
                    hipStream_t stream) {
  dim3 grid(unsigned((L.n_padded + 255) / 256), unsigned(n_fields));
  if (dtype == NOS_F64)
    hipLaunchKernelGGL((nos::retile_kernel<SRC, double>), grid, dim3(256), 0, stream, src, n_fields, L,
                       static_cast<double*>(dst));
  else
    hipLaunchKernelGGL((nos::retile_kernel<SRC, float>), grid, dim3(256), 0, stream, src, n_fields, L,
                       static_cast<float*>(dst));
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(NOS_ERR_HIP, "retile launch failed: %s", hipGetErrorString(e));
  return NOS_OK;
}

int dataset_new(nos_ctx* ctx, int kind, size_t n, int dtype, nos_dataset** out, nos_dataset** made) {
  if (!ctx || !out) return fail(NOS_ERR_INVALID_ARGUMENT, "ctx / out pointer is NULL");
  if (dtype != NOS_F64 && dtype != NOS_F32) return fail(NOS_ERR_INVALID_ARGUMENT, "unknown dtype %d", dtype);
  *out = nullptr;
  nos_dataset* ds = new (std::nothrow) nos_dataset();
  if (!ds) return fail(NOS_ERR_OUT_OF_MEMORY, "host allocation failed");
  ds->ctx = ctx;
  ds->kind = kind;
  ds->dtype = dtype;
  ds->n_fields = (kind == kKindNdt) ? NOS_NDT_PLANES : NOS_REPROJ_PLANES;
  ds->n = n;
  int rc = alloc_shards(ctx, ds);
  if (rc != NOS_OK) {
    nos_dataset_destroy(ds);
    return rc;
  }
  *made = ds;
  return NOS_OK;
}

int create_from_host_planes(nos_ctx* ctx, int kind, size_t n, const double* const* planes, int dtype,
                            nos_dataset** out) {
  if (!planes) return fail(NOS_ERR_INVALID_ARGUMENT, "planes is NULL");
  nos_dataset* ds = nullptr;
  int rc = dataset_new(ctx, kind, n, dtype, out, &ds);
  if (rc != NOS_OK) return rc;
  for (int f = 0; f < ds->n_fields; ++f)
    if (!planes[f] && n > 0) {
      nos_dataset_destroy(ds);
      return fail(NOS_ERR_INVALID_ARGUMENT, "plane %d is NULL", f);
    }
  size_t begin = 0;
  for (Shard& sh : ds->shards) {
    DeviceSlot& slot = ctx->slots[sh.slot];
    const size_t cnt = sh.layout.n;
    hipError_t e = hipSetDevice(slot.device);
    void* staging = nullptr;
    const size_t plane_bytes = cnt * sizeof(double);
    if (e == hipSuccess && cnt > 0) e = hipMalloc(&staging, plane_bytes * ds->n_fields);
    nos::PlanePtrs src{};
    for (int f = 0; f < ds->n_fields && e == hipSuccess && cnt > 0; ++f) {
      char* d = static_cast<char*>(staging) + plane_bytes * f;
      e = hipMemcpyAsync(d, planes[f] + begin, plane_bytes, hipMemcpyHostToDevice, slot.stream);
      src.p[f] = d;
    }
    if (e == hipSuccess) {
      rc = retile_dispatch<double>(src, ds->n_fields, sh.layout, sh.data, dtype, slot.stream);
      if (rc == NOS_OK) e = hipStreamSynchronize(slot.stream);
    }
    if (staging) (void)hipFree(staging);
    if (e != hipSuccess || rc != NOS_OK) {
      nos_dataset_destroy(ds);
      if (rc != NOS_OK) return rc;
      return fail(e == hipErrorOutOfMemory ? NOS_ERR_OUT_OF_MEMORY : NOS_ERR_HIP, "dataset upload failed: %s",
                  hipGetErrorString(e));
    }
    begin += cnt;
  }
  *out = ds;
  return NOS_OK;
}

int create_from_device_planes(nos_ctx* ctx, int kind, size_t n, const void* const* d_planes, int src_dtype,
                              int dtype, nos_dataset** out) {
  if (!d_planes) return fail(NOS_ERR_INVALID_ARGUMENT, "d_planes is NULL");
  if (!ctx || ctx->slots.size() != 1) return fail(NOS_ERR_UNSUPPORTED, "from_device needs a single-device context");
  if (src_dtype != NOS_F64 && src_dtype != NOS_F32) return fail(NOS_ERR_INVALID_ARGUMENT, "unknown src dtype");
  nos_dataset* ds = nullptr;
  int rc = dataset_new(ctx, kind, n, dtype, out, &ds);
  if (rc != NOS_OK) return rc;
  Shard& sh = ds->shards[0];
  DeviceSlot& slot = ctx->slots[0];
  nos::PlanePtrs src{};
  for (int f = 0; f < ds->n_fields; ++f) {
    if (!d_planes[f] && n > 0) {
      nos_dataset_destroy(ds);
      return fail(NOS_ERR_INVALID_ARGUMENT, "device plane %d is NULL", f);
    }
    src.p[f] = d_planes[f];
  }
  rc = (src_dtype == NOS_F64) ? retile_dispatch<double>(src, ds->n_fields, sh.layout, sh.data, dtype, slot.stream)
                              : retile_dispatch<float>(src, ds->n_fields, sh.layout, sh.data, dtype, slot.stream);
  hipError_t e = (rc == NOS_OK) ? hipStreamSynchronize(slot.stream) : hipSuccess;
  if (rc != NOS_OK || e != hipSuccess) {
    nos_dataset_destroy(ds);
    if (rc != NOS_OK) return rc;
    return fail(NOS_ERR_HIP, "retile failed: %s", hipGetErrorString(e));
  }
  *out = ds;
  return NOS_OK;
}

template <typename DST>
int unpack_launch(const unsigned char* d_rec, size_t stride, const nos::FieldOffsets& fo, int n_fields, size_t first,
                  size_t count, const nos::TiledLayout& L, void* dst, hipStream_t stream) {
  hipLaunchKernelGGL((nos::unpack_records_kernel<DST>), dim3(unsigned((count + 255) / 256)), dim3(256), 0, stream,
                     d_rec, uint64_t(stride), fo, n_fields, uint64_t(first), uint64_t(count), L,
                     static_cast<DST*>(dst));
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(NOS_ERR_HIP, "unpack launch failed: %s", hipGetErrorString(e));
  return NOS_OK;
}

template <typename DST>
int zero_pad_launch(int n_fields, const nos::TiledLayout& L, void* dst, hipStream_t stream) {
  const size_t pads = L.n_padded - L.n;
  if (pads == 0) return NOS_OK;
  hipLaunchKernelGGL((nos::zero_pad_kernel<DST>), dim3(unsigned((pads + 255) / 256)), dim3(256), 0, stream, n_fields, L,
                     static_cast<DST*>(dst));
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(NOS_ERR_HIP, "zero-pad launch failed: %s", hipGetErrorString(e));
  return NOS_OK;
}

// AoS ingestion: records are streamed in chunks through two device staging buffers so
// the H2D copy of chunk k+1 overlaps the unpack kernel of chunk k.
// Host-pack ingestion (first form: AoS → pinned SoA → H2D, double buffered): T host threads gather
// the n_fields used doubles out of every record into a pinned planar chunk (converted to the dataset's element type),
// the chunk's planes are copied straight into their final place in the planar layout while the threads pack the next
// chunk.  Moves 120 (60) instead of 304 bytes per NDT record over PCIe; pays when there are enough host threads, so it
// is chosen for large inputs only (see create_from_records).  Planar planes and tiled layouts alike.
// `pinned` is the staging image of one chunk: planar (tile_log2 = 0: field f of record j at f * chunk + j) or in the
// dataset's tiled order (record j of the chunk at (j >> T) * n_fields * 2^T + f * 2^T + (j mod 2^T); chunks start on
// tile boundaries), so that the image is one contiguous piece of the dataset.  [lo, lo + count) = this thread's records.
// Records are taken a cache line of OUTPUT at a time (8 doubles / 16 floats per field): the line's worth of every field is
// gathered into a small block first and leaves with full-line non-temporal stores — the staging image is written once and
// read only by the copy engine, so the destination lines need not be fetched for ownership first (4.2 instead of 5.4 GB of
// host memory traffic per 10 M NDT records) and 15 interleaved 8-byte store streams do not fight over the core's
// write-combining buffers.
template <typename T>
void pack_range(const unsigned char* host, size_t stride, const nos::FieldOffsets& fo, int n_fields, size_t first, size_t lo,
                size_t count, size_t chunk, int tile_log2, T* pinned) {
  const size_t tile = size_t(1) << tile_log2, mask = tile - 1;
  const size_t pitch = tile_log2 == 0 ? chunk : tile;
  auto dst_of = [&](size_t j) -> T* {
    return tile_log2 == 0 ? pinned + j : pinned + (j >> tile_log2) * (tile * size_t(n_fields)) + (j & mask);
  };
  auto one = [&](size_t j) {
    const unsigned char* rec = host + (first + j) * stride;
    T* dst = dst_of(j);
    for (int f = 0; f < n_fields; ++f) {
      double v;
      memcpy(&v, rec + fo.off[f], sizeof v);
      dst[size_t(f) * pitch] = T(v);
    }
  };
  [[maybe_unused]] constexpr size_t kLine = 64 / sizeof(T);  // records per output cache line
  size_t j = lo;
  const size_t end = lo + count;
#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
  if (n_fields <= 16 && (reinterpret_cast<uintptr_t>(pinned) & 63u) == 0 && pitch % kLine == 0) {
    for (; j < end && (j % kLine) != 0; ++j) one(j);  // up to the next line boundary of the image
    alignas(64) T block[16][kLine];
    for (; j + kLine <= end; j += kLine) {
      for (size_t r = 0; r < kLine; ++r) {
        const unsigned char* rec = host + (first + j + r) * stride;
        for (int f = 0; f < n_fields; ++f) {
          double v;
          memcpy(&v, rec + fo.off[f], sizeof v);
          block[f][r] = T(v);
        }
      }
      T* dst = dst_of(j);  // a line never straddles a tile: tiles are multiples of 1 024 records
      for (int f = 0; f < n_fields; ++f) {
        const __m128d* src = reinterpret_cast<const __m128d*>(block[f]);
        double* line = reinterpret_cast<double*>(dst + size_t(f) * pitch);
        _mm_stream_pd(line + 0, src[0]);
        _mm_stream_pd(line + 2, src[1]);
        _mm_stream_pd(line + 4, src[2]);
        _mm_stream_pd(line + 6, src[3]);
      }
    }
    _mm_sfence();  // the non-temporal stores are globally visible before this thread reports the chunk packed
  }
#endif
  for (; j < end; ++j) one(j);
}

int ingest_host_pack(nos_ctx* ctx, nos_dataset* ds, Shard& sh, const unsigned char* host, size_t stride,
                     const nos::FieldOffsets& fo, int threads) {
  DeviceSlot& slot = ctx->slots[sh.slot];
  const size_t cnt = sh.layout.n;
  const size_t es = elem_size(ds->dtype);
  const size_t chunk = size_t(256) << 10;  // records per chunk: 31 MB of fp64 planes
  const size_t need = chunk * size_t(ds->n_fields) * es;
  hipError_t e = hipSetDevice(slot.device);
  if (e == hipSuccess && slot.copy_stream == nullptr) e = hipStreamCreateWithFlags(&slot.copy_stream, hipStreamNonBlocking);
  for (int b = 0; b < 2 && e == hipSuccess; ++b)
    if (slot.pack_done[b] == nullptr) e = hipEventCreateWithFlags(&slot.pack_done[b], hipEventDisableTiming);
  if (e == hipSuccess && slot.pack_bytes < need) {
    for (int b = 0; b < 2; ++b) {
      if (slot.pack_pinned[b]) (void)hipHostFree(slot.pack_pinned[b]);
      slot.pack_pinned[b] = nullptr;
    }
    slot.pack_bytes = 0;
    for (int b = 0; b < 2 && e == hipSuccess; ++b) e = hipHostMalloc(&slot.pack_pinned[b], need, hipHostMallocDefault);
    if (e == hipSuccess) slot.pack_bytes = need;
  }
  // Worker threads live for the whole call; per chunk they are released by `go` (chunk number) and report through
  // `arrived`.  The calling thread waits for the pinned buffer to be free, releases the workers, waits for them, enqueues
  // the chunk's plane copies and moves on while those copies run.
  const size_t n_chunks = (cnt + chunk - 1) / chunk;
  std::atomic<long> go{-1};
  std::atomic<int> arrived{0};
  void* const pinned2[2] = {slot.pack_pinned[0], slot.pack_pinned[1]};
  const int n_fields = ds->n_fields;
  const bool f64 = ds->dtype == NOS_F64;
  const int tile_log2 = sh.layout.tile_stride == 0 ? 0 : int(sh.layout.tile_shift);  // 0 = planar planes
  std::vector<std::thread> pool;
  const int n_workers = (e == hipSuccess && n_chunks > 0) ? threads : 0;
  for (int w = 0; w < n_workers; ++w) {
    pool.emplace_back([&, w]() {
      for (size_t c = 0; c < n_chunks; ++c) {
        while (go.load(std::memory_order_acquire) < long(c)) std::this_thread::yield();
        if (go.load(std::memory_order_acquire) == LONG_MAX) return;  // the caller gave up
        const size_t first = c * chunk, count = std::min(chunk, cnt - first);
        const size_t per = (count + size_t(n_workers) - 1) / size_t(n_workers);
        const size_t lo = std::min(count, size_t(w) * per), hi = std::min(count, lo + per);
        if (lo < hi) {
          if (f64)
            pack_range<double>(host, stride, fo, n_fields, first, lo, hi - lo, chunk, tile_log2, static_cast<double*>(pinned2[c & 1]));
          else
            pack_range<float>(host, stride, fo, n_fields, first, lo, hi - lo, chunk, tile_log2, static_cast<float*>(pinned2[c & 1]));
        }
        arrived.fetch_add(1, std::memory_order_release);
      }
    });
  }
  bool used[2] = {false, false};
  for (size_t c = 0; c < n_chunks && e == hipSuccess; ++c) {
    const int buf = int(c & 1);
    const size_t first = c * chunk, count = std::min(chunk, cnt - first);
    if (used[buf]) e = hipEventSynchronize(slot.pack_done[buf]);  // its previous copies have left the pinned buffer
    if (e != hipSuccess) break;
    go.store(long(c), std::memory_order_release);
    while (arrived.load(std::memory_order_acquire) < int(c + 1) * n_workers) std::this_thread::yield();
    if (tile_log2 == 0) {
      for (int f = 0; f < n_fields && e == hipSuccess; ++f) {
        char* dst = static_cast<char*>(sh.data) + (size_t(f) * sh.layout.field_stride + first) * es;
        const char* src = static_cast<const char*>(slot.pack_pinned[buf]) + size_t(f) * chunk * es;
        e = hipMemcpyAsync(dst, src, count * es, hipMemcpyHostToDevice, slot.copy_stream);
      }
    } else {  // the chunk's tiles are one contiguous piece of the dataset (the pads of the last tile are zeroed below)
      const size_t tile = size_t(1) << tile_log2;
      const size_t tiles = (count + tile - 1) / tile;
      char* dst = static_cast<char*>(sh.data) + (first >> tile_log2) * sh.layout.tile_stride * es;
      e = hipMemcpyAsync(dst, slot.pack_pinned[buf], tiles * sh.layout.tile_stride * es, hipMemcpyHostToDevice, slot.copy_stream);
    }
    if (e == hipSuccess) e = hipEventRecord(slot.pack_done[buf], slot.copy_stream);
    used[buf] = true;
  }
  go.store(LONG_MAX, std::memory_order_release);  // releases workers still waiting (error path); no-op otherwise
  for (std::thread& th : pool) th.join();
  if (e == hipSuccess) e = hipStreamSynchronize(slot.copy_stream);
  if (e != hipSuccess)
    return fail(e == hipErrorOutOfMemory ? NOS_ERR_OUT_OF_MEMORY : NOS_ERR_HIP, "host-pack ingestion failed: %s", hipGetErrorString(e));
  int rc = (ds->dtype == NOS_F64) ? zero_pad_launch<double>(ds->n_fields, sh.layout, sh.data, slot.stream)
                                  : zero_pad_launch<float>(ds->n_fields, sh.layout, sh.data, slot.stream);
  if (rc != NOS_OK) return rc;
  NOS_HIP_CHECK(hipStreamSynchronize(slot.stream));
  return NOS_OK;
}

int create_from_records(nos_ctx* ctx, int kind, size_t n, const void* records, size_t stride,
                        const size_t* field_offsets, int dtype, nos_dataset** out) {
  if ((!records && n > 0) || !field_offsets) return fail(NOS_ERR_INVALID_ARGUMENT, "records / offsets is NULL");
  nos_dataset* ds = nullptr;
  int rc = dataset_new(ctx, kind, n, dtype, out, &ds);
  if (rc != NOS_OK) return rc;
  nos::FieldOffsets fo{};
  for (int f = 0; f < ds->n_fields; ++f) {
    if (field_offsets[f] + sizeof(double) > stride || (field_offsets[f] % sizeof(double)) != 0) {
      nos_dataset_destroy(ds);
      return fail(NOS_ERR_INVALID_ARGUMENT, "field offset %d out of record / misaligned", f);
    }
    fo.off[f] = uint32_t(field_offsets[f]);
  }
  if (stride % sizeof(double) != 0) {
    nos_dataset_destroy(ds);
    return fail(NOS_ERR_INVALID_ARGUMENT, "record stride must be a multiple of 8");
  }
  const size_t chunk_records = std::max<size_t>(1, (size_t(64) << 20) / stride);
  const unsigned char* host = static_cast<const unsigned char*>(records);
  // Which ingestion: "unpack" ships the raw records and unpacks on the device (no host work, 304 B/record over PCIe);
  // "pack" gathers on the host with a few threads and ships planes (120 / 60 B/record).  auto = pack for large planar
  // inputs when the host has threads to spare (NOS_INGEST=pack|unpack forces, NOS_INGEST_THREADS sets the count).
  const std::string mode = ctx->settings.ingest == 1 ? "pack" : (ctx->settings.ingest == 2 ? "unpack" : "auto");
  const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
  int pack_threads = ctx->settings.ingest_threads > 0 ? ctx->settings.ingest_threads : int(std::min(16u, hw / 2));
  // planar planes, or tiles that divide the 256 Ki-record chunk of the pack path (the fp32 default: 1 024-item tiles)
  const bool packable = ds->tile == 0 || (ds->tile <= (size_t(256) << 10) && ((size_t(256) << 10) % ds->tile) == 0);
  const bool use_pack = packable && pack_threads >= 1 &&
                        (mode == "pack" || (mode == "auto" && n >= size_t(800000) && pack_threads >= 8));
  size_t begin = 0;
  for (Shard& sh : ds->shards) {
    if (use_pack) {
      rc = ingest_host_pack(ctx, ds, sh, host + begin * stride, stride, fo, pack_threads);
      if (rc != NOS_OK) {
        nos_dataset_destroy(ds);
        return rc;
      }
      begin += sh.layout.n;
      continue;
    }
    DeviceSlot& slot = ctx->slots[sh.slot];
    const size_t cnt = sh.layout.n;
    hipError_t e = hipSetDevice(slot.device);
    // persistent per-device ingestion resources (stream, events, staging buffers sized to what this call needs)
    const size_t want_stage = std::min(chunk_records, std::max<size_t>(cnt, 1)) * stride;
    if (e == hipSuccess && slot.copy_stream == nullptr) e = hipStreamCreateWithFlags(&slot.copy_stream, hipStreamNonBlocking);
    for (int b = 0; b < 2 && e == hipSuccess; ++b)
      if (slot.ing_done[b] == nullptr) e = hipEventCreateWithFlags(&slot.ing_done[b], hipEventDisableTiming);
    if (e == hipSuccess && slot.ing_copied == nullptr) e = hipEventCreateWithFlags(&slot.ing_copied, hipEventDisableTiming);
    if (e == hipSuccess && slot.stage_bytes < want_stage) {
      for (int b = 0; b < 2; ++b) {
        if (slot.stage[b]) (void)hipFree(slot.stage[b]);
        slot.stage[b] = nullptr;
      }
      slot.stage_bytes = 0;
      const size_t grow = std::max(want_stage, size_t(4) << 20);
      for (int b = 0; b < 2 && e == hipSuccess; ++b) e = hipMalloc(&slot.stage[b], grow);
      if (e == hipSuccess) slot.stage_bytes = grow;
    }
    void** stage = slot.stage;
    hipEvent_t* done = slot.ing_done;
    hipStream_t copy_stream = slot.copy_stream;
    hipEvent_t copied = slot.ing_copied;
    int buf = 0;
    bool used[2] = {false, false};
    for (size_t first = 0; first < cnt && e == hipSuccess && rc == NOS_OK; first += chunk_records, buf ^= 1) {
      const size_t count = std::min(chunk_records, cnt - first);
      if (used[buf]) e = hipStreamWaitEvent(copy_stream, done[buf], 0);  // unpack of the previous use finished
      if (e == hipSuccess)
        e = hipMemcpyAsync(stage[buf], host + (begin + first) * stride, count * stride, hipMemcpyHostToDevice,
                           copy_stream);
      if (e == hipSuccess) e = hipEventRecord(copied, copy_stream);
      if (e == hipSuccess) e = hipStreamWaitEvent(slot.stream, copied, 0);
      if (e == hipSuccess) {
        rc = (dtype == NOS_F64)
                 ? unpack_launch<double>(static_cast<unsigned char*>(stage[buf]), stride, fo, ds->n_fields, first, count,
                                         sh.layout, sh.data, slot.stream)
                 : unpack_launch<float>(static_cast<unsigned char*>(stage[buf]), stride, fo, ds->n_fields, first, count,
                                        sh.layout, sh.data, slot.stream);
        if (rc == NOS_OK) e = hipEventRecord(done[buf], slot.stream);
        used[buf] = true;
      }
    }
    if (e == hipSuccess && rc == NOS_OK)
      rc = (dtype == NOS_F64) ? zero_pad_launch<double>(ds->n_fields, sh.layout, sh.data, slot.stream)
                              : zero_pad_launch<float>(ds->n_fields, sh.layout, sh.data, slot.stream);
    if (e == hipSuccess) e = hipStreamSynchronize(slot.stream);
    if (copy_stream) (void)hipStreamSynchronize(copy_stream);
    if (e != hipSuccess || rc != NOS_OK) {
      nos_dataset_destroy(ds);
      if (rc != NOS_OK) return rc;
      return fail(e == hipErrorOutOfMemory ? NOS_ERR_OUT_OF_MEMORY : NOS_ERR_HIP, "record ingestion failed: %s",
                  hipGetErrorString(e));
    }
    begin += cnt;
  }
  *out = ds;
  return NOS_OK;
}


int zero_pad(int dtype, int n_fields, const nos::TiledLayout& L, void* dst, hipStream_t stream) {
  return dtype == NOS_F64 ? zero_pad_launch<double>(n_fields, L, dst, stream) : zero_pad_launch<float>(n_fields, L, dst, stream);
}

int unpack_records(int dtype, const unsigned char* d_rec, size_t stride, const nos::FieldOffsets& fo, int n_fields,
                   size_t first, size_t count, const nos::TiledLayout& L, void* dst, hipStream_t stream) {
  return dtype == NOS_F64 ? unpack_launch<double>(d_rec, stride, fo, n_fields, first, count, L, dst, stream)
                          : unpack_launch<float>(d_rec, stride, fo, n_fields, first, count, L, dst, stream);
}

}  // namespace nosd

using namespace nosd;

// ====================================================================== C ABI

extern "C" {

namespace {
// defined next to the option table: a knob read from the environment outside its range goes back to its default
void drop_out_of_range_settings(nosd::Settings& st);
}  // namespace

int nos_ctx_create(const int* device_ids, int n_devices, nos_ctx** out_ctx) {
  if (!out_ctx) return fail(NOS_ERR_INVALID_ARGUMENT, "out_ctx is NULL");
  *out_ctx = nullptr;
  if (n_devices < 1 || n_devices > 64 || !device_ids) return fail(NOS_ERR_INVALID_ARGUMENT, "bad device list");
  int count = 0;
  hipError_t e = hipGetDeviceCount(&count);
  if (e != hipSuccess || count < 1)
    return fail(NOS_ERR_NO_DEVICE, "no usable HIP device (%s); this library has no CPU fallback",
                e != hipSuccess ? hipGetErrorString(e) : "device count is 0");
  for (int i = 0; i < n_devices; ++i)
    if (device_ids[i] < 0 || device_ids[i] >= count)
      return fail(NOS_ERR_INVALID_ARGUMENT, "device id %d out of range [0,%d)", device_ids[i], count);
  nos_ctx* ctx = new (std::nothrow) nos_ctx();
  if (!ctx) return fail(NOS_ERR_OUT_OF_MEMORY, "host allocation failed");
  ctx->slots.resize(n_devices);
  ctx->blocks_per_cu = env_int("NOS_BLOCKS_PER_CU", 0);
  ctx->variant = env_int("NOS_VARIANT", 0);
  {  // the only place the experiment knobs are read from the environment
    Settings& st = ctx->settings;
    st.plane_skew = env_int("NOS_PLANE_SKEW", st.plane_skew);
    st.sc1 = env_int("NOS_SC1", st.sc1);
    st.nt = env_int("NOS_NT", st.nt);
    st.fused = env_int("NOS_FUSED", st.fused);
    st.lm_fused = env_int("NOS_LM_FUSED", st.lm_fused);
    st.lm_window = env_int("NOS_LM_WINDOW", st.lm_window);
    st.lm_single = env_int("NOS_LM_SINGLE", st.lm_single);
    st.lm_cluster = env_int("NOS_LM_CLUSTER", st.lm_cluster);
    st.lm_cluster_max_blocks = env_int("NOS_LM_CLUSTER_MAX_BLOCKS", st.lm_cluster_max_blocks);
    st.pool = env_int("NOS_POOL", st.pool);
    st.tile_log2 = env_int("NOS_TILE_LOG2", int(kDefaultTileLog2));
    const char* ingest = getenv("NOS_INGEST");
    st.ingest = (ingest && !strcmp(ingest, "pack")) ? 1 : ((ingest && !strcmp(ingest, "unpack")) ? 2 : 0);
    st.ingest_threads = env_int("NOS_INGEST_THREADS", 0);
    st.indexed_bpc = env_int("NOS_INDEXED_BPC", st.indexed_bpc);
    st.match_dense = env_int("NOS_MATCH_DENSE", st.match_dense);
    st.map_compact_keys = env_int("NOS_MAP_COMPACT_KEYS", st.map_compact_keys);
    st.pgo_host_scalars = env_int("NOS_PGO_HOST_SCALARS", st.pgo_host_scalars);
    st.pgo_precond = env_int("NOS_PGO_PRECOND", st.pgo_precond);
    st.pgo_agg = env_int("NOS_PGO_AGG", st.pgo_agg);
    st.pgo_block = env_int("NOS_PGO_BLOCK", st.pgo_block);
    st.pgo_coarse_probe = env_int("NOS_PGO_COARSE_PROBE", st.pgo_coarse_probe);
    drop_out_of_range_settings(st);  // the same ranges nos_ctx_set_option enforces
  }
  for (int i = 0; i < n_devices; ++i) {
    DeviceSlot& s = ctx->slots[i];
    s.device = device_ids[i];
    s.pool_enabled = ctx->settings.pool != 0;
    hipDeviceProp_t prop;
    e = hipSetDevice(s.device);
    if (e == hipSuccess) e = hipGetDeviceProperties(&prop, s.device);
    if (e == hipSuccess) {
      s.num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
      e = hipStreamCreateWithFlags(&s.own_stream, hipStreamNonBlocking);
    }
    if (e == hipSuccess) e = hipMalloc(&s.partials, sizeof(double) * kMaxPartialRows * kMaxOut);
    if (e == hipSuccess) e = hipMalloc(&s.d_out, sizeof(double) * kMaxOut);
    if (e == hipSuccess) e = hipHostMalloc(&s.h_out, sizeof(double) * 64, hipHostMallocMapped);
    if (e == hipSuccess) memset(s.h_out, 0, sizeof(double) * 64);
    if (e == hipSuccess) e = hipHostGetDevicePointer(reinterpret_cast<void**>(&s.h_out_dev), s.h_out, 0);
    if (e == hipSuccess) e = hipMalloc(&s.d_lm, sizeof(nos::LmDevice));
    const size_t log_bytes = sizeof(double) * kLogSlots * nos::kLogEntryDoubles;
    if (e == hipSuccess) e = hipHostMalloc(&s.h_log, log_bytes, hipHostMallocMapped);
    if (e == hipSuccess) memset(s.h_log, 0, log_bytes);
    if (e == hipSuccess) e = hipHostGetDevicePointer(reinterpret_cast<void**>(&s.h_log_dev), s.h_log, 0);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&s.d_cluster), sizeof(nos::ClusterCtl));
    if (e == hipSuccess) e = hipMemset(s.d_cluster, 0, sizeof(nos::ClusterCtl));
    if (e == hipSuccess) e = hipHostMalloc(&s.h_hist, sizeof(double) * kHistCapacity, hipHostMallocMapped);
    if (e == hipSuccess) e = hipHostGetDevicePointer(reinterpret_cast<void**>(&s.h_hist_dev), s.h_hist, 0);
    if (e == hipSuccess) e = hipMalloc(&s.counter, 2048);  // top ticket + 8 group tickets, 128 bytes apart
    if (e == hipSuccess) e = hipMemset(s.counter, 0, 2048);
    if (e == hipSuccess) e = hipEventCreate(&s.ev0);
    if (e == hipSuccess) e = hipEventCreate(&s.ev1);
    if (e == hipSuccess) e = hipEventCreate(&s.ev2);
    s.stream = s.own_stream;
    if (e != hipSuccess) {
      nos_ctx_destroy(ctx);
      return fail(e == hipErrorOutOfMemory ? NOS_ERR_OUT_OF_MEMORY : NOS_ERR_HIP, "context setup failed on device %d: %s",
                  device_ids[i], hipGetErrorString(e));
    }
  }
  *out_ctx = ctx;
  return NOS_OK;
}

int nos_ctx_comm_destroy(nos_ctx* ctx) {
  nosd::CtxGuard guard_(ctx);  // one solve / accumulate / create at a time per context
  if (ctx)
    for (DeviceSlot& sl : ctx->slots) sl.cluster_paused_solves = 0, sl.cluster_next_pause = 64;  // a new communicator starts with every rank unpaused
  if (!ctx) return NOS_OK;
  if (ctx->comm != nullptr) {
    if (!ctx->slots.empty()) {
      (void)hipSetDevice(ctx->slots[0].device);
      (void)hipStreamSynchronize(ctx->slots[0].stream);
    }
    (void)Rccl()->CommDestroy(ctx->comm);
    ctx->comm = nullptr;
  }
  if (ctx->shm_host != nullptr) {
    if (!ctx->slots.empty()) {
      (void)hipSetDevice(ctx->slots[0].device);
      (void)hipStreamSynchronize(ctx->slots[0].stream);
    }
    for (size_t k = 0; k < ctx->ipc_peers.size(); ++k)
      if (ctx->ipc_peers[k] != nullptr && int(k) != ctx->comm_rank) (void)hipIpcCloseMemHandle(ctx->ipc_peers[k]);
    ctx->ipc_peers.clear();
    if (ctx->ipc_own) (void)hipFree(ctx->ipc_own);
    if (ctx->d_peers) (void)hipFree(ctx->d_peers);
    ctx->ipc_own = nullptr;
    ctx->d_peers = nullptr;
    (void)hipHostUnregister(ctx->shm_host);
    munmap(ctx->shm_host, ctx->shm_bytes);
    if (ctx->d_round) (void)hipFree(ctx->d_round);
    if (ctx->d_mail) (void)hipFree(ctx->d_mail);
    ctx->shm_host = nullptr;
    ctx->shm_dev = nullptr;
    ctx->d_round = nullptr;
    ctx->d_mail = nullptr;
  }
  ctx->comm_ranks = 1;
  ctx->comm_rank = 0;
  return NOS_OK;
}

int nos_ctx_destroy(nos_ctx* ctx) {
  if (!ctx) return NOS_OK;
  (void)nos_ctx_comm_destroy(ctx);
  for (DeviceSlot& s : ctx->slots) {
    (void)hipSetDevice(s.device);
    if (s.own_stream) {
      (void)hipStreamSynchronize(s.own_stream);
      (void)hipStreamDestroy(s.own_stream);
    }
    if (s.partials) (void)hipFree(s.partials);
    if (s.d_out) (void)hipFree(s.d_out);
    if (s.h_out) (void)hipHostFree(s.h_out);
    if (s.counter) (void)hipFree(s.counter);
    if (s.d_lm) (void)hipFree(s.d_lm);
    if (s.copy_stream) {
      (void)hipStreamSynchronize(s.copy_stream);
      (void)hipStreamDestroy(s.copy_stream);
    }
    for (int b = 0; b < 2; ++b) {
      if (s.stage[b]) (void)hipFree(s.stage[b]);
      if (s.ing_done[b]) (void)hipEventDestroy(s.ing_done[b]);
    }
    if (s.ing_copied) (void)hipEventDestroy(s.ing_copied);
    for (int b = 0; b < 2; ++b) {
      if (s.pack_pinned[b]) (void)hipHostFree(s.pack_pinned[b]);
      if (s.pack_done[b]) (void)hipEventDestroy(s.pack_done[b]);
    }
    for (auto& pe : s.pool) (void)hipFree(pe.ptr);
    s.pool.clear();
    if (s.h_log) (void)hipHostFree(s.h_log);
    if (s.h_hist) (void)hipHostFree(s.h_hist);
    if (s.d_cluster) (void)hipFree(s.d_cluster);
    if (s.ev0) (void)hipEventDestroy(s.ev0);
    if (s.ev1) (void)hipEventDestroy(s.ev1);
    if (s.ev2) (void)hipEventDestroy(s.ev2);
    for (hipEvent_t e : s.prof_events) (void)hipEventDestroy(e);
  }
  delete ctx;
  return NOS_OK;
}

int nos_ctx_num_devices(const nos_ctx* ctx) { return ctx ? int(ctx->slots.size()) : 0; }

int nos_ctx_set_stream(nos_ctx* ctx, int shard, void* hip_stream) {
  nosd::CtxGuard guard_(ctx);  // one solve / accumulate / create at a time per context
  if (!ctx || shard < 0 || shard >= int(ctx->slots.size())) return fail(NOS_ERR_INVALID_ARGUMENT, "bad ctx / shard");
  ctx->slots[shard].stream = hip_stream ? static_cast<hipStream_t>(hip_stream) : ctx->slots[shard].own_stream;
  return NOS_OK;
}

int nos_ctx_synchronize(nos_ctx* ctx) {
  nosd::CtxGuard guard_(ctx);  // one solve / accumulate / create at a time per context
  if (!ctx) return fail(NOS_ERR_INVALID_ARGUMENT, "ctx is NULL");
  for (DeviceSlot& s : ctx->slots) {
    NOS_HIP_CHECK(hipSetDevice(s.device));
    NOS_HIP_CHECK(hipStreamSynchronize(s.stream));
  }
  return NOS_OK;
}

namespace {
struct OptionEntry {
  const char* key;
  int nosd::Settings::*field;
  int lo, hi;
};
const OptionEntry kOptions[] = {
    // key, field, lowest and highest accepted value
    {"plane_skew", &nosd::Settings::plane_skew, 0, 1 << 20},
    {"sc1", &nosd::Settings::sc1, 0, 1},
    {"nt", &nosd::Settings::nt, -1, 1},
    {"fused", &nosd::Settings::fused, 0, 1},
    {"lm_fused", &nosd::Settings::lm_fused, 0, 1},
    {"lm_window", &nosd::Settings::lm_window, 1, nosd::kLogSlots - 2},
    {"lm_single", &nosd::Settings::lm_single, 0, 1},
    {"lm_cluster", &nosd::Settings::lm_cluster, 0, 5},
    {"lm_cluster_retry_ms", &nosd::Settings::lm_cluster_retry_ms, 0, 3600000},
    {"pool", &nosd::Settings::pool, 0, 1},
    {"tile_log2", &nosd::Settings::tile_log2, -1, 24},
    {"ingest", &nosd::Settings::ingest, 0, 2},
    {"ingest_threads", &nosd::Settings::ingest_threads, 0, 1024},
    {"indexed_bpc", &nosd::Settings::indexed_bpc, 1, 16},
    {"match_dense", &nosd::Settings::match_dense, 0, 1},
    {"map_compact_keys", &nosd::Settings::map_compact_keys, 0, 1},
    {"pgo_host_scalars", &nosd::Settings::pgo_host_scalars, 0, 1},
    {"pgo_precond", &nosd::Settings::pgo_precond, 0, 1},
    {"pgo_agg", &nosd::Settings::pgo_agg, 2, 1 << 20},
    {"pgo_block", &nosd::Settings::pgo_block, 0, 1},
    {"pgo_coarse_probe", &nosd::Settings::pgo_coarse_probe, 0, 1},
    {"map_fma_mask", &nosd::Settings::map_fma_mask, 0, (1 << 26) - 1},
    {"map_eigen_version", &nosd::Settings::map_eigen_version, 33, 34},
    {"debug_cluster_abort", &nosd::Settings::debug_cluster_abort, 0, 2},
    {"lm_cluster_max_blocks", &nosd::Settings::lm_cluster_max_blocks, 1, 256},
};
bool option_in_range(const OptionEntry& o, int value);
void drop_out_of_range_settings(nosd::Settings& st) {
  const nosd::Settings defaults;
  for (const OptionEntry& o : kOptions)
    if (!option_in_range(o, st.*(o.field))) {
      fprintf(stderr, "[nos-hip] NOS_%s = %d is outside [%d, %d]: ignored\n", o.key, st.*(o.field), o.lo, o.hi);
      st.*(o.field) = defaults.*(o.field);
    }
}
bool option_in_range(const OptionEntry& o, int value) {
  if (value < o.lo || value > o.hi) return false;
  if (!strcmp(o.key, "tile_log2")) return value <= 0 || value >= 10;  // -1 by element type, 0 planar, tiles of 2^10 … 2^24
  return true;
}
}  // namespace

int nos_ctx_set_option(nos_ctx* ctx, const char* key, int value) {
  if (!ctx || !key) return fail(NOS_ERR_INVALID_ARGUMENT, "ctx / key is NULL");
  nosd::CtxGuard guard_(ctx);
  for (const OptionEntry& o : kOptions)
    if (!strcmp(o.key, key)) {
      if (!option_in_range(o, value))
        return fail(NOS_ERR_INVALID_ARGUMENT, "option '%s' = %d is outside [%d, %d]", key, value, o.lo, o.hi);
      ctx->settings.*(o.field) = value;
      for (DeviceSlot& s : ctx->slots) s.pool_enabled = ctx->settings.pool != 0;
      return NOS_OK;
    }
  return fail(NOS_ERR_INVALID_ARGUMENT, "unknown option '%s'", key);
}

int nos_ctx_get_option(const nos_ctx* ctx, const char* key, int* value) {
  if (!ctx || !key || !value) return fail(NOS_ERR_INVALID_ARGUMENT, "ctx / key / value is NULL");
  nosd::CtxGuard guard_(ctx);
  for (const OptionEntry& o : kOptions)
    if (!strcmp(o.key, key)) {
      *value = ctx->settings.*(o.field);
      return NOS_OK;
    }
  return fail(NOS_ERR_INVALID_ARGUMENT, "unknown option '%s'", key);
}

// Where the HIP runtime and the collectives library mapped into this process come from, and whether that is the ROCm
// the library was built with.  One line of JSON.
int nos_runtime_info(char* buf, size_t capacity) {
  if (!buf || capacity == 0) return fail(NOS_ERR_INVALID_ARGUMENT, "buf is NULL");
  int runtime = 0, driver = 0;
  (void)hipRuntimeGetVersion(&runtime);
  (void)hipDriverGetVersion(&driver);
  std::string hip_file;
  const std::string hip_dir = dir_of_symbol(reinterpret_cast<const void*>(&hipGetDeviceCount), &hip_file);
  RcclApi* api = Rccl();
  int rccl_version = 0;
  if (api->ok && api->GetVersion) (void)api->GetVersion(&rccl_version);
  const std::string rccl_file = api->ok ? api->path : std::string();
  const size_t slash = rccl_file.rfind('/');
  const std::string rccl_dir = slash == std::string::npos ? std::string() : rccl_file.substr(0, slash);
  const int build = HIP_VERSION;  // major * 10^7 + minor * 10^5 + patch, same encoding as hipRuntimeGetVersion
  const int n = snprintf(buf, capacity,
                         "{\"build_hip_version\": %d, \"runtime_hip_version\": %d, \"driver_version\": %d, "
                         "\"hip_runtime_path\": \"%s\", \"rccl_path\": \"%s\", \"rccl_version\": %d, "
                         "\"same_rocm_tree\": %s, \"runtime_matches_build\": %s}",
                         build, runtime, driver, hip_file.c_str(), rccl_file.c_str(), rccl_version,
                         (!rccl_dir.empty() && rccl_dir == hip_dir) ? "true" : "false",
                         (build / 100000 == runtime / 100000) ? "true" : "false");
  if (n < 0 || size_t(n) >= capacity) return fail(NOS_ERR_INVALID_ARGUMENT, "buffer too small");
  return NOS_OK;
}

// Symbol (demangled) of the hot-path kernel launched last on `shard` of this context — the instantiation the library
// chose (problem, element type, loss, launch geometry / loop form), as rocprofv3 will list it.
int nos_ctx_last_kernel(const nos_ctx* ctx, int shard, char* buf, size_t capacity) {
  if (!ctx || !buf || capacity == 0) return fail(NOS_ERR_INVALID_ARGUMENT, "ctx / buf is NULL");
  nosd::CtxGuard guard_(ctx);
  if (shard < 0 || size_t(shard) >= ctx->slots.size()) return fail(NOS_ERR_INVALID_ARGUMENT, "bad shard index");
  buf[0] = 0;
  const DeviceSlot& slot = ctx->slots[shard];
  if (slot.last_kernel == nullptr) return NOS_OK;
  NOS_HIP_CHECK(hipSetDevice(slot.device));
  const char* mangled = hipKernelNameRefByPtr(slot.last_kernel, slot.stream);
  if (mangled == nullptr) return fail(NOS_ERR_HIP, "hipKernelNameRefByPtr returned NULL");
  int status = 0;
  char* dem = abi::__cxa_demangle(mangled, nullptr, nullptr, &status);
  snprintf(buf, capacity, "%s", (status == 0 && dem) ? dem : mangled);
  free(dem);
  return NOS_OK;
}

// Number of ranks RCCL itself reports for the context's communicator (ncclCommCount); 0 without an RCCL communicator.
int nos_ctx_comm_rccl_count(const nos_ctx* ctx, int* count) {
  if (!ctx || !count) return fail(NOS_ERR_INVALID_ARGUMENT, "ctx / count is NULL");
  nosd::CtxGuard guard_(ctx);
  *count = 0;
  if (ctx->comm == nullptr) return NOS_OK;
  RcclApi* api = Rccl();
  if (!api->ok || !api->CommCount) return fail(NOS_ERR_UNSUPPORTED, "ncclCommCount unavailable");
  NOS_RCCL_CHECK(api->CommCount(ctx->comm, count));
  return NOS_OK;
}

int nos_ctx_set_launch(nos_ctx* ctx, int blocks_per_cu, int variant) {
  nosd::CtxGuard guard_(ctx);  // one solve / accumulate / create at a time per context
  if (!ctx) return fail(NOS_ERR_INVALID_ARGUMENT, "ctx is NULL");
  if (blocks_per_cu < 0 || blocks_per_cu > 32 || variant < 0 || variant >= kNumVariants)
    return fail(NOS_ERR_INVALID_ARGUMENT, "launch override out of range");
  ctx->blocks_per_cu = blocks_per_cu;
  ctx->variant = variant;
  return NOS_OK;
}

int nos_ndt_dataset_create(nos_ctx* ctx, size_t n, const double* const planes[NOS_NDT_PLANES], int dtype,
                           nos_dataset** out_ds) {
  nosd::CtxGuard guard_(ctx);  // one solve / accumulate / create at a time per context
  return create_from_host_planes(ctx, kKindNdt, n, planes, dtype, out_ds);
}

int nos_reproj_dataset_create(nos_ctx* ctx, size_t n, const double* const planes[NOS_REPROJ_PLANES], int dtype,
                              nos_dataset** out_ds) {
  nosd::CtxGuard guard_(ctx);  // one solve / accumulate / create at a time per context
  return create_from_host_planes(ctx, kKindReproj, n, planes, dtype, out_ds);
}

int nos_ndt_dataset_create_from_device(nos_ctx* ctx, size_t n, const void* const d_planes[NOS_NDT_PLANES],
                                       int src_dtype, int dtype, nos_dataset** out_ds) {
  nosd::CtxGuard guard_(ctx);  // one solve / accumulate / create at a time per context
  return create_from_device_planes(ctx, kKindNdt, n, d_planes, src_dtype, dtype, out_ds);
}

int nos_reproj_dataset_create_from_device(nos_ctx* ctx, size_t n, const void* const d_planes[NOS_REPROJ_PLANES],
                                          int src_dtype, int dtype, nos_dataset** out_ds) {
  nosd::CtxGuard guard_(ctx);  // one solve / accumulate / create at a time per context
  return create_from_device_planes(ctx, kKindReproj, n, d_planes, src_dtype, dtype, out_ds);
}

int nos_ndt_dataset_create_from_records(nos_ctx* ctx, size_t n, const void* records, size_t stride_bytes,
                                        const size_t field_offsets[NOS_NDT_PLANES], int dtype,
                                        nos_dataset** out_ds) {
  nosd::CtxGuard guard_(ctx);  // one solve / accumulate / create at a time per context
  return create_from_records(ctx, kKindNdt, n, records, stride_bytes, field_offsets, dtype, out_ds);
}

int nos_reproj_dataset_create_from_records(nos_ctx* ctx, size_t n, const void* records, size_t stride_bytes,
                                           const size_t field_offsets[NOS_REPROJ_PLANES], int dtype,
                                           nos_dataset** out_ds) {
  nosd::CtxGuard guard_(ctx);  // one solve / accumulate / create at a time per context
  return create_from_records(ctx, kKindReproj, n, records, stride_bytes, field_offsets, dtype, out_ds);
}

int nos_dataset_destroy(nos_dataset* ds) {
  nosd::CtxGuard guard_(ds ? ds->ctx : nullptr);  // one solve / accumulate / create at a time per context
  if (!ds) return NOS_OK;
  for (Shard& sh : ds->shards) {
    if (sh.data || sh.index || sh.table) (void)hipSetDevice(ds->ctx->slots[sh.slot].device);
    if (sh.data) {
      // the stream may still be reading the buffer (asynchronous entry points): wait before it is handed on
      if (sh.pooled) {
        (void)hipStreamSynchronize(ds->ctx->slots[sh.slot].stream);
        pool_release(ds->ctx->slots[sh.slot], sh.data, sh.capacity);
      } else {
        (void)hipFree(sh.data);
      }
    }
    if (sh.index && !sh.one_block) (void)hipFree(sh.index);
    if (sh.table && !sh.one_block) (void)hipFree(sh.table);
  }
  delete ds;
  return NOS_OK;
}

int nos_dataset_set_simd_class(nos_dataset* ds, int on) {
  if (!ds) return fail(NOS_ERR_INVALID_ARGUMENT, "dataset is NULL");
  nosd::CtxGuard guard_(ds->ctx);
  ds->simd_class = on != 0 ? 1 : 0;
  return NOS_OK;
}
size_t nos_dataset_size(const nos_dataset* ds) { return ds ? ds->n : 0; }
int nos_dataset_dtype(const nos_dataset* ds) { return ds ? ds->dtype : -1; }
size_t nos_dataset_stream_bytes(const nos_dataset* ds) {
  if (!ds) return 0;
  if (ds->kind == kKindNdtIndexed)  // point (3 values) + one 4-byte voxel id per slot; the voxel table is cache resident
    return ds->n * (3 * elem_size(ds->dtype) + sizeof(int32_t) * size_t(ds->shards.empty() ? 0 : ds->shards[0].n_slots));
  return ds->n * size_t(ds->n_fields) * elem_size(ds->dtype);
}

int nos_ndt6_accumulate(nos_dataset* ds, const double R[9], const double t[3], const nos_loss* loss,
                        double out28[NOS_NDT6_OUT]) {
  nosd::CtxGuard guard_(ds ? ds->ctx : nullptr);  // one solve / accumulate / create at a time per context
  Request rq;
  int rc = build_request(6, ds, R, 9, t, 3, nullptr, 0.0, loss, &rq);
  if (rc != NOS_OK) return rc;
  if (!out28) return fail(NOS_ERR_INVALID_ARGUMENT, "out28 is NULL");
  return accumulate_sync(ds, rq, out28);
}

int nos_ndt3_accumulate(nos_dataset* ds, const double R2[4], const double t2[2], const nos_loss* loss,
                        double out10[NOS_NDT3_OUT]) {
  nosd::CtxGuard guard_(ds ? ds->ctx : nullptr);  // one solve / accumulate / create at a time per context
  Request rq;
  int rc = build_request(3, ds, R2, 4, t2, 2, nullptr, 0.0, loss, &rq);
  if (rc != NOS_OK) return rc;
  if (!out10) return fail(NOS_ERR_INVALID_ARGUMENT, "out10 is NULL");
  return accumulate_sync(ds, rq, out10);
}

int nos_reproj_accumulate(nos_dataset* ds, const double R[9], const double t[3], const double intr[4],
                          const nos_loss* loss, double min_depth, double out28[NOS_REPROJ_OUT]) {
  nosd::CtxGuard guard_(ds ? ds->ctx : nullptr);  // one solve / accumulate / create at a time per context
  Request rq;
  int rc = build_request(2, ds, R, 9, t, 3, intr, min_depth, loss, &rq);
  if (rc != NOS_OK) return rc;
  if (!out28) return fail(NOS_ERR_INVALID_ARGUMENT, "out28 is NULL");
  return accumulate_sync(ds, rq, out28);
}

int nos_ndt6_accumulate_async(nos_dataset* ds, const double R[9], const double t[3], const nos_loss* loss,
                              double* d_out28) {
  nosd::CtxGuard guard_(ds ? ds->ctx : nullptr);  // one solve / accumulate / create at a time per context
  Request rq;
  int rc = build_request(6, ds, R, 9, t, 3, nullptr, 0.0, loss, &rq);
  if (rc != NOS_OK) return rc;
  return accumulate_async(ds, rq, d_out28);
}

int nos_ndt3_accumulate_async(nos_dataset* ds, const double R2[4], const double t2[2], const nos_loss* loss,
                              double* d_out10) {
  nosd::CtxGuard guard_(ds ? ds->ctx : nullptr);  // one solve / accumulate / create at a time per context
  Request rq;
  int rc = build_request(3, ds, R2, 4, t2, 2, nullptr, 0.0, loss, &rq);
  if (rc != NOS_OK) return rc;
  return accumulate_async(ds, rq, d_out10);
}

int nos_reproj_accumulate_async(nos_dataset* ds, const double R[9], const double t[3], const double intr[4],
                                const nos_loss* loss, double min_depth, double* d_out28) {
  nosd::CtxGuard guard_(ds ? ds->ctx : nullptr);  // one solve / accumulate / create at a time per context
  Request rq;
  int rc = build_request(2, ds, R, 9, t, 3, intr, min_depth, loss, &rq);
  if (rc != NOS_OK) return rc;
  return accumulate_async(ds, rq, d_out28);
}

int nos_ndt6_solve(nos_dataset* ds, double R[9], double t[3], const nos_loss* loss, const nos_lm_options* options,
                   nos_lm_report* report) {
  nosd::CtxGuard guard_(ds ? ds->ctx : nullptr);  // one solve / accumulate / create at a time per context
  Request rq;
  int rc = build_request(6, ds, R, 9, t, 3, nullptr, 0.0, loss, &rq);
  if (rc != NOS_OK) return rc;
  return lm_solve(ds, rq, options, R, 9, t, 3, report);
}

int nos_ndt3_solve(nos_dataset* ds, double R2[4], double t2[2], const nos_loss* loss, const nos_lm_options* options,
                   nos_lm_report* report) {
  nosd::CtxGuard guard_(ds ? ds->ctx : nullptr);  // one solve / accumulate / create at a time per context
  Request rq;
  int rc = build_request(3, ds, R2, 4, t2, 2, nullptr, 0.0, loss, &rq);
  if (rc != NOS_OK) return rc;
  return lm_solve(ds, rq, options, R2, 4, t2, 2, report);
}

int nos_reproj_solve(nos_dataset* ds, double R[9], double t[3], const double intr[4], const nos_loss* loss,
                     double min_depth, const nos_lm_options* options, nos_lm_report* report) {
  nosd::CtxGuard guard_(ds ? ds->ctx : nullptr);  // one solve / accumulate / create at a time per context
  Request rq;
  int rc = build_request(2, ds, R, 9, t, 3, intr, min_depth, loss, &rq);
  if (rc != NOS_OK) return rc;
  return lm_solve(ds, rq, options, R, 9, t, 3, report);
}

int nos_ndt6_time_kernel(nos_dataset* ds, const double R[9], const double t[3], const nos_loss* loss, int repeats,
                         double* kernel_ms, double* total_ms) {
  nosd::CtxGuard guard_(ds ? ds->ctx : nullptr);  // one solve / accumulate / create at a time per context
  Request rq;
  int rc = build_request(6, ds, R, 9, t, 3, nullptr, 0.0, loss, &rq);
  if (rc != NOS_OK) return rc;
  return time_kernel(ds, rq, repeats, kernel_ms, total_ms);
}

int nos_ndt3_time_kernel(nos_dataset* ds, const double R2[4], const double t2[2], const nos_loss* loss, int repeats,
                         double* kernel_ms, double* total_ms) {
  nosd::CtxGuard guard_(ds ? ds->ctx : nullptr);  // one solve / accumulate / create at a time per context
  Request rq;
  int rc = build_request(3, ds, R2, 4, t2, 2, nullptr, 0.0, loss, &rq);
  if (rc != NOS_OK) return rc;
  return time_kernel(ds, rq, repeats, kernel_ms, total_ms);
}

int nos_reproj_time_kernel(nos_dataset* ds, const double R[9], const double t[3], const double intr[4],
                           const nos_loss* loss, double min_depth, int repeats, double* kernel_ms,
                           double* total_ms) {
  nosd::CtxGuard guard_(ds ? ds->ctx : nullptr);  // one solve / accumulate / create at a time per context
  Request rq;
  int rc = build_request(2, ds, R, 9, t, 3, intr, min_depth, loss, &rq);
  if (rc != NOS_OK) return rc;
  return time_kernel(ds, rq, repeats, kernel_ms, total_ms);
}

int nos_debug_lm_step(nos_ctx* ctx, int dof, const double* sums, const double settings[4], double state[22]) {
  nosd::CtxGuard guard_(ctx);  // one solve / accumulate / create at a time per context
  if (!ctx || !sums || !settings || !state || (dof != 6 && dof != 3)) return fail(NOS_ERR_INVALID_ARGUMENT, "bad argument");
  DeviceSlot& slot = ctx->slots[0];
  NOS_HIP_CHECK(hipSetDevice(slot.device));
  nos::LmDevice lmd{};
  for (int k = 0; k < 9; ++k) lmd.st.R[k] = state[k];
  for (int k = 0; k < 3; ++k) lmd.st.t[k] = state[9 + k];
  lmd.st.q.w = state[12], lmd.st.q.x = state[13], lmd.st.q.y = state[14], lmd.st.q.z = state[15];
  lmd.st.lambda = state[16], lmd.st.previous_cost = state[17], lmd.st.cost = state[18];
  lmd.st.iteration = int(state[19]), lmd.st.done = int(state[20]), lmd.st.ok = int(state[21]);
  lmd.settings.max_iterations = int(settings[0]);
  lmd.settings.gradient_tolerance = settings[1];
  lmd.settings.parameter_tolerance = settings[2];
  lmd.settings.float_schedule = int(settings[3]);
  const int n_out = dof == 6 ? 28 : 10;
  NOS_HIP_CHECK(hipMemcpyAsync(slot.d_lm, &lmd, sizeof lmd, hipMemcpyHostToDevice, slot.stream));
  NOS_HIP_CHECK(hipMemcpyAsync(slot.d_out, sums, sizeof(double) * n_out, hipMemcpyHostToDevice, slot.stream));
  NOS_HIP_CHECK(hipStreamSynchronize(slot.stream));  // lmd is a stack object
  if (dof == 6)
    hipLaunchKernelGGL((nos::lm_step_kernel<28>), dim3(1), dim3(64), 0, slot.stream, slot.d_out, slot.d_lm,
                       static_cast<double*>(nullptr), static_cast<unsigned long long*>(nullptr), 0ull);
  else
    hipLaunchKernelGGL((nos::lm_step_kernel<10>), dim3(1), dim3(64), 0, slot.stream, slot.d_out, slot.d_lm,
                       static_cast<double*>(nullptr), static_cast<unsigned long long*>(nullptr), 0ull);
  NOS_HIP_CHECK(hipGetLastError());
  NOS_HIP_CHECK(hipMemcpyAsync(&lmd, slot.d_lm, sizeof lmd, hipMemcpyDeviceToHost, slot.stream));
  NOS_HIP_CHECK(hipStreamSynchronize(slot.stream));
  for (int k = 0; k < 9; ++k) state[k] = lmd.st.R[k];
  for (int k = 0; k < 3; ++k) state[9 + k] = lmd.st.t[k];
  state[12] = lmd.st.q.w, state[13] = lmd.st.q.x, state[14] = lmd.st.q.y, state[15] = lmd.st.q.z;
  state[16] = lmd.st.lambda, state[17] = lmd.st.previous_cost, state[18] = lmd.st.cost;
  state[19] = lmd.st.iteration, state[20] = lmd.st.done, state[21] = lmd.st.ok;
  return NOS_OK;
}

int nos_comm_get_unique_id(unsigned char id[NOS_COMM_ID_BYTES]) {
  if (!id) return fail(NOS_ERR_INVALID_ARGUMENT, "id is NULL");
  RcclApi* api = Rccl();
  if (!api->ok) return fail(NOS_ERR_UNSUPPORTED, "librccl could not be loaded: %s", dlerror() ? dlerror() : "missing symbols");
  static_assert(NOS_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "unique id size");
  ncclUniqueId uid;
  NOS_RCCL_CHECK(api->GetUniqueId(&uid));
  memcpy(id, uid.internal, NOS_COMM_ID_BYTES);
  return NOS_OK;
}

int nos_ctx_comm_init(nos_ctx* ctx, int n_ranks, int rank, const unsigned char id[NOS_COMM_ID_BYTES]) {
  nosd::CtxGuard guard_(ctx);  // one solve / accumulate / create at a time per context
  if (!ctx || !id || n_ranks < 1 || rank < 0 || rank >= n_ranks) return fail(NOS_ERR_INVALID_ARGUMENT, "bad comm arguments");
  if (ctx->slots.size() != 1) return fail(NOS_ERR_UNSUPPORTED, "a communicator needs a single-device context");
  if (ctx->comm != nullptr) return fail(NOS_ERR_INVALID_ARGUMENT, "communicator already initialised");
  RcclApi* api = Rccl();
  if (!api->ok) return fail(NOS_ERR_UNSUPPORTED, "librccl could not be loaded");
  NOS_HIP_CHECK(hipSetDevice(ctx->slots[0].device));
  ncclUniqueId uid;
  memcpy(uid.internal, id, NOS_COMM_ID_BYTES);
  ncclComm_t comm = nullptr;
  NOS_RCCL_CHECK(api->CommInitRank(&comm, n_ranks, uid, rank));
  ctx->comm = comm;
  ctx->comm_ranks = n_ranks;
  return NOS_OK;
}

int nos_ctx_comm_size(const nos_ctx* ctx) { return (ctx && (ctx->comm || ctx->shm_dev)) ? ctx->comm_ranks : 0; }

}  // extern "C"

namespace {
// Both mailbox communicators: the handshake and control words always live in the POSIX shm segment; the SLOTS the kernels
// exchange through live there too (device_slots = false: bytes travel over PCIe to host memory) or in fine-grained device
// memory of every rank, exported with hipIpcGetMemHandle and opened by the peers (device_slots = true: a rank writes its
// sums straight into every peer's buffer — over xGMI between GPUs — and polls only its own memory).
int comm_init_mailbox(nos_ctx* ctx, int n_ranks, int rank, const char* shm_name, bool device_slots) {
  nosd::CtxGuard guard_(ctx);  // one solve / accumulate / create at a time per context
  if (!ctx || !shm_name || shm_name[0] != '/' || n_ranks < 1 || n_ranks > 64 || rank < 0 || rank >= n_ranks)
    return fail(NOS_ERR_INVALID_ARGUMENT, "bad comm arguments (name must start with '/', at most 64 ranks)");
  if (ctx->slots.size() != 1) return fail(NOS_ERR_UNSUPPORTED, "a communicator needs a single-device context");
  if (ctx->comm != nullptr || ctx->shm_dev != nullptr) return fail(NOS_ERR_INVALID_ARGUMENT, "communicator already initialised");
  DeviceSlot& slot = ctx->slots[0];
  NOS_HIP_CHECK(hipSetDevice(slot.device));
  size_t bytes = 0;
  // Attach with a handshake that cannot be fooled by a segment of that name left behind by a crashed run (whose flag
  // words would otherwise match the first round numbers and feed stale sums into the exchange):
  //   rank 0 unlinks the name, creates the segment EXCLUSIVELY (a fresh, zero-filled inode), publishes a random nonce and
  //   acknowledges every rank's own fresh random hello word with hello ^ nonce;
  //   rank k opens the name (retrying), writes its hello and accepts the mapping only when its acknowledgement shows up —
  //   a stale inode never acknowledges a fresh 64-bit random, so rank k drops it and opens the name again.
  // Bounded: NOS_SHM_ATTACH_TIMEOUT_MS (default 30 s) in total.  Header (after the slots): [0] nonce, [1..64] hello, [65..128] ack.
  const int kAttachTimeoutMs = std::max(100, env_int("NOS_SHM_ATTACH_TIMEOUT_MS", 30000));  // set-up path, not the solve path
  const size_t slots_bytes = size_t(n_ranks) * 2 * nos::kMailSlotDoubles * sizeof(double);
  // header (after the slots), in 8-byte words: [0] nonce, [1..64] hello, [65..128] ack, [129..192] ipc-ready, [193..704] 64 IPC handles of 64 bytes
  const size_t header_words = 1 + 64 + 64 + 64 + 64 * 8;
  static_assert(sizeof(hipIpcMemHandle_t) == 64, "IPC handle size");
  bytes = (slots_bytes + header_words * sizeof(unsigned long long) + 4095) & ~size_t(4095);
  auto now_ms = [] {
    return std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
  };
  auto fresh_random = [&]() -> unsigned long long {
    unsigned long long v = 0;
    std::random_device rd;
    while (v == 0) v = (static_cast<unsigned long long>(rd()) << 32) ^ rd() ^ (static_cast<unsigned long long>(getpid()) << 17);
    return v;
  };
  const long long deadline = now_ms() + kAttachTimeoutMs;
  void* host = MAP_FAILED;
  if (rank == 0) {
    (void)shm_unlink(shm_name);
    int fd = shm_open(shm_name, O_CREAT | O_EXCL | O_RDWR, 0600);
    if (fd < 0 && errno == EEXIST) {  // somebody re-created it in between: once more
      (void)shm_unlink(shm_name);
      fd = shm_open(shm_name, O_CREAT | O_EXCL | O_RDWR, 0600);
    }
    if (fd < 0) return fail(NOS_ERR_HIP, "shm_open(%s) failed: %s", shm_name, strerror(errno));
    if (ftruncate(fd, off_t(bytes)) != 0) {  // fresh pages read as zero = round 0 everywhere
      const int e = errno;
      close(fd);
      (void)shm_unlink(shm_name);
      return fail(NOS_ERR_HIP, "ftruncate(%s) failed: %s", shm_name, strerror(e));
    }
    host = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (host == MAP_FAILED) return fail(NOS_ERR_HIP, "mmap(%s) failed: %s", shm_name, strerror(errno));
    auto* hdr = reinterpret_cast<std::atomic<unsigned long long>*>(static_cast<char*>(host) + slots_bytes);
    const unsigned long long nonce = fresh_random();
    hdr[0].store(nonce, std::memory_order_release);
    for (int k = 1; k < n_ranks; ++k) {
      unsigned long long hello = 0;
      while ((hello = hdr[1 + k].load(std::memory_order_acquire)) == 0) {
        if (now_ms() > deadline) {
          munmap(host, bytes);
          return fail(NOS_ERR_HIP, "rank %d did not attach to the mailbox %s within %d ms", k, shm_name, kAttachTimeoutMs);
        }
        usleep(200);
      }
      hdr[65 + k].store(hello ^ nonce, std::memory_order_release);
    }
  } else {
    const unsigned long long hello = fresh_random();
    for (;;) {
      if (now_ms() > deadline)
        return fail(NOS_ERR_HIP, "mailbox %s: no acknowledgement from rank 0 within %d ms", shm_name, kAttachTimeoutMs);
      const int fd = shm_open(shm_name, O_RDWR, 0600);
      if (fd < 0) {
        usleep(500);
        continue;
      }
      struct stat st {};
      if (fstat(fd, &st) != 0 || size_t(st.st_size) < bytes) {  // not sized yet (or somebody else's segment)
        close(fd);
        usleep(500);
        continue;
      }
      void* m = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
      close(fd);
      if (m == MAP_FAILED) return fail(NOS_ERR_HIP, "mmap(%s) failed: %s", shm_name, strerror(errno));
      auto* hdr = reinterpret_cast<std::atomic<unsigned long long>*>(static_cast<char*>(m) + slots_bytes);
      hdr[1 + rank].store(hello, std::memory_order_release);
      bool acked = false;
      const long long until = std::min<long long>(deadline, now_ms() + 250);  // then look at the name again
      while (now_ms() <= until) {
        const unsigned long long nonce = hdr[0].load(std::memory_order_acquire);
        if (nonce != 0 && hdr[65 + rank].load(std::memory_order_acquire) == (hello ^ nonce)) {
          acked = true;
          break;
        }
        usleep(200);
      }
      if (acked) {
        host = m;
        break;
      }
      munmap(m, bytes);  // stale inode (or rank 0 not there yet): drop it and open the name again
    }
  }
  hipError_t e = hipHostRegister(host, bytes, hipHostRegisterMapped | hipHostRegisterPortable);
  void* dev = nullptr;
  if (e == hipSuccess) {
    e = hipHostGetDevicePointer(&dev, host, 0);
    if (e != hipSuccess) (void)hipHostUnregister(host);
  }
  unsigned long long* d_round = nullptr;
  if (e == hipSuccess) {
    e = hipMalloc(reinterpret_cast<void**>(&d_round), 2 * sizeof(unsigned long long));  // [0] round, [1] patient-until round
    if (e == hipSuccess) e = hipMemset(d_round, 0, 2 * sizeof(unsigned long long));
    if (e != hipSuccess) (void)hipHostUnregister(host);
  }
  if (e != hipSuccess) {
    if (d_round) (void)hipFree(d_round);
    munmap(host, bytes);
    return fail(NOS_ERR_HIP, "mapping the mailbox into the GPU failed: %s", hipGetErrorString(e));
  }
  *reinterpret_cast<volatile unsigned int*>(slot.h_out + kCommErrorSlot) = 0u;
  ctx->shm_host = host;
  ctx->shm_bytes = bytes;
  ctx->shm_dev = static_cast<double*>(dev);
  ctx->d_round = d_round;
  ctx->comm_ranks = n_ranks;
  ctx->comm_rank = rank;
  if (device_slots) {
    // every rank: its own [n_ranks][2][kMailSlotDoubles] buffer in fine-grained device memory (peers write into it across the
    // fabric while this GPU polls it: no cache may keep a stale copy), exported through the shm header and opened by the others
    auto* hdr = reinterpret_cast<std::atomic<unsigned long long>*>(static_cast<char*>(host) + slots_bytes);
    hipIpcMemHandle_t* handles = reinterpret_cast<hipIpcMemHandle_t*>(hdr + 193);
    double* own = nullptr;
    // second half: the granule slots of the one-launch loop's in-launch exchange (solve_cluster_kernel, stage 3)
    e = hipExtMallocWithFlags(reinterpret_cast<void**>(&own), 2 * slots_bytes, hipDeviceMallocFinegrained);
    if (e == hipSuccess) e = hipMemset(own, 0, 2 * slots_bytes);
    hipIpcMemHandle_t mine{};
    if (e == hipSuccess) e = hipIpcGetMemHandle(&mine, own);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e != hipSuccess) {
      if (own) (void)hipFree(own);
      (void)nos_ctx_comm_destroy(ctx);
      return fail(NOS_ERR_UNSUPPORTED, "device-memory mailbox: fine-grained allocation / IPC export failed: %s", hipGetErrorString(e));
    }
    ctx->ipc_own = own;
    memcpy(&handles[rank], &mine, sizeof mine);
    hdr[129 + rank].store(1ull, std::memory_order_release);
    ctx->ipc_peers.assign(size_t(n_ranks), nullptr);
    ctx->ipc_peers[size_t(rank)] = own;
    for (int k = 0; k < n_ranks; ++k) {
      if (k == rank) continue;
      while (hdr[129 + k].load(std::memory_order_acquire) == 0ull) {
        if (now_ms() > deadline) {
          (void)nos_ctx_comm_destroy(ctx);
          return fail(NOS_ERR_HIP, "device-memory mailbox %s: rank %d did not publish its IPC handle within %d ms", shm_name, k, kAttachTimeoutMs);
        }
        usleep(200);
      }
      hipIpcMemHandle_t theirs;
      memcpy(&theirs, &handles[k], sizeof theirs);
      void* p = nullptr;
      e = hipIpcOpenMemHandle(&p, theirs, hipIpcMemLazyEnablePeerAccess);
      if (e != hipSuccess) {
        (void)nos_ctx_comm_destroy(ctx);
        return fail(NOS_ERR_UNSUPPORTED, "device-memory mailbox: hipIpcOpenMemHandle(rank %d) failed: %s", k, hipGetErrorString(e));
      }
      ctx->ipc_peers[size_t(k)] = static_cast<double*>(p);
    }
    e = hipMalloc(reinterpret_cast<void**>(&ctx->d_peers), sizeof(double*) * size_t(n_ranks));
    if (e == hipSuccess) e = hipMemcpy(ctx->d_peers, ctx->ipc_peers.data(), sizeof(double*) * size_t(n_ranks), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
      (void)nos_ctx_comm_destroy(ctx);
      return fail(NOS_ERR_HIP, "device-memory mailbox: uploading the peer table failed: %s", hipGetErrorString(e));
    }
    // nobody may start exchanging (or leave and free its buffer) before every rank has opened every buffer
    hdr[129 + rank].store(2ull, std::memory_order_release);
    for (int k = 0; k < n_ranks; ++k)
      while (hdr[129 + k].load(std::memory_order_acquire) < 2ull) {
        if (now_ms() > deadline) {
          (void)nos_ctx_comm_destroy(ctx);
          return fail(NOS_ERR_HIP, "device-memory mailbox %s: rank %d did not finish attaching within %d ms", shm_name, k, kAttachTimeoutMs);
        }
        usleep(200);
      }
  }
  const nos::Mailbox mb = mailbox_of(ctx, slot);
  e = hipMalloc(reinterpret_cast<void**>(&ctx->d_mail), sizeof(nos::Mailbox));
  if (e == hipSuccess) e = hipMemcpy(ctx->d_mail, &mb, sizeof mb, hipMemcpyHostToDevice);
  if (e != hipSuccess) {
    (void)nos_ctx_comm_destroy(ctx);
    return fail(NOS_ERR_HIP, "uploading the mailbox descriptor failed: %s", hipGetErrorString(e));
  }
  return NOS_OK;
}

}  // namespace

extern "C" {

int nos_ctx_comm_init_shm(nos_ctx* ctx, int n_ranks, int rank, const char* shm_name) {
  return comm_init_mailbox(ctx, n_ranks, rank, shm_name, false);
}

int nos_ctx_comm_init_shm_device(nos_ctx* ctx, int n_ranks, int rank, const char* shm_name) {
  return comm_init_mailbox(ctx, n_ranks, rank, shm_name, true);
}

int nos_comm_shm_unlink(const char* shm_name) {
  if (!shm_name) return fail(NOS_ERR_INVALID_ARGUMENT, "name is NULL");
  if (shm_unlink(shm_name) != 0 && errno != ENOENT) return fail(NOS_ERR_HIP, "shm_unlink(%s) failed: %s", shm_name, strerror(errno));
  return NOS_OK;
}

int nos_ctx_comm_allreduce(nos_ctx* ctx, double* values, int count) {
  nosd::CtxGuard guard_(ctx);  // one solve / accumulate / create at a time per context
  if (!ctx || !values || count < 1 || count > kMaxOut) return fail(NOS_ERR_INVALID_ARGUMENT, "bad allreduce arguments");
  if (ctx->comm == nullptr && ctx->shm_dev == nullptr) return fail(NOS_ERR_INVALID_ARGUMENT, "no communicator");
  DeviceSlot& slot = ctx->slots[0];
  NOS_HIP_CHECK(hipSetDevice(slot.device));
  NOS_HIP_CHECK(hipMemcpyAsync(slot.d_out, values, sizeof(double) * count, hipMemcpyHostToDevice, slot.stream));
  if (ctx->shm_dev != nullptr) {
    hipLaunchKernelGGL(nos::mailbox_allreduce_kernel, dim3(1), dim3(64), 0, slot.stream, mailbox_of(ctx, slot), slot.d_out, count);
    NOS_HIP_CHECK(hipGetLastError());
  } else {
    NOS_RCCL_CHECK(Rccl()->AllReduce(slot.d_out, slot.d_out, size_t(count), ncclDouble, ncclSum, ctx->comm, slot.stream));
  }
  NOS_HIP_CHECK(hipMemcpyAsync(values, slot.d_out, sizeof(double) * count, hipMemcpyDeviceToHost, slot.stream));
  NOS_HIP_CHECK(hipStreamSynchronize(slot.stream));
  return check_mailbox_error(ctx, slot);
}

int nos_ctx_profile_begin(nos_ctx* ctx, int max_launches, int sample_every) {
  nosd::CtxGuard guard_(ctx);  // one solve / accumulate / create at a time per context
  if (!ctx || max_launches < 1 || max_launches > (1 << 20) || sample_every < 0)
    return fail(NOS_ERR_INVALID_ARGUMENT, "bad profile request");
  for (DeviceSlot& s : ctx->slots) {
    NOS_HIP_CHECK(hipSetDevice(s.device));
    if (sample_every == 0) {  // bracket form
      s.prof_used = 0;
      s.prof_every = 0;
      s.prof_launches = 0;
      s.prof_on = true;
      NOS_HIP_CHECK(hipEventRecord(s.ev0, s.stream));
      continue;
    }
    while (s.prof_events.size() < size_t(max_launches) * 2) {
      hipEvent_t e = nullptr;
      NOS_HIP_CHECK(hipEventCreate(&e));
      s.prof_events.push_back(e);
    }
    s.prof_used = 0;
    s.prof_every = sample_every;
    s.prof_launches = 0;
    s.prof_on = true;
  }
  return NOS_OK;
}

int nos_ctx_profile_end(nos_ctx* ctx, int* n_launches, double* mean_ms, double* min_ms, double* max_ms) {
  nosd::CtxGuard guard_(ctx);  // one solve / accumulate / create at a time per context
  if (!ctx) return fail(NOS_ERR_INVALID_ARGUMENT, "ctx is NULL");
  int count = 0;
  double sum = 0.0, lo = 1e300, hi = 0.0;
  for (DeviceSlot& s : ctx->slots) {
    const bool bracket = s.prof_on && s.prof_every == 0;
    s.prof_on = false;
    NOS_HIP_CHECK(hipSetDevice(s.device));
    if (bracket) {
      NOS_HIP_CHECK(hipEventRecord(s.ev1, s.stream));
      NOS_HIP_CHECK(hipEventSynchronize(s.ev1));
      float ms = 0.f;
      NOS_HIP_CHECK(hipEventElapsedTime(&ms, s.ev0, s.ev1));
      if (s.prof_launches > 0) {
        const double per = double(ms) / double(s.prof_launches);
        sum += per * double(s.prof_launches);
        lo = std::min(lo, per);
        hi = std::max(hi, per);
        count += int(s.prof_launches);
      }
      s.prof_every = 1;
      continue;
    }
    NOS_HIP_CHECK(hipStreamSynchronize(s.stream));
    for (size_t i = 0; i + 1 < s.prof_used; i += 2) {
      float ms = 0.f;
      NOS_HIP_CHECK(hipEventElapsedTime(&ms, s.prof_events[i], s.prof_events[i + 1]));
      sum += ms;
      lo = std::min(lo, double(ms));
      hi = std::max(hi, double(ms));
      ++count;
    }
    s.prof_used = 0;
  }
  if (n_launches) *n_launches = count;
  if (mean_ms) *mean_ms = count ? sum / count : 0.0;
  if (min_ms) *min_ms = count ? lo : 0.0;
  if (max_ms) *max_ms = count ? hi : 0.0;
  return NOS_OK;
}

const char* nos_status_string(int status) {
  switch (status) {
    case NOS_OK: return "ok";
    case NOS_ERR_INVALID_ARGUMENT: return "invalid argument";
    case NOS_ERR_NO_DEVICE: return "no HIP device (no CPU fallback)";
    case NOS_ERR_HIP: return "HIP runtime error";
    case NOS_ERR_OUT_OF_MEMORY: return "out of memory";
    case NOS_ERR_WRONG_KIND: return "dataset kind mismatch";
    case NOS_ERR_UNSUPPORTED: return "unsupported";
  }
  return "unknown status";
}

const char* nos_last_error(void) { return nosd::last_error_text(); }
#ifdef NOS_ALL_VARIANTS
const char* nos_version(void) { return "nos-hip 0.3 (gfx950, all launch geometries)"; }
#else
const char* nos_version(void) { return "nos-hip 0.3 (gfx950)"; }
#endif

}  // extern "C"

