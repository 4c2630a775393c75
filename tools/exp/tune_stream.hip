// tune_stream.hip — stand-alone geometry sweep of the streaming assemble kernel (tools/, not product code).
// Every configuration is its own template instantiation, i.e. its own kernel name: run under
//   rocprofv3 --kernel-trace --stats -- tools/_bin/tune_stream reproj f64
// and read the average duration per kernel from the stats file; the program itself prints event-bracketed trains.
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -o tools/_bin/tune_stream tools/exp/tune_stream.hip
#include "../../nonlinear_optimizer_for_slam_amd/csrc/nos_internal.hpp"

#include <cstdio>
#include <random>
#include <string>
#include <vector>

using namespace nos;

#define CK(x)                                                                      \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) {                                                        \
      fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                      \
      exit(1);                                                                     \
    }                                                                              \
  } while (0)

// Same loads, (almost) no arithmetic: what the launch costs when only the memory system works — the floor of the geometry
template <typename T>
struct StreamParams {
  T R[9];
  T t[3];
  T la, lb, lc;
};
template <typename T>
__device__ __forceinline__ void set_pose(StreamParams<T>&, const nos::LmDevice*) {}  // found by ADL from lm_prologue

template <typename T, int FIELDS>
struct StreamOnlyProblem {
  static constexpr int kFields = FIELDS;
  static constexpr int kOut = 28;
  using Params = StreamParams<T>;
  __device__ static __forceinline__ void item(const T (&x)[FIELDS], const Params&, bool, T (&acc)[28]) {
#pragma unroll
    for (int f = 0; f < FIELDS; ++f) acc[f] += x[f];
  }
};

struct Bench {
  TiledLayout L{};
  double* partials = nullptr;
  unsigned int* counter = nullptr;
  double* out = nullptr;
  int reps = 200;
  int num_cus = 256;
};

template <typename Problem, typename T, int ITEMS, int BLOCK, int MINW, bool NT, int PF>
void run(const Bench& b, const typename Problem::Params& P, int bpc, double* ref, const char* tag) {
  constexpr uint32_t kChunk = BLOCK * ITEMS;
  if (b.L.n_padded % kChunk != 0) {
    printf("%-34s skipped (layout)\n", tag);
    return;
  }
  const uint32_t n_chunks = uint32_t(b.L.n_padded / kChunk);
  int grid = int(std::min<uint64_t>(n_chunks, uint64_t(bpc) * b.num_cus));
  FusedFinal fin{};
  fin.counter = b.counter;
  fin.out_dev = b.out;
  fin.write_through = grid <= b.num_cus ? 1 : 0;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  auto launch = [&]() {
    hipLaunchKernelGGL((assemble_kernel<Problem, T, ITEMS, BLOCK, MINW, NT, PF>), dim3(grid), dim3(BLOCK), 0, 0, b.L, P, n_chunks,
                       b.partials, fin);
  };
  for (int i = 0; i < 20; ++i) launch();
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0, 0));
  for (int i = 0; i < b.reps; ++i) launch();
  CK(hipEventRecord(e1, 0));
  CK(hipEventSynchronize(e1));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  double got[28];
  CK(hipMemcpy(got, b.out, sizeof(double) * Problem::kOut, hipMemcpyDeviceToHost));
  double err = 0;
  if (ref[27] == 0.0 && ref[0] == 0.0)
    for (int k = 0; k < Problem::kOut; ++k) ref[k] = got[k];
  for (int k = 0; k < Problem::kOut; ++k) err = std::max(err, std::fabs(got[k] - ref[k]) / (std::fabs(ref[k]) + 1e-300));
  const double us = 1e3 * ms / b.reps;
  const double bytes = double(b.L.n) * Problem::kFields * sizeof(T);
  printf("%-34s grid %5d  %7.2f us/launch (train)  %6.2f TB/s  relerr %.1e\n", tag, grid, us, bytes / us * 1e-6, err);
  fflush(stdout);
}

template <typename T>
void bench_reproj(size_t n, int num_cus) {
  using PH = ReprojProblem<T, 2>;  // Huber
  std::mt19937_64 rng(20250912);
  std::uniform_real_distribution<double> U(0.0, 1.0);
  std::normal_distribution<double> N01(0.0, 1.0);
  Bench b;
  b.num_cus = num_cus;
  const size_t pad = 8192;
  b.L.n = n;
  b.L.n_padded = (n + pad - 1) / pad * pad;
  b.L.tile_stride = 0;
  b.L.field_stride = b.L.n_padded + 1088;
  b.L.tile_shift = 40;
  b.L.tile_mask = 0xFFFFFFFFu;
  std::vector<T> h(size_t(5) * b.L.field_stride, T(0));
  for (size_t i = 0; i < n; ++i) {
    const double z = 2.0 + 4.0 * U(rng), x = (U(rng) - 0.5) * z, y = (U(rng) - 0.5) * z * (2.0 / 3.0);
    h[0 * b.L.field_stride + i] = T(x);
    h[1 * b.L.field_stride + i] = T(y);
    h[2 * b.L.field_stride + i] = T(z);
    h[3 * b.L.field_stride + i] = T(525.0 * x / z + 320.0 + 0.5 * N01(rng));
    h[4 * b.L.field_stride + i] = T(525.0 * y / z + 240.0 + 0.5 * N01(rng));
  }
  T* d = nullptr;
  CK(hipMalloc(reinterpret_cast<void**>(&d), h.size() * sizeof(T)));
  CK(hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
  b.L.base = d;
  CK(hipMalloc(reinterpret_cast<void**>(&b.partials), size_t(8192) * 28 * sizeof(double)));
  CK(hipMalloc(reinterpret_cast<void**>(&b.counter), 4096));
  CK(hipMemset(b.counter, 0, 4096));
  CK(hipMalloc(reinterpret_cast<void**>(&b.out), 32 * sizeof(double)));
  ReprojParams<T> P{};
  const double c = std::cos(0.02), s = std::sin(0.02);
  const double R[9] = {c, -s, 0, s, c, 0, 0, 0, 1};
  for (int k = 0; k < 9; ++k) P.R[k] = T(R[k]);
  P.t[0] = T(0.01), P.t[1] = T(-0.02), P.t[2] = T(0.03);
  P.inv_fx = T(1.0 / 525.0), P.inv_fy = T(1.0 / 525.0), P.cx = T(320), P.cy = T(240);
  P.min_depth = T(0.03);
  set_reproj_rules(P, false);
  const double th = 1.0 / 525.0;
  P.la = T(th), P.lb = T(th * th), P.lc = T(2.0 * th);
  double ref[28] = {0};
#define RUN(ITEMS, BLOCK, MINW, PF, BPC) \
  run<PH, T, ITEMS, BLOCK, MINW, false, PF>(b, P, BPC, ref, "reproj items=" #ITEMS " block=" #BLOCK " minw=" #MINW " pf=" #PF " bpc=" #BPC)
  RUN(1, 512, 3, false, 1);
  RUN(2, 512, 2, false, 1);
  RUN(4, 512, 2, false, 1);
  RUN(1, 1024, 4, false, 1);
  RUN(2, 1024, 4, false, 1);
  RUN(4, 1024, 4, false, 1);
  RUN(1, 512, 4, false, 2);
  RUN(2, 512, 4, false, 2);
  RUN(4, 512, 4, false, 2);
  RUN(2, 256, 4, false, 4);
  RUN(4, 256, 4, false, 4);
  RUN(2, 512, 2, true, 1);
  RUN(4, 512, 2, true, 1);
  RUN(2, 1024, 4, true, 1);
  RUN(4, 1024, 4, true, 1);
  RUN(2, 512, 4, true, 2);
  RUN(8, 512, 2, false, 1);
  RUN(8, 256, 2, false, 2);
  RUN(8, 256, 2, false, 1);
  RUN(2, 512, 2, 2, 1);
  RUN(1, 512, 3, 3, 1);
  RUN(2, 512, 2, 3, 1);
  RUN(4, 512, 2, 3, 1);
  RUN(1, 1024, 4, 3, 1);
  RUN(2, 1024, 4, 3, 1);
#undef RUN
  {  // the floor of the geometry: the same loads with no arithmetic
    using PS = StreamOnlyProblem<T, 5>;
    typename PS::Params Q{};
    double r2[28] = {0};
    run<PS, T, 1, 512, 3, false, 0>(b, Q, 1, r2, "stream-only items=1 block=512");
    double r3[28] = {0};
    run<PS, T, 2, 512, 2, false, 0>(b, Q, 1, r3, "stream-only items=2 block=512");
    double r4[28] = {0};
    run<PS, T, 8, 512, 2, false, 0>(b, Q, 1, r4, "stream-only items=8 block=512");
    double r5[28] = {0};
    run<PS, T, 4, 512, 2, false, 1>(b, Q, 1, r5, "stream-only items=4 block=512 pf=1");
  }
  // size sweep of two geometries: per-launch time = fixed part + n x slope
  for (size_t m : {size_t(131072), size_t(524288), size_t(1048576), size_t(2000000), size_t(4000000), size_t(8000000)}) {
    if (m > n * 4) break;
  }
}

template <typename T>
void bench_ndt6(size_t n, int num_cus, int tile_log2) {
  using PE = Ndt6Problem<T, 1>;  // exponential
  std::mt19937_64 rng(20250912);
  std::normal_distribution<double> N01(0.0, 1.0);
  Bench b;
  b.num_cus = num_cus;
  b.reps = 60;
  const size_t pad = 8192;
  b.L.n = n;
  b.L.n_padded = (n + pad - 1) / pad * pad;
  if (tile_log2 > 0) {
    const size_t tile = size_t(1) << tile_log2;
    b.L.tile_stride = tile * 15;
    b.L.field_stride = tile;
    b.L.tile_shift = uint32_t(tile_log2);
    b.L.tile_mask = uint32_t(tile - 1);
  } else {
    b.L.tile_stride = 0;
    b.L.field_stride = b.L.n_padded + 1088;
    b.L.tile_shift = 40;
    b.L.tile_mask = 0xFFFFFFFFu;
  }
  const size_t elems = (b.L.tile_stride == 0 ? b.L.field_stride : b.L.n_padded) * 15;
  std::vector<T> h(elems);
  for (size_t i = 0; i < elems; ++i) h[i] = T(0.3 * N01(rng));  // values only steer exp(): any finite data times the same
  T* d = nullptr;
  CK(hipMalloc(reinterpret_cast<void**>(&d), h.size() * sizeof(T)));
  CK(hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
  b.L.base = d;
  CK(hipMalloc(reinterpret_cast<void**>(&b.partials), size_t(8192) * 28 * sizeof(double)));
  CK(hipMalloc(reinterpret_cast<void**>(&b.counter), 4096));
  CK(hipMemset(b.counter, 0, 4096));
  CK(hipMalloc(reinterpret_cast<void**>(&b.out), 32 * sizeof(double)));
  Ndt6Params<T> P{};
  const double c = std::cos(0.02), s = std::sin(0.02);
  const double R[9] = {c, -s, 0, s, c, 0, 0, 0, 1};
  for (int k = 0; k < 9; ++k) P.R[k] = T(R[k]);
  P.t[0] = T(0.01), P.t[1] = T(-0.02), P.t[2] = T(0.03);
  P.la = T(1), P.lb = T(1), P.lc = T(2);
  double ref[28] = {0};
  printf("layout: %s\n", tile_log2 > 0 ? "tiles of 1024" : "planar + skew");
#define RUNN(ITEMS, BLOCK, MINW, PF, BPC) \
  run<PE, T, ITEMS, BLOCK, MINW, true, PF>(b, P, BPC, ref, "ndt6 nt items=" #ITEMS " block=" #BLOCK " minw=" #MINW " pf=" #PF " bpc=" #BPC)
  if constexpr (sizeof(T) == 4) {
    RUNN(2, 512, 2, 1, 1);
    RUNN(2, 512, 2, 3, 1);
    RUNN(2, 512, 2, 2, 1);
    RUNN(2, 512, 2, 0, 1);
    RUNN(2, 256, 2, 2, 2);
    RUNN(1, 1024, 4, 1, 1);
    RUNN(1, 1024, 4, 2, 1);
    RUNN(2, 1024, 4, 1, 1);
    if (tile_log2 <= 0) {
      RUNN(4, 256, 2, 0, 1);
      RUNN(4, 256, 2, 1, 1);
      RUNN(4, 512, 2, 0, 1);
      RUNN(4, 512, 2, 1, 1);
    }
  } else {
    RUNN(1, 512, 3, 0, 1);
    RUNN(1, 512, 2, 3, 1);
    RUNN(1, 512, 2, 1, 1);
    RUNN(2, 512, 2, 0, 1);
  }
#undef RUNN
  {
    using PS = StreamOnlyProblem<T, 15>;
    typename PS::Params Q{};
    double r2[28] = {0};
    if constexpr (sizeof(T) == 4) {
      run<PS, T, 2, 512, 2, true, 1>(b, Q, 1, r2, "stream-only nt items=2 block=512 pf=1");
      double r3[28] = {0};
      run<PS, T, 2, 512, 2, true, 2>(b, Q, 1, r3, "stream-only nt items=2 block=512 pf=2");
    } else {
      run<PS, T, 1, 512, 3, true, 0>(b, Q, 1, r2, "stream-only nt items=1 block=512");
    }
  }
  CK(hipFree(d));
}

int main(int argc, char** argv) {
  const std::string what = argc > 1 ? argv[1] : "reproj";
  const std::string dtype = argc > 2 ? argv[2] : "f64";
  const size_t n = argc > 3 ? size_t(atoll(argv[3])) : 2000000;
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  printf("device %s, %d CUs; %s %s n=%zu\n", prop.name, prop.multiProcessorCount, what.c_str(), dtype.c_str(), n);
  if (what == "reproj") {
    if (dtype == "f64")
      bench_reproj<double>(n, prop.multiProcessorCount);
    else
      bench_reproj<float>(n, prop.multiProcessorCount);
  } else if (what == "ndt6") {
    if (dtype == "f64") {
      bench_ndt6<double>(n, prop.multiProcessorCount, 0);
    } else {
      bench_ndt6<float>(n, prop.multiProcessorCount, 10);
      bench_ndt6<float>(n, prop.multiProcessorCount, 0);
    }
  }
  return 0;
}
