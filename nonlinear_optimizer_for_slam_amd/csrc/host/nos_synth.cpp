// nos_synth.cpp — seeded synthetic workloads for the bench and the tests.
//
// The reference ships no data files: its tests build scenes in code
// (MDM/tests/simple_optimization_test.cc:170-294, REM/tests/simple_optimization_test.cc:115-160).
// For the BASELINE.json sizes (100 k … 80 M correspondences, 5 k … 200 k NDT voxels) this
// generator produces the same kind of input directly in the 15-plane / 5-plane form:
//   voxel v:  mean ~ U([-50,50]^2 x [-5,5]);  Q_v = rotation of a uniform unit quaternion;
//             eigenvalues l3 ~ U[0.02,0.10], l1,l2 ~ U[1e-4,l3] floored at 0.01*l3
//             (the flooring of …/simple_optimization_test.cc:268-273);
//             S_v = diag(l^-1/2) * Q_v   (same convention as :275-276)
//   point i:  voxel ~ U{0..V-1};  world = mean + Q_v^T (sqrt(l) .* N(0,I));  local = T_true^-1 world
//   T_true:   t = (-0.2, 0.123, 0.3) (test.cc:86), R = Rz(0.1) Ry(-0.03) Rx(0.02)
// Reprojection (REM/tests/…:43-61): z ~ U[2,6], x ~ U[-.5,.5] z, y ~ U[-1/3,1/3] z,
//   fx = fy = 525, cx = 320, cy = 240, pixel = project(T_true^-1 X) + N(0, 0.5^2) px,
//   5 % outliers uniform over 640x480, T_true: t = (-0.1, 0.123, -0.5), Rz(0.1).
// RNG: splitmix64-seeded xoshiro256**, one independent stream per block of 65536 items, so the
// output does not depend on the number of worker threads.
#include <algorithm>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <thread>
#include <vector>

namespace {

struct Rng {
  uint64_t s[4];
  static uint64_t SplitMix(uint64_t* x) {
    uint64_t z = (*x += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
  }
  explicit Rng(uint64_t seed) {
    uint64_t x = seed;
    for (int i = 0; i < 4; ++i) s[i] = SplitMix(&x);
  }
  static uint64_t Rotl(uint64_t v, int k) { return (v << k) | (v >> (64 - k)); }
  uint64_t Next() {
    const uint64_t r = Rotl(s[1] * 5, 7) * 9;
    const uint64_t t = s[1] << 17;
    s[2] ^= s[0];
    s[3] ^= s[1];
    s[1] ^= s[2];
    s[0] ^= s[3];
    s[2] ^= t;
    s[3] = Rotl(s[3], 45);
    return r;
  }
  double Uniform() { return double(Next() >> 11) * (1.0 / 9007199254740992.0); }  // [0,1), 53 bits
  double Uniform(double lo, double hi) { return lo + (hi - lo) * Uniform(); }
  double Normal() {
    double u1 = Uniform();
    if (u1 < 1e-300) u1 = 1e-300;
    const double u2 = Uniform();
    return std::sqrt(-2.0 * std::log(u1)) * std::cos(6.283185307179586476925 * u2);
  }
};

constexpr size_t kBlock = 65536;

uint64_t StreamSeed(uint64_t seed, uint64_t salt, uint64_t block) {
  uint64_t x = seed ^ (salt * 0xD1B54A32D192ED03ull) ^ (block * 0x9E3779B97F4A7C15ull);
  return Rng::SplitMix(&x);
}

void TruePoseNdt(double R[9], double t[3]) {
  const double cz = std::cos(0.1), sz = std::sin(0.1);
  const double cy = std::cos(-0.03), sy = std::sin(-0.03);
  const double cx = std::cos(0.02), sx = std::sin(0.02);
  // Rz * Ry * Rx
  R[0] = cz * cy;
  R[1] = cz * sy * sx - sz * cx;
  R[2] = cz * sy * cx + sz * sx;
  R[3] = sz * cy;
  R[4] = sz * sy * sx + cz * cx;
  R[5] = sz * sy * cx - cz * sx;
  R[6] = -sy;
  R[7] = cy * sx;
  R[8] = cy * cx;
  t[0] = -0.2;
  t[1] = 0.123;
  t[2] = 0.3;
}

void TruePoseReproj(double R[9], double t[3]) {
  const double c = std::cos(0.1), s = std::sin(0.1);
  R[0] = c;
  R[1] = -s;
  R[2] = 0;
  R[3] = s;
  R[4] = c;
  R[5] = 0;
  R[6] = 0;
  R[7] = 0;
  R[8] = 1;
  t[0] = -0.1;
  t[1] = 0.123;
  t[2] = -0.5;
}

struct Voxel {
  double mean[3];
  double Q[9];     // rotation, row-major
  double lam[3];   // eigenvalues
  double S[9];     // diag(lam^-1/2) * Q
};

void MakeVoxels(uint64_t seed, size_t v_count, std::vector<Voxel>* voxels) {
  voxels->resize(v_count);
  const size_t blocks = (v_count + kBlock - 1) / kBlock;
  for (size_t b = 0; b < blocks; ++b) {
    Rng rng(StreamSeed(seed, 1, b));
    const size_t end = std::min(v_count, (b + 1) * kBlock);
    for (size_t v = b * kBlock; v < end; ++v) {
      Voxel& vx = (*voxels)[v];
      vx.mean[0] = rng.Uniform(-50.0, 50.0);
      vx.mean[1] = rng.Uniform(-50.0, 50.0);
      vx.mean[2] = rng.Uniform(-5.0, 5.0);
      double q[4], n2 = 0.0;
      do {
        n2 = 0.0;
        for (int k = 0; k < 4; ++k) {
          q[k] = rng.Normal();
          n2 += q[k] * q[k];
        }
      } while (n2 < 1e-12);
      const double inv = 1.0 / std::sqrt(n2);
      const double w = q[0] * inv, x = q[1] * inv, y = q[2] * inv, z = q[3] * inv;
      vx.Q[0] = 1 - 2 * (y * y + z * z);
      vx.Q[1] = 2 * (x * y - w * z);
      vx.Q[2] = 2 * (x * z + w * y);
      vx.Q[3] = 2 * (x * y + w * z);
      vx.Q[4] = 1 - 2 * (x * x + z * z);
      vx.Q[5] = 2 * (y * z - w * x);
      vx.Q[6] = 2 * (x * z - w * y);
      vx.Q[7] = 2 * (y * z + w * x);
      vx.Q[8] = 1 - 2 * (x * x + y * y);
      const double l3 = rng.Uniform(0.02, 0.10);
      double l1 = rng.Uniform(1e-4, l3), l2 = rng.Uniform(1e-4, l3);
      l1 = std::max(l1, 0.01 * l3);
      l2 = std::max(l2, 0.01 * l3);
      vx.lam[0] = l1;
      vx.lam[1] = l2;
      vx.lam[2] = l3;
      for (int i = 0; i < 3; ++i) {
        const double sc = 1.0 / std::sqrt(vx.lam[i]);
        for (int j = 0; j < 3; ++j) vx.S[3 * i + j] = sc * vx.Q[3 * i + j];
      }
    }
  }
}

template <typename Fn>
void ParallelBlocks(size_t n, int threads, Fn&& fn) {
  const size_t blocks = (n + kBlock - 1) / kBlock;
  if (threads < 1) threads = 1;
  if (size_t(threads) > blocks) threads = int(blocks ? blocks : 1);
  std::vector<std::thread> pool;
  for (int w = 0; w < threads; ++w)
    pool.emplace_back([&, w]() {
      for (size_t b = size_t(w); b < blocks; b += size_t(threads)) fn(b, b * kBlock, std::min(n, (b + 1) * kBlock));
    });
  for (auto& th : pool) th.join();
}

}  // namespace

extern "C" {

// planes: 15 caller-allocated arrays of n doubles (plane order of include/nos.h).
// Returns 0 on success.  threads <= 0 → hardware concurrency.
int nos_synth_ndt_shard(uint64_t seed, size_t n, size_t n_voxels, size_t first_block, double* const planes[15],
                        int threads);

int nos_synth_ndt(uint64_t seed, size_t n, size_t n_voxels, double* const planes[15], int threads) {
  return nos_synth_ndt_shard(seed, n, n_voxels, 0, planes, threads);
}

// Same scene, but the point streams start at RNG block `first_block` (blocks of 65536 points):
// rank r of a sharded run passes first_block = r * ceil(n / 65536) and gets points that are
// disjoint from every other rank's while sharing the voxel map.
int nos_synth_ndt_shard(uint64_t seed, size_t n, size_t n_voxels, size_t first_block, double* const planes[15],
                        int threads) {
  if (!planes || n_voxels == 0) return 1;
  for (int k = 0; k < 15; ++k)
    if (!planes[k] && n > 0) return 1;
  if (threads <= 0) threads = int(std::min(32u, std::max(1u, std::thread::hardware_concurrency())));
  std::vector<Voxel> voxels;
  MakeVoxels(seed, n_voxels, &voxels);
  double Rt[9], tt[3];
  TruePoseNdt(Rt, tt);
  ParallelBlocks(n, threads, [&](size_t b, size_t begin, size_t end) {
    Rng rng(StreamSeed(seed, 2, first_block + b));
    for (size_t i = begin; i < end; ++i) {
      const size_t v = size_t(rng.Uniform() * double(n_voxels)) % n_voxels;
      const Voxel& vx = voxels[v];
      double zn[3], wld[3], d[3];
      for (int k = 0; k < 3; ++k) zn[k] = std::sqrt(vx.lam[k]) * rng.Normal();
      for (int k = 0; k < 3; ++k)  // mean + Q^T zn
        wld[k] = vx.mean[k] + vx.Q[k] * zn[0] + vx.Q[3 + k] * zn[1] + vx.Q[6 + k] * zn[2];
      for (int k = 0; k < 3; ++k) d[k] = wld[k] - tt[k];
      for (int k = 0; k < 3; ++k)  // R_true^T (world - t_true)
        planes[k][i] = Rt[k] * d[0] + Rt[3 + k] * d[1] + Rt[6 + k] * d[2];
      for (int k = 0; k < 3; ++k) planes[3 + k][i] = vx.mean[k];
      for (int k = 0; k < 9; ++k) planes[6 + k][i] = vx.S[k];
    }
  });
  return 0;
}

// planes: 5 caller-allocated arrays of n doubles (X, Y, Z, u, v).
int nos_synth_reproj(uint64_t seed, size_t n, double* const planes[5], int threads) {
  if (!planes) return 1;
  for (int k = 0; k < 5; ++k)
    if (!planes[k] && n > 0) return 1;
  if (threads <= 0) threads = int(std::min(32u, std::max(1u, std::thread::hardware_concurrency())));
  double Rt[9], tt[3];
  TruePoseReproj(Rt, tt);
  const double fx = 525.0, fy = 525.0, cx = 320.0, cy = 240.0;
  ParallelBlocks(n, threads, [&](size_t b, size_t begin, size_t end) {
    Rng rng(StreamSeed(seed, 3, b));
    for (size_t i = begin; i < end; ++i) {
      const double z = rng.Uniform(2.0, 6.0);
      const double X[3] = {rng.Uniform(-0.5, 0.5) * z, rng.Uniform(-1.0 / 3.0, 1.0 / 3.0) * z, z};
      double d[3], q[3];
      for (int k = 0; k < 3; ++k) d[k] = X[k] - tt[k];
      for (int k = 0; k < 3; ++k) q[k] = Rt[k] * d[0] + Rt[3 + k] * d[1] + Rt[6 + k] * d[2];
      double u = fx * q[0] / q[2] + cx + 0.5 * rng.Normal();
      double v = fy * q[1] / q[2] + cy + 0.5 * rng.Normal();
      const double outlier = rng.Uniform();
      const double ou = rng.Uniform(0.0, 640.0), ov = rng.Uniform(0.0, 480.0);
      if (outlier < 0.05) {
        u = ou;
        v = ov;
      }
      planes[0][i] = X[0];
      planes[1][i] = X[1];
      planes[2][i] = X[2];
      planes[3][i] = u;
      planes[4][i] = v;
    }
  });
  return 0;
}

// Synthetic pose graph of the BASELINE.json configs[4] shape: a smooth 3-D trajectory (0.5 m steps, small
// random turns) with odometry constraints i -> i+1 and `extra_per_pose` loop constraints from every pose to
// random poses 2..39 steps ahead; measurements = true relative pose + noise (1 cm / 5 mrad), initial
// estimate = truth + noise (5 cm / 20 mrad), pose 0 exact.  poses [n][7] = px py pz qw qx qy qz.
// ref / qry / meas must hold n - 1 + extra_per_pose * n entries; *n_edges receives the count written.
int nos_synth_pose_graph(uint64_t seed, size_t n, int extra_per_pose, double* poses_true, double* poses_init,
                         int32_t* ref, int32_t* qry, double* meas, size_t* n_edges) {
  if (!poses_true || !poses_init || !ref || !qry || !meas || !n_edges || n < 2 || extra_per_pose < 0) return 1;
  Rng rng(StreamSeed(seed, 7, 0));
  auto qmul = [](const double* a, const double* b, double* o) {
    o[0] = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
    o[1] = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
    o[2] = a[0] * b[2] + a[2] * b[0] + a[3] * b[1] - a[1] * b[3];
    o[3] = a[0] * b[3] + a[3] * b[0] + a[1] * b[2] - a[2] * b[1];
  };
  auto qexp = [](const double* w, double* o) {
    const double th = std::sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
    const double k = th < 1e-9 ? 0.5 : std::sin(0.5 * th) / th;
    o[0] = th < 1e-9 ? 1.0 : std::cos(0.5 * th);
    o[1] = k * w[0];
    o[2] = k * w[1];
    o[3] = k * w[2];
  };
  auto qnorm = [](double* q) {
    const double s = 1.0 / std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    for (int k = 0; k < 4; ++k) q[k] *= s;
  };
  auto rot = [](const double* q, const double* v, double* o, bool transpose) {
    const double w = q[0], x = q[1], y = q[2], z = q[3];
    const double R[9] = {1 - 2 * (y * y + z * z), 2 * (x * y - w * z),     2 * (x * z + w * y),
                         2 * (x * y + w * z),     1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
                         2 * (x * z - w * y),     2 * (y * z + w * x),     1 - 2 * (x * x + y * y)};
    for (int i = 0; i < 3; ++i)
      o[i] = transpose ? R[i] * v[0] + R[3 + i] * v[1] + R[6 + i] * v[2] : R[3 * i] * v[0] + R[3 * i + 1] * v[1] + R[3 * i + 2] * v[2];
  };
  double p[3] = {0, 0, 0}, q[4] = {1, 0, 0, 0};
  for (size_t i = 0; i < n; ++i) {
    double* t = poses_true + 7 * i;
    for (int k = 0; k < 3; ++k) t[k] = p[k];
    for (int k = 0; k < 4; ++k) t[3 + k] = q[k];
    double w[3] = {0.05 * rng.Normal(), 0.05 * rng.Normal(), 0.05 * rng.Normal()}, dq[4], qn[4], step[3];
    qexp(w, dq);
    qmul(q, dq, qn);
    qnorm(qn);
    for (int k = 0; k < 4; ++k) q[k] = qn[k];
    const double fwd[3] = {0.5, 0.0, 0.0};
    rot(q, fwd, step, false);
    for (int k = 0; k < 3; ++k) p[k] += step[k] + 0.02 * rng.Normal();
  }
  size_t m = 0;
  auto add_edge = [&](size_t a, size_t b) {
    const double* ta = poses_true + 7 * a;
    const double* tb = poses_true + 7 * b;
    double d[3] = {tb[0] - ta[0], tb[1] - ta[1], tb[2] - ta[2]}, tm[3];
    rot(ta + 3, d, tm, true);
    const double qa_conj[4] = {ta[3], -ta[4], -ta[5], -ta[6]};
    double rel[4], nz[4], w[3] = {0.005 * rng.Normal(), 0.005 * rng.Normal(), 0.005 * rng.Normal()}, out[4];
    qmul(qa_conj, tb + 3, rel);
    qexp(w, nz);
    qmul(rel, nz, out);
    qnorm(out);
    ref[m] = int32_t(a);
    qry[m] = int32_t(b);
    for (int k = 0; k < 3; ++k) meas[7 * m + k] = tm[k] + 0.01 * rng.Normal();
    for (int k = 0; k < 4; ++k) meas[7 * m + 3 + k] = out[k];
    ++m;
  };
  for (size_t i = 0; i + 1 < n; ++i) add_edge(i, i + 1);
  for (size_t i = 0; i < n; ++i)
    for (int k = 0; k < extra_per_pose; ++k) {
      const size_t j = i + 2 + size_t(rng.Uniform() * 38.0);
      if (j < n) add_edge(i, j);
    }
  for (size_t i = 0; i < n; ++i) {
    const double* t = poses_true + 7 * i;
    double* o = poses_init + 7 * i;
    if (i == 0) {
      for (int k = 0; k < 7; ++k) o[k] = t[k];
      continue;
    }
    for (int k = 0; k < 3; ++k) o[k] = t[k] + 0.05 * rng.Normal();
    double w[3] = {0.02 * rng.Normal(), 0.02 * rng.Normal(), 0.02 * rng.Normal()}, dq[4], qn[4];
    qexp(w, dq);
    qmul(t + 3, dq, qn);
    qnorm(qn);
    for (int k = 0; k < 4; ++k) o[3 + k] = qn[k];
  }
  *n_edges = m;
  return 0;
}

// which: 0 = NDT scene, 1 = reprojection scene.  R row-major.
// The room of the reference's NDT test drivers (MDM/tests/simple_optimization_test.cc:170-204, GenerateGlobalPoints): floor
// and four walls of a 7 x 5 x 2.5 m box sampled at 1 cm, the loop variables accumulated in floating point exactly as there
// (the captured runs under results/ depend on these very doubles).  Returns the number of points (954 605); writes at most
// `capacity` of them as xyz triples.  Synthetic INPUT of tests and bench, like the generators above.
size_t nos_synth_room_points(double* out, size_t capacity) {
  const double width = 5.0, length = 7.0, height = 2.5, step = 0.01;
  size_t n = 0;
  auto push = [&](double a, double b, double c) {
    if (out != nullptr && n < capacity) {
      out[3 * n + 0] = a;
      out[3 * n + 1] = b;
      out[3 * n + 2] = c;
    }
    ++n;
  };
  for (double x = -length / 2.0; x <= length / 2.0; x += step)
    for (double y = -width / 2.0; y <= width / 2.0; y += step) push(x, y, 0.0);
  for (double x = -length / 2.0; x <= length / 2.0; x += step)
    for (double z = 0.0; z <= height; z += step) {
      push(x, -width / 2.0, z);
      push(x, width / 2.0, z);
    }
  for (double y = -width / 2.0; y <= width / 2.0; y += step)
    for (double z = 0.0; z <= height; z += step) {
      push(length / 2.0, y, z);
      push(-length / 2.0, y, z);
    }
  return n;
}

void nos_synth_true_pose(int which, double R[9], double t[3]) {
  if (which == 0)
    TruePoseNdt(R, t);
  else
    TruePoseReproj(R, t);
}

}  // extern "C"
