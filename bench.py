#!/usr/bin/env python3
"""Headline benchmark: Gauss-Newton / LM iterations of the reference's pose solvers on MI355X.

    python bench.py --gpus N --steps K --warmup W [--problem ndt6|ndt3|reproj] [--dtype f64|f32] [--repeats R]

One "step" is one full LM iteration over the rank's resident correspondences: assemble kernel (residual +
analytic Jacobian + robust weight + 28- / 10-scalar reduction) → in-launch final reduce → [N > 1: one all-reduce of the
28 doubles] → damping + 6x6 (3x3) LDLT + pose update + lambda schedule.  The dataset is uploaded before the timed
region (inputs resident in HBM).  W untimed warm-up steps, then R trains of EXACTLY K steps, every train bracketed by
device synchronisation (+ a barrier for N > 1) on both sides, MAX over ranks; `value` comes from the MEDIAN train and
min / median / max are reported beside it.

Problems (BASELINE.json configs):
  ndt6    configs[1]  mahalanobis_distance_minimizer 6-DoF, 10 M correspondences / 200 k voxels per GPU, fp64,
                      ExponentialLossFunction(1,1) — the headline metric; x N GPUs = configs[3] (weak scaling)
  ndt3    same data through the planar (x, y, yaw) solver
  reproj  configs[2]  reprojection_error_minimizer, 2 M 3D<->2D correspondences, HuberLossFunction(1 px)
Synthetic data of SURVEY.md §8d; `value` = correspondences (residual blocks) per second of the whole job.

Prints ONE JSON line on rank 0.  The CPU oracle (oracle/) is imported only inside cpu_baseline(), as the thing
TIMED beside the GPU, never as part of the GPU path.
"""
import argparse
import ctypes
import json
import os
import re
import statistics
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0           # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
HBM_MEASURED_COPY_GBPS = 6290.0  # same guide: measured float4 copy
N_VOXELS = 200_000
EXP = ("exponential", 1.0, 1.0)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--repeats", type=int, default=5, help="trains of --steps timed steps (min / median / max reported)")
    ap.add_argument("--problem", default="ndt6", choices=["ndt6", "ndt3", "reproj"])
    ap.add_argument("--points", type=int, default=0, help="correspondences per GPU (0 = the config's size)")
    ap.add_argument("--dtype", default="f64", choices=["f64", "f32"])
    ap.add_argument("--layout", default="flat", choices=["flat", "indexed"],
                    help="flat = the reference's data model (the headline); indexed = additive voxel-indexed layout "
                         "(ndt6 only), reported separately")
    ap.add_argument("--loop", default="device", choices=["device", "host"],
                    help="device: LM loop resident on the GPU (nos_*_solve, the product default); host: the loop on the "
                         "host around nos_*_accumulate (north_star's literal arrangement)")
    ap.add_argument("--prewarm-ms", type=float, default=400.0,
                    help="untimed GPU activity (the same iteration) before the warm-up steps: the part needs ≈ 0.1-0.3 s "
                         "of load to reach its steady clocks; reported as prewarm_ms")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=1.2,
                    help="time budget of the longest CPU baseline leg of the headline (its thread-count sweep takes ≈ 4.5x this; "
                         "the other configurations and stages get one short leg each: ≤ 10 s of CPU baselines in the default run)")
    ap.add_argument("--no-stages", action="store_true",
                    help="skip the stage lines (pose graph, matcher, map build, ingestion, reference wrappers: bench_stages.py)")
    ap.add_argument("--no-strong-baseline", action="store_true",
                    help="skip the 80 M-correspondence single-GPU pass (denominator of the 8-GPU strong-scaling claim)")
    ap.add_argument("--strong-points", type=int, default=80_000_000)
    ap.add_argument("--no-cold", action="store_true", help="skip the cold (evicted) / warm single-launch measurement")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the short trains of the other single-GPU configurations (other_configs in the JSON line)")
    return ap.parse_args()


# ----------------------------------------------------------------------------------------------------------- workloads

class Workload:
    """One problem of the path: data, device-resident and host loops, algorithmic bytes, CPU baselines."""

    def __init__(self, args, pkg):
        self.args = args
        self.pkg = pkg
        self.dtype = args.dtype
        self.elem = 8 if args.dtype == "f64" else 4
        self.host = pkg["synth"].host_lib()
        self.lossobj = pkg["solvers"].make_loss(self.loss)
        self.pose_t = np.zeros(3)
        self.pose_R = np.eye(3).reshape(-1).copy()
        self.rep = np.zeros(5)

    def reset_pose(self):
        self.pose_t[:] = 0.0
        self.pose_R[:] = np.eye(3).reshape(-1)

    # the device-resident loop: tolerances 0 → exactly k iterations execute
    def _check(self, r, k):
        if not r["ok"] or r["iterations"] != k or r["launches"] not in (1, k):  # 1: one-launch form at small sizes
            raise RuntimeError("device LM loop failed: %r" % (r,))
        self.rep[:4] = [r["iterations"], r["printed_cost"], r["last_cost"], r["final_lambda"]]
        self.launches_of_last_solve = int(r["launches"])

    def _host_check(self, ok, k, rep):
        if not ok or int(rep[0]) != k:
            raise RuntimeError("host LM loop failed: ok=%s iterations=%s status=%s" % (ok, rep[0], rep[4]))


class Ndt6(Workload):
    name, default_points, fields, residual_dim, n_out = "ndt6", 10_000_000, 15, 3, 28
    metric = "ndt6_gauss_newton_residual_blocks_per_sec"
    loss = EXP
    loss_name = "ExponentialLossFunction(1,1)"
    config_ref = "BASELINE.json configs[1]"

    def describe(self, n, world):
        return ("mahalanobis_distance_minimizer 6-DoF %s, %d points / %d NDT voxels per GPU (BASELINE.json configs[1]; "
                "x%d GPUs = configs[3] shape)" % (self.dtype, n, N_VOXELS, world))

    def kernel_name(self):
        return "nos::assemble_kernel<Ndt6Problem<%s, exponential>>" % ("double" if self.dtype == "f64" else "float")

    def planes(self, n, rank=0):
        blocks_per_rank = (n + 65535) // 65536
        return self.pkg["synth"].ndt_planes(n, N_VOXELS, first_block=rank * blocks_per_rank)

    def dataset(self, ctx, planes):
        if self.args.layout == "indexed":
            _, first, inverse = np.unique(planes[3], return_index=True, return_inverse=True)
            return self.pkg["NdtIndexedDataset"].from_arrays(ctx, planes[0:3], inverse.astype(np.int32)[None, :],
                                                             planes[3:6, first].T.copy(), planes[6:15, first].T.copy(),
                                                             self.dtype, sort_by_voxel=True)
        return self.pkg["NdtDataset"].from_planes(ctx, planes, self.dtype)

    def iterate_device(self, ds, k):
        self.pose_R, self.pose_t, r = ds.solve6(self.pose_R, self.pose_t, self.loss, max_iterations=k,
                                                gradient_tolerance=0.0, parameter_tolerance=0.0)
        self._check(r, k)

    def iterate_host(self, ds, k, t=None, R=None, rep=None):
        t = self.pose_t if t is None else t
        R = self.pose_R if R is None else R
        rep = self.rep if rep is None else rep
        dp = self.pkg["_lib"].c_double_p
        ok = self.host.nos_host_ndt6_iterate(ds._h, ctypes.byref(self.lossobj), ctypes.c_int(k), t.ctypes.data_as(dp),
                                             R.ctypes.data_as(dp), rep.ctypes.data_as(dp))
        self._host_check(ok, k, rep)

    def accumulate(self, ds):
        return ds.accumulate6(np.eye(3), np.zeros(3), self.loss)

    def pose_error(self, t=None, R=None):
        return float(np.max(np.abs(np.asarray(self.pose_t if t is None else t) - self.pkg["synth"].true_pose("ndt")[1])))

    def cpu_leg_fp64_avx(self, oracle):
        R, t = np.eye(3), np.zeros(3)
        return lambda p64, threads: oracle.avx_ndt6_accumulate_f64(p64, R, t, self.loss, threads=threads)

    def cpu_legs(self, oracle, planes):
        R, t = np.eye(3), np.zeros(3)
        return (lambda p32, threads: oracle.avx_ndt6_accumulate(p32, R, t, self.loss, threads=threads),
                lambda sub: oracle.ndt6_accumulate(sub, R, t, self.loss),
                "restates MDM/..._analytic_simd_various.cc:1300-1447 (SolveFloatIntrinsicAligned), reference thread partition "
                "MDM/..._analytic_simd.cc:55-76", "restates MDM/..._analytic.cc:12-52")


class Ndt3(Ndt6):
    name, n_out = "ndt3", 10
    metric = "ndt3_gauss_newton_residual_blocks_per_sec"
    cpu_leg_fp64_avx = None  # the reference has no fp64 AVX variant of the planar class

    def describe(self, n, world):
        return ("mahalanobis_distance_minimizer 3-DoF (planar) %s, %d points / %d NDT voxels per GPU — the data of "
                "BASELINE.json configs[1] through MahalanobisDistanceMinimizerAnalytic3DOF's path" % (self.dtype, n, N_VOXELS))

    def kernel_name(self):
        return "nos::assemble_kernel<Ndt3Problem<%s, exponential>>" % ("double" if self.dtype == "f64" else "float")

    def dataset(self, ctx, planes):
        return self.pkg["NdtDataset"].from_planes(ctx, planes, self.dtype)

    def iterate_device(self, ds, k):
        R2 = np.array([self.pose_R[0], self.pose_R[1], self.pose_R[3], self.pose_R[4]])
        R2, t2, r = ds.solve3(R2, self.pose_t[:2].copy(), self.loss, max_iterations=k, gradient_tolerance=0.0,
                              parameter_tolerance=0.0)
        self.pose_R[[0, 1, 3, 4]] = R2
        self.pose_t[:2] = t2
        self._check(r, k)

    def iterate_host(self, ds, k, t=None, R=None, rep=None):
        t = self.pose_t if t is None else t
        R = self.pose_R if R is None else R
        rep = self.rep if rep is None else rep
        dp = self.pkg["_lib"].c_double_p
        ok = self.host.nos_host_ndt3_iterate(ds._h, ctypes.byref(self.lossobj), ctypes.c_int(k), t.ctypes.data_as(dp),
                                             R.ctypes.data_as(dp), rep.ctypes.data_as(dp))
        self._host_check(ok, k, rep)

    def accumulate(self, ds):
        return ds.accumulate3(np.eye(2), np.zeros(2), self.loss)

    def pose_error(self, t=None, R=None):
        # the generator's true pose is a full 6-DoF pose: a planar solver cannot reach it; report the planar part only
        tt = self.pkg["synth"].true_pose("ndt")[1]
        return float(np.max(np.abs(np.asarray(self.pose_t if t is None else t)[:2] - tt[:2])))

    def cpu_legs(self, oracle, planes):
        R2, t2 = np.eye(2), np.zeros(2)
        return (lambda p32, threads: oracle.avx_ndt3_accumulate(p32, R2, t2, self.loss, threads=threads),
                lambda sub: oracle.ndt3_accumulate(sub, R2, t2, self.loss),
                "restates MDM/..._analytic_3dof_simd.cc:85-158 (single-threaded in the reference: the thread fan-out "
                "reuses the 6-DoF partition)", "restates MDM/..._analytic_3dof.cc:36-69,110-139")


class Reproj(Workload):
    name, default_points, fields, residual_dim, n_out = "reproj", 2_000_000, 5, 2, 28
    metric = "reprojection_gauss_newton_residual_blocks_per_sec"
    loss_name = "HuberLossFunction(1 px = 1/525)"
    config_ref = "BASELINE.json configs[2]"

    @property
    def loss(self):
        return ("huber", self.pkg["synth"].REPROJ_HUBER_THRESHOLD)

    def describe(self, n, world):
        return ("reprojection_error_minimizer %s, %d 3D<->2D correspondences, Huber loss (BASELINE.json configs[2])"
                % (self.dtype, n))

    def kernel_name(self):
        return "nos::assemble_kernel<ReprojProblem<%s, huber>>" % ("double" if self.dtype == "f64" else "float")

    def planes(self, n, rank=0):
        return self.pkg["synth"].reproj_planes(n)

    def dataset(self, ctx, planes):
        return self.pkg["ReprojDataset"].from_planes(ctx, planes, self.dtype)

    def iterate_device(self, ds, k):
        self.pose_R, self.pose_t, r = ds.solve(self.pose_R, self.pose_t, self.pkg["synth"].REPROJ_INTR4, self.loss,
                                               max_iterations=k, gradient_tolerance=0.0, parameter_tolerance=0.0)
        self._check(r, k)

    def iterate_host(self, ds, k, t=None, R=None, rep=None):
        t = self.pose_t if t is None else t
        R = self.pose_R if R is None else R
        rep = self.rep if rep is None else rep
        dp = self.pkg["_lib"].c_double_p
        intr = np.array(self.pkg["synth"].REPROJ_INTR4)
        ok = self.host.nos_host_reproj_iterate(ds._h, intr.ctypes.data_as(dp), ctypes.byref(self.lossobj),
                                               ctypes.c_double(0.03), ctypes.c_int(k), t.ctypes.data_as(dp),
                                               R.ctypes.data_as(dp), rep.ctypes.data_as(dp))
        self._host_check(ok, k, rep)

    def accumulate(self, ds):
        return ds.accumulate(np.eye(3), np.zeros(3), self.pkg["synth"].REPROJ_INTR4, self.loss)

    def pose_error(self, t=None, R=None):
        # the solver estimates the INVERSE of the generator's pose: compare -R^T t with the true translation
        R = (self.pose_R if R is None else R).reshape(3, 3)
        tt = np.asarray(self.pose_t if t is None else t)
        return float(np.max(np.abs(-R.T @ tt - self.pkg["synth"].true_pose("reproj")[1])))

    def cpu_legs(self, oracle, planes):
        R, t = np.eye(3), np.zeros(3)
        intr = np.array(self.pkg["synth"].REPROJ_INTR4)
        return (lambda p32, threads: oracle.avx_reproj_accumulate(p32, R, t, intr, self.loss, threads=threads),
                lambda sub: oracle.reproj_accumulate(sub, R, t, intr, self.loss),
                "restates REM/..._analytic_simd.cc:55-138 (single-threaded in the reference: the thread fan-out reuses "
                "the 6-DoF partition)", "restates REM/..._analytic.cc:31-64,107-162")


WORKLOADS = {"ndt6": Ndt6, "ndt3": Ndt3, "reproj": Reproj}


# --------------------------------------------------------------------------------------------------------- CPU baseline

def cpu_baseline(work, planes, seconds):
    """The reference's CPU paths restated (oracle/), timed on this host: AVX2+FMA fp32 lanes over the host's cores with
    the reference's thread partition, the same on one thread, and the scalar fp64 class on one core.  Rank 0 at N = 1
    only; the oracle is the thing being TIMED AS A BASELINE here, never part of the GPU path."""
    from oracle import loader as oracle
    n = planes.shape[1]
    avx_fn, scalar_fn, avx_cite, scalar_cite = work.cpu_legs(oracle, planes)
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    # default: every core this process may use (north_star: "the box's host cores"); NOS_BENCH_CPU_THREADS caps it
    cores = max(1, min(avail, int(os.environ.get("NOS_BENCH_CPU_THREADS", str(avail)))))
    p32 = planes.astype(np.float32)
    avx_fn(p32[:, :80_000], cores)  # warm the pages / threads

    def timed(fn, budget, min_passes=2):
        passes, t0 = 0, time.perf_counter()
        while True:
            fn()
            passes += 1
            el = time.perf_counter() - t0
            if el >= budget and passes >= min_passes:
                return passes, el

    passes, el = timed(lambda: avx_fn(p32, cores), seconds)
    avx = {"value": n * passes / el, "unit": "corr/s", "cores": cores, "kind": "port",
           "sample": "%d full passes over the same %d-correspondence workload, AVX2+FMA fp32 lanes (%s), %d threads of %d "
                     "visible cores" % (passes, n, avx_cite, cores, avail)}
    if cores > 32:  # the same at 32 threads (round 1's figure, comparable across boxes)
        p32n, el32 = timed(lambda: avx_fn(p32, 32), 0.4 * seconds)
        avx["threads32_value"] = n * p32n / el32
    passes1, el1 = timed(lambda: avx_fn(p32, 1), 0.4 * seconds)
    avx["one_thread_value"] = n * passes1 / el1
    # the baseline is the FASTEST of the thread counts tried (small workloads lose time to the thread fan-out)
    avx["all_cores_value"] = avx["value"]
    best = max([(avx["value"], cores), (avx.get("threads32_value", 0.0), 32), (avx["one_thread_value"], 1)])
    if best[1] != cores:
        avx["value"], avx["cores"] = best
        avx["sample"] += "; reported value = the fastest thread count tried (%d)" % best[1]
    try:
        with open("/proc/cpuinfo") as f:
            avx["cpu_model"] = next(l.split(":", 1)[1].strip() for l in f if l.startswith("model name"))
    except Exception:  # noqa: BLE001
        avx["cpu_model"] = "unknown"
    del p32
    # the same-precision baseline of an fp64 GPU line: SolveDouble's 4-lane fp64 loop (6-DoF NDT only)
    avx64 = None
    fp64_fn = work.cpu_leg_fp64_avx(oracle) if getattr(work, "cpu_leg_fp64_avx", None) is not None else None
    if fp64_fn is not None:
        fp64_fn(planes[:, :80_000], cores)
        tried = {}
        for threads, share in ((cores, 0.6), (32, 0.3), (1, 0.3)):
            if threads in tried or threads > cores:
                continue
            pk, ek = timed(lambda: fp64_fn(planes, threads), share * seconds)
            tried[threads] = n * pk / ek
        best_threads = max(tried, key=tried.get)
        avx64 = {"value": tried[best_threads], "unit": "corr/s", "cores": best_threads, "kind": "port",
                 "by_threads": {str(k): v for k, v in sorted(tried.items())},
                 "sample": "full passes over the same %d-correspondence workload, AVX2+FMA 4-lane fp64 (restates the inner loop "
                           "of SolveDouble, MDM/..._analytic_simd_various.cc:42-134, on planar planes — the reference gathers "
                           "its 304-byte records every iteration and runs it on ONE thread; the thread fan-out here is the 6-DoF "
                           "SIMD class's partition on multiples of 4); value = the fastest thread count tried (%d of %d visible "
                           "cores)" % (n, best_threads, avail)}
    ns = min(n, 4_000_000)
    sub = np.ascontiguousarray(planes[:, :ns])
    passes, el = timed(lambda: scalar_fn(sub), seconds)
    scalar = {"value": ns * passes / el, "unit": "corr/s", "cores": 1, "kind": "port",
              "sample": "%d passes over the first %d correspondences, scalar fp64 (%s)" % (passes, ns, scalar_cite)}
    return avx, scalar, avx64


def cpu_baseline_single(work, planes, budget):
    """ONE short CPU leg for the lines of the other configurations: the same-precision AVX restatement at min(32, cores)
    threads (the fastest count of the headline's sweep on every box so far); the thread-count sweep is the headline's."""
    from oracle import loader as oracle
    n = planes.shape[1]
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    threads = max(1, min(32, avail, int(os.environ.get("NOS_BENCH_CPU_THREADS", str(avail)))))
    fp64_fn = work.cpu_leg_fp64_avx(oracle) if (work.dtype == "f64" and getattr(work, "cpu_leg_fp64_avx", None) is not None) else None
    if fp64_fn is not None:
        fn, data, what = (lambda: fp64_fn(planes, threads)), planes, "AVX2+FMA 4-lane fp64 (SolveDouble restated)"
    else:
        avx_fn, _, avx_cite, _ = work.cpu_legs(oracle, planes)
        p32 = planes.astype(np.float32)
        fn, data, what = (lambda: avx_fn(p32, threads)), p32, "AVX2+FMA fp32 lanes (%s)" % avx_cite
    fn()
    passes, t0 = 0, time.perf_counter()
    while passes < 2 or time.perf_counter() - t0 < budget:
        fn()
        passes += 1
    el = time.perf_counter() - t0
    return {"value": n * passes / el, "unit": "corr/s", "cores": threads, "kind": "port",
            "sample": "%d full passes over the same %d correspondences, %s, %d threads" % (passes, n, what, threads)}


class _StdoutToStderr:
    """RCCL prints a version banner on stdout at communicator creation; keep stdout for the one JSON line by pointing
    fd 1 at stderr while communicators are being set up."""

    def __enter__(self):
        sys.stdout.flush()
        self._saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self._saved, 1)
        os.close(self._saved)


def check_runtime(api, need_rccl):
    """One ROCm in the process: the HIP runtime mapped in must be the one libnos_hip.so was built with (major.minor) and
    — when RCCL is used — librccl must come from the same directory.  A mixed process (e.g. torch's bundled runtime under
    a library built with the system hipcc) is refused loudly; NOS_BENCH_ALLOW_SKEW=1 turns the refusal into a warning."""
    info = api.runtime_info()
    problems = []
    if not info["runtime_matches_build"]:
        problems.append("HIP runtime %d (%s) is not the ROCm libnos_hip.so was built with (%d)"
                        % (info["runtime_hip_version"], info["hip_runtime_path"], info["build_hip_version"]))
    if need_rccl and not info["same_rocm_tree"]:
        problems.append("librccl (%s) and the HIP runtime (%s) come from different ROCm trees"
                        % (info["rccl_path"], info["hip_runtime_path"]))
    if problems:
        msg = "[bench] ROCm version skew in this process: " + "; ".join(problems)
        if os.environ.get("NOS_BENCH_ALLOW_SKEW", "0") != "1":
            raise SystemExit(msg + " — refusing to measure (load libnos_hip.so before torch so that the system runtime "
                                   "is the one mapped in, or set NOS_BENCH_ALLOW_SKEW=1)")
        print(msg, file=sys.stderr)
    info["skew"] = bool(problems)
    return info


def dist_free_default(args):
    """True for the default headline run (ndt6 fp64 flat, configs[1] size, device loop): the run that also carries short
    trains of the other single-GPU configurations."""
    return (args.problem == "ndt6" and args.dtype == "f64" and args.layout == "flat" and args.loop == "device"
            and args.points in (0, 10_000_000) and not args.no_other_configs
            and os.environ.get("NOS_BENCH_FORCE_DIST", "0") != "1")


def profile_order(path):
    """Sort key for committed profile summaries: newest round first; within a round the untagged set ("r04_") is the final one."""
    m = re.match(r"r(\d+)([a-z]*)_", os.path.basename(path))
    return (int(m.group(1)), m.group(2) == "", m.group(2)) if m else (0, False, "")


def valu_floor_ms(problem, dtype, n, layout="flat"):
    """VALU-issue floor of one pass: VALU instructions per correspondence from the newest committed SQ counter pass of the
    streaming kernel of this problem (SQ_INSTS_VALU), 4 cycles per wave instruction, 1024 SIMDs, 2.4 GHz.  The
    voxel-indexed layout has its own kernel and its own counter pass (tools/profile_stages.sh, stage `indexed`: 10 M
    points, fp64, one slot)."""
    import glob
    if layout == "indexed":
        for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_indexed_summary.json")), key=profile_order, reverse=True):
            try:
                prof = json.load(open(path))
            except Exception:  # noqa: BLE001
                continue
            for name, k in (prof.get("kernels") or {}).items():
                insts = (k.get("sq") or {}).get("SQ_INSTS_VALU")
                if "assemble_indexed_kernel" in name and insts and dtype == "f64":
                    ipc = insts * 64.0 / 10_000_000
                    return ipc * n / 64.0 * 4.0 / (1024.0 * 2.4e9) * 1e3, ipc, os.path.relpath(path, ROOT)
        return None, None, None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench_*summary*.json")), key=profile_order, reverse=True):
        try:
            prof = json.load(open(path))
        except Exception:  # noqa: BLE001
            continue
        ipc = (prof.get("sq_derived") or {}).get("valu_instructions_per_correspondence")
        if ipc and prof.get("dtype") == dtype and prof.get("problem", "ndt6") == problem and "solve_cluster" not in prof.get("kernel", ""):
            return ipc * n / 64.0 * 4.0 / (1024.0 * 2.4e9) * 1e3, ipc, os.path.relpath(path, ROOT)
    return None, None, None


def summarize(values):
    return {"min": min(values), "median": statistics.median(values), "max": max(values), "n": len(values)}


def main():
    args = parse_args()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if world > 1 and (args.problem != "ndt6" or args.layout != "flat"):
        raise SystemExit("N > 1 is the sharded 6-DoF NDT configuration (configs[3]); --problem %s runs at N = 1" % args.problem)
    # libnos_hip.so FIRST: the process then runs on the ROCm the library was built with (the system one) and the library
    # binds the librccl that sits next to that runtime.  torch comes in afterwards and only as launcher-side plumbing
    # (gloo process group for the barrier / max-over-ranks / the 128-byte RCCL id): it never touches the GPU here.
    from nonlinear_optimizer_for_slam_amd import (Context, NdtDataset, NdtIndexedDataset, ReprojDataset, _lib, api,
                                                  distributed, solvers, synth)
    pkg = {"Context": Context, "NdtDataset": NdtDataset, "NdtIndexedDataset": NdtIndexedDataset,
           "ReprojDataset": ReprojDataset, "_lib": _lib, "api": api, "distributed": distributed, "solvers": solvers,
           "synth": synth}
    shared_gpu = os.environ.get("NOS_BENCH_SHARED_GPU", "0") == "1"  # rehearsal: every rank on device 0 of a one-GPU box
    force_dist = os.environ.get("NOS_BENCH_FORCE_DIST", "0") == "1"  # exercise the N > 1 code path with one rank
    if shared_gpu:
        local_rank = 0
    ctx = Context((local_rank,))
    multi = world > 1 or force_dist
    runtime = check_runtime(api, need_rccl=multi and not shared_gpu)

    dist = torch = None
    if multi:
        import torch
        import torch.distributed as dist
        if force_dist and world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
        with _StdoutToStderr():
            dist.init_process_group("gloo", rank=rank, world_size=world)
        runtime_after = api.runtime_info()
        if runtime_after["hip_runtime_path"] != runtime["hip_runtime_path"]:
            raise SystemExit("[bench] importing torch changed the HIP runtime of the process: %r" % (runtime_after,))

    work = WORKLOADS[args.problem](args, pkg)
    n_local = args.points if args.points > 0 else work.default_points
    planes = work.planes(n_local, rank)
    ds = work.dataset(ctx, planes)
    other_wanted = (world == 1 and dist_free_default(args))
    keep_planes = rank == 0 and world == 1 and (not args.no_cpu_baseline or other_wanted)
    if not keep_planes:
        del planes
        planes = None

    def fence():
        ctx.synchronize()
        if dist is not None:
            dist.barrier()

    def max_over_ranks(x):
        if dist is None:
            return x
        tt = torch.tensor([x], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        return float(tt.item())

    def run_trains(iterate, bracket, work=work, steps=args.steps, warmup=args.warmup, repeats=args.repeats,
                   prewarm_target_ms=args.prewarm_ms):
        """prewarm → warm-up → R trains of K steps.  → dict(ms_per_step stats, kernel ms stats, launches)."""
        t_pre = time.perf_counter()
        iterate(50)
        fence()
        est = max_over_ranks(time.perf_counter() - t_pre)  # identical on every rank: the exchange inside is collective,
        rounds = int(min(400, max(0, prewarm_target_ms * 1e-3 / max(est, 1e-6) - 1)))  # so the COUNT must be agreed on
        for _ in range(rounds):
            iterate(50)
        fence()
        prewarm_ms = 1e3 * (time.perf_counter() - t_pre)
        work.reset_pose()
        if warmup > 0:
            iterate(warmup)
        fence()
        ms, kms, launches = [], [], 0
        for _ in range(max(1, repeats)):
            work.reset_pose()  # every train does the same work: K iterations from the initial pose
            fence()
            if os.environ.get("NOS_BENCH_NO_EVENTS", "0") != "1":
                ctx.profile_begin(steps + 8, sample_every=0 if bracket else 4)
            t0 = time.perf_counter()
            iterate(steps)
            fence()
            elapsed = max_over_ranks(time.perf_counter() - t0)
            n_timed, k_mean, _, _ = ctx.profile_end()
            ms.append(1e3 * elapsed / steps)
            if n_timed > 0:
                kms.append(k_mean)
                launches += n_timed
        return {"ms_per_step": summarize(ms), "kernel_ms": summarize(kms) if kms else None, "launches_timed": launches,
                "prewarm_ms": prewarm_ms, "final_translation_error_m": work.pose_error(), "lm_last_cost": float(work.rep[2])}

    legs = {}
    comm_mode = "none"
    comm_details = {}
    if dist is None:
        if args.loop == "device":
            legs["main"] = run_trains(lambda k: work.iterate_device(ds, k), bracket=True)
        else:
            legs["main"] = run_trains(lambda k: work.iterate_host(ds, k), bracket=False)
    else:
        # N > 1: the 28 doubles of every iteration are exchanged by (a) ONE RCCL all-reduce issued by libnos_hip.so on the
        # launch stream (north_star's arrangement, the headline `value`) and (b) the in-launch shared-memory mailbox; BOTH
        # are brought up, self-tested and timed for R trains of K steps, and both results are printed.
        # NOS_BENCH_COMM = rccl | mailbox | torch restricts the set.
        want = os.environ.get("NOS_BENCH_COMM", "both")
        every = ["rccl-native", "mailbox_device_one_launch", "mailbox_device", "mailbox"]
        order = {"both": every, "auto": every, "rccl": ["rccl-native"], "mailbox": ["mailbox"], "mailbox_device": ["mailbox_device"],
                 "mailbox_device_one_launch": ["mailbox_device_one_launch"], "torch": []}.get(want, every)
        if shared_gpu:
            order = [c for c in order if c != "rccl-native"]  # RCCL refuses two ranks on one device
            ctx.set_option("lm_cluster_max_blocks", max(1, 256 // world))  # all ranks' one-launch grids resident together

        def bring_up(candidate):
            ok, micros, seen = 1.0, 0.0, 0
            try:
                with _StdoutToStderr():
                    if candidate == "mailbox":
                        ctx.comm_init_shm_from_torch()
                    elif candidate.startswith("mailbox_device"):
                        ctx.comm_init_shm_from_torch(device_memory=True)
                    else:
                        ctx.comm_init_from_torch()
                        seen = ctx.comm_rccl_count
                    for k in range(120):  # both parities of the double-buffered mailbox, ranks out of step
                        if k == 20:
                            t_probe = time.perf_counter()
                        got = ctx.comm_allreduce([rank + 1.0, 1.0])
                        if abs(got[0] - world * (world + 1) / 2.0) > 1e-12 or abs(got[1] - world) > 1e-12:
                            raise RuntimeError("%s self-test mismatch: %r" % (candidate, got))
                    micros = 1e6 * (time.perf_counter() - t_probe) / 100.0
            except Exception as exc:  # noqa: BLE001
                print("[bench] %s unavailable on rank %d: %s" % (candidate, rank, exc), file=sys.stderr)
                ok = 0.0
            vote = torch.tensor([ok, -micros], dtype=torch.float64)
            dist.all_reduce(vote, op=dist.ReduceOp.MIN)  # all ranks ok, and the slowest rank's time
            if float(vote[0].item()) > 0.5:
                return {"allreduce_call_us": -float(vote[1].item()),
                        "ranks_seen": seen if not candidate.startswith("mailbox") else ctx.comm_size}
            if ctx.comm_size > 0:
                ctx.comm_destroy()
            return None

        failed_legs = {}
        for candidate in order:
            up = bring_up(candidate)
            if up is None:
                continue
            it = (lambda k: work.iterate_device(ds, k)) if args.loop == "device" else (lambda k: work.iterate_host(ds, k))
            # the device-memory mailbox keeps the one-launch loop (its exchange is a stage of the in-launch all-reduce);
            # "mailbox_device" times the same transport under the launch-per-iteration loop (lm_cluster = 0), round 3's form
            keep = ctx.get_option("lm_cluster")
            if candidate == "mailbox_device":
                ctx.set_option("lm_cluster", 0)
            # A leg that fails on this rank (a peer that never arrives: every in-launch wait is bounded and ends in an error)
            # must not take the whole line with it: the ranks vote, a leg counts only if it finished on ALL of them, and the
            # next candidate starts from a fresh communicator.
            leg, failure = None, ""
            try:
                leg = run_trains(it, bracket=(candidate.startswith("mailbox") and args.loop == "device"))
            except Exception as exc:  # noqa: BLE001
                failure = "%s: %s" % (type(exc).__name__, exc)
                print("[bench] leg %s failed on rank %d: %s" % (candidate, rank, failure), file=sys.stderr)
            finally:
                ctx.set_option("lm_cluster", keep)
            try:
                ctx.comm_destroy()
            except Exception:  # noqa: BLE001
                pass
            vote = torch.tensor([1.0 if leg is not None else 0.0], dtype=torch.float64)
            dist.all_reduce(vote, op=dist.ReduceOp.MIN)
            if float(vote[0].item()) < 0.5:
                failed_legs[candidate] = failure or "failed on another rank"
                continue
            leg.update(up)
            leg["launches_per_train"] = getattr(work, "launches_of_last_solve", None)
            if candidate == "rccl-native":
                leg["ncclCommCount"] = up["ranks_seen"]
            legs[candidate] = leg
        mailboxes = [c for c in legs if c.startswith("mailbox")]
        if "rccl-native" in legs:
            comm_mode = "rccl-native"
        elif mailboxes:  # RCCL did not come up: the fastest mailbox form measured, not the first in a list
            comm_mode = min(mailboxes, key=lambda c: legs[c]["ms_per_step"]["median"])
        else:
            # last resort: torch.distributed all_reduce (gloo here) from a Python callback around nos_ndt6_accumulate
            comm_mode = "torch.distributed(gloo)"
            out = np.zeros(28)

            def local(R, t):
                out[:] = ds.accumulate6(R, t, work.loss)
                return torch.from_numpy(out)

            asm = distributed.ShardedAssembler(local)
            pose = solvers.Pose()

            def iterate_torch(k):
                r = distributed.solve_ndt6(asm, solvers.Options(k, 0.0, 0.0), pose)
                if r.iterations != k:
                    raise RuntimeError("LM loop ended after %d of %d iterations" % (r.iterations, k))
                work.rep[:4] = [r.iterations, r.printed_cost, r.last_cost, r.final_lambda]
                work.pose_t[:], work.pose_R[:] = pose.t, pose.R.reshape(-1)

            legs[comm_mode] = run_trains(iterate_torch, bracket=False)
        legs["main"] = legs[comm_mode]
        comm_details = {k: {"ms_per_step": v["ms_per_step"], "allreduce_call_us": v.get("allreduce_call_us"),
                            "ranks_seen": v.get("ranks_seen"), "ncclCommCount": v.get("ncclCommCount"),
                            "launches_per_train": v.get("launches_per_train")}
                        for k, v in legs.items() if k != "main"}
        for k, why in failed_legs.items():
            comm_details[k] = {"failed": why}

    main_leg = legs["main"]
    launched_kernel = ctx.last_kernel()  # the instantiation the library chose for the timed trains, by its symbol
    ms_med = main_leg["ms_per_step"]["median"]
    n_total = n_local * world
    value = n_total / (ms_med * 1e-3)
    bytes_per_launch = ds.stream_bytes  # flat: n * fields * sizeof(element); indexed: n * (3 * elem + 4)
    kern = main_leg["kernel_ms"]
    k_med = kern["median"] if kern else 0.0
    achieved = bytes_per_launch / (k_med * 1e-3) / 1e9 if k_med > 0 else 0.0
    device_loop = args.loop == "device" and comm_mode != "torch.distributed(gloo)"
    bracket = dist is None and args.loop == "device" or (comm_mode.startswith("mailbox") and args.loop == "device")

    # ---- second placement of the loop at N = 1 (host loop beside the device loop), one train
    host_loop = None
    if dist is None and args.loop == "device":
        ht, hR, hrep = np.zeros(3), np.eye(3).reshape(-1).copy(), np.zeros(5)
        work.iterate_host(ds, max(args.warmup, 1), ht, hR, hrep)
        ht[:], hR[:] = 0.0, np.eye(3).reshape(-1)
        ctx.synchronize()
        th = time.perf_counter()
        work.iterate_host(ds, args.steps, ht, hR, hrep)
        ctx.synchronize()
        eh = time.perf_counter() - th
        host_loop = {"ms_per_step": 1e3 * eh / args.steps, "value": n_local * args.steps / eh, "unit": "corr/s",
                     "final_translation_error_m": work.pose_error(ht, hR),
                     "note": "same K steps, LM loop on the host around nos_%s_accumulate (north_star's literal arrangement)"
                             % args.problem}

    # ---- cold (evicted from the 256 MiB Infinity Cache) vs warm single launches, for datasets that fit the cache
    cold_warm = None
    if dist is None and not args.no_cold and args.layout == "flat" and bytes_per_launch <= (200 << 20):
        evict_n = 4_500_000  # 540 MB of fp64 planes streamed through the cache between two timed launches
        evictor = NdtDataset.from_planes(ctx, synth.ndt_planes(evict_n, 10_000), "f64")

        def single_launch_ms(evict, reps=24):
            out = []
            for _ in range(reps):
                if evict:
                    evictor.accumulate6(np.eye(3), np.zeros(3), None)
                ctx.profile_begin(4, sample_every=1)
                work.accumulate(ds)
                _, k_mean, _, _ = ctx.profile_end()
                out.append(k_mean)
            return summarize(out[4:])

        warm, cold = single_launch_ms(False), single_launch_ms(True)
        evictor.close()
        cold_warm = {
            "how": "hipEvent pair on the launch stream around ONE assemble launch (launch gap included); cold = a 540 MB "
                   "streaming pass over another dataset between two timed launches (evicts the Infinity Cache), warm = "
                   "launches back to back on the resident data",
            "warm_ms": warm, "cold_ms": cold,
            "warm_GBps": bytes_per_launch / (warm["median"] * 1e-3) / 1e9,
            "cold_GBps": bytes_per_launch / (cold["median"] * 1e-3) / 1e9,
        }

    # ---- the other single-GPU configurations, short trains with the same bracket timing (driver-timed lines instead of
    # builder-run ones): configs[2], the configs[0] shape on the GPU, the fp32 classes' storage, the planar solver
    other_configs = None
    cpu_leg_seconds = 0.0  # wall time of the CPU legs of other_configs (solver configurations here, stages in bench_stages)
    if dist is None and other_wanted and planes is not None:
        other_configs = {}
        specs = [("reproj_f64_2M (BASELINE.json configs[2])", "reproj", "f64", 2_000_000, "flat"),
                 ("ndt6_f64_100k (BASELINE.json configs[0] shape on the GPU)", "ndt6", "f64", 100_000, "flat"),
                 ("ndt6_f32_10M (configs[1] data, fp32 storage = the reference's SIMD classes)", "ndt6", "f32", n_local, "flat"),
                 ("ndt3_f64_10M (configs[1] data through the planar solver)", "ndt3", "f64", n_local, "flat"),
                 ("indexed_10M (configs[1] data in the voxel-indexed layout, one slot, fp64)", "ndt6", "f64", n_local, "indexed")]
        for label, problem, dtype, points, layout in specs:
            a2 = argparse.Namespace(**vars(args))
            a2.problem, a2.dtype, a2.points, a2.layout = problem, dtype, points, layout
            w2 = WORKLOADS[problem](a2, pkg)
            p2 = w2.planes(points, 0) if problem == "reproj" else (planes if points == n_local else np.ascontiguousarray(planes[:, :points]))
            d2 = w2.dataset(ctx, p2)
            k2 = 40 if points >= 1_000_000 else 200
            leg = run_trains(lambda k, w2=w2, d2=d2: w2.iterate_device(d2, k), bracket=True, work=w2, steps=k2, warmup=10,
                             repeats=3, prewarm_target_ms=60.0)
            sym = ctx.last_kernel()
            b2 = d2.stream_bytes
            med = leg["ms_per_step"]["median"]
            floor, ipc, src = valu_floor_ms(problem, dtype, points, layout)
            entry = {"ms_per_step": leg["ms_per_step"], "steps": k2, "trains": 3, "points": points, "dtype": dtype,
                     "bytes_per_step": b2, "frac_hbm": b2 / (med * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                     "value": points / (med * 1e-3), "unit": "corr/s",
                     "valu_floor_ms": floor, "valu_instructions_per_corr": ipc, "valu_source": src,
                     "launches_per_train": getattr(w2, "launches_of_last_solve", None), "kernel": sym,
                     "kernel_ms_bracket": leg["kernel_ms"], "final_translation_error_m": leg["final_translation_error_m"],
                     "timing": "as the headline: trains of K steps bracketed by device synchronisation; median of 3"}
            entry["roofline"] = {"bound": "hbm", "achieved": b2 / (med * 1e-3) / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                                 "frac": entry["frac_hbm"], "traffic": None,
                                 "note": ("resident on chip: no HBM traffic per iteration after the first; the bound is vector-ALU "
                                          "issue + the in-launch all-reduce (valu_floor_ms)") if b2 <= 200e6 and layout == "flat" else
                                         "streamed from HBM every iteration"}
            if layout == "indexed":
                entry["bytes_note"] = ("ADDITIVE voxel-indexed layout: its own bytes (24 B point + 4 B voxel id per point; the 25 MB "
                                       "voxel table is cache resident) — never compared with the 120-byte roofline of the flat "
                                       "layout; the bound is vector-ALU issue (valu_floor_ms)")
            if not args.no_cpu_baseline:
                t_cpu = time.perf_counter()
                entry["cpu_baseline"] = cpu_baseline_single(w2, planes if points == n_local and problem != "reproj" else p2, 0.25)
                cpu_leg_seconds += time.perf_counter() - t_cpu
            other_configs[label] = entry
            d2.close()
            del p2

    # ---- BASELINE.json configs[4] (pose graph) and the stages either side of the hot path: matcher, map build,
    # ingestion, the reference's two published test wrappers — bench_stages.py, same run, same clock
    stages = None
    if other_configs is not None and not args.no_stages:
        import bench_stages
        want_cpu = not args.no_cpu_baseline
        stages = {}
        stages.update(bench_stages.stage_pgo(ctx, pkg, want_cpu))
        other_configs["pgo_1M_poses_4M_constraints (BASELINE.json configs[4])"] = stages.pop(
            "pgo_1M_poses_4M_constraints (BASELINE.json configs[4])")
        stages.update(bench_stages.stage_matcher(ctx, pkg, planes, want_cpu))
        stages.update(bench_stages.stage_ingest(ctx, pkg, planes, want_cpu))
        stages.update(bench_stages.stage_mapbuild(ctx, pkg, want_cpu))
        stages.update(bench_stages.stage_reference_wrappers(ctx, pkg, want_cpu))

    # ---- the denominator of the strong-scaling claim: configs[3]'s 80 M correspondences on ONE GPU
    strong = None
    if (dist is None and not args.no_strong_baseline and args.problem == "ndt6" and args.layout == "flat"
            and args.dtype == "f64" and args.points in (0, 10_000_000)):
        ds.close()
        big_n = args.strong_points
        big = np.empty((15, big_n))
        per = 10_000_000
        for r8 in range((big_n + per - 1) // per):  # the 8 ranks' blocks of configs[3], side by side
            lo, hi = r8 * per, min(big_n, (r8 + 1) * per)
            big[:, lo:hi] = synth.ndt_planes(hi - lo, N_VOXELS, first_block=r8 * ((per + 65535) // 65536))
        big_ds = NdtDataset.from_planes(ctx, big, "f64")
        del big
        steps80 = max(10, args.steps // 8)
        work.reset_pose()
        work.iterate_device(big_ds, 5)
        samples = []
        for _ in range(3):
            work.reset_pose()
            ctx.synchronize()
            t0 = time.perf_counter()
            work.iterate_device(big_ds, steps80)
            ctx.synchronize()
            samples.append(1e3 * (time.perf_counter() - t0) / steps80)
        strong = {"points": big_n, "ms_per_step": statistics.median(samples), "ms_per_step_min": min(samples),
                  "ms_per_step_max": max(samples), "steps": steps80, "trains": 3,
                  "value": big_n / (statistics.median(samples) * 1e-3), "unit": "corr/s",
                  "final_translation_error_m": work.pose_error(),
                  "note": "configs[3]'s 80 M correspondences (the 8 ranks' blocks) resident on ONE GPU, same device loop: "
                          "T1 of the strong-scaling ratio; 8 GPUs at 10 M each take the N = 8 line's ms_per_step"}
        big_ds.close()
        ds = None

    result = {
        "metric": work.metric,
        "value": value,
        "unit": "corr/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": ms_med,
        "ms_per_step_trains": main_leg["ms_per_step"],
        "prewarm_ms": main_leg["prewarm_ms"],
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": args.dtype,
        "data": "synthetic",
        "config": {
            "workload": work.describe(n_local, world) + ("" if args.layout == "flat" else
                                                        " — ADDITIVE voxel-indexed layout, not the 120-B/corr headline"),
            "problem": args.problem, "layout": args.layout,
            "points_per_gpu": n_local, "total_points": n_total,
            "loss": work.loss_name,
            "parallelism": "corr-shard x%d, all-reduce %d f64" % (world, work.n_out),
            "collective": comm_mode,
            "collectives_timed": comm_details,
            "value_is": ("median of %d trains of %d steps" % (main_leg["ms_per_step"]["n"], args.steps))
                        + ("" if world == 1 else "; exchange = %s%s" % (
                            comm_mode, " (north_star: one RCCL all-reduce of the 28 doubles per iteration)"
                            if comm_mode == "rccl-native" else "")),
            "loop": "device" if device_loop else "host",
            # 1 = the whole K-step train ran inside ONE launch (data resident on chip, or streamed from HBM every
            # iteration when it does not fit); K = one launch per iteration
            "launches_per_train": getattr(work, "launches_of_last_solve", None) if device_loop else None,
            "step": ("LM iteration, device resident: pass over the data + grid-wide reduce%s + LDLT / pose update / "
                     "lambda schedule on the GPU (one launch for the whole loop when launches_per_train = 1, else one per "
                     "iteration with the next launch already queued)"
                     % ("" if world == 1 else (" + in-launch mailbox all-reduce" if comm_mode.startswith("mailbox")
                                               else " + RCCL all-reduce(28 f64) + step kernel")))
                    if device_loop else
                    ("LM iteration: assemble kernel + final reduce%s + readback + host LDLT / pose update"
                     % (" + all-reduce(28 f64)" if world > 1 else "")),
        },
        "scalar_residuals_per_sec": work.residual_dim * value,
        "gn_iters_per_sec": 1e3 / ms_med,
        "final_translation_error_m": main_leg["final_translation_error_m"],
        "lm_last_cost": main_leg["lm_last_cost"],
        "runtime": runtime,
        "roofline": {
            "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBPS, "traffic": None,
            "kernel": launched_kernel,  # nos_ctx_last_kernel: the symbol rocprofv3 --kernel-trace lists for these launches
            "kernel_note": ("whole LM loop in ONE launch; kernel_ms = launch duration / iterations"
                            if device_loop and getattr(work, "launches_of_last_solve", 0) == 1 else "one launch per LM iteration"),
            "kernel_ms": kern, "kernel_ms_mean": k_med,
            "launches_timed": main_leg["launches_timed"], "algorithmic_bytes_per_launch": bytes_per_launch,
            "bytes_per_corr": bytes_per_launch / max(n_local, 1),
            "frac_of_measured_copy_6290": achieved / HBM_MEASURED_COPY_GBPS,
            "note": "frac is against the 8000 GB/s spec peak; the guide's 6290 GB/s is a float4 COPY (read + write sharing "
                    "the bus) — a read-only stream with non-temporal loads sustains more (its measured range for streaming "
                    "reads is 6.0-6.8 TB/s); datasets below 256 MiB are Infinity-Cache resident when warm (see cold_warm)",
            "timing": ("one hipEvent pair on the launch stream bracketing each back-to-back train of timed launches; duration "
                       "per launch = train / launches (upper bound: includes the in-launch reduce, the LM step and any gap); "
                       "median over the trains") if bracket else
                      "hipEvent pairs on the launch stream around every 4th assemble launch of the timed steps",
        },
    }
    # HBM traffic per launch comes from rocprofv3 PMC passes of this same command (they cannot be collected from inside
    # the process): taken from the newest committed summary for this problem / size / dtype, with its provenance.
    try:
        import glob
        one_launch = device_loop and getattr(work, "launches_of_last_solve", 0) == 1
        for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench*summary*.json")), key=profile_order, reverse=True):
            prof = json.load(open(path))
            if (prof.get("points_per_gpu") == n_local and prof.get("dtype") == args.dtype and args.layout == "flat"
                    and prof.get("problem", "ndt6") == args.problem
                    and bool(prof.get("one_launch_loop", False)) == bool(one_launch) and "traffic_bytes_per_launch" in prof):
                result["roofline"]["traffic"] = prof["traffic_bytes_per_launch"]
                result["roofline"]["traffic_source"] = (
                    "NOT measured in this run: copied from the committed rocprofv3 profile %s (2 x FETCH_SIZE + WRITE_SIZE of "
                    "separate --pmc passes, gfx950 x2 correction, per LM iteration; profiled commit %s)"
                    % (os.path.relpath(path, ROOT), prof.get("commit", "unrecorded")))
                break
    except Exception:  # a missing / malformed summary only loses the optional field
        pass
    # The other bound: vector-ALU issue.  VALU instructions per correspondence come from the committed SQ counter pass
    # (SQ_INSTS_VALU) of the streaming kernel of this problem; one wave instruction occupies a SIMD for 4 cycles
    # (MI355X_MICROARCH.md), 1024 SIMDs, 2.4 GHz.  Where this floor is close to the measured iteration the HBM fraction above
    # is not the figure of merit (the resident reprojection solve: data on chip, no loads at all).
    try:
        import glob
        for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench_*summary*.json")), key=profile_order, reverse=True):
            prof = json.load(open(path))
            ipc = (prof.get("sq_derived") or {}).get("valu_instructions_per_correspondence")
            if (ipc and prof.get("dtype") == args.dtype and prof.get("problem", "ndt6") == args.problem
                    and "solve_cluster" not in prof.get("kernel", "")):
                floor_ms = ipc * n_local / 64.0 * 4.0 / (1024.0 * 2.4e9) * 1e3
                result["roofline"]["valu"] = {
                    "instructions_per_corr": ipc, "floor_ms": floor_ms,
                    "floor_over_ms_per_step": floor_ms / ms_med,
                    "source": "NOT measured in this run: SQ_INSTS_VALU of %s; 4 cycles per wave instruction, 1024 SIMDs, 2.4 GHz"
                              % os.path.relpath(path, ROOT)}
                break
    except Exception:
        pass
    if host_loop is not None:
        result["host_loop"] = host_loop
    if cold_warm is not None:
        result["cold_warm"] = cold_warm
    if strong is not None:
        result["strong_baseline"] = strong
    if world > 1:
        for name, leg in legs.items():
            if name != "main":
                result[name.replace("-native", "")] = {"ms_per_step": leg["ms_per_step"]["median"],
                                                      "ms_per_step_trains": leg["ms_per_step"],
                                                      "value": n_total / (leg["ms_per_step"]["median"] * 1e-3),
                                                      "ncclCommCount": leg.get("ncclCommCount")}
        result["scaling_note"] = ("weak scaling: every rank keeps %d correspondences; N > 1 numbers exist only where this "
                                  "line was produced on N GPUs" % n_local)
    if other_configs is not None:
        result["other_configs"] = other_configs
    if keep_planes and not args.no_cpu_baseline:
        avx, scalar, avx64 = cpu_baseline(work, planes, args.cpu_seconds)
        result["cpu_baseline"] = avx
        result["cpu_baseline_scalar_fp64"] = scalar
        result["gpu_over_cpu_avx"] = value / avx["value"]
        if avx64 is not None:
            result["cpu_baseline_fp64_avx"] = avx64
        # same-precision ratio: an fp64 GPU line against the fp64 AVX loop, an fp32 one against the fp32 lanes
        same = avx64 if (args.dtype == "f64" and avx64 is not None) else avx
        result["gpu_over_cpu"] = {"value": value / same["value"], "cpu_leg": "cpu_baseline_fp64_avx" if same is avx64 else "cpu_baseline",
                                  "cpu_threads": same["cores"],
                                  "note": "same precision on both sides; a GPU / CPU ratio says nothing about kernel quality, "
                                          "the roofline fraction does"}
    if stages is not None:
        # the stage lines are entries of other_configs like the solver configurations (same run, same clock)
        other_configs.update(stages)
        result["stage_entries_of_other_configs"] = sorted(stages)
        import bench_stages
        result["cpu_seconds_of_the_stage_legs"] = cpu_leg_seconds + bench_stages.CPU_LEG_SECONDS[0]
    # LAST key, <= 1 KB: one [ms, fraction of the stage's bound] pair per line of this run, so that a log tail shows them all
    def sig(x):
        return float("%.4g" % x)
    summary = {"ndt6_f64_10M": [sig(ms_med), sig(result["roofline"]["frac"])]}
    if host_loop is not None:
        summary["host_loop"] = [sig(host_loop["ms_per_step"]), sig(bytes_per_launch / (host_loop["ms_per_step"] * 1e-3) / 1e9 / HBM_PEAK_GBPS)]
    if strong is not None:
        summary["strong_80M"] = [sig(strong["ms_per_step"]), sig(strong["points"] * 120 / (strong["ms_per_step"] * 1e-3) / 1e9 / HBM_PEAK_GBPS)]
    short = {"reproj_f64_2M": "reproj_2M", "ndt6_f64_100k": "ndt6_100k", "ndt6_f32_10M": "ndt6_f32", "ndt3_f64_10M": "ndt3",
             "indexed_10M": "indexed"}
    for label, entry in (other_configs or {}).items():
        key = short.get(label.split(" ")[0])
        if key is not None:
            summary[key] = [sig(entry["ms_per_step"]["median"]), sig(entry["frac_hbm"])]
    short = {"pgo_linearize": "pgo_lin", "pgo_matvec": "pgo_mv", "pgo_pcg_iteration": "pgo_it", "matcher_10M_unsorted": "match_uns",
             "matcher_10M_cell_sorted": "match", "matcher_10M_cell_sorted_ids": "match_ids", "mapbuild_10M_100k_voxels": "map_100k",
             "mapbuild_10M_796k_voxels": "map_796k", "mapbuild_reference_scene_exact": "map_ref", "mapbuild_reference_scene_wave_parallel":
             "map_ref_wave", "ingest_10M_records_raw": "ing_raw", "ingest_10M_records_host_pack": "ing_pack", "ingest_10M_planes": "ing_planes"}
    for label, entry in (stages or {}).items():
        if label in short:
            summary[short[label]] = [sig(entry["ms"]["median"]), sig(entry["roofline"]["frac"])]
    if stages is not None:
        w = stages["reference_wrapper_ndt"]
        summary["wrap_ndt"] = [sig(w["ms"]["median"]), int(w["cost_lines_equal_the_captured_run"])]
        w = stages["reference_wrapper_reproj"]
        summary["wrap_reproj"] = [sig(w["ms"]["median"]), int(w["cost_line_equals_the_captured_run"])]
    summary["_"] = "[ms, frac of the line's bound (hbm | pcie; wrappers: 1 = COST lines equal the captured run)]"
    result["summary"] = summary
    if rank == 0:
        print(json.dumps(result), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
