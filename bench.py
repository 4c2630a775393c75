#!/usr/bin/env python3
"""Headline benchmark: Gauss-Newton / LM iterations of the 6-DoF NDT solver on MI355X.

    python bench.py --gpus N --steps K --warmup W

One "step" is one full LM iteration of MahalanobisDistanceMinimizer (6-DoF, fp64, robust
ExponentialLossFunction(1,1)) over the rank's resident correspondences: assemble kernel (residual +
analytic Jacobian + weight + 28-scalar reduction) → final reduce → [N>1: one RCCL all-reduce of the 28
doubles] → 224-byte readback → host damping + 6x6 LDLT + pose update + lambda schedule.  The dataset is
uploaded before the timed region (inputs resident in HBM).

Workload: BASELINE.json configs[1] — 10 M synthetic correspondences over 200 k NDT voxels per GPU
(weak scaling: configs[3] is 8 x 10 M = 80 M over 8 GPUs), generator of SURVEY.md §8d.

Prints ONE JSON line on rank 0.  `value` = correspondences (3-vector residual blocks) processed per
second by the whole job; scalar residuals/s = 3x that; GN iterations/s = steps / elapsed.
"""
import argparse
import ctypes
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0          # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
HBM_MEASURED_COPY_GBPS = 6290.0  # same guide: measured float4 copy
BYTES_PER_CORR = {"f64": 120, "f32": 60}  # 15 planes x sizeof(element), SURVEY.md §8d
N_VOXELS = 200_000
LOSS = ("exponential", 1.0, 1.0)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--points", type=int, default=10_000_000, help="correspondences per GPU")
    ap.add_argument("--dtype", default="f64", choices=["f64", "f32"])
    ap.add_argument("--layout", default="flat", choices=["flat", "indexed"],
                    help="flat = the reference's data model (120 B/corr, the headline); indexed = additive "
                         "voxel-indexed layout (28 B/point fp64), reported separately")
    ap.add_argument("--loop", default="device", choices=["device", "host"],
                    help="device: LM loop resident on the GPU (nos_ndt6_solve, the product default); "
                         "host: the loop on the host around nos_ndt6_accumulate")
    ap.add_argument("--prewarm-ms", type=float, default=400.0,
                    help="untimed GPU activity (the same iteration) before the W warm-up steps: the part needs ≈ 0.1-0.3 s "
                         "of load to reach its steady clocks (tools/clock_ramp_probe.py); reported as prewarm_ms")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=8.0, help="time budget per CPU baseline leg")
    return ap.parse_args()


def cpu_baseline(planes, seconds):
    """The reference's CPU paths restated (oracle/), timed on this host: AVX2+FMA fp32 over all
    cores with the reference's thread partition, and the scalar fp64 class on one core.
    Runs on rank 0 at N=1 only; the oracle is used here as the thing being TIMED AS A BASELINE,
    never as part of the GPU path."""
    from oracle import loader as oracle
    n = planes.shape[1]
    R = np.eye(3)
    t = np.zeros(3)
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    # a one-GPU box shares its host with other jobs: stay within a fair CPU share unless told otherwise
    cores = max(1, min(avail, int(os.environ.get("NOS_BENCH_CPU_THREADS", "32"))))
    p32 = planes.astype(np.float32)
    oracle.avx_ndt6_accumulate(p32[:, :80_000], R, t, LOSS, threads=cores)  # warm the pool / pages
    passes, t0 = 0, time.perf_counter()
    while True:
        oracle.avx_ndt6_accumulate(p32, R, t, LOSS, threads=cores)
        passes += 1
        el = time.perf_counter() - t0
        if el >= seconds and passes >= 2:
            break
    avx = {"value": n * passes / el, "unit": "corr/s", "cores": cores, "kind": "port",
           "sample": "%d full passes over the same %d-correspondence workload, AVX2+FMA fp32 lanes "
                     "(restates ..._analytic_simd_various.cc:1300-1447), %d threads of %d visible cores, reference "
                     "thread partition" % (passes, n, cores, avail)}
    # the same AVX2 path on ONE thread (SURVEY §8d asks for T = all and T = 1)
    passes1, t0 = 0, time.perf_counter()
    while True:
        oracle.avx_ndt6_accumulate(p32, R, t, LOSS, threads=1)
        passes1 += 1
        el1 = time.perf_counter() - t0
        if el1 >= 0.4 * seconds and passes1 >= 2:
            break
    avx["one_thread_value"] = n * passes1 / el1
    try:
        with open("/proc/cpuinfo") as f:
            avx["cpu_model"] = next(l.split(":", 1)[1].strip() for l in f if l.startswith("model name"))
    except Exception:  # noqa: BLE001
        avx["cpu_model"] = "unknown"
    del p32
    ns = min(n, 4_000_000)
    sub = np.ascontiguousarray(planes[:, :ns])
    passes, t0 = 0, time.perf_counter()
    while True:
        oracle.ndt6_accumulate(sub, R, t, LOSS)
        passes += 1
        el = time.perf_counter() - t0
        if el >= seconds and passes >= 2:
            break
    scalar = {"value": ns * passes / el, "unit": "corr/s", "cores": 1, "kind": "port",
              "sample": "%d passes over the first %d correspondences, scalar fp64 "
                        "(restates ..._analytic.cc:12-52)" % (passes, ns)}
    return avx, scalar


class _StdoutToStderr:
    """RCCL prints a version banner on stdout at communicator creation; keep stdout for the one
    JSON line by pointing fd 1 at stderr while communicators are being set up."""

    def __enter__(self):
        sys.stdout.flush()
        self._saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self._saved, 1)
        os.close(self._saved)


def main():
    args = parse_args()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    from nonlinear_optimizer_for_slam_amd import (Context, NdtDataset, NdtIndexedDataset, _lib, distributed, solvers,
                                                  synth)

    # torch is only the launcher-side plumbing for N > 1 (process group, barrier, max-over-ranks); the N = 1 path
    # does not touch it, so the bench does not depend on torch's own view of the GPU.
    torch = None
    dist = None
    force_dist = os.environ.get("NOS_BENCH_FORCE_DIST", "0") == "1"  # exercise the N>1 code path on one GPU
    # NOS_BENCH_SHARED_GPU=1: rehearsal of the N > 1 path on a one-GPU box — every rank uses device 0 and the launcher
    # side runs over gloo (RCCL refuses two ranks on one device); the data path (mailbox exchange) is the real one.
    shared_gpu = os.environ.get("NOS_BENCH_SHARED_GPU", "0") == "1"
    tdev = "cpu" if shared_gpu else "cuda"
    if shared_gpu:
        local_rank = 0
    if world > 1 or force_dist:
        import torch
        import torch.distributed as dist
        if force_dist and world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
        with _StdoutToStderr():
            if shared_gpu:
                dist.init_process_group("gloo", rank=rank, world_size=world)
            else:
                torch.cuda.set_device(local_rank)
                dist.init_process_group("nccl", rank=rank, world_size=world,
                                        device_id=torch.device("cuda", local_rank))
                probe = torch.ones(1, device="cuda")
                dist.all_reduce(probe)  # creates torch's communicator (and its banner) now
                torch.cuda.synchronize()

    n_local = args.points
    blocks_per_rank = (n_local + 65535) // 65536
    planes = synth.ndt_planes(n_local, N_VOXELS, first_block=rank * blocks_per_rank)
    ctx = Context((local_rank,))
    if args.layout == "indexed":
        # same correspondences, stored as {point, voxel id} + voxel table; the generator's voxel of a point is
        # identified by its mean (x coordinate is unique per voxel)
        _, first, inverse = np.unique(planes[3], return_index=True, return_inverse=True)
        ds = NdtIndexedDataset.from_arrays(ctx, planes[0:3], inverse.astype(np.int32)[None, :], planes[3:6, first].T.copy(),
                                           planes[6:15, first].T.copy(), args.dtype, sort_by_voxel=True)
    else:
        ds = NdtDataset.from_planes(ctx, planes, args.dtype)

    # Data-path exchange for N > 1 (the 28 doubles of every iteration), best first; each candidate is brought up and
    # self-tested on every rank and adopted only if ALL ranks succeed (collective MIN vote), else the next one is tried:
    #   "mailbox"     sums exchanged INSIDE the assemble launch through a shared-memory mailbox (nos_ctx_comm_init_shm):
    #                 no extra kernel, no RCCL call, no host step per iteration; the LM loop is device resident
    #   "rccl-native" ncclAllReduce issued by libnos_hip.so on the launch stream + a one-wave step kernel
    #   "torch.distributed"  all_reduce from a Python callback around nos_ndt6_accumulate_async (host loop)
    # NOS_BENCH_COMM = auto (default: probe both, keep the faster) | mailbox | rccl | torch.
    comm_mode = "none"
    comm_probe = {}
    if dist is not None:
        comm_mode = "torch.distributed"
        want = os.environ.get("NOS_BENCH_COMM", "auto")
        candidates = {"auto": ["mailbox", "rccl-native"], "mailbox": ["mailbox"], "rccl": ["rccl-native"],
                      "torch": []}.get(want, [])

        def bring_up(candidate):
            """Collective: init + self-test on every rank, unanimous vote.  → mean µs per all-reduce call, or None."""
            ok, micros = 1.0, 0.0
            try:
                with _StdoutToStderr():
                    if candidate == "mailbox":
                        ctx.comm_init_shm_from_torch()
                    else:
                        ctx.comm_init_from_torch()
                    for k in range(120):  # many rounds: both parities of the double-buffered mailbox, ranks out of step
                        if k == 20:
                            t_probe = time.perf_counter()
                        got = ctx.comm_allreduce([rank + 1.0, 1.0])
                        if abs(got[0] - world * (world + 1) / 2.0) > 1e-12 or abs(got[1] - world) > 1e-12:
                            raise RuntimeError("%s self-test mismatch: %r" % (candidate, got))
                    micros = 1e6 * (time.perf_counter() - t_probe) / 100.0
            except Exception as exc:  # noqa: BLE001
                print("[bench] %s unavailable on rank %d: %s" % (candidate, rank, exc), file=sys.stderr)
                ok = 0.0
            vote = torch.tensor([ok, -micros], dtype=torch.float64, device=tdev)
            dist.all_reduce(vote, op=dist.ReduceOp.MIN)   # all ranks ok, and the slowest rank's time
            if float(vote[0].item()) > 0.5:
                return -float(vote[1].item())
            if ctx.comm_size > 0:
                ctx.comm_destroy()  # came up here but not everywhere: drop it
            return None

        # "auto": bring each candidate up, time 100 all-reduce calls of the 28-double payload's kind on THIS machine
        # (same host-side wrapping for both, so the difference is the exchange itself), keep the faster one — the RCCL form
        # is charged the ≈ 4 µs of its extra step kernel that the probe does not see.  Identical decision on every rank
        # (the votes are all-reduced).
        for candidate in candidates:
            micros = bring_up(candidate)
            if micros is not None:
                comm_probe[candidate] = micros
                ctx.comm_destroy()
        if comm_probe:
            def cost(c):
                return comm_probe[c] + (4.0 if c == "rccl-native" else 0.0)
            best = min(comm_probe, key=cost)
            if bring_up(best) is not None:
                comm_mode = best
    if not (rank == 0 and world == 1 and not args.no_cpu_baseline):
        del planes
        planes = None

    host = synth.host_lib()
    loss = solvers.make_loss(LOSS)
    pose_t = np.zeros(3)
    pose_R = np.eye(3).reshape(-1).copy()
    rep = np.zeros(5)

    if comm_mode != "torch.distributed" and args.loop == "device":
        def iterate(k):
            nonlocal pose_t, pose_R
            # tolerances 0: no convergence exit, exactly k iterations execute
            pose_R, pose_t, r = ds.solve6(pose_R, pose_t, LOSS, max_iterations=k, gradient_tolerance=0.0,
                                          parameter_tolerance=0.0)
            if not r["ok"] or r["iterations"] != k or r["launches"] not in (1, k):  # 1: small --points, one-launch form
                raise RuntimeError("device LM loop failed: %r" % (r,))
            rep[:4] = [r["iterations"], r["printed_cost"], r["last_cost"], r["final_lambda"]]
    elif comm_mode != "torch.distributed":
        def iterate(k):
            ok = host.nos_host_ndt6_iterate(ds._h, ctypes.byref(loss), ctypes.c_int(k),
                                            pose_t.ctypes.data_as(_lib.c_double_p),
                                            pose_R.ctypes.data_as(_lib.c_double_p),
                                            rep.ctypes.data_as(_lib.c_double_p))
            if not ok or int(rep[0]) != k:
                raise RuntimeError("LM loop failed: ok=%s iterations=%s status=%s" % (ok, rep[0], rep[4]))
    else:
        ctx.use_torch_stream()
        out = torch.zeros(28, dtype=torch.float64, device="cuda")

        def local(R, t):
            ds.accumulate6_async(R, t, LOSS, out)
            return out

        asm = distributed.ShardedAssembler(local)
        pose = solvers.Pose()

        def iterate(k):
            nonlocal pose_t, pose_R
            r = distributed.solve_ndt6(asm, solvers.Options(k, 0.0, 0.0), pose)
            if r.iterations != k:
                raise RuntimeError("LM loop ended after %d of %d iterations" % (r.iterations, k))
            rep[:4] = [r.iterations, r.printed_cost, r.last_cost, r.final_lambda]
            pose_t, pose_R = pose.t, pose.R.reshape(-1)

    def fence():
        ctx.synchronize()
        if dist is not None:
            if not shared_gpu:
                torch.cuda.synchronize()
            dist.barrier()

    t_pre = time.perf_counter()
    # a fixed number of iterations (≈ 0.2 ms each at the default size), identical on every rank: the exchange inside
    # the iterations is collective, a time-based loop would let ranks disagree on the count
    for _ in range(int(args.prewarm_ms / 0.2) // 50):
        iterate(50)
    fence()
    prewarm_ms = 1e3 * (time.perf_counter() - t_pre)
    pose_t[:] = 0.0                       # the warm-up and the timed steps start from the initial pose again
    pose_R[:] = np.eye(3).reshape(-1)
    if args.warmup > 0:
        iterate(args.warmup)
    fence()
    # Kernel duration for the roofline, from HIP events on the launch stream over the timed region.  Device-resident
    # loop at N = 1: the launches form one back-to-back train, so ONE event pair brackets the whole train (an event
    # between two queued kernels would serialise their dispatch) and the duration per launch is train / launches — an
    # upper bound of the kernel's own duration.  Host loop / N > 1: an event pair around every 4th assemble launch.
    bracket = comm_mode in ("none", "mailbox") and args.loop == "device"
    if os.environ.get("NOS_BENCH_NO_EVENTS", "0") != "1":
        ctx.profile_begin(args.steps + 8, sample_every=0 if bracket else 4)
    t0 = time.perf_counter()
    iterate(args.steps)
    fence()
    elapsed = time.perf_counter() - t0
    n_timed, k_mean_ms, k_min_ms, k_max_ms = ctx.profile_end()
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=tdev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    # Second, untimed-for-the-metric pass at N = 1: the same K steps with the loop ON THE HOST (6x6 LDLT, damping and pose
    # update on the CPU around nos_ndt6_accumulate — the arrangement BASELINE.json's north_star describes literally), so
    # both placements of the loop are on record from one run.
    host_loop = None
    if world == 1 and dist is None and args.loop == "device":
        ht, hR, hrep = np.zeros(3), np.eye(3).reshape(-1).copy(), np.zeros(5)

        def iterate_host(k):
            ok = host.nos_host_ndt6_iterate(ds._h, ctypes.byref(loss), ctypes.c_int(k), ht.ctypes.data_as(_lib.c_double_p),
                                            hR.ctypes.data_as(_lib.c_double_p), hrep.ctypes.data_as(_lib.c_double_p))
            if not ok or int(hrep[0]) != k:
                raise RuntimeError("host LM loop failed: ok=%s iterations=%s status=%s" % (ok, hrep[0], hrep[4]))

        iterate_host(max(args.warmup, 1))
        ctx.synchronize()
        th = time.perf_counter()
        iterate_host(args.steps)
        ctx.synchronize()
        eh = time.perf_counter() - th
        host_loop = {"ms_per_step": 1e3 * eh / args.steps, "value": n_local * args.steps / eh, "unit": "corr/s",
                     "final_translation_error_m": float(np.max(np.abs(ht - synth.true_pose("ndt")[1]))),
                     "note": "same K steps, LM loop on the host around nos_ndt6_accumulate (north_star's literal arrangement)"}

    n_total = n_local * world
    value = n_total * args.steps / elapsed
    bytes_per_launch = ds.stream_bytes  # flat: n * 120 (fp64) / 60 (fp32); indexed: n * (3 * elem + 4)
    achieved = bytes_per_launch / (k_mean_ms * 1e-3) / 1e9 if k_mean_ms > 0 else 0.0
    Rt, tt_true = synth.true_pose("ndt")
    pose_err = float(np.max(np.abs(np.asarray(pose_t) - tt_true)))

    result = {
        "metric": "ndt6_gauss_newton_residual_blocks_per_sec",
        "value": value,
        "unit": "corr/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps,
        "prewarm_ms": prewarm_ms,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": args.dtype,
        "data": "synthetic",
        "config": {
            "workload": "mahalanobis_distance_minimizer 6-DoF %s, %d points / %d NDT voxels per GPU "
                        "(BASELINE.json configs[1]; x%d GPUs = configs[3] shape)%s"
                        % (args.dtype, n_local, N_VOXELS, world,
                           "" if args.layout == "flat" else " — ADDITIVE voxel-indexed layout, not the 120-B/corr headline"),
            "layout": args.layout,
            "points_per_gpu": n_local, "total_points": n_total, "voxels": N_VOXELS,
            "loss": "ExponentialLossFunction(1,1)", "parallelism": "corr-shard x%d, all-reduce 28 f64" % world,
            "collective": comm_mode,
            "collective_probe_us_per_allreduce_call": comm_probe,
            "loop": args.loop if comm_mode != "torch.distributed" else "host",
            "step": ("LM iteration, device resident: assemble kernel + in-launch final reduce%s + 6x6 LDLT / pose update "
                     "/ lambda schedule on the GPU, next launch already queued"
                     % ("" if world == 1 else (" + in-launch mailbox all-reduce(28 f64)" if comm_mode == "mailbox"
                                               else " + RCCL all-reduce(28 f64) + step kernel")))
                    if (args.loop == "device" and comm_mode != "torch.distributed") else
                    ("LM iteration: assemble kernel + final reduce%s + 224 B readback + host 6x6 LDLT/pose update"
                     % (" + RCCL all-reduce(28 f64)" if world > 1 else "")),
        },
        "scalar_residuals_per_sec": 3.0 * value,
        "gn_iters_per_sec": args.steps / elapsed,
        "final_translation_error_m": pose_err,
        "lm_last_cost": float(rep[2]),
        "roofline": {
            "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBPS, "traffic": None,
            "kernel": "nos::assemble_kernel<Ndt6Problem<%s, exponential>>" % ("double" if args.dtype == "f64" else "float"),
            "kernel_ms_mean": k_mean_ms, "kernel_ms_min": k_min_ms, "kernel_ms_max": k_max_ms,
            "launches_timed": n_timed, "algorithmic_bytes_per_launch": bytes_per_launch,
            "bytes_per_corr": bytes_per_launch / max(n_local, 1),
            "frac_of_measured_copy_6290": achieved / HBM_MEASURED_COPY_GBPS,
            "timing": ("one hipEvent pair on the launch stream bracketing the back-to-back train of the timed steps' "
                       "launches; duration per launch = train / launches (upper bound: includes the in-launch reduce, "
                       "the LM step and any gap)") if bracket else
                      "hipEvent pairs on the launch stream around every 4th assemble launch of the timed steps",
        },
    }
    # HBM traffic per launch comes from rocprofv3 PMC passes of this same command (they cannot be
    # collected from inside the process); use the committed summary when it is for this workload.
    try:
        import glob
        for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench_summary.json")), reverse=True):
            prof = json.load(open(path))
            if prof.get("points_per_gpu") == n_local and prof.get("dtype") == args.dtype and args.layout == "flat":
                result["roofline"]["traffic"] = prof["traffic_bytes_per_launch"]
                result["roofline"]["traffic_source"] = (
                    "%s: 2 x FETCH_SIZE + WRITE_SIZE of separate rocprofv3 --pmc passes (gfx950 x2 correction)"
                    % os.path.relpath(path, ROOT))
                break
    except Exception:  # a missing / malformed summary only loses the optional field
        pass
    if host_loop is not None:
        result["host_loop"] = host_loop
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        avx, scalar = cpu_baseline(planes, args.cpu_seconds)
        result["cpu_baseline"] = avx
        result["cpu_baseline_scalar_fp64"] = scalar
        result["gpu_over_cpu_avx"] = value / avx["value"]
    if rank == 0:
        print(json.dumps(result), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
