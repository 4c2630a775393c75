"""LM-iteration time of nos_*_solve: the one-launch loop (data resident on chip, or streamed every iteration beyond the
on-chip capacity; key resident_us_per_iteration in both cases) against one launch per iteration, over problem sizes.

usage: python tools/measure_resident.py  → one JSON line per case."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nonlinear_optimizer_for_slam_amd import Context, NdtDataset, ReprojDataset, synth  # noqa: E402

ctx = Context((0,))
EXP = ("exponential", 1.0, 1.0)
HUB = ("huber", synth.REPROJ_HUBER_THRESHOLD)


def per_iter_us(solve, k=400, reps=5):
    solve(50)
    out = []
    for _ in range(reps):
        ctx.synchronize()
        t0 = time.perf_counter()
        r = solve(k)
        ctx.synchronize()
        out.append(1e6 * (time.perf_counter() - t0) / k)
        assert r[2]["iterations"] == k and r[2]["ok"], r[2]
    return float(np.median(out)), r[2]["launches"]


cases = [("reproj", "f64", n) for n in (131_072, 500_000, 1_000_000, 2_000_000)] + [("reproj", "f32", 2_000_000), ("reproj", "f32", 3_000_000)] + \
        [("ndt6", "f64", n) for n in (100_000, 131_072, 262_144, 500_000, 786_432)] + [("ndt6", "f32", 900_000), ("ndt3", "f64", 500_000)] + \
        [("ndt6", "f64", 1_000_000), ("ndt6", "f64", 4_000_000), ("ndt6", "f32", 4_000_000), ("reproj", "f64", 4_000_000)]  # streamed
for kind, dtype, n in cases:
    if kind == "reproj":
        ds = ReprojDataset.from_planes(ctx, synth.reproj_planes(n), dtype)
        fn = lambda k: ds.solve(np.eye(3), np.zeros(3), synth.REPROJ_INTR4, HUB, max_iterations=k, gradient_tolerance=0.0, parameter_tolerance=0.0)  # noqa: E731
    else:
        ds = NdtDataset.from_planes(ctx, synth.ndt_planes(n, max(1, n // 50)), dtype)
        if kind == "ndt6":
            fn = lambda k: ds.solve6(np.eye(3), np.zeros(3), EXP, max_iterations=k, gradient_tolerance=0.0, parameter_tolerance=0.0)  # noqa: E731
        else:
            fn = lambda k: ds.solve3(np.eye(2), np.zeros(2), EXP, max_iterations=k, gradient_tolerance=0.0, parameter_tolerance=0.0)  # noqa: E731
    res, l1 = per_iter_us(fn)
    with ctx.options(lm_cluster=0):
        per, l2 = per_iter_us(fn)
    cap = {("ndt6", "f64"): 786_432, ("ndt3", "f64"): 786_432, ("ndt6", "f32"): 917_504, ("reproj", "f64"): 2_097_152,
           ("reproj", "f32"): 3_145_728}[(kind, dtype)]
    print(json.dumps({"problem": kind, "dtype": dtype, "n": n, "one_launch_form": "resident on chip" if n <= cap else "streamed",
                      "resident_us_per_iteration": res, "resident_launches": l1,
                      "launch_per_iteration_us": per, "launches": l2, "algorithmic_MB": ds.stream_bytes / 1e6}), flush=True)
    ds.close()
