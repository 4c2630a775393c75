"""Launch-geometry A/B of the launch-per-pass assemble kernel inside the library (needs an all-variants build):
    NOS_HIP_LIB=tools/_bin/libnos_hip_all.so python tools/tune_variants.py
Per (element type, layout, geometry): kernel-only time (hipEvents), kernel + in-launch final reduce, and the LM iteration of
the launch-per-iteration device loop (lm_cluster = 0), 10 M correspondences."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nonlinear_optimizer_for_slam_amd import Context, NdtDataset, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
planes = synth.ndt_planes(n, max(1, n // 50))
EXP = ("exponential", 1.0, 1.0)
cases = [("f64", 0, v) for v in (0, 5, 4, 2)] + [("f32", 10, v) for v in (8, 1, 7)] + [("f32", 0, v) for v in (0, 1, 7)]
for dtype, tile, variant in cases:
    ctx = Context((0,))
    ctx.set_option("tile_log2", tile)
    ds = NdtDataset.from_planes(ctx, planes, dtype)
    ctx.set_launch(0, variant)
    try:
        k, tot = ds.time_kernel6(np.eye(3), np.zeros(3), EXP, repeats=40)
        with ctx.options(lm_cluster=0):
            ds.solve6(np.eye(3), np.zeros(3), EXP, max_iterations=60, gradient_tolerance=0.0, parameter_tolerance=0.0)
            out = []
            for _ in range(3):
                ctx.synchronize()
                t0 = time.perf_counter()
                ds.solve6(np.eye(3), np.zeros(3), EXP, max_iterations=200, gradient_tolerance=0.0, parameter_tolerance=0.0)
                ctx.synchronize()
                out.append(1e3 * (time.perf_counter() - t0) / 200)
        print("%s tile=%2d variant=%2d  kernel %.4f ms (%.2f TB/s)  +final %.4f ms  launch-per-iteration LM loop %.4f ms  [%s]"
              % (dtype, tile, variant, k, ds.stream_bytes / k / 1e9, tot, float(np.median(out)), ctx.last_kernel()[:90]), flush=True)
    except Exception as exc:  # noqa: BLE001
        print("ERR", dtype, tile, variant, exc, flush=True)
    ds.close()
    ctx.close()
