"""ms per LM iteration of the 10 M-correspondence fp64 6-DoF case in its three loop forms (one launch, one launch per
iteration, host loop around nos_ndt6_accumulate) — for same-box A/B of two builds:  NOS_HIP_LIB=<lib> python tools/measure_headline_forms.py"""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
from nonlinear_optimizer_for_slam_amd import Context, NdtDataset, synth
ctx = Context((0,))
EXP = ("exponential", 1.0, 1.0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
for dtype in ("f64", "f32"):
    ds = NdtDataset.from_planes(ctx, synth.ndt_planes(n, n // 50), dtype)
    def solve(k):
        return ds.solve6(np.eye(3), np.zeros(3), EXP, max_iterations=k, gradient_tolerance=0.0, parameter_tolerance=0.0)
    def timed(k=200, reps=5):
        solve(100)
        out = []
        for _ in range(reps):
            ctx.synchronize(); t0 = time.perf_counter(); r = solve(k); ctx.synchronize()
            out.append(1e3 * (time.perf_counter() - t0) / k)
            assert r[2]["iterations"] == k and r[2]["ok"]
        return min(out), float(np.median(out))
    one = timed()
    with ctx.options(lm_cluster=0):
        per = timed()
    R = np.eye(3); t = np.zeros(3)
    ds.accumulate6(R, t, EXP)
    ctx.synchronize(); t0 = time.perf_counter()
    for _ in range(200):
        ds.accumulate6(R, t, EXP)
    acc = 1e3 * (time.perf_counter() - t0) / 200
    print("%s ndt6 %s n=%d  one-launch %.5f / %.5f ms   launch-per-iteration %.5f / %.5f ms   blocking accumulate %.5f ms   [%s]"
          % (os.path.basename(os.environ.get("NOS_HIP_LIB", "libnos_hip.so")), dtype, n, one[0], one[1], per[0], per[1], acc, ctx.last_kernel()[:50]), flush=True)
    ds.close()
