"""LM-iteration time of the device-resident solve forms at small / mid sizes (dataset resident, tolerances 0)."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
from nonlinear_optimizer_for_slam_amd import Context, NdtDataset, synth
ctx = Context((0,))
EXP = ("exponential", 1.0, 1.0)
for n in (600, 2_900, 9_400, 18_700, 37_700, 75_000, 100_000, 131_000):
    ds = NdtDataset.from_planes(ctx, synth.ndt_planes(n, max(10, n // 40)), "f64")
    row = []
    for label, env in (("default", {}), ("cluster off", {"NOS_LM_CLUSTER": "0", "NOS_LM_SINGLE": "0"})):
        for k, v in env.items():
            os.environ[k] = v
        args = dict(max_iterations=200, gradient_tolerance=0.0, parameter_tolerance=0.0)
        ds.solve6(np.eye(3), np.zeros(3), EXP, **args)
        best = 1e9
        for _ in range(5):
            t0 = time.perf_counter()
            r = ds.solve6(np.eye(3), np.zeros(3), EXP, **args)
            best = min(best, time.perf_counter() - t0)
        row.append("%s %.2f us/iter (%d launches)" % (label, 1e6 * best / 200, r[2]["launches"]))
        for k in env:
            del os.environ[k]
    print("n=%7d: %s" % (n, " | ".join(row)), flush=True)
    ds.close()
