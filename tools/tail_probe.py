import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
from nonlinear_optimizer_for_slam_amd import Context, NdtDataset, synth
ctx = Context((0,))
for n in (100_000, 1_000_000, 10_000_000):
    planes = synth.ndt_planes(n, max(1, n // 50))
    ds = NdtDataset.from_planes(ctx, planes, "f64")
    for rep in range(3):
        k, tot = ds.time_kernel6(np.eye(3), np.zeros(3), ("exponential", 1.0, 1.0), repeats=200 if n < 5e6 else 50)
        print("n=%d SC1=%s kernel %.2f us fused %.2f us tail %.2f us" % (n, os.environ.get("NOS_SC1", "1"), 1e3 * k, 1e3 * tot, 1e3 * (tot - k)), flush=True)
    ds.close()
