import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
from nonlinear_optimizer_for_slam_amd import Context, NdtDataset, synth
ctx = Context((0,))
for n in (630, 100_000, 10_000_000):
    planes = synth.ndt_planes(n, max(1, n // 50))
    ds = NdtDataset.from_planes(ctx, planes, "f64")
    print("n =", n, file=sys.stderr, flush=True)
    ds.solve6(np.eye(3), np.zeros(3), ("exponential", 1.0, 1.0), max_iterations=8, gradient_tolerance=0.0, parameter_tolerance=0.0)
    ds.close()
