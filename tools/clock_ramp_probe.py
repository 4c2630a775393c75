"""How the LM-iteration time evolves from a cold start of the process (GPU clock / power ramp): chunks of 100 iterations."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
from nonlinear_optimizer_for_slam_amd import Context, NdtDataset, synth
ctx = Context((0,))
planes = synth.ndt_planes(10_000_000, 200_000)
ds = NdtDataset.from_planes(ctx, planes, "f64")
t_start = time.perf_counter()
for chunk in range(40):
    t0 = time.perf_counter()
    ds.solve6(np.eye(3), np.zeros(3), ("exponential", 1.0, 1.0), max_iterations=100, gradient_tolerance=0.0, parameter_tolerance=0.0)
    t1 = time.perf_counter()
    print("t=%.3f s  chunk %2d: %.4f ms/iter" % (t1 - t_start, chunk, 1e3 * (t1 - t0) / 100), flush=True)
