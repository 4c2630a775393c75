"""Does the iteration time depend on where the dataset lands in memory?  Re-create the dataset several times in one
process (with differently sized dummy allocations in between) and time 300 iterations each."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
from nonlinear_optimizer_for_slam_amd import Context, NdtDataset, synth
ctx = Context((0,))
planes = synth.ndt_planes(10_000_000, 200_000)
keep = []
for trial in range(8):
    ds = NdtDataset.from_planes(ctx, planes, "f64")
    args = dict(max_iterations=100, gradient_tolerance=0.0, parameter_tolerance=0.0)
    ds.solve6(np.eye(3), np.zeros(3), ("exponential", 1.0, 1.0), **args)
    t0 = time.perf_counter()
    for _ in range(3):
        ds.solve6(np.eye(3), np.zeros(3), ("exponential", 1.0, 1.0), **args)
    dt = (time.perf_counter() - t0) / 300
    k, _ = ds.time_kernel6(np.eye(3), np.zeros(3), ("exponential", 1.0, 1.0), repeats=50)
    print("trial %d: %.4f ms/iter, back-to-back unfused kernel %.4f ms" % (trial, 1e3 * dt, k), flush=True)
    if trial % 2 == 0:
        keep.append(ds)          # keep it alive: the next dataset lands elsewhere
    else:
        ds.close()
