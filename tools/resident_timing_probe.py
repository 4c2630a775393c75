"""Per-phase time of one LM iteration inside the resident one-launch solve (workgroup 0's view), from the
-DNOS_LM_TIMING build of libnos_hip.so:  NOS_HIP_LIB=tools/_bin/libnos_hip_timing.so python tools/resident_timing_probe.py"""
import os
import sys
sys.path.insert(0, os.getcwd())
import numpy as np
from nonlinear_optimizer_for_slam_amd import Context, NdtDataset, ReprojDataset, synth
ctx = Context((0,))
for n in (3_000, 100_000, 131_072, 500_000):
    ds = NdtDataset.from_planes(ctx, synth.ndt_planes(n, max(1, n // 50)), "f64")
    print("ndt6 f64 n =", n, file=sys.stderr, flush=True)
    for _ in range(2):
        ds.solve6(np.eye(3), np.zeros(3), ("exponential", 1.0, 1.0), max_iterations=200, gradient_tolerance=0.0, parameter_tolerance=0.0)
    ds.close()
for n in (131_072, 2_000_000):
    ds = ReprojDataset.from_planes(ctx, synth.reproj_planes(n), "f64")
    print("reproj f64 n =", n, file=sys.stderr, flush=True)
    for _ in range(2):
        ds.solve(np.eye(3), np.zeros(3), synth.REPROJ_INTR4, ("huber", synth.REPROJ_HUBER_THRESHOLD), max_iterations=200,
                 gradient_tolerance=0.0, parameter_tolerance=0.0)
    ds.close()
