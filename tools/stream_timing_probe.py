"""Phases of ONE launch of the launch-per-iteration loop (device time stamps of a -DNOS_LM_TIMING build):
  NOS_HIP_LIB=tools/_bin/libnos_hip_timing.so python tools/stream_timing_probe.py
The library prints the [lm-timing] lines on stderr; this script only drives the solves."""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
from nonlinear_optimizer_for_slam_amd import Context, NdtDataset, ReprojDataset, synth
ctx = Context((0,))
HUB = ("huber", synth.REPROJ_HUBER_THRESHOLD)
with ctx.options(lm_cluster=0, lm_single=0):
    for kind, n, dt in (("reproj", 2_000_000, "f64"), ("ndt", 100_000, "f64"), ("ndt", 10_000_000, "f64")):
        print("problem =", kind, n, dt, file=sys.stderr, flush=True)
        if kind == "reproj":
            ds = ReprojDataset.from_planes(ctx, synth.reproj_planes(n), dt)
            for _ in range(2):
                ds.solve(np.eye(3), np.zeros(3), synth.REPROJ_INTR4, HUB, max_iterations=8, gradient_tolerance=0.0,
                         parameter_tolerance=0.0)
        else:
            ds = NdtDataset.from_planes(ctx, synth.ndt_planes(n, max(1, n // 50)), dt)
            for _ in range(2):
                ds.solve6(np.eye(3), np.zeros(3), ("exponential", 1.0, 1.0), max_iterations=8, gradient_tolerance=0.0,
                          parameter_tolerance=0.0)
        ds.close()
