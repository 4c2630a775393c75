"""Scan-to-map rounds at scale (10 M scan points, 200 k voxels): flat records vs voxel-indexed layout."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
from nonlinear_optimizer_for_slam_amd import Context, api, pipeline, solvers, synth
ctx = Context((0,))
n, v = 10_000_000, 200_000
planes = synth.ndt_planes(n, v)
_, first = np.unique(planes[3], return_index=True)
m = api.NdtMap(ctx, planes[3:6, first].T.copy(), planes[6:15, first].T.copy(), None, 1.0)
pts = planes[0:3].T.copy()
del planes
t0 = time.perf_counter(); sc = api.Scan(ctx, pts, sort_cell=1.0); t_scan = time.perf_counter() - t0
print("scan upload + cell sort: %.1f ms" % (1e3 * t_scan))
opt = solvers.Options()
for indexed in (False, True):
    for rep in range(2):
        t0 = time.perf_counter()
        pose, rounds, outer = pipeline.scan_to_map(ctx, m, sc, options=opt, indexed=indexed)
        dt = time.perf_counter() - t0
    its = sum(r["iterations"] for r in rounds)
    # (the synthetic voxels overlap at random, so nearest-mean matching from 0.4 m away is ambiguous and the rounds run to
    # their iteration caps — this measures throughput; registration accuracy is tested on the room scene)
    print("%-8s: %d rounds, %d LM iterations, %.1f ms total = %.2f ms per round; t = %s"
          % ("indexed" if indexed else "flat", len(rounds), its, 1e3 * dt, 1e3 * dt / len(rounds), np.round(pose.t, 9)), flush=True)
