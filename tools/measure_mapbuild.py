"""NDT map build on the GPU (nos_ndt_map_build): wall time for the reference-like room scene and a large synthetic cloud."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
from nonlinear_optimizer_for_slam_amd import Context, api
ctx = Context((0,))
rng = np.random.default_rng(3)
for n, extent, res in ((954_605, (20.0, 14.0, 6.0), 1.0), (10_000_000, (100.0, 100.0, 10.0), 1.0), (10_000_000, (100.0, 100.0, 10.0), 0.5)):
    pts = rng.uniform(-0.5, 0.5, size=(n, 3)) * np.array(extent)
    for stats in (True, False):
        for rep in range(3):
            t0 = time.perf_counter()
            m, _stats = api.NdtMap.build(ctx, pts, voxel_resolution=res, search_radius_sq=1.0, return_stats=stats)
            dt = time.perf_counter() - t0
            nv = len(m)
            m.close()
        print("n=%9d extent %s res %.1f: %d valid voxels, build %.1f ms (%.1f M points/s)%s"
              % (n, extent, res, nv, 1e3 * dt, n / dt / 1e6, "" if stats else "  [statistics stay on the device]"), flush=True)
