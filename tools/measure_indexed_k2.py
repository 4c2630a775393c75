"""K = 2 voxel-indexed kernel timing (two voxel slots per point), fp64 and fp32."""
import os, sys
import numpy as np
ROOT = os.getcwd()
sys.path.insert(0, ROOT)
from nonlinear_optimizer_for_slam_amd import Context, NdtIndexedDataset, synth  # noqa: E402
n, v = 5_000_000, 200_000
planes = synth.ndt_planes(n, v)
uniq, first, inv = np.unique(planes[3], return_index=True, return_inverse=True)
means, S = planes[3:6, first].T.copy(), planes[6:15, first].T.copy()
idx = np.stack([inv.astype(np.int32), ((inv + 1) % len(first)).astype(np.int32)])
ctx = Context((0,))
for dtype in ("f64", "f32"):
    ds = NdtIndexedDataset.from_arrays(ctx, planes[0:3], idx, means, S, dtype, True)
    for rep in range(3):
        k, tot = ds.time_kernel6(np.eye(3), np.zeros(3), ("exponential", 1.0, 1.0), repeats=30)
    print("K=2 %s: kernel %.4f ms fused %.4f ms (%d points, 2 slots)" % (dtype, k, tot, n), flush=True)
    ds.close()
