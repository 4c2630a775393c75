"""Launch-geometry sweep for the assemble kernels on the real GPU (tuning aid, not a test).

usage: python tools/tune_ndt6.py [n] [dtype] — prints kernel ms and achieved GB/s per setting.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from nonlinear_optimizer_for_slam_amd import Context, NdtDataset, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
dtypes = sys.argv[2].split(",") if len(sys.argv) > 2 else ["f64", "f32"]
tiles = [int(x) for x in (sys.argv[3].split(",") if len(sys.argv) > 3 else ["12", "0", "10"])]
planes = synth.ndt_planes(n, max(1, n // 50))
R = np.eye(3)
t = np.zeros(3)
loss = ("exponential", 1.0, 1.0)
for dtype in dtypes:
    for tile in tiles:
        os.environ["NOS_TILE_LOG2"] = str(tile)
        ctx = Context((0,))
        ds = NdtDataset.from_planes(ctx, planes, dtype)
        gb = ds.stream_bytes / 1e9
        for nt in (0, 1):
            ctx.set_option("nt", nt)
            for variant in [int(v) for v in os.environ.get("TUNE_VARIANTS", "0,1,2,3,4,5,6,7,8").split(",")]:
                for bpc in (0, 1, 2, 3, 4):
                    ctx.set_launch(bpc, variant)
                    try:
                        k, tot = ds.time_kernel6(R, t, loss, repeats=int(os.environ.get("TUNE_REPEATS", "10")))
                    except Exception as exc:  # noqa: BLE001
                        print("ERR", dtype, tile, nt, variant, bpc, exc, flush=True)
                        continue
                    print("%s tile=%2d nt=%d variant=%d bpc=%d  kernel %.4f ms  %.0f GB/s  (+final %.4f ms)"
                          % (dtype, tile, nt, variant, bpc, k, gb / (k * 1e-3), tot), flush=True)
        ds.close()
        ctx.close()
