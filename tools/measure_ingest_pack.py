"""Host-pack ingestion of 10 M 304-byte records (fp64 and fp32 datasets), best of 5; NOS_HIP_LIB selects the build."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
from nonlinear_optimizer_for_slam_amd import Context, NdtDataset, synth
ctx = Context((0,))
n = 10_000_000
planes = synth.ndt_planes(n, n // 50)
rec = np.zeros((n, 38))
rec[:, 0:3] = planes[0:3].T
rec[:, 16:19] = planes[3:6].T
for i in range(3):
    for j in range(3):
        rec[:, 28 + 3 * j + i] = planes[6 + 3 * i + j]
offs = [0, 8, 16, 128, 136, 144] + [224 + 8 * (3 * j + i) for i in range(3) for j in range(3)]
for dtype in ("f64", "f32"):
    for threads in (8, 16):
        best = 1e9
        with ctx.options(ingest=1, ingest_threads=threads):
            for _ in range(5):
                t0 = time.perf_counter()
                ds = NdtDataset.from_records(ctx, rec, 304, offs, dtype)
                best = min(best, time.perf_counter() - t0)
                ds.close()
        print("host pack %s, %2d threads: %.2f ms (%.1f GB/s of records)" % (dtype, threads, 1e3 * best, rec.nbytes / best / 1e9), flush=True)
