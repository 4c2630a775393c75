"""Record ingestion (304-byte AoS → device dataset): device-side unpack of raw records vs host pack to pinned planes."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
from nonlinear_optimizer_for_slam_amd import Context, NdtDataset, synth, api
ctx = Context((0,))
for n in (100_000, 1_000_000, 10_000_000):
    planes = synth.ndt_planes(n, max(10, n // 50))
    rec = np.zeros((n, 38))
    rec[:, 0:3] = planes[0:3].T
    rec[:, 16:19] = planes[3:6].T
    for i in range(3):
        for j in range(3):
            rec[:, 28 + 3 * j + i] = planes[6 + 3 * i + j]
    offs = [0, 8, 16, 128, 136, 144] + [224 + 8 * (3 * j + i) for i in range(3) for j in range(3)]
    ref = None
    for mode, threads in (("unpack", 0), ("pack", 4), ("pack", 8), ("pack", 16), ("pack", 32), ("auto", 0)):
        os.environ["NOS_INGEST"] = mode
        if threads:
            os.environ["NOS_INGEST_THREADS"] = str(threads)
        else:
            os.environ.pop("NOS_INGEST_THREADS", None)
        best = 1e9
        for _ in range(4):
            t0 = time.perf_counter()
            ds = NdtDataset.from_records(ctx, rec, 304, offs, "f64")
            best = min(best, time.perf_counter() - t0)
            got = ds.accumulate6(np.eye(3), np.zeros(3), ("exponential", 1.0, 1.0))
            ds.close()
        if ref is None:
            ref = got
        assert np.array_equal(got, ref), mode
        print("n=%9d %-6s threads %2d: %.2f ms (%.1f GB/s of records)" % (n, mode, threads, 1e3 * best, rec.nbytes / best / 1e9), flush=True)
