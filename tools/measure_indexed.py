"""Voxel-indexed vs flat layout on the configs[1] workload (10 M points / 200 k voxels)."""
import ctypes, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nonlinear_optimizer_for_slam_amd import Context, NdtDataset, NdtIndexedDataset, _lib, solvers, synth  # noqa: E402

n, v = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000, 200_000
planes = synth.ndt_planes(n, v)
uniq, first, inv = np.unique(planes[3], return_index=True, return_inverse=True)
means, S, idx = planes[3:6, first].T.copy(), planes[6:15, first].T.copy(), inv.astype(np.int32)[None, :]
host = synth.host_lib()
loss = ("exponential", 1.0, 1.0)
l = solvers.make_loss(loss)
for dtype in ("f64", "f32"):
    for sort in (True,):
        for bpc in (1, 2, 3, 4):
            os.environ["NOS_INDEXED_BPC"] = str(bpc)
            ctx = Context((0,))
            t0 = time.perf_counter(); ds = NdtIndexedDataset.from_arrays(ctx, planes[0:3], idx, means, S, dtype, sort); tc = time.perf_counter() - t0
            k, tot = ds.time_kernel6(np.eye(3), np.zeros(3), loss, repeats=30)
            pt, pR, rep = np.zeros(3), np.eye(3).reshape(-1).copy(), np.zeros(5)
            def run(kk):
                host.nos_host_ndt6_iterate(ds._h, ctypes.byref(l), ctypes.c_int(kk), pt.ctypes.data_as(_lib.c_double_p), pR.ctypes.data_as(_lib.c_double_p), rep.ctypes.data_as(_lib.c_double_p))
            run(10); t0 = time.perf_counter(); run(100); it = (time.perf_counter() - t0) / 100
            print("indexed %s sort=%d bpc=%d: create %.0f ms, kernel %.4f ms, fused %.4f ms, LM iteration %.4f ms = %.1f G corr/s, %.0f GB/s of %d B/pt"
                  % (dtype, sort, bpc, 1e3 * tc, k, tot, 1e3 * it, n / it / 1e9, ds.stream_bytes / k / 1e6, ds.stream_bytes // n), flush=True)
            ds.close(); ctx.close()
