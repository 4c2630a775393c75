"""Matcher + ingestion throughput on the GPU box (tuning aid)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nonlinear_optimizer_for_slam_amd import Context, NdtDataset, api, synth  # noqa: E402

ctx = Context((0,))
n, v = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000, 200_000
planes = synth.ndt_planes(n, v)
# voxel map = unique (mean, S) of the synthetic scene: regenerate voxels by taking the first hit of each voxel
_, first = np.unique(planes[3], return_index=True)
means = planes[3:6, first].T.copy()
S = planes[6:15, first].T.copy()
pts = planes[0:3].T.copy()
Rt, tt = synth.true_pose("ndt")
t0 = time.perf_counter(); m = api.NdtMap(ctx, means, S, None, 1.0); t1 = time.perf_counter()
sc = api.Scan(ctx, pts); t2 = time.perf_counter()
print("map build+upload %.1f ms (%d voxels), scan upload %.1f ms (%d pts, %.2f GB/s)" % (1e3*(t1-t0), len(m), 1e3*(t2-t1), n, n*24/(t2-t1)/1e9))
for dtype in ("f64", "f32"):
    for rep in range(3):
        t0 = time.perf_counter(); ds, nm = m.match(sc, Rt, tt, 2, dtype); dt = time.perf_counter() - t0
        print("match %s: %.2f ms, %d matches of %d slots, %.1f M points/s" % (dtype, 1e3*dt, nm, len(ds), n/dt/1e6))
        ds.close()
t0 = time.perf_counter(); sc2 = api.Scan(ctx, pts, sort_cell=1.0); t1 = time.perf_counter()
print("scan upload + sort by cell: %.1f ms" % (1e3 * (t1 - t0)))
for rep in range(3):
    t0 = time.perf_counter(); ds, nm = m.match(sc2, Rt, tt, 2, "f64"); dt = time.perf_counter() - t0
    print("match f64 (cell-sorted scan): %.2f ms, %d matches, %.1f M points/s" % (1e3*dt, nm, n/dt/1e6))
    ds.close()
for rep in range(3):
    t0 = time.perf_counter(); ds, nm = m.match_indexed(sc2, Rt, tt, 2, "f64", sort_by_voxel=False); dt = time.perf_counter() - t0
    print("match_indexed f64 (cell-sorted scan, no voxel sort): %.2f ms, %d matches" % (1e3*dt, nm))
    ds.close()
# ingestion paths
for dtype in ("f64",):
    t0 = time.perf_counter(); ds = NdtDataset.from_planes(ctx, planes, dtype); dt = time.perf_counter() - t0
    print("from_planes %s: %.1f ms, %.2f GB/s host->dataset" % (dtype, 1e3*dt, planes.nbytes/dt/1e9)); ds.close()
    rec = np.zeros((n, 38)); rec[:, 0:3] = planes[0:3].T; rec[:, 16:19] = planes[3:6].T
    offs = [0, 8, 16, 128, 136, 144] + [224 + 8*(3*j+i) for i in range(3) for j in range(3)]
    for i in range(3):
        for j in range(3):
            rec[:, 28 + 3*j + i] = planes[6 + 3*i + j]
    t0 = time.perf_counter(); ds = NdtDataset.from_records(ctx, rec, 304, offs, dtype); dt = time.perf_counter() - t0
    print("from_records(304 B AoS) %s: %.1f ms, %.2f GB/s of records, %.1f M corr/s" % (dtype, 1e3*dt, rec.nbytes/dt/1e9, n/dt/1e6)); ds.close()
