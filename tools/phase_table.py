"""Per-phase timing table of a cold Solve(), in the spirit of the reference's TimeChecker report
(nonlinear_optimizer/time_checker.cc:44-76: calls / min / max / avg / std / total per named scope).

Phases of MahalanobisDistanceMinimizerHip::Solve on n correspondences:
  ingest_records   AoS records → device dataset (H2D of 304-byte records + unpack kernel)
  ingest_planes    planar host arrays → device dataset (H2D + re-tile)
  lm_iteration     one LM iteration: assemble kernel + in-launch reduce + 224-byte zero-copy readback + host 6x6 step
  kernel_only      the assemble kernel alone (hipEvent pairs)
  solve_40         40 LM iterations on the resident dataset (SolvePrepared)
usage: python tools/phase_table.py [n]
"""
import ctypes
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nonlinear_optimizer_for_slam_amd import Context, NdtDataset, _lib, solvers, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
planes = synth.ndt_planes(n, max(1, n // 50))
rec = np.zeros((n, 38))
rec[:, 0:3] = planes[0:3].T
rec[:, 16:19] = planes[3:6].T
for i in range(3):
    for j in range(3):
        rec[:, 28 + 3 * j + i] = planes[6 + 3 * i + j]
offs = [0, 8, 16, 128, 136, 144] + [224 + 8 * (3 * j + i) for i in range(3) for j in range(3)]
ctx = Context((0,))
host = synth.host_lib()
loss = ("exponential", 1.0, 1.0)
l = solvers.make_loss(loss)
samples = {k: [] for k in ("ingest_records", "ingest_planes", "lm_iteration", "kernel_only", "solve_40")}
for rep in range(6):
    t0 = time.perf_counter(); ds = NdtDataset.from_records(ctx, rec, 304, offs, "f64"); samples["ingest_records"].append(time.perf_counter() - t0); ds.close()
    t0 = time.perf_counter(); ds = NdtDataset.from_planes(ctx, planes, "f64"); samples["ingest_planes"].append(time.perf_counter() - t0)
    k, _ = ds.time_kernel6(np.eye(3), np.zeros(3), loss, repeats=20)
    samples["kernel_only"].append(k * 1e-3)
    pt, pR, r5 = np.zeros(3), np.eye(3).reshape(-1).copy(), np.zeros(5)

    def run(kk):
        host.nos_host_ndt6_iterate(ds._h, ctypes.byref(l), ctypes.c_int(kk), pt.ctypes.data_as(_lib.c_double_p),
                                   pR.ctypes.data_as(_lib.c_double_p), r5.ctypes.data_as(_lib.c_double_p))

    run(5)
    t0 = time.perf_counter(); run(100); samples["lm_iteration"].append((time.perf_counter() - t0) / 100)
    pt[:] = 0; pR[:] = np.eye(3).reshape(-1)
    t0 = time.perf_counter(); run(40); samples["solve_40"].append(time.perf_counter() - t0)
    ds.close()
print("------------ Time Analysis (n = %d correspondences, fp64) ------------" % n)
for name, v in samples.items():
    v = np.array(v[1:]) * 1e3  # drop the first (warm-up) sample
    print("%-16s calls: %d   min: %10.4f [ms]   max: %10.4f [ms]   avg: %10.4f [ms]   std: %8.4f [ms]" % (name, v.size, v.min(), v.max(), v.mean(), v.std()))
