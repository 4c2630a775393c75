"""Kernel / LM-iteration timings for every BASELINE.json single-GPU config (tuning + DESIGN.md table).

usage: python tools/measure_configs.py  → one JSON line per config on stdout.
"""
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from nonlinear_optimizer_for_slam_amd import Context, NdtDataset, ReprojDataset, _lib, solvers, synth  # noqa: E402

host = synth.host_lib()
ctx = Context((0,))


def lm_iter_ms(fn, k=200, warm=20):
    fn(warm)
    t0 = time.perf_counter()
    fn(k)
    return 1e3 * (time.perf_counter() - t0) / k


def ndt_case(name, n, voxels, dtype, loss):
    planes = synth.ndt_planes(n, voxels)
    ds = NdtDataset.from_planes(ctx, planes, dtype)
    del planes
    R, t = np.eye(3), np.zeros(3)
    k6, tot6 = ds.time_kernel6(R, t, loss, repeats=50)
    k3, tot3 = ds.time_kernel3(np.eye(2), np.zeros(2), loss, repeats=50)
    l = solvers.make_loss(loss)
    pt, pR, rep = np.zeros(3), np.eye(3).reshape(-1).copy(), np.zeros(5)

    def run(k):
        host.nos_host_ndt6_iterate(ds._h, ctypes.byref(l), ctypes.c_int(k), pt.ctypes.data_as(_lib.c_double_p),
                                   pR.ctypes.data_as(_lib.c_double_p), rep.ctypes.data_as(_lib.c_double_p))

    it = lm_iter_ms(run)

    def run_dev(k):
        ds.solve6(np.eye(3), np.zeros(3), loss, max_iterations=k, gradient_tolerance=0.0, parameter_tolerance=0.0)

    it_dev = lm_iter_ms(run_dev)
    b = ds.stream_bytes
    print(json.dumps({"config": name, "n": n, "dtype": dtype, "loss": loss[0] if loss else "none",
                      "ndt6_kernel_ms": k6, "ndt6_kernel_GBps": b / k6 / 1e6, "ndt6_fused_ms": tot6,
                      "ndt3_kernel_ms": k3, "ndt3_kernel_GBps": b / k3 / 1e6,
                      "ndt6_lm_iteration_ms": it, "ndt6_corr_per_s": n / it * 1e3,
                      "ndt6_device_loop_iteration_ms": it_dev, "ndt6_device_loop_corr_per_s": n / it_dev * 1e3}), flush=True)
    ds.close()


def reproj_case(name, n, dtype, loss):
    planes = synth.reproj_planes(n)
    ds = ReprojDataset.from_planes(ctx, planes, dtype)
    del planes
    R, t = np.eye(3), np.zeros(3)
    k, tot = ds.time_kernel(R, t, synth.REPROJ_INTR4, loss, repeats=200)
    l = solvers.make_loss(loss)
    intr = np.array(synth.REPROJ_INTR4)
    pt, pR, rep = np.zeros(3), np.eye(3).reshape(-1).copy(), np.zeros(5)

    def run(kk):
        host.nos_host_reproj_iterate(ds._h, intr.ctypes.data_as(_lib.c_double_p), ctypes.byref(l),
                                     ctypes.c_double(0.03), ctypes.c_int(kk), pt.ctypes.data_as(_lib.c_double_p),
                                     pR.ctypes.data_as(_lib.c_double_p), rep.ctypes.data_as(_lib.c_double_p))

    it = lm_iter_ms(run, 500, 50)

    def run_dev(kk):
        ds.solve(np.eye(3), np.zeros(3), synth.REPROJ_INTR4, loss, max_iterations=kk, gradient_tolerance=0.0,
                 parameter_tolerance=0.0)

    it_dev = lm_iter_ms(run_dev, 500, 50)
    b = ds.stream_bytes
    print(json.dumps({"config": name, "n": n, "dtype": dtype, "loss": loss[0] if loss else "none",
                      "reproj_kernel_ms": k, "reproj_kernel_GBps": b / k / 1e6, "reproj_fused_ms": tot,
                      "reproj_lm_iteration_ms": it, "reproj_corr_per_s": n / it * 1e3,
                      "reproj_device_loop_iteration_ms": it_dev, "reproj_device_loop_corr_per_s": n / it_dev * 1e3}), flush=True)
    ds.close()


EXP = ("exponential", 1.0, 1.0)
ndt_case("configs[0] 100k/5k", 100_000, 5_000, "f64", EXP)
ndt_case("configs[1] 10M/200k", 10_000_000, 200_000, "f64", EXP)
ndt_case("configs[1] 10M/200k fp32 storage", 10_000_000, 200_000, "f32", EXP)
ndt_case("configs[1] 10M/200k no loss", 10_000_000, 200_000, "f64", None)
ndt_case("80M single GPU (configs[3] strong-scaling baseline)", 80_000_000, 200_000, "f64", EXP)
reproj_case("configs[2] 2M huber", 2_000_000, "f64", ("huber", synth.REPROJ_HUBER_THRESHOLD))
reproj_case("configs[2] 2M huber fp32 storage", 2_000_000, "f32", ("huber", synth.REPROJ_HUBER_THRESHOLD))
reproj_case("reference scene size 630", 630, "f64", EXP)
