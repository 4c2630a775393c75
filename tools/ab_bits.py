"""Bit fingerprint of the sums and of device-loop solves (A/B aid: run once per build of libnos_hip.so, e.g.
NOS_HIP_LIB=tools/_bin/libnos_hip_bperm.so python tools/ab_bits.py, and diff the two outputs)."""
import hashlib
import os
import sys

sys.path.insert(0, os.getcwd())
import numpy as np  # noqa: E402

from nonlinear_optimizer_for_slam_amd import Context, NdtDataset, ReprojDataset, synth  # noqa: E402


def fp(*arrays):
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(np.asarray(a, dtype=np.float64)).tobytes())
    return h.hexdigest()[:16]


ctx = Context((0,))
loss = ("exponential", 1.0, 1.0)
for dtype in ("f64", "f32"):
    for n in (1, 63, 1000, 100_000, 131_072, 1_000_003):
        ds = NdtDataset.from_planes(ctx, synth.ndt_planes(n, max(1, n // 50)), dtype)
        a6 = ds.accumulate6(np.eye(3), np.array([0.01, -0.02, 0.03]), loss)
        a3 = ds.accumulate3(np.eye(2), np.array([0.01, -0.02]), loss)
        s6 = ds.solve6(np.eye(3), np.zeros(3), loss, max_iterations=12)
        print("ndt", dtype, n, fp(a6), fp(a3), fp(s6[0], s6[1], s6[2]["last_cost"]), s6[2]["iterations"])
        ds.close()
        ds = ReprojDataset.from_planes(ctx, synth.reproj_planes(n), dtype)
        ar = ds.accumulate(np.eye(3), np.array([0.01, -0.02, 0.03]), synth.REPROJ_INTR4, ("huber", synth.REPROJ_HUBER_THRESHOLD))
        sr = ds.solve(np.eye(3), np.zeros(3), synth.REPROJ_INTR4, ("huber", synth.REPROJ_HUBER_THRESHOLD), max_iterations=12)
        print("reproj", dtype, n, fp(ar), fp(sr[0], sr[1], sr[2]["last_cost"]), sr[2]["iterations"])
        ds.close()
ctx.close()
