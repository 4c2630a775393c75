"""Seeded synthetic workloads — wraps csrc/host/nos_synth.cpp.

Host-side test/bench input generation only; no GPU and no oracle involved.
"""
import ctypes
import os

import numpy as np

from . import _lib

SEED = 20250912
_host = None


def host_lib():
    global _host
    if _host is None:
        if not os.path.exists(_lib.LIB_HOST):
            raise ImportError("libnos_host.so is missing at %s — run `python __graft_entry__.py`" % _lib.LIB_HOST)
        _lib.hip_lib()  # dependency of libnos_host.so, load it first from its in-tree path
        _host = ctypes.CDLL(_lib.LIB_HOST)
        _host.nos_synth_true_pose.restype = None
        _host.nos_host_sizeof_ndt_correspondence.restype = ctypes.c_size_t
        _host.nos_synth_room_points.restype = ctypes.c_size_t
    return _host


def _out_planes(count, n):
    planes = np.empty((count, n), dtype=np.float64)
    arr = (_lib.c_double_p * count)(*[planes[k].ctypes.data_as(_lib.c_double_p) for k in range(count)])
    return planes, arr


def ndt_planes(n, n_voxels, seed=SEED, threads=0, first_block=0):
    """[15, n] float64: point(3), mean(3), sqrt-information(9, row-major).

    first_block shifts the point RNG streams (blocks of 65536 points) so that ranks of a sharded
    run draw disjoint points over the same voxel map."""
    planes, arr = _out_planes(15, n)
    rc = host_lib().nos_synth_ndt_shard(ctypes.c_uint64(seed), ctypes.c_size_t(n), ctypes.c_size_t(n_voxels),
                                        ctypes.c_size_t(first_block), arr, ctypes.c_int(threads))
    if rc != 0:
        raise RuntimeError("nos_synth_ndt failed: %d" % rc)
    return planes


def reproj_planes(n, seed=SEED, threads=0):
    """[5, n] float64: X, Y, Z, u, v.  Intrinsics: fx = fy = 525, cx = 320, cy = 240."""
    planes, arr = _out_planes(5, n)
    rc = host_lib().nos_synth_reproj(ctypes.c_uint64(seed), ctypes.c_size_t(n), arr, ctypes.c_int(threads))
    if rc != 0:
        raise RuntimeError("nos_synth_reproj failed: %d" % rc)
    return planes


REPROJ_INTRINSICS = (525.0, 525.0, 320.0, 240.0)           # fx, fy, cx, cy
REPROJ_INTR4 = (1.0 / 525.0, 1.0 / 525.0, 320.0, 240.0)    # inv_fx, inv_fy, cx, cy (C ABI order)
REPROJ_HUBER_THRESHOLD = 1.0 / 525.0                        # 1 px in normalised image coordinates


def true_pose(which="ndt"):
    """(R [3,3], t [3]) of the scene's ground truth; reprojection solves for its inverse."""
    R = np.zeros(9)
    t = np.zeros(3)
    host_lib().nos_synth_true_pose(0 if which == "ndt" else 1, R.ctypes.data_as(_lib.c_double_p),
                                   t.ctypes.data_as(_lib.c_double_p))
    return R.reshape(3, 3), t


def pose_graph(n_poses, extra_per_pose=3, seed=SEED):
    """Synthetic pose graph (configs[4] shape).  → dict(true [n,7], init [n,7], ref, qry, meas [m,7], fixed)."""
    cap = n_poses - 1 + extra_per_pose * n_poses
    true = np.zeros((n_poses, 7))
    init = np.zeros((n_poses, 7))
    ref = np.zeros(cap, dtype=np.int32)
    qry = np.zeros(cap, dtype=np.int32)
    meas = np.zeros((cap, 7))
    m = ctypes.c_size_t()
    ip = ctypes.POINTER(ctypes.c_int32)
    rc = host_lib().nos_synth_pose_graph(
        ctypes.c_uint64(seed), ctypes.c_size_t(n_poses), ctypes.c_int(extra_per_pose),
        true.ctypes.data_as(_lib.c_double_p), init.ctypes.data_as(_lib.c_double_p), ref.ctypes.data_as(ip),
        qry.ctypes.data_as(ip), meas.ctypes.data_as(_lib.c_double_p), ctypes.byref(m))
    if rc != 0:
        raise RuntimeError("nos_synth_pose_graph failed: %d" % rc)
    m = m.value
    fixed = np.zeros(n_poses, dtype=np.uint8)
    fixed[0] = 1
    return {"true": true, "init": init, "ref": ref[:m].copy(), "qry": qry[:m].copy(), "meas": meas[:m].copy(),
            "fixed": fixed}


# ---- the reference's own NDT test scene (input of the captured runs under results/)

def room_points():
    """[954605, 3]: GenerateGlobalPoints of the reference's NDT test drivers (MDM/tests/simple_optimization_test.cc:170-204)."""
    lib = host_lib()
    n = lib.nos_synth_room_points(None, ctypes.c_size_t(0))
    out = np.zeros((n, 3))
    lib.nos_synth_room_points(out.ctypes.data_as(_lib.c_double_p), ctypes.c_size_t(n))
    return out


def _harness_voxel_keys(points, inv_res):
    """ComputeVoxelKey of the harness (…test.cc:283-294): zig-zag of the floored coordinates, two Cantor pairings."""
    k = np.floor(points * inv_res).astype(np.int64)
    k = np.where(k >= 0, 2 * k, -2 * k - 1)
    xy = (k[:, 0] + k[:, 1]) * (k[:, 0] + k[:, 1] + 1) // 2 + k[:, 1]
    return ((xy + k[:, 2]) * (xy + k[:, 2] + 1) // 2 + k[:, 2]).astype(np.uint64)


def room_scan(points, voxel_size=0.1, t_true=(-0.2, 0.123, 0.3), yaw_true=0.1):
    """FilterPoints (first point of every voxel of edge voxel_size, in point order; …test.cc:206-234) followed by
    WarpPoints(true_pose^-1) (:85-92): the scan of the reference's "simple" test by default (results/maha_amd64_simple.txt).
    → (local points [n, 3], R_true [3, 3], t_true [3])."""
    keys = _harness_voxel_keys(points, 1.0 / voxel_size)
    _, first = np.unique(keys, return_index=True)
    f = points[np.sort(first)]
    c, s = np.cos(yaw_true), np.sin(yaw_true)
    Rt = np.array([[c, -s, 0.0], [s, c, 0.0], [0.0, 0.0, 1.0]])
    tt = np.asarray(t_true, dtype=np.float64)
    return (Rt.T @ (f - tt).T).T, Rt, tt


def reference_reprojection_scene():
    """The reference's reprojection test scene (REM/tests/simple_optimization_test.cc:43-61,115-160): 30 x 21 planar grid at
    z = 3, loop variables accumulated in floating point as there, exact projections through true_pose^-1.
    → (planes [5, 630], (fx, fy, cx, cy), R_true, t_true)."""
    pts = []
    x = -1.5
    while x <= 1.5:
        y = -1.0
        while y <= 1.0:
            pts.append((x, y, 3.0))
            y += 0.1
        x += 0.1
    pts = np.array(pts)
    fx = fy = 525.0
    cx, cy = 320.0, 240.0
    c, s = np.cos(0.1), np.sin(0.1)
    Rt = np.array([[c, -s, 0.0], [s, c, 0.0], [0.0, 0.0, 1.0]])
    tt = np.array([-0.1, 0.123, -0.5])
    q = (Rt.T @ (pts - tt).T).T
    inv_z = 1.0 / q[:, 2]
    planes = np.stack([pts[:, 0], pts[:, 1], pts[:, 2], fx * q[:, 0] * inv_z + cx, fy * q[:, 1] * inv_z + cy])
    return planes, (fx, fy, cx, cy), Rt, tt
