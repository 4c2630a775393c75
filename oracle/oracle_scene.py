"""CPU restatement of the reference's NDT test harness — TEST INFRASTRUCTURE.

Follows nonlinear_optimizer/mahalanobis_distance_minimizer/tests/simple_optimization_test.cc:
  GenerateGlobalPoints :170-204   room 7 x 5 x 2.5 m, 1 cm grid, floor + 4 walls (loop variables
                                  accumulated in floating point exactly as there)
  FilterPoints         :206-222   first point per voxel key
  ComputeVoxelKey      :283-294   Cantor pairing of folded integer voxel coordinates
  UpdateNdtMap         :236-281   count / sum / moment (moment starts at IDENTITY, MDM/types.h:14),
                                  mean, covariance, eigen-decomposition, eigenvalue flooring,
                                  sqrt_information = diag(eigvals^-1/2) * eigenvectors
  MatchPointCloud      :296-342   2 nearest valid voxel means within squared distance 1.0
Known-answer values of the reference's captured runs (results/maha_amd64.txt:1-2): 954605 global
points, 96 voxels; 9356 points after the 0.1 m filter.

Eigen's SelfAdjointEigenSolver is replaced by numpy.linalg.eigh (same ascending eigenvalue order;
eigenvector signs may differ, which changes S = D^-1/2 V for voxels with off-diagonal covariance:
end-to-end NDT numbers are therefore a sanity band, not bit goldens — DESIGN.md §5).
"""
import numpy as np


def generate_global_points():
    width, length, height, step = 5.0, 7.0, 2.5, 0.01

    def frange(a, b):
        out = []
        v = a
        while v <= b:
            out.append(v)
            v += step
        return np.array(out)

    xs = frange(-length / 2.0, length / 2.0)
    ys = frange(-width / 2.0, width / 2.0)
    zs = frange(0.0, height)
    pts = []
    # floor
    X, Y = np.meshgrid(xs, ys, indexing="ij")
    pts.append(np.stack([X.ravel(), Y.ravel(), np.zeros(X.size)], axis=1))
    # left/right wall: for x, for z: (x, y, z), (x, -y, z)
    y = -width / 2.0
    X, Z = np.meshgrid(xs, zs, indexing="ij")
    a = np.stack([X.ravel(), np.full(X.size, y), Z.ravel()], axis=1)
    b = np.stack([X.ravel(), np.full(X.size, -y), Z.ravel()], axis=1)
    pts.append(np.stack([a, b], axis=1).reshape(-1, 3))
    # front/back wall: for y, for z: (-x, y, z), (x, y, z)
    x = -length / 2.0
    Y, Z = np.meshgrid(ys, zs, indexing="ij")
    a = np.stack([np.full(Y.size, -x), Y.ravel(), Z.ravel()], axis=1)
    b = np.stack([np.full(Y.size, x), Y.ravel(), Z.ravel()], axis=1)
    pts.append(np.stack([a, b], axis=1).reshape(-1, 3))
    return np.concatenate(pts, axis=0)


def voxel_keys(points, inv_res):
    k = np.floor(points * inv_res).astype(np.int64)
    k = np.where(k >= 0, 2 * k, -2 * k - 1)
    xk, yk, zk = k[:, 0], k[:, 1], k[:, 2]
    # the reference evaluates (x+y)*(x+y+1)/2 + y in `int`; values stay far below 2^31 here
    xy = (xk + yk) * (xk + yk + 1) // 2 + yk
    return ((xy + zk) * (xy + zk + 1) // 2 + zk).astype(np.uint64)


def filter_points(points, voxel_size):
    keys = voxel_keys(points, 1.0 / voxel_size)
    _, first = np.unique(keys, return_index=True)
    return points[np.sort(first)]


def canonical_signs(U):
    """Eigenvector sign convention of the GPU map build: the first component whose magnitude is within
    1e-6 of the largest is positive."""
    U = U.copy()
    for k in range(U.shape[1]):
        mag = np.abs(U[:, k])
        i = int(np.nonzero(mag >= mag.max() * (1.0 - 1e-6))[0][0])  # first of the (near-)largest components
        if U[i, k] < 0:
            U[:, k] = -U[:, k]
    return U


def canonical_eigenbasis(w, U):
    """Tie rule of the GPU map build: when two eigenvalues coincide (relative 1e-9) the eigenbasis is the
    Householder reflection mapping e_0 (two LARGE eigenvalues tie) or e_2 (two SMALL ones tie) onto the
    distinct eigenvector; all three tie → identity.  Otherwise U is returned unchanged."""
    tol = 1e-9 * abs(w[2])
    tie_hi, tie_lo = abs(w[2] - w[1]) <= tol, abs(w[1] - w[0]) <= tol
    if tie_hi and tie_lo:
        return np.eye(3)
    if not (tie_hi or tie_lo):
        return U
    col = 0 if tie_hi else 2
    n = U[:, col].copy()
    if n[col] > 0:
        n = -n
    hv = -n
    hv[col] += 1.0
    return np.eye(3) - 2.0 * np.outer(hv, hv) / float(hv @ hv)


def build_ndt_map(points, voxel_resolution=1.0, canonical=False, proper_transpose=False):
    """→ dict(means [V,3], sqrt_infos [V,3,3], valid [V], keys [V], cells [V,3], eigvals [V,3]) in
    first-seen voxel order.  canonical=True applies canonical_signs to the eigenvectors."""
    keys = voxel_keys(points, 1.0 / voxel_resolution)
    uniq, first, inv = np.unique(keys, return_index=True, return_inverse=True)
    order = np.argsort(first)  # first-seen order
    rank = np.empty_like(order)
    rank[order] = np.arange(order.size)
    vid = rank[inv]
    V = uniq.size
    count = np.bincount(vid, minlength=V)
    s = np.zeros((V, 3))
    np.add.at(s, vid, points)
    moment = np.tile(np.eye(3), (V, 1, 1))  # NDT::moment starts at Identity (MDM/types.h:14)
    np.add.at(moment, vid, points[:, :, None] * points[:, None, :])
    means = np.zeros((V, 3))
    S = np.tile(np.eye(3), (V, 1, 1))
    valid = np.zeros(V, dtype=bool)
    eigvals = np.zeros((V, 3))
    cells_all = np.floor(points * (1.0 / voxel_resolution)).astype(np.int64)
    cells = np.zeros((V, 3), dtype=np.int64)
    cells[vid] = cells_all
    for v in range(V):
        if count[v] < 5:
            continue
        mean = s[v] / count[v]
        cov = moment[v] / count[v] - np.outer(mean, mean)
        w, U = np.linalg.eigh(cov)
        if canonical:
            U = canonical_eigenbasis(w, canonical_signs(U))
        eigvals[v] = w
        if w[2] < 0.01:
            # the reference `return`s here (harness bug); not reproduced
            continue
        w = w.copy()
        w[0] = max(w[0], w[2] * 0.01)
        w[1] = max(w[1], w[2] * 0.01)
        means[v] = mean
        S[v] = np.diag(1.0 / np.sqrt(w)) @ (U.T if proper_transpose else U)
        valid[v] = True
    return {"means": means, "sqrt_infos": S, "valid": valid, "keys": uniq[order], "count": count, "cells": cells,
            "eigvals": eigvals}


def match_point_cloud(means, sqrt_infos, valid, local_points, R, t, radius_sq=1.0, max_neighbors=2):
    """Brute-force restatement of MatchPointCloud.  → (planes [15, 2n] with zero records for absent
    neighbours — the slot convention of nos_ndt_match —, number of matches, index array [n, 2] (-1 = none))."""
    means = np.asarray(means, dtype=np.float64).reshape(-1, 3)
    S = np.asarray(sqrt_infos, dtype=np.float64).reshape(-1, 9)
    ok = np.ones(means.shape[0], dtype=bool) if valid is None else np.asarray(valid, dtype=bool)
    ids = np.nonzero(ok)[0]
    R = np.asarray(R, dtype=np.float64).reshape(3, 3)
    t = np.asarray(t, dtype=np.float64).reshape(3)
    P = np.asarray(local_points, dtype=np.float64).reshape(-1, 3)
    n = P.shape[0]
    # same operation order as the kernel: R[0]*x + R[1]*y + R[2]*z + t
    q = np.stack([R[r, 0] * P[:, 0] + R[r, 1] * P[:, 1] + R[r, 2] * P[:, 2] + t[r] for r in range(3)], axis=1)
    idx = -np.ones((n, 2), dtype=np.int64)
    chunk = max(1, 4_000_000 // max(1, ids.size))
    for b in range(0, n, chunk):
        qq = q[b:b + chunk]
        ex = qq[:, None, 0] - means[None, ids, 0]
        ey = qq[:, None, 1] - means[None, ids, 1]
        ez = qq[:, None, 2] - means[None, ids, 2]
        d = ex * ex + ey * ey + ez * ez
        d = np.where(d < radius_sq, d, np.inf)
        k = min(2, ids.size)
        # stable sort → ties broken by original voxel id (ids ascending)
        part = np.argsort(d, axis=1, kind="stable")[:, :k]
        dsel = np.take_along_axis(d, part, axis=1)
        sel = np.where(np.isfinite(dsel), ids[part], -1)
        idx[b:b + chunk, :k] = sel
    if max_neighbors < 2:
        idx[:, 1] = -1
    planes = np.zeros((15, 2 * n))
    for k in range(2):
        m = idx[:, k] >= 0
        cols = 2 * np.nonzero(m)[0] + k
        planes[0:3, cols] = P[m].T
        planes[3:6, cols] = means[idx[m, k]].T
        planes[6:15, cols] = S[idx[m, k]].T
    return planes, int((idx >= 0).sum()), idx


def match_point_cloud_kdtree(means, valid, world_points, radius_sq=1.0, max_neighbors=2, workers=1):
    """MatchPointCloud's search as the reference runs it (MDM/tests/simple_optimization_test.cc:296-342): a k-d tree over
    the valid voxel means, per point a radius search on SQUARED L2 distance < radius with max_neighbors = 2, sorted.  FLANN
    (flann::KDTreeSingleIndex, un-vendored, absent here) is stood in for by scipy.spatial.cKDTree — the same data
    structure and query; used as the CPU timing baseline of the matcher stage and checked against the brute-force
    restatement above.  → (index array [n, 2] of original voxel ids, -1 = none; tree build seconds; query seconds).

    Ties at equal distance are broken by tree order here and by voxel id in match_point_cloud: callers that compare the
    two use inputs without exact ties."""
    import time
    from scipy.spatial import cKDTree
    means = np.asarray(means, dtype=np.float64).reshape(-1, 3)
    ok = np.ones(means.shape[0], dtype=bool) if valid is None else np.asarray(valid, dtype=bool)
    ids = np.nonzero(ok)[0]
    t0 = time.perf_counter()
    tree = cKDTree(means[ids])
    t1 = time.perf_counter()
    q = np.asarray(world_points, dtype=np.float64).reshape(-1, 3)
    k = min(max(int(max_neighbors), 1), 2)
    # `distance_upper_bound` is exclusive like FLANN's `dist < radius` test; distances are Euclidean, so sqrt(radius)
    d, j = tree.query(q, k=2, distance_upper_bound=float(np.sqrt(radius_sq)), workers=workers)
    t2 = time.perf_counter()
    idx = np.where(np.isfinite(d), ids[np.minimum(j, ids.size - 1)], -1).astype(np.int64)
    if k < 2:
        idx[:, 1] = -1
    return idx, t1 - t0, t2 - t1


# ----------------------------------------------------------------------------------------------
# Bit-level restatement (oracle/scene_oracle.c): the accumulation of UpdateNdtMap and Eigen's
# SelfAdjointEigenSolver<Matrix3d>, which together decide the captured COST lines.

import ctypes as _ct
import os as _os

_scene_lib = None

# multiply-add sites of scene_oracle.c (bits of fma_mask)
SITE_TRIDIAG, SITE_SQRT1P, SITE_QR, SITE_Q, SITE_MOMENT_SHIFT, SITE_COV_SHIFT = 4, 8, 16, 32, 8, 17

# The setting that reproduces the reference's captured runs (found by tests/golden/make_ndt_scene_golden.py).
REFERENCE_EIGEN_VERSION = 34
# solver sites all contracted; moment += p p^T contracted where Eigen's packet-of-two evaluation through the
# temporary keeps the product in a register (column-major linear elements 0, 1, 6, 7, 8 = row-major 0, 3, 2, 5, 8);
# covariance (lazy outer product, no temporary) contracted everywhere.
REFERENCE_FMA_MASK = (4 | 8 | 16 | 32) | ((1 | 4 | 8 | 32 | 256) << 8) | (0x1ff << 17)


def scene_lib():
    global _scene_lib
    if _scene_lib is None:
        from oracle import loader
        loader.build()
        path = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "build", "libnos_scene_oracle.so")
        if not _os.path.exists(path):
            loader.build(force=True)
        lib = _ct.CDLL(path)
        lib.scene_generate_global_points.restype = _ct.c_size_t
        lib.scene_build_ndt_map.restype = _ct.c_long
        lib.scene_eigen_selfadjoint3.restype = _ct.c_int
        _scene_lib = lib
    return _scene_lib


def _p(a, t=_ct.c_double):
    return a.ctypes.data_as(_ct.POINTER(t))


def generate_global_points_c():
    lib = scene_lib()
    n = lib.scene_generate_global_points(None, _ct.c_size_t(0))
    out = np.zeros((n, 3))
    lib.scene_generate_global_points(_p(out), _ct.c_size_t(n))
    return out


def eigen_selfadjoint3(A, version=None, fma_mask=None):
    """Eigen::SelfAdjointEigenSolver<Matrix3d>(A) → (eigenvalues ascending [3], eigenvectors as columns [3,3])."""
    version = REFERENCE_EIGEN_VERSION if version is None else version
    fma_mask = REFERENCE_FMA_MASK if fma_mask is None else fma_mask
    A = np.ascontiguousarray(A, dtype=np.float64).reshape(9)
    w = np.zeros(3)
    U = np.zeros(9)
    rc = scene_lib().scene_eigen_selfadjoint3(_p(A), _ct.c_int(version), _ct.c_int(fma_mask), _p(w), _p(U))
    if rc != 0:
        raise RuntimeError("NoConvergence")
    return w, U.reshape(3, 3).T.copy()  # column-major → V[i, k]


def build_ndt_map_eigen(points, voxel_resolution=1.0, version=None, fma_mask=None, max_voxels=4096):
    """UpdateNdtMap with Eigen's solver restated bit for bit; same dict as build_ndt_map (first-seen order)."""
    version = REFERENCE_EIGEN_VERSION if version is None else version
    fma_mask = REFERENCE_FMA_MASK if fma_mask is None else fma_mask
    pts = np.ascontiguousarray(points, dtype=np.float64)
    keys = np.zeros(max_voxels, dtype=np.uint64)
    counts = np.zeros(max_voxels, dtype=np.int32)
    means = np.zeros((max_voxels, 3))
    S = np.zeros((max_voxels, 3, 3))
    valid = np.zeros(max_voxels, dtype=np.int32)
    evals = np.zeros((max_voxels, 3))
    evecs = np.zeros((max_voxels, 3, 3))
    V = scene_lib().scene_build_ndt_map(
        _p(pts), _ct.c_size_t(pts.shape[0]), _ct.c_double(voxel_resolution), _ct.c_int(version), _ct.c_int(fma_mask),
        _ct.c_size_t(max_voxels), _p(keys, _ct.c_uint64), _p(counts, _ct.c_int), _p(means), _p(S), _p(valid, _ct.c_int),
        _p(evals), _p(evecs))
    if V < 0:
        raise RuntimeError("max_voxels too small")
    return {"means": means[:V].copy(), "sqrt_infos": S[:V].copy(), "valid": valid[:V].astype(bool), "keys": keys[:V].copy(),
            "count": counts[:V].copy(), "eigvals": evals[:V].copy(), "eigvecs": evecs[:V].copy()}


# ----------------------------------------------------------------------------------------------
# The reference's captured NDT runs (results/*.txt) as reproducible scenes.

# name → (filter voxel size, true translation, true yaw, dof, file:lines of the captured run)
CAPTURED_RUNS = {
    # MDM/tests/simple_optimization_test.cc:72-92 → results/maha_amd64_simple.txt:9-14,24
    "simple_6dof": (0.1, (-0.2, 0.123, 0.3), 0.1, 6),
    # MDM/tests/3dof_6dof_comparison_test.cc:64-88 → results/maha_3_vs_6_amd64.txt:6-11,33
    "planar_3dof": (0.1, (-0.15, 0.05, 0.0), 0.2, 3),
    # same scene, 6-DoF class → results/maha_3_vs_6_amd64.txt:18-24,35
    "planar_6dof": (0.1, (-0.15, 0.05, 0.0), 0.2, 6),
    # MDM/tests/simd_implementation_comparison_test.cc:71-91 → results/maha_amd64.txt:3-8,54-55
    "dense_6dof": (0.05, (-0.321, 0.123, 0.013), 0.123, 6),
}


def captured_run_scan(points, name):
    """FilterPoints + WarpPoints(true_pose^-1) of the named captured run → (local points [n,3], R_true, t_true)."""
    res, tt, yaw, _ = CAPTURED_RUNS[name]
    f = filter_points(points, res)
    c, s = np.cos(yaw), np.sin(yaw)
    Rt = np.array([[c, -s, 0.0], [s, c, 0.0], [0.0, 0.0, 1.0]])
    tt = np.array(tt)
    return (Rt.T @ (f - tt).T).T, Rt, tt


def compact_correspondences(planes, idx, stride=1):
    """The reference's correspondence list: point order, nearest then second nearest, absent neighbours skipped
    (MatchPointCloud, …test.cc:320-340), truncated to floor(N/stride)*stride items.

    stride = 4 is what the captured runs need: today's 3-DoF scalar class processes floor(N/4)*4 correspondences
    (MDM/…_analytic_3dof.cc:33-36) and the captured 6-DoF scalar lines (results/maha_amd64_simple.txt:10-13,
    maha_3_vs_6_amd64.txt:19-23, maha_amd64.txt:4-7 — the last identical to the 4-wide `SolveDoubleMatrix` lines
    :10-13) are reproduced digit for digit with the same truncation and by nothing else, i.e. they were captured from
    a revision whose 6-DoF scalar loop truncated as well."""
    cols = np.nonzero((idx >= 0).reshape(-1))[0]
    n = (cols.size // stride) * stride
    return np.ascontiguousarray(planes[:, cols[:n]]), cols.size


def captured_run_icp(solve_round, ndt_map, local_points, stride=4, max_outer=10):
    """OptimizePoseAnalytic / OptimizePoseAnalytic3dof (…/simple_optimization_test.cc:474-503): ≤ 10 rounds of
    {MatchPointCloud, Solve}, stop when |Δt| < 1e-5 and |vec(Δq)| < 1e-5.

    solve_round(planes [15, n], R, t) → (R, t, printed_cost, iterations).  Returns (R, t, rounds, outer_iter) with
    rounds = [(printed_cost, iterations, n_matches)]."""
    from oracle import loader
    R, t = np.eye(3), np.zeros(3)
    lastR, lastt = R.copy(), t.copy()
    rounds = []
    outer = 0
    for outer in range(max_outer):
        planes, _, idx = match_point_cloud(ndt_map["means"], ndt_map["sqrt_infos"], ndt_map["valid"], local_points, R, t)
        planes, n_matches = compact_correspondences(planes, idx, stride)
        R, t, printed, iters = solve_round(planes, R, t)
        R, t = np.asarray(R, dtype=np.float64).reshape(3, 3), np.asarray(t, dtype=np.float64).reshape(3)
        rounds.append((float(printed), int(iters), int(n_matches)))
        dR, dt = R.T @ lastR, R.T @ (lastt - t)
        q = loader.quat_from_matrix(dR)
        if np.linalg.norm(dt) < 1e-5 and np.linalg.norm(q[1:]) < 1e-5:
            break
        lastR, lastt = R.copy(), t.copy()
    else:
        outer = max_outer
    return R, t, rounds, outer


def printed(x):
    """A double as `std::cerr << x` prints it (6 significant digits)."""
    return "%.6g" % x
