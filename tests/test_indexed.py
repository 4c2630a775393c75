"""Voxel-indexed NDT datasets (additive layout): same sums as the flat layout / the oracle."""
import numpy as np
import pytest

from oracle import oracle_scene as scene
from tests import helpers

pytestmark = pytest.mark.gpu
LOSSES = [None, ("exponential", 1.0, 1.0), ("huber", 1.2)]
R_TEST = helpers.rot_xyz(0.01, -0.02, 0.05)
T_TEST = np.array([-0.1, 0.05, 0.2])


def _random_indexed(n, v, k, seed, frac_missing=0.1):
    rng = np.random.default_rng(seed)
    means = rng.uniform(-20, 20, size=(v, 3))
    S = rng.normal(size=(v, 3, 3)) * 3.0
    idx = rng.integers(0, v, size=(k, n)).astype(np.int32)
    idx[rng.uniform(size=(k, n)) < frac_missing] = -1
    pts = (means[np.maximum(idx[0], 0)] + rng.normal(scale=0.3, size=(n, 3))).T.copy()
    # the equivalent flat planes (one column per (point, slot) with a valid voxel)
    cols = []
    for kk in range(k):
        ok = idx[kk] >= 0
        cols.append(np.concatenate([pts[:, ok], means[idx[kk, ok]].T, S[idx[kk, ok]].reshape(-1, 9).T], axis=0))
    return pts, idx, means, S, np.concatenate(cols, axis=1)


@pytest.mark.parametrize("n,v,k", [(1, 1, 1), (1000, 7, 1), (50_003, 900, 2)])
@pytest.mark.parametrize("loss", LOSSES)
@pytest.mark.parametrize("sort", [False, True])
def test_indexed_matches_oracle(ctx, oracle, n, v, k, loss, sort):
    from nonlinear_optimizer_for_slam_amd import NdtIndexedDataset
    pts, idx, means, S, flat = _random_indexed(n, v, k, n + v)
    for dtype, rtol in (("f64", 1e-10), ("f32", 1e-4)):  # fp32 table of A = S^T S: measured up to 2.4e-5 on a single item
        ds = NdtIndexedDataset.from_arrays(ctx, pts, idx, means, S, dtype, sort)
        assert len(ds) == n and ds.stream_bytes == n * ((24 if dtype == "f64" else 12) + 4 * k)
        helpers.assert_normal_equations_close(ds.accumulate6(R_TEST, T_TEST, loss),
                                              oracle.ndt6_accumulate(flat, R_TEST, T_TEST, loss), 6, rtol)
        c, s = np.cos(0.07), np.sin(0.07)
        R2, t2 = np.array([[c, -s], [s, c]]), np.array([-0.15, 0.1])
        helpers.assert_normal_equations_close(ds.accumulate3(R2, t2, loss), oracle.ndt3_accumulate(flat, R2, t2, loss), 3, rtol)
        ds.close()


def test_indexed_all_slots_missing_gives_zero(ctx):
    from nonlinear_optimizer_for_slam_amd import NdtIndexedDataset
    pts = np.random.default_rng(0).normal(size=(3, 500))
    idx = -np.ones((2, 500), dtype=np.int32)
    ds = NdtIndexedDataset.from_arrays(ctx, pts, idx, np.zeros((3, 3)), np.tile(np.eye(3), (3, 1, 1)))
    assert np.all(ds.accumulate6(np.eye(3), np.zeros(3), ("exponential", 1.0, 1.0)) == 0.0)
    ds.close()


def test_indexed_matcher_equals_flat_matcher(ctx, oracle):
    """nos_ndt_match_indexed and nos_ndt_match describe the same correspondences: identical sums (to rounding),
    identical match counts — on the reference's room scene and through a whole scan-to-map solve."""
    from nonlinear_optimizer_for_slam_amd import api, solvers
    pts = scene.generate_global_points()
    m = scene.build_ndt_map(pts, 1.0)
    local = scene.filter_points(pts, 0.1) + np.array([0.1, -0.05, 0.02])
    gm = api.NdtMap(ctx, m["means"], m["sqrt_infos"], m["valid"], 1.0)
    sc = api.Scan(ctx, local)
    loss = ("exponential", 1.0, 1.0)
    flat, n_flat = gm.match(sc, np.eye(3), np.zeros(3), 2, "f64")
    for sort in (False, True):
        ind, n_ind = gm.match_indexed(sc, np.eye(3), np.zeros(3), 2, "f64", sort)
        assert n_ind == n_flat and len(ind) == local.shape[0]
        helpers.assert_normal_equations_close(ind.accumulate6(R_TEST, T_TEST, loss), flat.accumulate6(R_TEST, T_TEST, loss), 6, 1e-11)
        ind.close()
    ind, _ = gm.match_indexed(sc, np.eye(3), np.zeros(3), 2, "f64", True)
    solver = solvers.MahalanobisDistanceMinimizerHip()
    solver.SetLossFunction(loss)
    pa, pb = solvers.Pose(), solvers.Pose()
    assert solver.SolveDataset(solvers.Options(), flat, pa)
    it_flat = solver.report.iterations
    assert solver.SolveDataset(solvers.Options(), ind, pb)
    assert solver.report.iterations == it_flat
    dt, dq = helpers.pose_delta(pa.R, pa.t, pb.R, pb.t)
    assert dt < 1e-9 and dq < 1e-9
    for h in (flat, ind, sc, gm):
        h.close()


def test_indexed_full_size_matches_flat(ctx):
    """configs[1] shape (10 M points / 200 k voxels): the indexed dataset built from the synthetic scene's voxel
    assignment gives the flat dataset's sums."""
    from nonlinear_optimizer_for_slam_amd import NdtDataset, NdtIndexedDataset, synth
    n, v = 10_000_000, 200_000
    planes = synth.ndt_planes(n, v)
    # recover the voxel table / ids of the generator from the flat planes (mean_x identifies the voxel)
    uniq, first, inv = np.unique(planes[3], return_index=True, return_inverse=True)
    means = planes[3:6, first].T.copy()
    S = planes[6:15, first].T.copy()
    idx = inv.astype(np.int32)[None, :]
    loss = ("exponential", 1.0, 1.0)
    flat = NdtDataset.from_planes(ctx, planes, "f64")
    want = flat.accumulate6(R_TEST, T_TEST, loss)
    flat.close()
    ind = NdtIndexedDataset.from_arrays(ctx, planes[0:3], idx, means, S, "f64", True)
    helpers.assert_normal_equations_close(ind.accumulate6(R_TEST, T_TEST, loss), want, 6, 1e-11)
    ind.close()


def test_indexed_empty_and_download(ctx):
    from nonlinear_optimizer_for_slam_amd import NdtIndexedDataset, api
    ds = NdtIndexedDataset.from_arrays(ctx, np.zeros((3, 0)), np.zeros((1, 0), dtype=np.int32), np.zeros((1, 3)),
                                       np.eye(3)[None])
    assert len(ds) == 0 and np.all(ds.accumulate6(np.eye(3), np.zeros(3), None) == 0.0)
    ds.close()
    pts = np.arange(30, dtype=np.float64).reshape(3, 10)
    idx = np.array([[3, 1, 2, 0, 1, 3, 2, 0, -1, 1]], dtype=np.int32)
    ds = NdtIndexedDataset.from_arrays(ctx, pts, idx, np.zeros((4, 3)), np.tile(np.eye(3), (4, 1, 1)), "f64", True)
    got = api.download(ds)            # points come back in voxel-sorted order (ids 0,0,1,1,1,2,2,3,3,none)
    order = np.argsort(np.where(idx[0] < 0, 1 << 30, idx[0]), kind="stable")
    assert np.array_equal(got, pts[:, order])
    ds.close()
