"""GPU parity tests (-m gpu): the HIP path, called through the C ABI, against the CPU oracle on
identical seeded inputs.  Tolerances (written here, used below):

  fp64 kernels vs fp64 oracle, per-iteration H/g/cost : 1e-10 scaled
  fp32 kernels vs fp64 oracle, per-iteration H/g/cost : 3e-6 scaled — MEASURED worst case 1.1e-6 (NDT, exponential
                                                         loss; 1.0e-6 of it is the rounding of the INPUTS to fp32, which
                                                         the reference's SIMD classes share, MDM/..._analytic_simd.cc:25-27),
                                                         reprojection 5.1e-7 (tools/measure_fp32_error.py,
                                                         profiles/r02_fp32_error.jsonl)
  final pose, fp32 datasets vs fp64 oracle              : 2e-7 in t, 1e-8 in q — measured 5.1e-8 / 7.4e-10 on configs[0];
                                                         SURVEY.md §8(d) allows 2e-5 (the reference's own fp32-vs-fp64
                                                         gap, results/maha_amd64_simple.txt:24-25)
  final pose, fp64 LM loop                              : 1e-6 (north star), expected ~1e-10
"""
import os
import numpy as np
import pytest

from nonlinear_optimizer_for_slam_amd import NdtDataset, ReprojDataset, Context, solvers, synth
from tests import helpers

pytestmark = pytest.mark.gpu

RTOL_F64 = 1e-10
RTOL_F32 = 3e-6
LOSSES = [None, ("exponential", 1.0, 1.0), ("huber", 1.2)]
R_TEST = helpers.rot_xyz(0.01, -0.02, 0.05)
T_TEST = np.array([-0.1, 0.05, 0.2])


@pytest.mark.parametrize("n", [1, 63, 1000, 4097, 100_003])
@pytest.mark.parametrize("loss", LOSSES)
def test_ndt6_f64_matches_oracle(ctx, oracle, n, loss):
    planes = synth.ndt_planes(n, max(1, n // 20))
    ds = NdtDataset.from_planes(ctx, planes, "f64")
    assert len(ds) == n and ds.stream_bytes == n * 120
    got = ds.accumulate6(R_TEST, T_TEST, loss)
    want = oracle.ndt6_accumulate(planes, R_TEST, T_TEST, loss)
    helpers.assert_normal_equations_close(got, want, 6, RTOL_F64)
    ds.close()


@pytest.mark.parametrize("loss", LOSSES)
def test_ndt6_f32_matches_oracle_within_fp32(ctx, oracle, loss):
    planes = synth.ndt_planes(50_000, 2500)
    ds = NdtDataset.from_planes(ctx, planes, "f32")
    assert ds.stream_bytes == 50_000 * 60
    got = ds.accumulate6(R_TEST, T_TEST, loss)
    want = oracle.ndt6_accumulate(planes, R_TEST, T_TEST, loss)
    helpers.assert_normal_equations_close(got, want, 6, RTOL_F32)
    ds.close()


@pytest.mark.parametrize("dtype,rtol", [("f64", RTOL_F64), ("f32", RTOL_F32)])
@pytest.mark.parametrize("loss", LOSSES)
def test_ndt3_matches_oracle(ctx, oracle, dtype, rtol, loss):
    planes = synth.ndt_planes(30_001, 1500)
    c, s = np.cos(0.07), np.sin(0.07)
    R2 = np.array([[c, -s], [s, c]])
    t2 = np.array([-0.15, 0.1])
    ds = NdtDataset.from_planes(ctx, planes, dtype)
    got = ds.accumulate3(R2, t2, loss)
    want = oracle.ndt3_accumulate(planes, R2, t2, loss)
    helpers.assert_normal_equations_close(got, want, 3, rtol)
    ds.close()


@pytest.mark.parametrize("dtype,rtol", [("f64", RTOL_F64), ("f32", 2e-6)])
@pytest.mark.parametrize("loss", [None, ("exponential", 1.0, 1.0), ("huber", synth.REPROJ_HUBER_THRESHOLD)])
def test_reproj_matches_oracle(ctx, oracle, dtype, rtol, loss):
    planes = synth.reproj_planes(40_003)
    planes[2, :100] = -2.0  # behind the camera → dropped by the depth test (REM/..._analytic.cc:119-123)
    R = helpers.rot_xyz(0.0, 0.01, -0.08)
    t = np.array([0.08, -0.1, 0.4])
    ds = ReprojDataset.from_planes(ctx, planes, dtype)
    got = ds.accumulate(R, t, synth.REPROJ_INTR4, loss)
    want = oracle.reproj_accumulate(planes, R, t, synth.REPROJ_INTR4, loss)
    helpers.assert_normal_equations_close(got, want, 6, rtol)
    ds.close()


@pytest.mark.parametrize("n", [0, 1, 511, 513, 131_071, 262_145, 393_217, 450_001, 700_003, 1_048_583])
def test_reprojection_ping_pong_kernel_at_every_trip_count(ctx, oracle, n):
    """The launch-per-pass reprojection kernel is the ping-pong form (two register buffers, loop unrolled twice, peeled
    tail): with one 512-thread workgroup per CU a workgroup takes 0, 1, 2, 3, 4, 5-6 or 8-9 chunks at these sizes — every
    path through the prologue, the unrolled body (odd and even trip counts) and the epilogue — and the sums must be the
    oracle's, with behind-camera points and a ragged tail; both element types."""
    planes = synth.reproj_planes(max(n, 1))[:, :n]
    if n > 200:
        planes[2, 50:150] = -2.0
    R = helpers.rot_xyz(0.0, 0.01, -0.08)
    t = np.array([0.08, -0.1, 0.4])
    loss = ("huber", synth.REPROJ_HUBER_THRESHOLD)
    want = oracle.reproj_accumulate(planes, R, t, synth.REPROJ_INTR4, loss) if n > 0 else np.zeros(28)
    for dtype, rtol in (("f64", RTOL_F64), ("f32", 2e-6)):
        ds = ReprojDataset.from_planes(ctx, planes, dtype)
        got = ds.accumulate(R, t, synth.REPROJ_INTR4, loss)
        if n == 0:
            assert np.all(got == 0.0)
        else:
            helpers.assert_normal_equations_close(got, want, 6, rtol)
            assert ", 3>" in ctx.last_kernel() and "ReprojProblem" in ctx.last_kernel()  # the ping-pong instantiation ran
        assert np.array_equal(got, ds.accumulate(R, t, synth.REPROJ_INTR4, loss))  # bit-repeatable
        ds.close()


def test_empty_dataset_gives_zero_sums(ctx):
    planes = np.zeros((15, 0))
    ds = NdtDataset.from_planes(ctx, planes, "f64")
    out = ds.accumulate6(np.eye(3), np.zeros(3), ("exponential", 1.0, 1.0))
    assert np.all(out == 0.0)
    ds.close()


def test_repeated_launches_are_bit_identical(ctx):
    planes = synth.ndt_planes(300_000, 5000)
    ds = NdtDataset.from_planes(ctx, planes, "f64")
    a = ds.accumulate6(R_TEST, T_TEST, ("exponential", 1.0, 1.0))
    for _ in range(3):
        assert np.array_equal(a, ds.accumulate6(R_TEST, T_TEST, ("exponential", 1.0, 1.0)))
    ds.close()


@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_ingestion_paths_agree_bit_for_bit(ctx, dtype):
    """host planes / device planes (torch) / array-of-structures records → same dataset (fp64: planar planes, fp32: the
    1024-item tiles), and it reads back as what went in."""
    import torch
    if not torch.cuda.is_available():  # torch is plumbing for device memory here, not the product under test
        pytest.skip("torch cannot see the GPU on this box (libnos_hip can): device-plane ingestion not exercised")
    n = 20_011
    planes = synth.ndt_planes(n, 800)
    loss = ("exponential", 1.0, 1.0)
    a = NdtDataset.from_planes(ctx, planes, dtype)
    want = a.accumulate6(R_TEST, T_TEST, loss)
    dev = torch.from_numpy(planes).cuda()
    b = NdtDataset.from_device_planes(ctx, dev, dtype)
    assert np.array_equal(want, b.accumulate6(R_TEST, T_TEST, loss))
    # the reference's 304-byte Correspondence (MDM/types.h:11-26) with Eigen's column-major 3x3:
    # point @0, ndt.mean @128, ndt.sqrt_information @224
    rec = np.zeros((n, 38), dtype=np.float64)
    rec[:, 0:3] = planes[0:3].T
    rec[:, 16:19] = planes[3:6].T
    for i in range(3):
        for j in range(3):
            rec[:, 28 + 3 * j + i] = planes[6 + 3 * i + j]
    offs = [0, 8, 16, 128, 136, 144] + [224 + 8 * (3 * j + i) for i in range(3) for j in range(3)]
    c = NdtDataset.from_records(ctx, rec, 304, offs, dtype)
    assert len(c) == n
    assert np.array_equal(want, c.accumulate6(R_TEST, T_TEST, loss))
    from nonlinear_optimizer_for_slam_amd import api
    back = api.download(a)
    assert np.array_equal(back, planes if dtype == "f64" else planes.astype(np.float32).astype(np.float64))
    for d in (a, b, c):
        d.close()


def test_two_shards_on_one_device_match_single_shard(oracle):
    """Single-process fan-out (context listing the device twice): contiguous shards, host sum."""
    planes = synth.ndt_planes(70_001, 3000)
    loss = ("exponential", 1.0, 1.0)
    c2 = Context((0, 0))
    ds = NdtDataset.from_planes(c2, planes, "f64")
    got = ds.accumulate6(R_TEST, T_TEST, loss)
    want = oracle.ndt6_accumulate(planes, R_TEST, T_TEST, loss)
    helpers.assert_normal_equations_close(got, want, 6, RTOL_F64)
    ds.close()
    c2.close()


def test_async_result_in_torch_tensor_matches_sync(ctx):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("torch cannot see the GPU on this box (libnos_hip can)")
    planes = synth.ndt_planes(50_000, 2000)
    loss = ("exponential", 1.0, 1.0)
    ds = NdtDataset.from_planes(ctx, planes, "f64")
    want = ds.accumulate6(R_TEST, T_TEST, loss)
    out = torch.zeros(28, dtype=torch.float64, device="cuda")
    ctx.use_torch_stream()
    ds.accumulate6_async(R_TEST, T_TEST, loss, out)
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy(), want)
    ctx.set_stream(0)
    ds.close()


def test_every_launch_geometry_and_layout_gives_the_same_sums(oracle):
    """Every compiled launch geometry on every dataset layout (planar planes, 1024- and 4096-item tiles; the fp32 default
    is the 1024-item tile, the fp64 default planar).  A geometry whose chunk does not divide the tile, or that is not
    compiled for the element type, must be refused with an error — never run.  The default build carries fp64 geometries
    0, 1, 3, 7 and fp32 1, 11 (0 = the library's choice); a `make ALL_VARIANTS=1` build (tools/, nos_version() says so)
    all of them."""
    from nonlinear_optimizer_for_slam_amd import _lib
    all_variants = b"all launch geometries" in _lib.hip_lib().nos_version()
    compiled = {"f64": set(range(9)) if all_variants else {0, 1, 3, 7}, "f32": {0, 1, 2, 3, 4, 5, 6, 9, 11, 12, 13} if all_variants else {0, 1, 11}}
    planes = synth.ndt_planes(423_457, 4000)  # 3-4 chunks per workgroup at one 512-thread workgroup per CU: loop bodies run
    loss = ("exponential", 1.0, 1.0)
    want = oracle.ndt6_accumulate(planes, R_TEST, T_TEST, loss)
    c = Context((0,))
    ran = refused = 0
    for tile in (-1, 0, 10, 12):
        c.set_option("tile_log2", tile)
        for dtype, rtol in (("f64", RTOL_F64), ("f32", RTOL_F32)):
            ds = NdtDataset.from_planes(c, planes, dtype)
            for variant in range(14):
                for bpc in (0, 1, 4):
                    c.set_launch(bpc, variant)
                    try:
                        got = ds.accumulate6(R_TEST, T_TEST, loss)
                    except RuntimeError:
                        # not compiled for the element type, or a 2048-item chunk (fp32 geometries 6, 12) on a 1024-item tile
                        assert variant not in compiled[dtype] or (dtype == "f32" and variant in (6, 12) and tile in (-1, 10))
                        refused += 1
                        continue
                    assert variant in compiled[dtype]
                    helpers.assert_normal_equations_close(got, want, 6, rtol)
                    ran += 1
            ds.close()
    c.close()
    assert ran >= 4 * 3 * (len(compiled["f64"]) + len(compiled["f32"]) - 1) and refused > 0


# ---------------------------------------------------------------- Solve() through the C++ classes

def test_reprojection_known_answer_on_gpu():
    """End-to-end golden of the reference (results/reproj_amd64.txt:5,8): ReprojectionErrorMinimizerHip
    from identity, ExponentialLossFunction(1,1), default Options → `COST: 2.33228e-11, iter: 6`
    and the true pose."""
    planes, intr, Rt, tt = helpers.reference_reprojection_scene()
    solver = solvers.ReprojectionErrorMinimizerHip()
    solver.SetLossFunction(("exponential", 1.0, 1.0))
    pose = solvers.Pose()
    assert solver.Solve(solvers.Options(), planes, intr, pose)
    assert solver.report.iterations == 6
    assert "%.5g" % solver.report.printed_cost == "2.3323e-11"
    inv = pose.inverse()
    np.testing.assert_allclose(inv.t, tt, atol=5e-7)
    np.testing.assert_allclose(inv.R, Rt, atol=1e-7)


@pytest.mark.parametrize("loss", [("exponential", 1.0, 1.0), None])
def test_ndt6_solve_matches_oracle_final_pose(oracle, loss):
    """BASELINE.json configs[0] shape: 100k points / 5k voxels, 6-DoF, default Options."""
    planes = synth.ndt_planes(100_000, 5000)
    want = oracle.ndt6_solve(planes, np.zeros(3), np.eye(3), loss=loss, linear_solver=1)
    solver = solvers.MahalanobisDistanceMinimizerHip()
    solver.SetLossFunction(loss)
    pose = solvers.Pose()
    assert solver.Solve(solvers.Options(), planes, pose)
    assert solver.report.iterations == want["iterations"]
    dt, dq = helpers.pose_delta(pose.R, pose.t, want["R"], want["t"])
    assert dt < 1e-6 and dq < 1e-6, (dt, dq)   # north-star tolerance
    assert dt < 1e-9 and dq < 1e-9, (dt, dq)   # what fp64 actually delivers
    Rt, tt = synth.true_pose("ndt")
    dt, dq = helpers.pose_delta(pose.R, pose.t, Rt, tt)
    assert dt < 5e-3 and dq < 2e-3


def test_ndt6_solve_f32_tracks_fp64_oracle(oracle):
    planes = synth.ndt_planes(100_000, 5000)
    loss = ("exponential", 1.0, 1.0)
    want = oracle.ndt6_solve(planes, np.zeros(3), np.eye(3), loss=loss, linear_solver=1)
    solver = solvers.MahalanobisDistanceMinimizerHip(dtype="f32")
    solver.SetLossFunction(loss)
    pose = solvers.Pose()
    assert solver.Solve(solvers.Options(), planes, pose)
    dt, dq = helpers.pose_delta(pose.R, pose.t, want["R"], want["t"])
    assert solver.report.iterations == want["iterations"]
    assert dt < 2e-7 and dq < 1e-8, (dt, dq)


def test_ndt3_solve_matches_oracle(oracle):
    planes = synth.ndt_planes(60_000, 3000)
    want = oracle.ndt3_solve(planes, np.zeros(3), np.eye(3), loss=("exponential", 1.0, 1.0))
    solver = solvers.MahalanobisDistanceMinimizerHip3DOF()
    solver.SetLossFunction(("exponential", 1.0, 1.0))
    pose = solvers.Pose()
    assert solver.Solve(solvers.Options(), planes, pose)
    assert solver.report.iterations == want["iterations"]
    np.testing.assert_allclose(pose.t, want["t"], atol=1e-9)
    np.testing.assert_allclose(pose.R, want["R"], atol=1e-9)


def test_reproj_solve_huber_matches_oracle(oracle):
    """BASELINE.json configs[2] shape (scaled to 200k): Huber loss, noisy pixels, 5 % outliers."""
    planes = synth.reproj_planes(200_000)
    loss = ("huber", synth.REPROJ_HUBER_THRESHOLD)
    want = oracle.reproj_solve(planes, synth.REPROJ_INTR4, np.zeros(3), np.eye(3), loss=loss, linear_solver=1)
    solver = solvers.ReprojectionErrorMinimizerHip()
    solver.SetLossFunction(loss)
    pose = solvers.Pose()
    assert solver.Solve(solvers.Options(), planes, synth.REPROJ_INTRINSICS, pose)
    assert solver.report.iterations == want["iterations"]
    dt, dq = helpers.pose_delta(pose.R, pose.t, want["R"], want["t"])
    assert dt < 1e-8 and dq < 1e-8, (dt, dq)
    Rt, tt = synth.true_pose("reproj")
    inv = pose.inverse()
    assert np.max(np.abs(inv.t - tt)) < 2e-3


def test_prepared_dataset_resolve_is_repeatable():
    planes = synth.ndt_planes(50_000, 2500)
    solver = solvers.MahalanobisDistanceMinimizerHip()
    solver.SetLossFunction(("exponential", 1.0, 1.0))
    p1 = solvers.Pose()
    assert solver.Solve(solvers.Options(), planes, p1)
    p2 = solvers.Pose()
    assert solver.Solve(solvers.Options(), planes, p2, repeat_solves=3)
    assert np.array_equal(p1.t, p2.t) and np.array_equal(p1.R, p2.R)


# ---------------------------------------------------------------- full-size properties

@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_full_size_additivity_and_permutation(ctx, dtype):
    """BASELINE.json configs[1] size (10 M points / 200 k voxels), too big for the scalar oracle to
    be the only check: the sums are additive over any split of the correspondences and invariant
    under permutation, so   A(all) == A(first part) + A(rest)   and   A(all) == A(reversed)."""
    n = 10_000_000
    planes = synth.ndt_planes(n, 200_000)
    loss = ("exponential", 1.0, 1.0)
    rtol = 1e-11 if dtype == "f64" else 2e-5
    whole = NdtDataset.from_planes(ctx, planes, dtype)
    a_all = whole.accumulate6(R_TEST, T_TEST, loss)
    whole.close()
    cut = 3_333_333
    parts = np.zeros(28)
    for sl in (slice(0, cut), slice(cut, n)):
        ds = NdtDataset.from_planes(ctx, np.ascontiguousarray(planes[:, sl]), dtype)
        parts += ds.accumulate6(R_TEST, T_TEST, loss)
        ds.close()
    helpers.assert_normal_equations_close(parts, a_all, 6, rtol)
    rev = NdtDataset.from_planes(ctx, np.ascontiguousarray(planes[:, ::-1]), dtype)
    helpers.assert_normal_equations_close(rev.accumulate6(R_TEST, T_TEST, loss), a_all, 6, rtol)
    rev.close()


def test_full_size_sample_against_oracle(ctx, oracle):
    """A 1 M-correspondence slice of the 10 M workload against the scalar oracle directly."""
    planes = synth.ndt_planes(1_000_000, 200_000)
    loss = ("exponential", 1.0, 1.0)
    ds = NdtDataset.from_planes(ctx, planes, "f64")
    got = ds.accumulate6(R_TEST, T_TEST, loss)
    want = oracle.ndt6_accumulate(planes, R_TEST, T_TEST, loss)
    helpers.assert_normal_equations_close(got, want, 6, RTOL_F64)
    ds.close()


def test_native_rccl_single_rank_communicator(oracle):
    """The in-library RCCL all-reduce path (one process per GPU) with a 1-rank communicator:
    same sums as the plain path, self-test all-reduce is the identity."""
    from nonlinear_optimizer_for_slam_amd.api import new_unique_id
    planes = synth.ndt_planes(65_000, 3000)
    loss = ("exponential", 1.0, 1.0)
    c = Context((0,))
    ds = NdtDataset.from_planes(c, planes, "f64")
    want = ds.accumulate6(R_TEST, T_TEST, loss)
    assert c.comm_size == 0
    from nonlinear_optimizer_for_slam_amd import _lib
    try:
        c.comm_init(1, 0, new_unique_id())
    except _lib.NosError as exc:  # RCCL itself refusing to start on a box is an environment problem, not ours
        ds.close()
        c.close()
        pytest.skip("RCCL could not create a 1-rank communicator on this box: %s" % exc)
    assert c.comm_size == 1
    np.testing.assert_array_equal(c.comm_allreduce([3.0, 4.5]), [3.0, 4.5])
    got = ds.accumulate6(R_TEST, T_TEST, loss)
    assert np.array_equal(got, want)
    helpers.assert_normal_equations_close(got, oracle.ndt6_accumulate(planes, R_TEST, T_TEST, loss), 6, RTOL_F64)
    ds.close()
    c.close()


@pytest.mark.parametrize("n", [3000, 400_000])
def test_in_launch_final_reduce_never_reads_stale_rows(n):
    """Hand-off stress for the fused final reduce (Guideline-16 protocol): alternate two poses so every
    launch overwrites the block rows with different values; a stale row (missed release/acquire) would
    reproduce the other pose's contribution.  Every launch must be bit-identical to the first launch of
    its pose, for several launch geometries (1 … 8 blocks per CU)."""
    planes = synth.ndt_planes(n, max(1, n // 40))
    loss = ("exponential", 1.0, 1.0)
    c = Context((0,))
    ds = NdtDataset.from_planes(c, planes, "f64")
    RA, tA = R_TEST, T_TEST
    RB, tB = helpers.rot_xyz(-0.03, 0.02, -0.06), np.array([0.3, -0.2, -0.1])
    for variant, bpc in ((0, 0), (1, 8), (3, 4)):
        c.set_launch(bpc, variant)
        a0 = ds.accumulate6(RA, tA, loss)
        b0 = ds.accumulate6(RB, tB, loss)
        assert not np.array_equal(a0, b0)
        for _ in range(150):
            assert np.array_equal(ds.accumulate6(RA, tA, loss), a0)
            assert np.array_equal(ds.accumulate6(RB, tB, loss), b0)
    ds.close()
    c.close()


def test_c_abi_argument_errors_are_reported_not_crashed(ctx):
    import ctypes
    from nonlinear_optimizer_for_slam_amd import _lib
    lib = _lib.hip_lib()
    planes = synth.ndt_planes(100, 5)
    ds = NdtDataset.from_planes(ctx, planes, "f64")
    rp = ReprojDataset.from_planes(ctx, synth.reproj_planes(100), "f64")
    with pytest.raises(_lib.NosError) as e1:      # wrong dataset kind
        rp_as_ndt = NdtDataset(ctx, rp._h)
        try:
            rp_as_ndt.accumulate6(np.eye(3), np.zeros(3))
        finally:
            rp_as_ndt._h = None
    assert e1.value.status == 5
    with pytest.raises(_lib.NosError) as e2:      # loss parameters validated like loss_function.h:24-25
        ds.accumulate6(np.eye(3), np.zeros(3), ("exponential", -1.0, 1.0))
    assert e2.value.status == 1
    with pytest.raises(_lib.NosError):
        ds.accumulate6(np.eye(3), np.zeros(3), ("huber", 0.0))
    with pytest.raises(_lib.NosError):            # dtype
        NdtDataset.from_planes(ctx, planes, 7) if False else _lib.check(
            lib.nos_ndt_dataset_create(ctx.handle, 0, None, 7, ctypes.byref(ctypes.c_void_p())), "create")
    assert lib.nos_ndt6_accumulate(None, None, None, None, None) == 1
    assert b"NULL" in lib.nos_last_error()
    ds.close()
    rp.close()


def test_options_are_range_checked_and_two_contexts_share_a_device(oracle):
    """(a) nos_ctx_set_option refuses values outside an option's range instead of storing nonsense (a negative
    plane skew used to become a huge stride); (b) the dynamic-LDS grant of the resident solve is asked for per launch, not
    remembered in a process-wide static: a SECOND context on the device runs a resident solve that needs > 64 KB of LDS
    right after the first one did, and a launch-geometry query names what ran."""
    from nonlinear_optimizer_for_slam_amd import _lib
    a = Context((0,))
    for key, bad in (("plane_skew", -5), ("lm_cluster", 9), ("lm_window", 0), ("tile_log2", 3), ("pgo_agg", 1),
                     ("map_eigen_version", 35), ("debug_cluster_abort", 3), ("nt", -2), ("lm_cluster_max_blocks", 0)):
        with pytest.raises(_lib.NosError) as err:
            a.set_option(key, bad)
        assert err.value.status == 1, key
    a.set_option("plane_skew", 0)
    assert a.get_option("plane_skew") == 0
    a.set_option("plane_skew", 1088)
    with pytest.raises(_lib.NosError):
        a.set_option("no_such_option", 1)
    n = 700_000  # 6 correspondences per lane: three of them in LDS (3 x 12 x 512 x 8 B = 147 KB of dynamic LDS)
    planes = synth.ndt_planes(n, 7000)
    loss = ("exponential", 1.0, 1.0)
    want = oracle.ndt6_solve(planes, np.zeros(3), np.eye(3), loss=loss, linear_solver=1)
    b = Context((0,))
    for c in (a, b, a):
        ds = NdtDataset.from_planes(c, planes, "f64")
        R, t, rep = ds.solve6(np.eye(3), np.zeros(3), loss, max_iterations=40)
        assert rep["launches"] == 1 and rep["iterations"] == want["iterations"]
        assert "solve_cluster_kernel<nos::Ndt6Problem<double, 1>, double, 512, 3, 3" in c.last_kernel()
        dt, dq = helpers.pose_delta(R.reshape(3, 3), t, want["R"], want["t"])
        assert dt < 1e-9 and dq < 1e-9
        ds.close()
    b.close()
    a.close()


def test_eighty_million_correspondences_on_one_gpu(ctx):
    """Largest BASELINE.json size (configs[3]: 80 M correspondences) resident on ONE device as fp32 storage
    (4.8 GB): additivity over a 3-way split, so the single-GPU strong-scaling baseline is known to be right."""
    n = 80_000_000
    loss = ("exponential", 1.0, 1.0)
    total = np.zeros(28)
    parts = []
    whole_planes = []
    for k, cnt in enumerate((30_000_000, 30_000_000, 20_000_000)):
        planes = synth.ndt_planes(cnt, 200_000, first_block=k * 500)
        ds = NdtDataset.from_planes(ctx, planes, "f32")
        total += ds.accumulate6(R_TEST, T_TEST, loss)
        ds.close()
        whole_planes.append(planes)
    whole = NdtDataset.from_planes(ctx, np.concatenate(whole_planes, axis=1), "f32")
    del whole_planes
    assert len(whole) == n and whole.stream_bytes == n * 60
    helpers.assert_normal_equations_close(whole.accumulate6(R_TEST, T_TEST, loss), total, 6, 2e-5)
    whole.close()


def test_eighty_million_correspondences_fp64_on_one_gpu_the_strong_scaling_baseline(ctx, oracle):
    """BASELINE.json configs[3] as ONE fp64 dataset (9.6 GB): the denominator of the 8-GPU strong-scaling claim
    (bench.py `strong_baseline`).  The data is what the 8 ranks of `bench.py --gpus 8` hold (rank r: 10 M correspondences
    from generator block r * 153).  Checked by additivity — the sums over the whole equal the sums of the 8 rank shards
    (= what the all-reduce delivers) — and by the oracle on a 1 M-correspondence slice straddling two ranks."""
    per, ranks = 10_000_000, 8
    loss = ("exponential", 1.0, 1.0)
    blocks_per_rank = (per + 65535) // 65536
    big = np.empty((15, per * ranks))
    total = np.zeros(28)
    for r in range(ranks):
        planes = synth.ndt_planes(per, 200_000, first_block=r * blocks_per_rank)
        big[:, r * per:(r + 1) * per] = planes
        ds = NdtDataset.from_planes(ctx, planes, "f64")
        total += ds.accumulate6(R_TEST, T_TEST, loss)
        ds.close()
    whole = NdtDataset.from_planes(ctx, big, "f64")
    assert len(whole) == per * ranks and whole.stream_bytes == per * ranks * 120
    helpers.assert_normal_equations_close(whole.accumulate6(R_TEST, T_TEST, loss), total, 6, 1e-11)
    lo, hi = 3 * per - 500_000, 3 * per + 500_000
    sl = NdtDataset.from_planes(ctx, np.ascontiguousarray(big[:, lo:hi]), "f64")
    helpers.assert_normal_equations_close(sl.accumulate6(R_TEST, T_TEST, loss),
                                          oracle.ndt6_accumulate(np.ascontiguousarray(big[:, lo:hi]), R_TEST, T_TEST, loss), 6, 1e-10)
    sl.close()
    del big
    # the device-resident loop on it converges to the generator's pose like the 10 M case does
    R, t, rep = whole.solve6(np.eye(3), np.zeros(3), loss, max_iterations=40)
    assert rep["ok"] and np.max(np.abs(t - synth.true_pose("ndt")[1])) < 5e-4
    whole.close()


def test_cpp_class_multi_shard_and_fp32_solve_agree_with_single_shard():
    """HipOptions.device_ids with the device listed twice (single-process fan-out inside Solve()) and dtype."""
    planes = synth.ndt_planes(80_003, 4000)
    loss = ("exponential", 1.0, 1.0)
    ref = solvers.MahalanobisDistanceMinimizerHip()
    ref.SetLossFunction(loss)
    p0 = solvers.Pose()
    assert ref.Solve(solvers.Options(), planes, p0)
    two = solvers.MahalanobisDistanceMinimizerHip(device_ids=(0, 0))
    two.SetLossFunction(loss)
    two.SetMultiThreadExecutor(object())  # accepted, ignored
    p1 = solvers.Pose()
    assert two.Solve(solvers.Options(), planes, p1)
    assert two.report.iterations == ref.report.iterations
    dt, dq = helpers.pose_delta(p0.R, p0.t, p1.R, p1.t)
    assert dt < 1e-10 and dq < 1e-10
    f32 = solvers.MahalanobisDistanceMinimizerHip3DOF(dtype="f32")
    f32.SetLossFunction(loss)
    f64 = solvers.MahalanobisDistanceMinimizerHip3DOF()
    f64.SetLossFunction(loss)
    pa, pb = solvers.Pose(), solvers.Pose()
    assert f32.Solve(solvers.Options(), planes, pa) and f64.Solve(solvers.Options(), planes, pb)
    assert np.max(np.abs(pa.t - pb.t)) < 1e-6 and np.max(np.abs(pa.R - pb.R)) < 1e-6


# ---------------------------------------------------------------- more full-size properties (BASELINE.json sizes)

def test_full_size_reprojection_additivity_and_oracle_sample(ctx, oracle):
    """BASELINE.json configs[2]: 2 M reprojection correspondences, Huber.  Additive over a split, invariant under
    reversal, and a 250 k slice agrees with the scalar oracle."""
    n = 2_000_000
    planes = synth.reproj_planes(n)
    loss = ("huber", synth.REPROJ_HUBER_THRESHOLD)
    whole = ReprojDataset.from_planes(ctx, planes, "f64")
    a_all = whole.accumulate(R_TEST, T_TEST, synth.REPROJ_INTR4, loss)
    whole.close()
    parts = np.zeros(28)
    for sl in (slice(0, 700_001), slice(700_001, n)):
        ds = ReprojDataset.from_planes(ctx, np.ascontiguousarray(planes[:, sl]), "f64")
        parts += ds.accumulate(R_TEST, T_TEST, synth.REPROJ_INTR4, loss)
        ds.close()
    helpers.assert_normal_equations_close(parts, a_all, 6, 1e-11)
    rev = ReprojDataset.from_planes(ctx, np.ascontiguousarray(planes[:, ::-1]), "f64")
    helpers.assert_normal_equations_close(rev.accumulate(R_TEST, T_TEST, synth.REPROJ_INTR4, loss), a_all, 6, 1e-11)
    rev.close()
    sub = np.ascontiguousarray(planes[:, :250_000])
    ds = ReprojDataset.from_planes(ctx, sub, "f64")
    helpers.assert_normal_equations_close(ds.accumulate(R_TEST, T_TEST, synth.REPROJ_INTR4, loss),
                                          oracle.reproj_accumulate(sub, R_TEST, T_TEST, synth.REPROJ_INTR4, loss), 6, RTOL_F64)
    ds.close()


def test_full_size_planar_additivity(ctx):
    """The 3-DoF sums at the configs[1] size: additive over a split and invariant under reversal."""
    n = 10_000_000
    planes = synth.ndt_planes(n, 200_000)
    loss = ("exponential", 1.0, 1.0)
    R2 = np.array([[np.cos(0.05), -np.sin(0.05)], [np.sin(0.05), np.cos(0.05)]])
    t2 = np.array([-0.1, 0.05])
    whole = NdtDataset.from_planes(ctx, planes, "f64")
    a_all = whole.accumulate3(R2, t2, loss)
    whole.close()
    parts = np.zeros(10)
    for sl in (slice(0, 4_000_003), slice(4_000_003, n)):
        ds = NdtDataset.from_planes(ctx, np.ascontiguousarray(planes[:, sl]), "f64")
        parts += ds.accumulate3(R2, t2, loss)
        ds.close()
    helpers.assert_normal_equations_close(parts, a_all, 3, 1e-11)


def test_full_size_device_loop_equals_host_loop_and_reaches_the_true_pose(ctx):
    """configs[1] end to end: the device-resident loop and the host loop on the same 10 M-correspondence dataset stop at
    the same iteration with the same pose, and that pose is the generator's true pose up to the noise floor."""
    planes = synth.ndt_planes(10_000_000, 200_000)
    loss = ("exponential", 1.0, 1.0)
    ds = NdtDataset.from_planes(ctx, planes, "f64")
    del planes
    R, t, rep = ds.solve6(np.eye(3), np.zeros(3), loss, max_iterations=100)
    host = solvers.MahalanobisDistanceMinimizerHip(device_loop=False)
    host.SetLossFunction(loss)
    pose = solvers.Pose()
    opt = solvers.Options()
    opt.max_iterations = 100
    assert host.SolveDataset(opt, ds, pose)
    assert rep["ok"] and rep["iterations"] == host.report.iterations
    dt, dq = helpers.pose_delta(R.reshape(3, 3), t, pose.R, pose.t)
    assert dt < 1e-10 and dq < 1e-10, (dt, dq)
    Rt, tt = synth.true_pose("ndt")
    dt, dq = helpers.pose_delta(R.reshape(3, 3), t, Rt, tt)
    assert dt < 5e-4 and dq < 2e-4, (dt, dq)
    ds.close()


def test_repeated_cold_solves_do_not_leak_device_memory():
    """The buffer pool parks at most 8 buffers per device: 300 cold Solve() calls of changing size must leave the free
    device memory where a handful of calls left it."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("torch cannot see the GPU on this box (environment): no mem_get_info")
    solver = solvers.MahalanobisDistanceMinimizerHip()
    solver.SetLossFunction(("exponential", 1.0, 1.0))
    sizes = [3_000, 50_000, 20_000, 120_000, 7_000, 400_000]
    cache = {n: synth.ndt_planes(n, max(10, n // 50)) for n in sizes}
    opt = solvers.Options()
    opt.max_iterations = 5

    def sweep(rounds):
        for r in range(rounds):
            for n in sizes:
                assert solver.Solve(opt, cache[n], solvers.Pose())

    sweep(3)
    torch.cuda.synchronize()
    free_before, _ = torch.cuda.mem_get_info()
    sweep(50)
    torch.cuda.synchronize()
    free_after, _ = torch.cuda.mem_get_info()
    assert free_before - free_after < (256 << 20), (free_before, free_after)


@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_host_pack_ingestion_equals_device_unpack(ctx, dtype):
    """Both ingestion forms of the 304-byte records (host threads pack pinned planes | raw records unpacked on the
    device) must build the same dataset bit for bit, chunk borders and tails included (600 001 records = 3 pack chunks)."""
    import os
    n = 600_001
    planes = synth.ndt_planes(n, 9000)
    rec = np.zeros((n, 38))
    rec[:, 0:3] = planes[0:3].T
    rec[:, 16:19] = planes[3:6].T
    for i in range(3):
        for j in range(3):
            rec[:, 28 + 3 * j + i] = planes[6 + 3 * i + j]
    offs = [0, 8, 16, 128, 136, 144] + [224 + 8 * (3 * j + i) for i in range(3) for j in range(3)]
    got = {}
    for mode in ("unpack", "pack"):
        with ctx.options(ingest={"pack": 1, "unpack": 2}[mode], ingest_threads=5):
            ds = NdtDataset.from_records(ctx, rec, 304, offs, dtype)
        from nonlinear_optimizer_for_slam_amd import api
        got[mode] = (api.download(ds), ds.accumulate6(R_TEST, T_TEST, ("exponential", 1.0, 1.0)))
        ds.close()
    assert np.array_equal(got["pack"][0], got["unpack"][0])
    assert np.array_equal(got["pack"][1], got["unpack"][1])
    if dtype == "f64":
        assert np.array_equal(got["pack"][0], planes)


@pytest.mark.gpu
def test_two_threads_each_owning_a_solver_get_the_single_thread_answers(oracle):
    """the drop-in solver objects share one context per device list (AcquireRuntime); every C-ABI entry point
    now holds the context's lock, so threads that each own a solver object take turns instead of racing on the
    per-slot sequence words / partial rows / loop state.  The reference's solver objects are independent per instance.
    Two threads x 12 solves each on different data (one-launch form, launch-per-iteration form and a reprojection
    solve interleaved) must reproduce, bit for bit, what each gets alone."""
    import threading
    from nonlinear_optimizer_for_slam_amd import solvers, synth
    jobs = {
        "ndt_small": ("ndt", synth.ndt_planes(20_000, 700), ("exponential", 1.0, 1.0)),
        "ndt_large": ("ndt", synth.ndt_planes(300_000, 9000), ("huber", 0.8)),
        "reproj": ("reproj", synth.reproj_planes(50_000), ("huber", synth.REPROJ_HUBER_THRESHOLD)),
    }

    def solve_once(kind, planes, loss):
        if kind == "ndt":
            s = solvers.MahalanobisDistanceMinimizerHip()
            s.SetLossFunction(loss)
            pose = solvers.Pose()
            assert s.Solve(solvers.Options(), planes, pose)
        else:
            s = solvers.ReprojectionErrorMinimizerHip()
            s.SetLossFunction(loss)
            pose = solvers.Pose()
            assert s.Solve(solvers.Options(), planes, synth.REPROJ_INTRINSICS, pose)
        return pose.R.copy(), pose.t.copy(), s.report.iterations, s.report.printed_cost

    alone = {k: solve_once(*v) for k, v in jobs.items()}
    errors = []

    def worker(order):
        try:
            for _ in range(4):
                for k in order:
                    R, t, it, cost = solve_once(*jobs[k])
                    if not (np.array_equal(R, alone[k][0]) and np.array_equal(t, alone[k][1]) and it == alone[k][2]
                            and cost == alone[k][3]):
                        errors.append((k, it, alone[k][2], float(np.max(np.abs(t - alone[k][1])))))
        except Exception as exc:  # noqa: BLE001
            errors.append(repr(exc))

    threads = [threading.Thread(target=worker, args=(o,)) for o in
               (("ndt_small", "reproj", "ndt_large"), ("ndt_large", "ndt_small", "reproj"))]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors[:5]


# ---------------------------------------------------------------- the bench line

def test_bench_line_carries_the_contract_fields():
    """`python bench.py` prints ONE JSON line; the driver and the judge read fixed keys from it.  A short run (1 M
    correspondences, small CPU sample) must carry every one of them with sane values: throughput = points / time,
    roofline fraction = algorithmic bytes / kernel time / 8 TB/s, a CPU baseline with its core count and sample."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "30", "--warmup", "5", "--repeats", "2",
           "--points", "1000000", "--prewarm-ms", "50", "--cpu-seconds", "0.5", "--no-strong-baseline", "--no-cold"]
    run = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=root)
    assert run.returncode == 0, run.stderr[-2000:]
    lines = [l for l in run.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines  # one line on stdout, everything else on stderr
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 30 and d["warmup"] == 5 and d["higher_is_better"] is True
    assert d["dtype"] == "f64" and d["vs_baseline"] is None and "synthetic" in d["data"] and "workload" in d["config"]
    assert abs(d["value"] - 1_000_000 / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]
    r = d["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel"):
        assert key in r, key
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and 0.0 < r["frac"] < 1.0
    assert abs(r["algorithmic_bytes_per_launch"] - 120 * 1_000_000) < 1  # 120 B per fp64 NDT correspondence
    assert "Ndt6Problem<double" in r["kernel"]  # the symbol the library launched, not a description
    c = d["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in c, key
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0


@pytest.mark.gpu
def test_default_bench_line_carries_every_stage_with_roofline_and_cpu_baseline():
    """`python bench.py` with no flags — the command the driver runs: besides the headline, `other_configs` holds one
    driver-timed entry per single-GPU configuration and per stage either side of the hot path, each with a
    roofline object priced on DESIGN.md §3.1's algorithmic bytes; the stages with a CPU baseline of the same work; the
    reference's two wrappers with their COST lines checked; `summary` is the LAST key and stays below 1 KB; the CPU legs of
    the stages take ≤ 10 s together."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    run = subprocess.run([sys.executable, os.path.join(root, "bench.py")], capture_output=True, text=True, timeout=900, cwd=root)
    assert run.returncode == 0, run.stderr[-2000:]
    lines = [l for l in run.stdout.splitlines() if l.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert list(d)[-1] == "summary" and len(json.dumps(d["summary"])) <= 1024
    oc = d["other_configs"]
    stages = ["pgo_linearize", "pgo_matvec", "pgo_pcg_iteration", "matcher_10M_unsorted", "matcher_10M_cell_sorted",
              "matcher_10M_cell_sorted_ids", "ingest_10M_records_raw", "ingest_10M_records_host_pack", "ingest_10M_planes",
              "mapbuild_10M_100k_voxels", "mapbuild_10M_796k_voxels", "mapbuild_reference_scene_exact",
              "mapbuild_reference_scene_wave_parallel"]
    for key in stages:
        r = oc[key]["roofline"]
        assert r["bound"] in ("hbm", "pcie") and 0.0 < r["frac"] <= 1.05 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9, key
        assert oc[key]["algorithmic_bytes"] > 0
    for key in ("pgo_linearize", "matcher_10M_cell_sorted", "ingest_10M_records_host_pack", "mapbuild_10M_100k_voxels",
                "mapbuild_reference_scene_exact", "reference_wrapper_ndt", "reference_wrapper_reproj"):
        c = oc[key]["cpu_baseline"]
        assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0 and c["sample"], key
    assert oc["reference_wrapper_ndt"]["cost_lines_equal_the_captured_run"] is True
    assert oc["reference_wrapper_reproj"]["cost_line_equals_the_captured_run"] is True
    assert oc["pgo_pcg_iteration"]["launches_per_iteration"] <= 6 and oc["pgo_pcg_iteration"]["ms"]["median"] <= 0.55
    solver = [k for k in oc if k.split(" ")[0] in ("reproj_f64_2M", "ndt6_f64_100k", "ndt6_f32_10M", "ndt3_f64_10M", "indexed_10M")]
    assert len(solver) == 5 and all("roofline" in oc[k] and "cpu_baseline" in oc[k] for k in solver)
    assert d["cpu_seconds_of_the_stage_legs"] <= 10.0
