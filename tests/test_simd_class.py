"""The semantics of the reference's fp32 ("SIMD") solver classes, selectable per dataset (nos_dataset_set_simd_class,
HipOptions::simd_class): tail drop to floor(N/8)*8, float lambda / previous_cost in the NDT loops, depth > 0 mask on the
weight only for reprojection.

What can and cannot be pinned: the lane arithmetic of those classes comes from the un-vendored `simd_helper` library —
above all its `simd::exp`, which the 6-DoF class's Exponential loss goes through (NO/loss_function.h:37) and whose
approximation error shows up as a systematic +4e-6 on the printed costs.  The 3-DoF class evaluates the loss with the
scalar `std::exp` per lane (MDM/..._3dof_simd.cc:131-137) and IS reproduced to every printed digit.  So, from the runs the
reference holds (results/maha_amd64_simple.txt:15-20, maha_3_vs_6_amd64.txt:12-17,25-31, maha_amd64.txt:16-51):
iteration counts, outer-iteration counts and the `COST: 3.40282e+38, iter: 0` lines (FLT_MAX: the float previous_cost
of a loop that converged in its first pass) are asserted exactly, the 3-DoF lines as strings, the 6-DoF costs to 1e-5
(measured 4.3e-6) and the poses to 1e-6 (measured 4.5e-7; the reference's own fp32 variants differ among themselves from
the 5th printed digit on, results/maha_amd64.txt:56-66).
"""
import json
import os

import numpy as np
import pytest

from nonlinear_optimizer_for_slam_amd import NdtDataset, ReprojDataset, solvers, synth
from oracle import oracle_np
from oracle import oracle_scene as scene
from tests import helpers

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
KNOWN = json.load(open(os.path.join(HERE, "golden", "reference_known_answers.json")))["captured_ndt_simd_runs"]
EXP = ("exponential", 1.0, 1.0)


@pytest.fixture(scope="module")
def points():
    return scene.generate_global_points_c()


@pytest.fixture(scope="module")
def ndt_map(points):
    return scene.build_ndt_map_eigen(points, 1.0)


@pytest.mark.parametrize("dtype", ["f32", "f64"])
@pytest.mark.parametrize("name", ["simple_6dof", "planar_3dof", "planar_6dof", "dense_6dof"])
def test_simd_class_semantics_track_the_captured_fp32_runs(oracle, points, ndt_map, name, dtype):
    local, _, _ = scene.captured_run_scan(points, name)
    dof = scene.CAPTURED_RUNS[name][3]
    cls = solvers.MahalanobisDistanceMinimizerHip if dof == 6 else solvers.MahalanobisDistanceMinimizerHip3DOF

    def solve(planes, R, t):
        s = cls(dtype=dtype, simd_class=True)  # the class drops the tail itself: all N correspondences go in
        s.SetLossFunction(EXP)
        pose = solvers.Pose(R, t)
        assert s.Solve(solvers.Options(), planes, pose)
        return pose.R, pose.t, s.report.printed_cost, s.report.iterations

    R, t, rounds, outer = scene.captured_run_icp(solve, ndt_map, local, stride=1)
    want = KNOWN[name]
    assert outer == want["outer_iter"]
    assert [i for _, i, _ in rounds] == [i for _, i in want["cost_lines"]]
    for (cost, _, _), (text, _) in zip(rounds, want["cost_lines"]):
        if text == "3.40282e+38":  # FLT_MAX: previous_cost is a float in these classes
            assert scene.printed(cost) == text
        elif dof == 3:
            assert scene.printed(cost) == text
        else:
            assert abs(cost - float(text)) <= 1e-5 * float(text) + 0.06, (name, cost, text)  # 0.06: the printed rounding
    q = oracle.quat_from_matrix(R)
    pose = np.array([t[0], t[1], t[2], q[1], q[2], q[3], q[0]])
    assert np.max(np.abs(pose - np.array(want["final_pose"]))) < 1e-6


def test_scalar_and_simd_semantics_differ_where_the_reference_classes_do(points, ndt_map):
    """First round of the simple scene: scalar class 17438.4 (floor(N/4)*4 of the captured build), fp32 class the same
    digits; second round 17394.5 against 17390.7 — the tail the fp32 class drops."""
    local, _, _ = scene.captured_run_scan(points, "simple_6dof")
    got = {}
    for simd in (False, True):
        def solve(planes, R, t, simd=simd):
            s = solvers.MahalanobisDistanceMinimizerHip(dtype="f64", simd_class=simd)
            s.SetLossFunction(EXP)
            pose = solvers.Pose(R, t)
            assert s.Solve(solvers.Options(), planes, pose)
            return pose.R, pose.t, s.report.printed_cost, s.report.iterations
        _, _, rounds, _ = scene.captured_run_icp(solve, ndt_map, local, stride=4 if not simd else 1, max_outer=2)
        got[simd] = [c for c, _, _ in rounds]
    assert scene.printed(got[False][1]) == "17394.5" and abs(got[True][1] - 17390.7) < 0.25


@pytest.mark.parametrize("loss", [None, ("huber", synth.REPROJ_HUBER_THRESHOLD), EXP])
def test_reprojection_simd_class_mask_and_cost_rules(ctx, oracle, loss):
    n = 40_000
    planes = synth.reproj_planes(n)
    planes[2, :900] = -np.abs(planes[2, :900]) - 0.5      # behind the camera: weight 0, loss still counted
    planes[0:2, 900:910] = 0.01                           # ten points on the axis at world depth ≈ 0.015:
    planes[2, 900:910] = -0.385                           # 0 < depth < 0.03 counts for the fp32 class, not for the scalar one
    R = helpers.rot_xyz(0.0, 0.01, -0.08)
    t = np.array([0.08, -0.1, 0.4])
    intr = np.array(synth.REPROJ_INTR4, dtype=np.float64)
    want = oracle_np.reproj_accumulate_simd_class(planes, R, t, intr, loss)
    scalar_rules = oracle.reproj_accumulate(planes, R, t, intr, loss)
    assert abs(want[27] - scalar_rules[27]) > 1e-3 * abs(want[27])  # the two rule sets really differ on this input
    for dtype, rtol in (("f64", 1e-10), ("f32", 1e-4)):
        ds = ReprojDataset.from_planes(ctx, planes, dtype).set_simd_class(True)
        helpers.assert_normal_equations_close(ds.accumulate(R, t, intr, loss), want, 6, rtol)
        ds.set_simd_class(False)
        helpers.assert_normal_equations_close(ds.accumulate(R, t, intr, loss), scalar_rules, 6, rtol if dtype == "f64" else 1e-4)
        ds.close()
    # the AVX2 restatement of the class (oracle/nos_oracle_avx.c, fp32 lanes and fp32 sums) agrees to its own accuracy
    avx = oracle.avx_reproj_accumulate(planes.astype(np.float32), R, t, intr, loss)
    helpers.assert_normal_equations_close(avx, want, 6, 5e-3)


def test_ndt_lambda_schedule_runs_in_float_under_simd_class(ctx):
    planes = synth.ndt_planes(40_000, 2000)
    ds = NdtDataset.from_planes(ctx, planes, "f32")
    R0, t0 = np.eye(3), np.zeros(3)
    plain = ds.solve6(R0, t0, EXP, max_iterations=7, gradient_tolerance=0.0, parameter_tolerance=0.0)[2]
    ds.set_simd_class(True)
    simd = ds.solve6(R0, t0, EXP, max_iterations=7, gradient_tolerance=0.0, parameter_tolerance=0.0)[2]
    # MDM/..._analytic_simd.cc:38-39,99-101 replayed on the costs the run itself reports: lambda = 0.001f, then
    # x (cost > previous_cost ? 2.0 : 0.6) with the product in double stored to float, clamped to [1e-6f, 1e-2f]
    lam, prev = np.float32(0.001), np.float32(np.finfo(np.float32).max)
    for c in simd["cost_history"]:
        lam = np.float32(np.float64(lam) * (2.0 if c > np.float64(prev) else 0.6))
        lam = min(max(lam, np.float32(1e-6)), np.float32(1e-2))
        prev = np.float32(c)
    assert len(simd["cost_history"]) == 7 and simd["final_lambda"] == float(lam)
    assert float(np.float32(simd["final_lambda"])) == simd["final_lambda"]     # a float value …
    assert float(np.float32(plain["final_lambda"])) != plain["final_lambda"]   # … which the double schedule's is not
    assert simd["printed_cost"] == float(np.float32(simd["printed_cost"]))  # previous_cost is a float too
    ds.close()
    # a loop that stops in its first pass prints FLT_MAX / DBL_MAX (results/maha_3_vs_6_amd64.txt:23,30)
    ds = NdtDataset.from_planes(ctx, planes, "f64")
    rep = ds.solve6(R0, t0, EXP, max_iterations=5, gradient_tolerance=1e30)[2]
    assert rep["iterations"] == 0 and rep["printed_cost"] == np.finfo(np.float64).max
    rep = ds.set_simd_class(True).solve6(R0, t0, EXP, max_iterations=5, gradient_tolerance=1e30)[2]
    assert rep["iterations"] == 0 and rep["printed_cost"] == float(np.finfo(np.float32).max)
    ds.close()


@pytest.mark.parametrize("dof", [6, 3])
def test_tail_drop_with_an_executor_follows_each_class(oracle, dof):
    """N = 8 * 3 * 41 + 21: not a multiple of 8 T for T = 3.  The 6-DoF SIMD class splits over its executor's threads and
    uses T * floor(floor(N/8)/T) * 8 correspondences (MDM/..._analytic_simd.cc:46-69); the 3-DoF SIMD class has no executor
    and always uses floor(N/8)*8 (MDM/..._analytic_3dof_simd.cc:83-86), whatever thread count the caller configured.  One
    LM iteration from the identity: the printed cost is the cost of exactly the correspondences the class keeps."""
    T = 3
    n = 8 * T * 41 + 21
    planes = synth.ndt_planes(n, 50)
    keep = {6: T * ((n // 8) // T) * 8, 3: (n // 8) * 8}[dof]
    assert {6: 984, 3: 1000}[dof] == keep and keep < n
    cls = solvers.MahalanobisDistanceMinimizerHip if dof == 6 else solvers.MahalanobisDistanceMinimizerHip3DOF
    s = cls(dtype="f64", simd_class=True, simd_class_threads=T)
    s.SetLossFunction(EXP)
    pose = solvers.Pose()
    assert s.Solve(solvers.Options(max_iterations=1, gradient_tolerance=0.0, parameter_tolerance=0.0), planes, pose)
    acc = oracle.ndt6_accumulate if dof == 6 else oracle.ndt3_accumulate
    R0 = np.eye(3) if dof == 6 else np.eye(2)
    t0 = np.zeros(3) if dof == 6 else np.zeros(2)
    want = acc(planes[:, :keep], R0, t0, EXP)[-1]
    other = acc(planes[:, :{6: 1000, 3: 984}[dof]], R0, t0, EXP)[-1]
    got = s.report.printed_cost  # previous_cost after the one iteration = its cost (kept as a float in these classes)
    assert abs(got - want) <= 1e-6 * want and abs(got - other) > 1e-4 * want
