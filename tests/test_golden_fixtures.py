"""Committed golden fixtures (tests/golden/, made by tests/golden/make_golden.py): the oracle (CPU) and the HIP path
(GPU) are both held to them, so neither can drift silently; the reference's own known answers live in
reference_known_answers.json with their results/*.txt file:line."""
import json
import os

import numpy as np
import pytest

from tests import helpers

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
RL = {"none": None, "exponential": ("exponential", 1.0, 1.0)}


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(HERE, "hotpath_small.npz"))


@pytest.fixture(scope="module")
def known():
    return json.load(open(os.path.join(HERE, "reference_known_answers.json")))


def _losses(kind):
    from nonlinear_optimizer_for_slam_amd import synth
    if kind == "reproj":
        return {"none": None, "exponential": ("exponential", 1.0, 1.0), "huber": ("huber", synth.REPROJ_HUBER_THRESHOLD)}
    return {"none": None, "exponential": ("exponential", 1.0, 1.0), "huber": ("huber", 1.2)}


def test_fixture_inputs_are_what_the_generator_produces(gold):
    from nonlinear_optimizer_for_slam_amd import synth
    assert np.array_equal(gold["ndt_planes"], synth.ndt_planes(1000, 50, seed=20250912))
    assert np.array_equal(gold["reproj_planes"], synth.reproj_planes(1000, seed=20250912))


def test_oracle_reproduces_the_golden_sums(oracle, gold):
    from nonlinear_optimizer_for_slam_amd import synth
    for name, loss in _losses("ndt").items():
        np.testing.assert_allclose(oracle.ndt6_accumulate(gold["ndt_planes"], gold["R"], gold["t"], loss), gold["ndt6_" + name], rtol=1e-13, atol=1e-9)
        np.testing.assert_allclose(oracle.ndt3_accumulate(gold["ndt_planes"], gold["R2"], gold["t2"], loss), gold["ndt3_" + name], rtol=1e-13, atol=1e-9)
    for name, loss in _losses("reproj").items():
        np.testing.assert_allclose(oracle.reproj_accumulate(gold["reproj_planes"], gold["Rr"], gold["tr"], synth.REPROJ_INTR4, loss),
                                   gold["reproj_" + name], rtol=1e-13, atol=1e-12)
    sol = oracle.ndt6_solve(gold["ndt_planes"], np.zeros(3), np.eye(3), loss=("exponential", 1.0, 1.0), linear_solver=1)
    assert sol["iterations"] == int(gold["ndt6_solve_meta"][0])
    np.testing.assert_allclose(sol["t"], gold["ndt6_solve_t"], atol=1e-12)


def test_reference_known_answers_file_matches_the_tests_that_use_them(known, oracle):
    planes, (fx, fy, cx, cy), _, _ = helpers.reference_reprojection_scene()
    assert planes.shape[1] == known["scene_counts"]["reprojection_points"]["value"]
    res = oracle.reproj_solve(planes, [1 / fx, 1 / fy, cx, cy], np.zeros(3), np.eye(3), loss=tuple(known["reprojection_analytic"]["loss"]))
    assert "COST: %.6g, iter: %d" % (res["printed_cost"], res["iterations"]) == known["reprojection_analytic"]["cost_line"]


@pytest.mark.gpu
def test_gpu_reproduces_the_golden_sums(ctx, gold):
    from nonlinear_optimizer_for_slam_amd import NdtDataset, ReprojDataset, synth
    nd = NdtDataset.from_planes(ctx, gold["ndt_planes"], "f64")
    rp = ReprojDataset.from_planes(ctx, gold["reproj_planes"], "f64")
    for name, loss in _losses("ndt").items():
        helpers.assert_normal_equations_close(nd.accumulate6(gold["R"], gold["t"], loss), gold["ndt6_" + name], 6, 1e-10)
        helpers.assert_normal_equations_close(nd.accumulate3(gold["R2"], gold["t2"], loss), gold["ndt3_" + name], 3, 1e-10)
    for name, loss in _losses("reproj").items():
        helpers.assert_normal_equations_close(rp.accumulate(gold["Rr"], gold["tr"], synth.REPROJ_INTR4, loss), gold["reproj_" + name], 6, 1e-10)
    nd.close()
    rp.close()


@pytest.mark.gpu
def test_gpu_solve_reproduces_the_golden_poses(gold):
    from nonlinear_optimizer_for_slam_amd import solvers, synth
    s = solvers.MahalanobisDistanceMinimizerHip()
    s.SetLossFunction(("exponential", 1.0, 1.0))
    pose = solvers.Pose()
    assert s.Solve(solvers.Options(), gold["ndt_planes"], pose)
    assert s.report.iterations == int(gold["ndt6_solve_meta"][0])
    dt, dq = helpers.pose_delta(pose.R, pose.t, gold["ndt6_solve_R"], gold["ndt6_solve_t"])
    assert dt < 1e-9 and dq < 1e-9
    r = solvers.ReprojectionErrorMinimizerHip()
    r.SetLossFunction(("huber", synth.REPROJ_HUBER_THRESHOLD))
    pose = solvers.Pose()
    assert r.Solve(solvers.Options(), gold["reproj_planes"], synth.REPROJ_INTRINSICS, pose)
    assert r.report.iterations == int(gold["reproj_solve_meta"][0])
    dt, dq = helpers.pose_delta(pose.R, pose.t, gold["reproj_solve_R"], gold["reproj_solve_t"])
    assert dt < 1e-8 and dq < 1e-8
