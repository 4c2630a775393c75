"""CPU-only tests: the C ABI library loads and exports every declared symbol, fails loudly
without a GPU, and the C++ host layer (loss descriptor recovery, damped step, LM loop) agrees
with the oracle when the oracle is injected as the accumulate callback."""
import ctypes
import os
import re

import numpy as np
import pytest

from nonlinear_optimizer_for_slam_amd import _lib, distributed, solvers, synth
from tests import helpers

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "nos.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(nos_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_symbol_of_the_header():
    lib = _lib.hip_lib()
    declared = _declared_symbols()
    assert len(declared) >= 25
    missing = [s for s in declared if not hasattr(lib, s)]
    assert not missing, missing
    assert sorted(_lib.C_ABI_SYMBOLS) == declared
    assert b"gfx950" in lib.nos_version()


def test_no_gpu_means_loud_failure_not_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from nonlinear_optimizer_for_slam_amd import Context
    with pytest.raises(_lib.NosError) as ei:
        Context((0,))
    assert ei.value.status == 2  # NOS_ERR_NO_DEVICE
    assert "no CPU fallback" in str(ei.value)
    # the C++ drop-in class reports failure (Solve() == false) instead of computing on the CPU
    planes = synth.ndt_planes(64, 4)
    solver = solvers.MahalanobisDistanceMinimizerHip()
    pose = solvers.Pose()
    assert solver.Solve(solvers.Options(), planes, pose) is False
    assert np.array_equal(pose.R, np.eye(3)) and np.all(pose.t == 0)


@pytest.mark.parametrize("loss", [None, ("exponential", 1.0, 1.0), ("exponential", 0.3, 2.5),
                                  ("huber", 1.0), ("huber", synth.REPROJ_HUBER_THRESHOLD)])
def test_loss_descriptor_recovery_through_private_members(loss):
    ok, kind, a, b = solvers.describe_loss(loss)
    assert ok
    if loss is None:
        assert kind == 0
    elif loss[0] == "exponential":
        assert kind == 1 and a == loss[1] and abs(b - loss[2]) <= 2e-16 * loss[2]
    else:
        assert kind == 2 and a == loss[1]


def test_loss_constructors_validate_like_the_reference():
    # loss_function.h:24-25,53-54 throw std::out_of_range; the shim turns that into failure
    assert solvers.describe_loss(("exponential", -1.0, 1.0))[0] is False
    assert solvers.describe_loss(("huber", 0.0))[0] is False


def test_damped_step_matches_oracle(oracle):
    planes = synth.ndt_planes(3000, 100)
    out28 = oracle.ndt6_accumulate(planes, np.eye(3), np.zeros(3), ("exponential", 1.0, 1.0))
    step = np.zeros(6)
    ok = synth.host_lib().nos_host_damped_step6(out28.ctypes.data_as(_lib.c_double_p), ctypes.c_double(1e-3),
                                               step.ctypes.data_as(_lib.c_double_p))
    assert ok == 1
    for solver in (0, 1):
        np.testing.assert_allclose(step, oracle.lm_step6(out28, 1e-3, solver), rtol=1e-9, atol=1e-13)
    out10 = oracle.ndt3_accumulate(planes, np.eye(2), np.zeros(2), None)
    step3 = np.zeros(3)
    ok = synth.host_lib().nos_host_damped_step3(out10.ctypes.data_as(_lib.c_double_p), ctypes.c_double(1e-3),
                                               step3.ctypes.data_as(_lib.c_double_p))
    assert ok == 1
    np.testing.assert_allclose(step3, oracle.lm_step3(out10, 1e-3), rtol=1e-9, atol=1e-13)


def test_damped_step_rejects_indefinite_matrix():
    out28 = np.zeros(28)
    out28[21:27] = 1.0  # H = 0
    step = np.zeros(6)
    assert synth.host_lib().nos_host_damped_step6(out28.ctypes.data_as(_lib.c_double_p), ctypes.c_double(1e-3),
                                                  step.ctypes.data_as(_lib.c_double_p)) == 0


class _OracleAssembler:
    """Test double: the oracle plays the role of the GPU accumulate."""

    def __init__(self, fn):
        self.fn = fn
        self.calls = 0

    def accumulate(self, R, t):
        self.calls += 1
        return self.fn(R, t)


def test_cpp_lm_loop_with_oracle_accumulate_matches_oracle_solve(oracle):
    planes = synth.ndt_planes(6000, 300)
    loss = ("exponential", 1.0, 1.0)
    want = oracle.ndt6_solve(planes, np.zeros(3), np.eye(3), loss=loss, linear_solver=1)
    asm = _OracleAssembler(lambda R, t: oracle.ndt6_accumulate(planes, R, t, loss))
    pose = solvers.Pose()
    rep = distributed.solve_ndt6(asm, solvers.Options(), pose)
    assert rep.iterations == want["iterations"]
    assert asm.calls == min(want["iterations"] + 1, 40)
    dt, dq = helpers.pose_delta(pose.R, pose.t, want["R"], want["t"])
    assert dt < 1e-10 and dq < 1e-10
    assert abs(rep.printed_cost - want["printed_cost"]) <= 1e-9 * abs(want["printed_cost"])


def test_cpp_lm3_loop_with_oracle_accumulate_matches_oracle_solve(oracle):
    planes = synth.ndt_planes(6000, 300)
    # planar scene: keep only yaw + xy of the truth so the 3-DoF model can fit
    want = oracle.ndt3_solve(planes, np.zeros(3), np.eye(3), loss=None)
    asm = _OracleAssembler(lambda R2, t2: oracle.ndt3_accumulate(planes, R2, t2, None))
    pose = solvers.Pose()
    rep = distributed.solve_ndt3(asm, solvers.Options(), pose)
    assert rep.iterations == want["iterations"]
    np.testing.assert_allclose(pose.t[:2], want["t"][:2], atol=1e-10)
    np.testing.assert_allclose(pose.R[:2, :2], want["R"][:2, :2], atol=1e-10)
    assert pose.t[2] == 0.0 and pose.R[2, 2] == 1.0  # z / roll / pitch pass through


def test_reprojection_known_answer_through_cpp_lm_loop(oracle):
    """The C++ host LM loop reproduces the reference's captured run (results/reproj_amd64.txt:5)
    when fed the oracle's sums: `COST: 2.33228e-11, iter: 6`."""
    planes, (fx, fy, cx, cy), _, _ = helpers.reference_reprojection_scene()
    intr = [1 / fx, 1 / fy, cx, cy]
    loss = ("exponential", 1.0, 1.0)
    asm = _OracleAssembler(lambda R, t: oracle.reproj_accumulate(planes, R, t, intr, loss))
    pose = solvers.Pose()
    rep = distributed.solve_ndt6(asm, solvers.Options(), pose)
    assert rep.iterations == 6
    assert "%.6g" % rep.printed_cost == "2.33228e-11"
    inv = pose.inverse()
    np.testing.assert_allclose(inv.t, [-0.1, 0.123, -0.5], atol=5e-7)


def test_shard_ranges_cover_everything_once():
    for n in (0, 1, 7, 8, 1000, 10_000_019):
        for world in (1, 2, 3, 8):
            spans = [distributed.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            for (a0, a1), (b0, b1) in zip(spans, spans[1:]):
                assert a1 == b0 and a0 <= a1 and b0 <= b1


def test_synthetic_generator_is_thread_count_independent_and_seeded():
    a = synth.ndt_planes(200_000, 5000, threads=1)
    b = synth.ndt_planes(200_000, 5000, threads=7)
    assert np.array_equal(a, b)
    c = synth.ndt_planes(1000, 50, seed=1)
    d = synth.ndt_planes(1000, 50, seed=2)
    assert not np.array_equal(c, d)
    # S = diag(lambda^-1/2) Q  → S S^T = diag(1/lambda), eigenvalue flooring 0.01 * l3
    S = a[6:15, 0].reshape(3, 3)
    G = S @ S.T
    assert np.allclose(G - np.diag(np.diag(G)), 0.0, atol=1e-9 * np.max(G))
    lam = 1.0 / np.diag(G)
    assert 0.02 <= lam[2] <= 0.10 and lam[0] >= 0.01 * lam[2] * (1 - 1e-12)


def test_reference_record_layout_is_304_bytes():
    # MDM/types.h:11-26: 24 B point + 280 B NDT; the stand-in keeps the same record size
    assert synth.host_lib().nos_host_sizeof_ndt_correspondence() == 304


def test_header_is_plain_c(tmp_path):
    """include/nos.h is the drop-in boundary: it must compile as C99 (no C++ / torch types in signatures) and a C
    program must link against libnos_hip.so using nothing but that header."""
    import subprocess
    src = tmp_path / "abi.c"
    src.write_text(
        '#include <stdio.h>\n#include "nos.h"\n'
        "int main(void) {\n"
        "  nos_ctx* ctx = 0; int dev = 0;\n"
        "  int rc = nos_ctx_create(&dev, 1, &ctx);\n"
        '  printf("%s|%d|%s\\n", nos_version(), rc, nos_status_string(rc));\n'
        "  if (rc == NOS_OK) nos_ctx_destroy(ctx);\n"
        "  return 0;\n}\n")
    exe = tmp_path / "abi"
    csrc = os.path.join(ROOT, "nonlinear_optimizer_for_slam_amd", "csrc")
    subprocess.check_call(["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"),
                           str(src), "-o", str(exe), "-L", csrc, "-lnos_hip", "-Wl,-rpath," + csrc])
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120).stdout
    assert "gfx950" in out and ("|0|ok" in out or "|2|no HIP device" in out), out
