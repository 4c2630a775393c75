"""Pose-graph optimisation: oracle self-checks on the CPU, GPU parity against the oracle.

Parity status: the reference's analytic PGO is an empty loop and results/ holds no PGO run, so this row is
unpinned against reference outputs; it is pinned by the literal residual restatement, finite differences of
its Jacobians and a direct sparse solve (oracle/oracle_pgo.py)."""
import numpy as np
import pytest

from oracle import oracle_pgo as op


def _fd_jacobians(pr, qr, pq, qq, tm, qm, h=1e-6):
    Jr = np.zeros((6, 6))
    Jq = np.zeros((6, 6))
    for k in range(6):
        d = np.zeros(6)
        d[k] = h

        def res(sign, which):
            pr2, qr2, pq2, qq2 = pr.copy(), qr.copy(), pq.copy(), qq.copy()
            dd = sign * d
            if which == 0:
                pr2 = pr + dd[:3]
                qr2 = op.qmul(qr, op.qexp(dd[3:]))
            else:
                pq2 = pq + dd[:3]
                qq2 = op.qmul(qq, op.qexp(dd[3:]))
            return op.edge_residual(pr2, qr2, pq2, qq2, tm, qm)[0]

        Jr[:, k] = (res(+1, 0) - res(-1, 0)) / (2 * h)
        Jq[:, k] = (res(+1, 1) - res(-1, 1)) / (2 * h)
    return Jr, Jq


def test_analytic_jacobians_match_finite_differences():
    rng = np.random.default_rng(1)
    for _ in range(20):
        qr, qq, qm = (op.qexp(rng.normal(scale=0.8, size=3)) for _ in range(3))
        pr, pq, tm = rng.normal(size=3), rng.normal(size=3), rng.normal(size=3)
        r, e = op.edge_residual(pr, qr, pq, qq, tm, qm)
        Jr, Jq = op.edge_jacobians(qr, tm, qm, e)
        Fr, Fq = _fd_jacobians(pr, qr, pq, qq, tm, qm)
        np.testing.assert_allclose(Jr, Fr, atol=2e-8)
        np.testing.assert_allclose(Jq, Fq, atol=2e-8)


def test_residual_is_zero_for_consistent_measurement():
    rng = np.random.default_rng(2)
    qr, qq = op.qexp(rng.normal(size=3)), op.qexp(rng.normal(size=3))
    pr, pq = rng.normal(size=3), rng.normal(size=3)
    tm = op.qrot(qr).T @ (pq - pr)
    qm = op.qmul(op.qconj(qr), qq)
    r, _ = op.edge_residual(pr, qr, pq, qq, tm, qm)
    np.testing.assert_allclose(r, 0.0, atol=1e-14)


def test_oracle_solves_the_reference_demo_and_switches_off_the_outlier():
    """pose_graph_optimizer/tests/simple_optimization_test.cc: 80 poses, 79 + 4 constraints, last loop constraint
    is an identity outlier with a free switch → poses return to the truth, the outlier's switch → ~0."""
    true, noisy, ref, qry, meas, free = op.reference_test_scene()
    fixed = np.zeros(80, dtype=bool)
    fixed[0] = True
    g = op.Graph(noisy, ref, qry, meas, None, free, fixed)
    c0 = g.cost()
    it, hist = g.optimize(max_iterations=60, gradient_tolerance=1e-12, parameter_tolerance=1e-12)
    assert g.cost() < 1e-6 * c0
    np.testing.assert_allclose(g.poses[:, :3], true[:, :3], atol=2e-4)
    assert abs(g.sw[82]) < 1e-3 and np.all(np.abs(g.sw[79:82] - 1.0) < 1e-3)


# ------------------------------------------------------------------------------------------ GPU

def _graph_pair(ctx, d, switch_free=None, switch_init=None):
    from nonlinear_optimizer_for_slam_amd import pgo
    cpu = op.Graph(d["init"], d["ref"], d["qry"], d["meas"], switch_init, switch_free, d["fixed"])
    gpu = pgo.PoseGraph(ctx, d["init"], d["ref"], d["qry"], d["meas"], switch_init, switch_free, d["fixed"])
    return cpu, gpu


@pytest.mark.gpu
@pytest.mark.parametrize("n,with_switches", [(2, False), (50, False), (400, True)])
def test_gpu_linearisation_matches_explicit_assembly(ctx, n, with_switches):
    d = op.random_graph(n, 3, seed=n)
    m = d["ref"].size
    free = init = None
    if with_switches:
        free = (np.arange(m) >= n - 1).astype(np.uint8)          # loop constraints carry free switches
        init = np.where(free, 0.7, 1.0)
    cpu, gpu = _graph_pair(ctx, d, free, init)
    H, g, cost = cpu.linearize()
    c_gpu, gnorm = gpu.linearize()
    assert abs(c_gpu - cost) <= 1e-12 * cost
    g_gpu = gpu.vector("gradient")
    np.testing.assert_allclose(g_gpu, g, rtol=0, atol=1e-11 * np.max(np.abs(g)))
    assert abs(gnorm - np.linalg.norm(g)) <= 1e-12 * np.linalg.norm(g)
    # diagonal blocks
    hd = gpu.vector("hdiag").reshape(21, n)
    Hd = H.toarray() if n <= 400 else None
    k = 0
    for r in range(6):
        for c in range(r, 6):
            want = np.array([Hd[r * n + i, c * n + i] for i in range(n)])
            np.testing.assert_allclose(hd[k], want, rtol=0, atol=1e-11 * np.max(np.abs(Hd)))
            k += 1
    # matrix-free product against the explicit matrix, damped and undamped
    rng = np.random.default_rng(0)
    for lam in (0.0, 1e-3):
        x = rng.normal(size=6 * n + m)
        if free is None:
            x[6 * n:] = 0.0
        else:
            x[6 * n:][free == 0] = 0.0
        for i in np.nonzero(d["fixed"])[0]:
            x[[kk * n + i for kk in range(6)]] = 0.0
        want = cpu.damped(H, lam) @ x
        got = gpu.matvec(lam, x)
        np.testing.assert_allclose(got, want, rtol=0, atol=1e-11 * np.max(np.abs(want)))
    gpu.close()


@pytest.mark.gpu
def test_gpu_pcg_step_matches_direct_sparse_solve(ctx):
    d = op.random_graph(300, 3, seed=5)
    cpu, gpu = _graph_pair(ctx, d)
    H, g, _ = cpu.linearize()
    gpu.linearize()
    for lam in (1e-3, 1e-6):
        want = cpu.solve_step(H, g, lam)
        it, res, step = gpu.solve(lam, 2000, 1e-13)
        got = gpu.vector("step")
        assert res <= 1e-12
        np.testing.assert_allclose(got, want, rtol=0, atol=1e-8 * np.max(np.abs(want)))
        assert abs(step - np.linalg.norm(want)) <= 1e-8 * np.linalg.norm(want)
    gpu.close()


@pytest.mark.gpu
def test_gpu_lm_loop_tracks_the_oracle_loop(ctx):
    d = op.random_graph(200, 3, seed=9)
    cpu, gpu = _graph_pair(ctx, d)
    it_c, hist_c = cpu.optimize(max_iterations=15, gradient_tolerance=1e-10, parameter_tolerance=1e-10)
    it_g, hist_g = gpu.optimize(max_iterations=15, gradient_tolerance=1e-10, parameter_tolerance=1e-10,
                                pcg_iterations=3000, pcg_tolerance=1e-13)
    assert it_c == it_g
    for a, b in zip(hist_c, hist_g):
        assert abs(a[0] - b[0]) <= 1e-7 * max(a[0], 1e-12) + 1e-16
    poses, _ = gpu.state()
    np.testing.assert_allclose(poses, cpu.poses, atol=1e-8)
    # and it actually optimises: the cost collapses to the measurement-noise floor
    assert hist_g[-1][0] < 0.02 * hist_g[0][0]
    gpu.close()


@pytest.mark.gpu
def test_gpu_reference_demo_with_outlier(ctx):
    """The reference's 80-pose demo on the GPU path: truth recovered, outlier loop constraint switched off."""
    from nonlinear_optimizer_for_slam_amd import pgo
    true, noisy, ref, qry, meas, free = op.reference_test_scene()
    fixed = np.zeros(80, dtype=np.uint8)
    fixed[0] = 1
    g = pgo.PoseGraph(ctx, noisy, ref, qry, meas, None, free, fixed)
    c0, _ = g.linearize()
    g.optimize(max_iterations=60, gradient_tolerance=1e-12, parameter_tolerance=1e-12, pcg_iterations=3000,
               pcg_tolerance=1e-12)
    c1, _ = g.linearize()
    poses, sw = g.state()
    assert c1 < 1e-6 * c0
    np.testing.assert_allclose(poses[:, :3], true[:, :3], atol=2e-4)
    assert abs(sw[82]) < 1e-3 and np.all(np.abs(sw[79:82] - 1.0) < 1e-3)
    g.close()


@pytest.mark.gpu
def test_cpp_drop_in_class_on_the_reference_demo():
    """PoseGraphOptimizerHip through the reference's call sequence (pose_graph_optimizer/tests/
    simple_optimization_test.cc:124-139): SetPose x80, SetPoseConstant(0), SetConstraint x83, Solve(options)."""
    from nonlinear_optimizer_for_slam_amd import pgo, solvers
    true, noisy, ref, qry, meas, free = op.reference_test_scene()
    poses = [noisy[i].copy() for i in range(80)]
    opt = pgo.PoseGraphOptimizerHip()
    for i in range(80):
        opt.SetPose(10 * i + 3, poses[i])            # arbitrary integer keys, like the reference's bimap
    opt.SetPoseConstant(3)
    for e in range(83):
        opt.SetConstraint(10 * int(ref[e]) + 3, 10 * int(qry[e]) + 3, meas[e], is_loop=bool(free[e]))
    assert opt.Solve(solvers.Options(60, 1e-12, 1e-12))
    got = np.stack(poses)
    np.testing.assert_allclose(got[:, :3], true[:, :3], atol=2e-4)
    np.testing.assert_allclose(np.abs(got[:, 3]), 1.0, atol=1e-6)
    assert abs(opt.switch_parameters[82]) < 1e-3
    assert opt.report["final_cost"] < 1e-6 * opt.report["initial_cost"]


@pytest.mark.gpu
def test_large_graph_generator_and_gpu_loop(ctx):
    """configs[4] shape scaled to 20 k poses / ~80 k constraints: the GPU LM loop reduces the cost to the
    measurement-noise floor and agrees with the explicit sparse oracle on the first linearisation."""
    from nonlinear_optimizer_for_slam_amd import pgo, synth
    d = synth.pose_graph(20_000, 3)
    assert d["ref"].size > 75_000
    g = pgo.PoseGraph(ctx, d["init"], d["ref"], d["qry"], d["meas"], None, None, d["fixed"])
    c0, g0 = g.linearize()
    it, hist = g.optimize(max_iterations=10, gradient_tolerance=1e-9, parameter_tolerance=1e-9, pcg_iterations=400,
                          pcg_tolerance=1e-8)
    c1, g1 = g.linearize()
    assert c1 < 0.02 * c0 and g1 < 1e-3 * g0
    g.close()


@pytest.mark.gpu
def test_configs4_full_size_every_pcg_solve_converges_and_the_loop_descends(ctx):
    """BASELINE.json configs[4] at FULL size: 1 M poses / ~4 M relative-pose constraints.  With the two-level
    preconditioner (rigid-motion coarse space over aggregates of consecutive poses + block-Jacobi, csrc/pgo_coarse_kernels.hpp)
    EVERY linear solve of the LM loop reaches 1e-6 well inside the 300-iteration cap (round 1's block-Jacobi PCG ran into
    the cap from the third LM iteration on), the cost never increases and the gradient norm falls by more than 1e-3."""
    from nonlinear_optimizer_for_slam_amd import pgo, synth
    d = synth.pose_graph(1_000_000, 3)
    assert 3_900_000 < d["ref"].size < 4_100_000
    g = pgo.PoseGraph(ctx, d["init"], d["ref"], d["qry"], d["meas"], None, None, d["fixed"])
    it, hist = g.optimize(max_iterations=6, gradient_tolerance=1e-9, parameter_tolerance=1e-9, pcg_iterations=300,
                          pcg_tolerance=1e-6)
    costs = [h[0] for h in hist]
    assert all(h[3] < 200 and h[4] <= 1e-6 for h in hist), [(h[3], h[4]) for h in hist]   # iterations, relative residual
    assert all(b <= a * (1 + 1e-12) for a, b in zip(costs, costs[1:])), costs
    c_end, g_end = g.linearize()
    assert c_end <= costs[-1] and c_end < 3e-3 * costs[0] and g_end < 1e-3 * hist[0][1], (c_end, g_end, hist[0])
    # the same linearisation through round 1's preconditioner alone: does not get there
    with ctx.options(pgo_precond=0):
        it_bj, res_bj, _ = g.solve(1e-6, 300, 1e-6)
    it_2l, res_2l, _ = g.solve(1e-6, 300, 1e-6)
    assert it_bj == 300 and res_bj > 1e-6 and it_2l < 200 and res_2l <= 1e-6, (it_bj, res_bj, it_2l, res_2l)
    g.close()


@pytest.mark.gpu
def test_two_level_preconditioner_is_a_preconditioner_not_a_different_solve(ctx):
    """Same converged step with and without the coarse level (it changes the iteration count, not the answer), on a graph
    with loop closures that span many aggregates (their coupling is dropped from the coarse operator only)."""
    d = op.random_graph(3000, 3, seed=4)
    rng = np.random.default_rng(0)
    extra_ref = rng.integers(0, 1500, 40)
    extra_qry = extra_ref + rng.integers(800, 1400, 40)       # long loop closures
    ref = np.concatenate([d["ref"], extra_ref]).astype(np.int32)
    qry = np.concatenate([d["qry"], extra_qry]).astype(np.int32)
    meas = np.concatenate([d["meas"], d["meas"][:40]])        # inconsistent on purpose: only the linear algebra matters
    from nonlinear_optimizer_for_slam_amd import pgo
    g = pgo.PoseGraph(ctx, d["init"], ref, qry, meas, None, None, d["fixed"])
    g.linearize()
    it2, res2, _ = g.solve(1e-3, 3000, 1e-12)
    x2 = g.vector("step").copy()
    with ctx.options(pgo_precond=0):
        it1, res1, _ = g.solve(1e-3, 3000, 1e-12)
    x1 = g.vector("step").copy()
    assert res1 <= 1e-12 and res2 <= 1e-12 and it2 < it1, (it1, res1, it2, res2)
    np.testing.assert_allclose(x2, x1, rtol=0, atol=1e-6 * np.max(np.abs(x1)))
    g.close()


@pytest.mark.gpu
def test_pcg_with_device_resident_scalars_equals_host_scalar_pcg(ctx):
    """The CG scalars (alpha, beta, r.z) live on the device by default and the host looks at |r| every 8th iteration;
    the context option pgo_host_scalars = 1 restores the per-iteration readback.  Same arithmetic: at a fixed iteration count (a multiple
    of 8, tolerance 0) the two must agree to rounding, and both must reach the direct solve."""
    import os
    d = op.random_graph(400, 3, seed=9)
    cpu, gpu = _graph_pair(ctx, d)
    H, g, _ = cpu.linearize()
    gpu.linearize()
    it_dev, res_dev, _ = gpu.solve(1e-4, 64, 0.0)
    x_dev = gpu.vector("step").copy()
    with ctx.options(pgo_host_scalars=1):
        it_host, res_host, _ = gpu.solve(1e-4, 64, 0.0)
    x_host = gpu.vector("step").copy()
    assert it_dev == it_host == 64
    np.testing.assert_allclose(x_dev, x_host, rtol=0, atol=1e-12 * np.max(np.abs(x_host)))
    assert res_dev == pytest.approx(res_host, rel=1e-6)
    it, res, _ = gpu.solve(1e-4, 4000, 1e-13)
    want = cpu.solve_step(H, g, 1e-4)
    np.testing.assert_allclose(gpu.vector("step"), want, rtol=0, atol=1e-8 * np.max(np.abs(want)))
    assert it % 8 == 0 or it == 4000 or res <= 1e-13
    gpu.close()


@pytest.mark.gpu
@pytest.mark.parametrize("with_switches", [False, True])
def test_block_local_product_equals_owner_computes_product(ctx, with_switches):
    """Round 4's product visits every constraint once per block of 128 poses (entry lists built by nos_pgo_create, the two
    s J^T v contributions parked in LDS slots and added in adjacency order); round 2's walks every pose's constraints and
    re-derives each constraint at both ends.  Same sums in the same order — they differ only in where s J^T v is rounded —
    on a graph whose blocks have boundary constraints, fixed poses and (second case) free switches; a block boundary that
    falls inside the last, partly filled block is part of the graph size chosen."""
    from nonlinear_optimizer_for_slam_amd import pgo
    n = 128 * 9 + 37
    d = op.random_graph(n, 3, seed=12)
    m = d["ref"].size
    fixed = d["fixed"].copy()
    fixed[[200, 513]] = 1
    free = init = None
    if with_switches:
        free = (np.arange(m) % 3 == 1).astype(np.uint8)
        init = np.where(free, 0.8, 1.0)
    rng = np.random.default_rng(3)
    x = rng.normal(size=6 * n + m)
    if free is None:
        x[6 * n:] = 0.0
    else:
        x[6 * n:][free == 0] = 0.0
    for i in np.nonzero(fixed)[0]:
        x[[k * n + i for k in range(6)]] = 0.0
    got = {}
    for block in (1, 0):
        with ctx.options(pgo_block=block):
            g = pgo.PoseGraph(ctx, d["init"], d["ref"], d["qry"], d["meas"], init, free, fixed)
        g.linearize()
        got[block] = (g.matvec(1e-3, x), g.solve(1e-3, 400, 1e-10), g.vector("step").copy())
        g.close()
    y1, y0 = got[1][0], got[0][0]
    np.testing.assert_allclose(y1, y0, rtol=0, atol=1e-13 * np.max(np.abs(y0)))
    assert got[1][1][0] == got[0][1][0]                      # PCG iterations
    np.testing.assert_allclose(got[1][2], got[0][2], rtol=0, atol=1e-9 * np.max(np.abs(got[0][2])))


@pytest.mark.gpu
def test_coarse_operator_assembled_directly_equals_the_probed_one(ctx):
    """The coarse operator A_c = P^T H' P of the two-level preconditioner: round 2 probed it with 18 masked matrix-free
    products per solve, round 4 assembles its three block diagonals in three sweeps over the constraints
    (pgo_coarse_assemble_kernel).  Same operator up to rounding: the PCG needs the same number of iterations and reaches the
    same step, with loop closures that span many aggregates (dropped from H' in both forms) and fixed poses in the graph."""
    from nonlinear_optimizer_for_slam_amd import pgo
    d = op.random_graph(4000, 3, seed=6)
    rng = np.random.default_rng(1)
    extra_ref = rng.integers(0, 2000, 30)
    extra_qry = extra_ref + rng.integers(900, 1900, 30)
    ref = np.concatenate([d["ref"], extra_ref]).astype(np.int32)
    qry = np.concatenate([d["qry"], extra_qry]).astype(np.int32)
    meas = np.concatenate([d["meas"], d["meas"][:30]])
    fixed = d["fixed"].copy()
    fixed[[777, 2048]] = 1
    g = pgo.PoseGraph(ctx, d["init"], ref, qry, meas, None, None, fixed)
    g.linearize()
    res = {}
    for probe in (0, 1):
        with ctx.options(pgo_coarse_probe=probe):
            it, rel, _ = g.solve(1e-3, 2000, 1e-11)
        res[probe] = (it, rel, g.vector("step").copy())
    assert res[0][1] <= 1e-11 and res[1][1] <= 1e-11
    assert abs(res[0][0] - res[1][0]) <= 8, (res[0][0], res[1][0])   # the host looks at |r| every 8th iteration
    np.testing.assert_allclose(res[0][2], res[1][2], rtol=0, atol=1e-7 * np.max(np.abs(res[1][2])))
    with ctx.options(pgo_precond=0):
        it_bj, _, _ = g.solve(1e-3, 4000, 1e-11)
    assert res[0][0] < it_bj                                        # and it is still a preconditioner worth having
    g.close()
