"""configs[4]: pose-graph optimisation, 1 M poses / ~4 M constraints on one MI355X (timing aid)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nonlinear_optimizer_for_slam_amd import Context, pgo, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
t0 = time.perf_counter(); d = synth.pose_graph(n, 3); t1 = time.perf_counter()
print("generator: %.2f s, %d poses, %d constraints" % (t1 - t0, n, d["ref"].size), flush=True)
ctx = Context((0,))
t0 = time.perf_counter(); g = pgo.PoseGraph(ctx, d["init"], d["ref"], d["qry"], d["meas"], None, None, d["fixed"]); t1 = time.perf_counter()
print("nos_pgo_create (SoA + adjacency + upload): %.1f ms" % (1e3 * (t1 - t0)), flush=True)
for rep in range(3):
    t0 = time.perf_counter(); c, gn = g.linearize(); dt = time.perf_counter() - t0
    print("linearize: %.3f ms  cost %.6g |g| %.4g  (%.1f M constraint evaluations/s)" % (1e3 * dt, c, gn, 2 * d["ref"].size / dt / 1e6), flush=True)
x = np.random.default_rng(0).normal(size=g.n_unknowns)
for iters in (50, 200):
    t0 = time.perf_counter(); it, res, step = g.solve(1e-3, iters, 0.0); dt = time.perf_counter() - t0
    print("pcg: %d iterations in %.2f ms = %.3f ms/iteration, rel residual %.3e" % (it, 1e3 * dt, 1e3 * dt / max(it, 1), res), flush=True)
for mode in (1, 0):  # PCG on the same linearisation: two-level preconditioner, then round 1's block-Jacobi
    ctx.set_option("pgo_precond", mode)
    for lam in (1e-3, 1e-6):
        t0 = time.perf_counter(); it, res, step = g.solve(lam, 300, 1e-6); dt = time.perf_counter() - t0
        print("pcg (%s, lambda %g): %d iterations in %.2f ms = %.3f ms/iteration, rel residual %.3e"
              % ("two-level" if mode else "block-Jacobi", lam, it, 1e3 * dt, 1e3 * dt / max(it, 1), res), flush=True)
ctx.set_option("pgo_precond", 1)
t0 = time.perf_counter()
it, hist = g.optimize(max_iterations=10, gradient_tolerance=1e-6, parameter_tolerance=1e-6, pcg_iterations=300, pcg_tolerance=1e-6)
dt = time.perf_counter() - t0
for h in hist:
    print("  LM: cost %.6g |g| %.3e |step| %.3e pcg %d (res %.1e)" % h)
c, gn = g.linearize()
print("LM loop: %d iterations in %.1f ms; final cost %.6g |g| %.3e" % (it + 1, 1e3 * dt, c, gn), flush=True)
poses, _ = g.state()
print("max |p - p_true| before %.3f after %.3f" % (np.max(np.abs(d["init"][:, :3] - d["true"][:, :3])), np.max(np.abs(poses[:, :3] - d["true"][:, :3]))))
