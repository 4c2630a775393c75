"""Soak: bit-repeatability of the three solve forms and of accumulate over many launches (hand-off protocols)."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
from nonlinear_optimizer_for_slam_amd import Context, NdtDataset, synth
EXP = ("exponential", 1.0, 1.0)
ctx = Context((0,))
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
sets = {n: NdtDataset.from_planes(ctx, synth.ndt_planes(n, max(10, n // 40), seed=n), "f64") for n in (700, 9_400, 60_000, 125_000, 400_000, 3_000_000)}
first = {}
counts = {}
t_end = time.time() + budget
R_test = np.array([[0.9987, -0.0499, -0.0199], [0.0501, 0.9987, 0.0095], [0.0194, -0.0105, 0.9998]])
rounds = 0
while time.time() < t_end:
    for n, ds in sets.items():
        for form, env in (("default", {}), ("per-iteration", {"NOS_LM_CLUSTER": "0", "NOS_LM_SINGLE": "0"})):
            os.environ.update(env)
            R, t, r = ds.solve6(np.eye(3), np.zeros(3), EXP, max_iterations=25)
            for k in env:
                del os.environ[k]
            key = (R.tobytes(), t.tobytes(), r["iterations"], r["cost_history"].tobytes())
            first.setdefault((n, form), key)
            if key != first[(n, form)]:
                print("MISMATCH solve", n, form, rounds, flush=True)
                sys.exit(1)
            counts[(n, form)] = counts.get((n, form), 0) + 1
        out = ds.accumulate6(R_test, [-0.1, 0.05, 0.2], EXP).tobytes()
        first.setdefault((n, "acc"), out)
        if out != first[(n, "acc")]:
            print("MISMATCH accumulate", n, rounds, flush=True)
            sys.exit(1)
        counts[(n, "acc")] = counts.get((n, "acc"), 0) + 1
    rounds += 1
    if rounds % 200 == 0:
        print("round", rounds, flush=True)
print("soak ok: %d rounds; calls per case: %s" % (rounds, sorted(set(counts.values()))))
