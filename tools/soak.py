"""Soak: bit-repeatability of the solve forms (one-launch solve — resident or streamed — with the tagged all-reduce, stage 1
through the XCD's L2 or through sc1 stores; the counter form of it; one launch per iteration) and of accumulate over many launches — the hand-off protocols under test.

usage: python tools/soak.py [seconds]"""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
from nonlinear_optimizer_for_slam_amd import Context, NdtDataset, ReprojDataset, synth
EXP = ("exponential", 1.0, 1.0)
HUB = ("huber", synth.REPROJ_HUBER_THRESHOLD)
ctx = Context((0,))
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
sets = {("ndt", n): NdtDataset.from_planes(ctx, synth.ndt_planes(n, max(10, n // 40), seed=n), "f64")
        for n in (700, 9_400, 60_000, 125_000, 400_000, 3_000_000)}
sets[("ndt32", 800_000)] = NdtDataset.from_planes(ctx, synth.ndt_planes(800_000, 20_000, seed=5), "f32")
sets[("reproj", 2_000_000)] = ReprojDataset.from_planes(ctx, synth.reproj_planes(2_000_000), "f64")
first = {}
counts = {}
t_end = time.time() + budget
R_test = np.array([[0.9987, -0.0499, -0.0199], [0.0501, 0.9987, 0.0095], [0.0194, -0.0105, 0.9998]])
FORMS = (("default", {}), ("sc1-stage1", {"lm_cluster": 5}), 
         ("per-iteration", {"lm_cluster": 0, "lm_single": 0}))
rounds = 0
while time.time() < t_end:
    for (kind, n), ds in sets.items():
        for form, opts in FORMS:
            with ctx.options(**opts):
                if kind == "reproj":
                    R, t, r = ds.solve(np.eye(3), np.zeros(3), synth.REPROJ_INTR4, HUB, max_iterations=25)
                else:
                    R, t, r = ds.solve6(np.eye(3), np.zeros(3), EXP, max_iterations=25)
            key = (R.tobytes(), t.tobytes(), r["iterations"], r["cost_history"].tobytes())
            first.setdefault((kind, n, form), key)
            if key != first[(kind, n, form)] or not r["ok"]:
                print("MISMATCH solve", kind, n, form, rounds, flush=True)
                sys.exit(1)
            counts[(kind, n, form)] = counts.get((kind, n, form), 0) + 1
            # the two stage-1 transports of the tagged all-reduce add the same numbers in the same order
            if form == "sc1-stage1" and key != first[(kind, n, "default")]:
                print("MISMATCH between stage-1 transports", kind, n, rounds, flush=True)
                sys.exit(1)
        if kind != "reproj":
            out = ds.accumulate6(R_test, [-0.1, 0.05, 0.2], EXP).tobytes()
            first.setdefault((kind, n, "acc"), out)
            if out != first[(kind, n, "acc")]:
                print("MISMATCH accumulate", kind, n, rounds, flush=True)
                sys.exit(1)
    rounds += 1
    if rounds % 200 == 0:
        print("round", rounds, flush=True)
print("soak ok: %d rounds x %d datasets x %d solve forms, every result bit-identical to its first occurrence; calls per case: %s"
      % (rounds, len(sets), len(FORMS), sorted(set(counts.values()))))
