"""Fingerprint of the pose-graph solver on one build of libnos_hip.so: run it once per build (NOS_HIP_LIB=… selects another
build) on the same box and diff the two outputs — iteration counts, costs and sha256 of the PCG step must be identical when a
change only re-arranges launches (round 4: 25 → 6 launches per PCG iteration).

usage: python tools/ab_pgo.py [n_poses] > a.json ; NOS_HIP_LIB=tools/_bin/libnos_hip_r03.so python tools/ab_pgo.py [n_poses] > b.json
"""
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nonlinear_optimizer_for_slam_amd import Context, pgo, synth  # noqa: E402

out = {"cases": []}
for n in ([int(a) for a in sys.argv[1:]] or [3_000, 100_000, 1_000_000]):
    d = synth.pose_graph(n, 3)
    m = d["ref"].size
    rng = np.random.default_rng(5)
    free = np.zeros(m, dtype=np.uint8)
    for with_switches in (False, True):
        if with_switches:
            free[n - 1:] = (rng.uniform(size=m - (n - 1)) < 0.2).astype(np.uint8)  # a fifth of the non-chain constraints
        ctx = Context((0,))
        g = pgo.PoseGraph(ctx, d["init"], d["ref"], d["qry"], d["meas"], None, free if with_switches else None, d["fixed"])
        case = {"poses": n, "constraints": int(m), "free_switches": int(free.sum()) if with_switches else 0}
        cost, gnorm = g.linearize()
        case["linearize"] = [cost.hex(), gnorm.hex()]
        case["solves"] = []
        for precond in (1, 0):
            ctx.set_option("pgo_precond", precond)
            for lam in (1e-3, 1e-6):
                g.solve(lam, 8, 0.0)
                ctx.synchronize()
                t0 = time.perf_counter()
                it, res, step = g.solve(lam, 300, 1e-6)
                dt = time.perf_counter() - t0
                x = g.vector("step")
                case["solves"].append({"precond": precond, "lambda": lam, "iterations": it, "rel_residual": float(res).hex(),
                                       "step_norm": float(step).hex(), "step_sha256": hashlib.sha256(x.tobytes()).hexdigest(),
                                       "ms_per_iteration": 1e3 * dt / max(it, 1)})
        ctx.set_option("pgo_precond", 1)
        lm_it, hist = g.optimize(max_iterations=3, gradient_tolerance=1e-6, parameter_tolerance=1e-6, pcg_iterations=300,
                                 pcg_tolerance=1e-6)
        c1, g1 = g.linearize()
        case["lm"] = {"pcg_iterations": [int(h[3]) for h in hist], "costs": [float(h[0]).hex() for h in hist],
                      "cost_after": float(c1).hex(), "cost_after_decimal": c1}
        out["cases"].append(case)
        g.close()
        ctx.close()
# timings differ from run to run: print them apart from the fingerprint
timing = [[s.pop("ms_per_iteration") for s in c["solves"]] for c in out["cases"]]
print(json.dumps(out, indent=1, sort_keys=True))
print(json.dumps({"lib": os.environ.get("NOS_HIP_LIB", "in-tree"), "ms_per_pcg_iteration": timing}), file=sys.stderr)
