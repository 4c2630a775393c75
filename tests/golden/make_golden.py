"""Generates the committed golden fixtures under tests/golden/.

  reference_known_answers.json  values transcribed from the reference's captured runs (results/*.txt) with
                                their file:line — the only outputs of the reference available offline
                                (it cannot be built here: Eigen / Ceres / FLANN / simd_helper are absent).
  hotpath_small.npz             seeded synthetic inputs (generator of csrc/host/nos_synth.cpp) and the CPU
                                oracle's outputs for them: per-call {H upper | g | cost} for every loss and
                                LM-loop results.  The GPU path and the oracle are both tested against it, so
                                a silent change of either shows up.

Run from the repository root:  python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from nonlinear_optimizer_for_slam_amd import synth  # noqa: E402
from oracle import loader as oracle  # noqa: E402
from tests import helpers  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))

known = {
    "source": "ChanghyeonKim93/nonlinear_optimizer_for_slam @ 2025-09-12, results/*.txt (captured stderr of its test drivers)",
    "scene_counts": {
        "global_points": {"value": 954605, "at": "results/maha_amd64.txt:1"},
        "ndt_voxels": {"value": 96, "at": "results/maha_amd64.txt:2"},
        "reprojection_points": {"value": 630, "at": "results/reproj_amd64.txt:1"},
    },
    "reprojection_analytic": {
        "at": "results/reproj_amd64.txt:5,8,10",
        "cost_line": "COST: 2.33228e-11, iter: 6",
        "true_pose_txyz_qxyzw": [-0.1, 0.123, -0.5, 0.0, 0.0, 0.0499792, 0.99875],
        "final_pose_txyz_qxyzw": [-0.1, 0.123, -0.5, -2.38636e-09, 5.42421e-11, 0.0499792, 0.99875],
        "loss": ["exponential", 1.0, 1.0],
    },
    "ndt_analytic_simple": {
        "at": "results/maha_amd64_simple.txt:10-14,24,26",
        "cost_lines": [[17438.4, 40], [17394.5, 40], [17490.6, 20], [17490.7, 2]],
        "outer_iter": 3,
        "final_pose_txyz_qxyzw": [-0.196416, 0.121469, 0.304836, -0.000156768, -0.00124237, 0.0499568, 0.998751],
        "true_pose_txyz_qxyzw": [-0.2, 0.123, 0.3, 0.0, 0.0, 0.0499792, 0.99875],
        "note": "band only: depends on Eigen's eigenvector conventions through S = D^-1/2 V (DESIGN.md §9)",
    },
    "fp32_vs_fp64_pose_gap": {"at": "results/maha_amd64_simple.txt:24-25", "translation": 1.1e-5, "quaternion": 6e-6},
}
json.dump(known, open(os.path.join(HERE, "reference_known_answers.json"), "w"), indent=1)

LOSSES = {"none": None, "exponential": ("exponential", 1.0, 1.0), "huber": ("huber", 1.2)}
R = helpers.rot_xyz(0.01, -0.02, 0.05)
t = np.array([-0.1, 0.05, 0.2])
c, s = np.cos(0.07), np.sin(0.07)
R2 = np.array([[c, -s], [s, c]])
t2 = np.array([-0.15, 0.1])
out = {"R": R, "t": t, "R2": R2, "t2": t2}
ndt = synth.ndt_planes(1000, 50, seed=20250912)
rep = synth.reproj_planes(1000, seed=20250912)
out["ndt_planes"] = ndt
out["reproj_planes"] = rep
for name, loss in LOSSES.items():
    out["ndt6_" + name] = oracle.ndt6_accumulate(ndt, R, t, loss)
    out["ndt3_" + name] = oracle.ndt3_accumulate(ndt, R2, t2, loss)
rl = {"none": None, "exponential": ("exponential", 1.0, 1.0), "huber": ("huber", synth.REPROJ_HUBER_THRESHOLD)}
Rr = helpers.rot_xyz(0.0, 0.01, -0.08)
tr = np.array([0.08, -0.1, 0.4])
out["Rr"], out["tr"] = Rr, tr
for name, loss in rl.items():
    out["reproj_" + name] = oracle.reproj_accumulate(rep, Rr, tr, synth.REPROJ_INTR4, loss)
sol = oracle.ndt6_solve(ndt, np.zeros(3), np.eye(3), loss=LOSSES["exponential"], linear_solver=1)
out["ndt6_solve_t"], out["ndt6_solve_R"] = sol["t"], sol["R"]
out["ndt6_solve_meta"] = np.array([sol["iterations"], sol["printed_cost"], sol["last_cost"]])
sol = oracle.reproj_solve(rep, synth.REPROJ_INTR4, np.zeros(3), np.eye(3), loss=rl["huber"], linear_solver=1)
out["reproj_solve_t"], out["reproj_solve_R"] = sol["t"], sol["R"]
out["reproj_solve_meta"] = np.array([sol["iterations"], sol["printed_cost"], sol["last_cost"]])
np.savez_compressed(os.path.join(HERE, "hotpath_small.npz"), **out)
print("wrote", sorted(os.listdir(HERE)))
