"""GPU-resident scan-to-map registration: match → solve → re-match, nothing returns to the host
between matching and solving except the pose.

Mirrors the outer loop of the reference's test drivers (OptimizePoseAnalytic,
nonlinear_optimizer/mahalanobis_distance_minimizer/tests/simple_optimization_test.cc:474-503):
up to 10 rounds of {MatchPointCloud at the current pose, Solve()}, stopping when the pose changed by
less than 1e-5 in translation and in the quaternion vector part.
"""
import numpy as np

from .api import NdtMap, Scan
from .solvers import MahalanobisDistanceMinimizerHip, MahalanobisDistanceMinimizerHip3DOF, Options, Pose


def _quat_vec_norm(R):
    """|vec(q)| of the rotation matrix R = sin(angle / 2)."""
    c = (np.trace(R) - 1.0) / 2.0
    c = min(1.0, max(-1.0, c))
    return float(np.sqrt(max(0.0, (1.0 - c) / 2.0)))


def scan_to_map(ctx, ndt_map, scan, initial_pose=None, loss=("exponential", 1.0, 1.0), options=None,
                max_outer_iterations=10, dof=6, dtype="f64", on_solve=None, indexed=False, keep_multiple=None,
                device_loop=True):
    """ndt_map: api.NdtMap, scan: api.Scan.  → (Pose, list of per-round dicts, outer_iter) — outer_iter as the
    reference prints it (index of the round that met the stopping test, or max_outer_iterations).

    indexed=True: the matcher emits voxel ids instead of 120-byte records (nos_ndt_match_indexed) and the solver runs on
    the voxel-indexed layout — 2-3x less memory traffic per LM iteration for large scans (sort the scan by cell first:
    api.Scan(..., sort_cell=...)); same sums, same pose.

    keep_multiple=k: every round uses only the first floor(N/k)*k of its N matches, as the reference's classes do on their
    correspondence vector (k = 4: scalar 3-DoF class, MDM/..._analytic_3dof.cc:33-36, and the revision of the 6-DoF class
    behind results/*.txt; k = 8: the SIMD classes) — done on the device (nos_dataset_drop_last_matches).

    on_solve(round, report, n_matches) is called after every inner Solve (e.g. to print the
    reference's `COST: ..., iter: ...` lines)."""
    if indexed and keep_multiple:
        raise ValueError("keep_multiple (the tail drop of the reference's classes) is implemented for the flat layout only: "
                         "a voxel-indexed dataset has no per-match records to clear (nos_dataset_drop_last_matches)")
    pose = Pose() if initial_pose is None else Pose(initial_pose.R, initial_pose.t)
    last = Pose(pose.R, pose.t)
    options = options or Options()
    solver = (MahalanobisDistanceMinimizerHip3DOF if dof == 3 else MahalanobisDistanceMinimizerHip)(
        device_ids=ctx.device_ids, dtype=dtype, device_loop=device_loop)
    solver.SetLossFunction(loss)
    rounds = []
    outer = 0
    for outer in range(max_outer_iterations):
        n_used = None
        if indexed:
            dataset, n_matches = ndt_map.match_indexed(scan, pose.R, pose.t, 2, dtype, sort_by_voxel=False)
        else:
            dataset, n_matches = ndt_map.match(scan, pose.R, pose.t, 2, dtype)
            if keep_multiple:
                n_used = n_matches - n_matches % int(keep_multiple)
                dataset.drop_last_matches(n_matches - n_used)
        try:
            if not solver.SolveDataset(options, dataset, pose):
                raise RuntimeError("SolveDataset failed (status %d)" % solver.report.status)
        finally:
            dataset.close()
        rounds.append({"matches": n_matches, "used": n_matches if n_used is None else n_used,  # matched / summed by the solve
                       "iterations": solver.report.iterations, "printed_cost": solver.report.printed_cost})
        if on_solve is not None:
            on_solve(outer, solver.report, n_matches)
        dR = pose.R.T @ last.R                  # optimized_pose.inverse() * last_optimized_pose
        dt = pose.R.T @ (last.t - pose.t)
        if np.linalg.norm(dt) < 1e-5 and _quat_vec_norm(dR) < 1e-5:
            break
        last = Pose(pose.R, pose.t)
    else:
        outer = max_outer_iterations  # never met the stopping test: the reference's loop variable ends at the bound
    return pose, rounds, outer
