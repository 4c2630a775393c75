"""Driver-timed lines for the stages either side of the hot path and for BASELINE.json configs[4]:
pose-graph sweeps, GPU matcher, NDT map build, record ingestion, the voxel-indexed layout and the reference's two
published test wrappers.  Called by bench.py at N = 1 (the default run); every entry carries

    ms                 {min, median, max, n} wall time of the stage (blocking C-ABI calls, device idle before and after)
    algorithmic_bytes  the bytes the stage has to move, as defined in DESIGN.md §4 (one table)
    bound / frac       which resource bounds it and the fraction of that resource's peak the stage reaches
    cpu_baseline       the oracle's restatement of the SAME stage timed on this host on a bounded sample (kind "port")

The oracle (oracle/) is imported only inside the cpu_* functions below — as the thing timed beside the GPU or, for the
reference wrappers, as the checker of the COST lines — never on the GPU path.
"""
import json
import os
import statistics
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
HBM_PEAK_GBPS = 8000.0   # MI355X_MICROARCH.md: HBM3E spec peak
PCIE_PEAK_GBPS = 63.0    # same guide: PCIe Gen5 x16 host link (spec)


def summarize(values):
    return {"min": min(values), "median": statistics.median(values), "max": max(values), "n": len(values)}


def timed_ms(ctx, fn, reps=3, warm=1):
    """Wall milliseconds of fn() (a blocking call), device synchronised on both sides."""
    out = []
    for k in range(warm + reps):
        ctx.synchronize()
        t0 = time.perf_counter()
        r = fn()
        ctx.synchronize()
        dt = 1e3 * (time.perf_counter() - t0)
        if hasattr(r, "close"):
            r.close()
        elif isinstance(r, tuple) and r and hasattr(r[0], "close"):
            r[0].close()
        if k >= warm:
            out.append(dt)
    return summarize(out)


def roof(entry, nbytes, ms, peak_gbps=HBM_PEAK_GBPS, bound="hbm", traffic=None):
    """The entry's roofline object, same keys as the headline's: achieved = algorithmic bytes / measured time.
    traffic = HBM-side bytes per launch of the dominant kernel from the committed stage profile, when there is one."""
    entry["algorithmic_bytes"] = int(nbytes)
    achieved = nbytes / (ms * 1e-3) / 1e9
    entry["roofline"] = {"bound": bound, "achieved": achieved, "peak": peak_gbps, "unit": "GB/s", "frac": achieved / peak_gbps,
                         "traffic": traffic}
    return entry


def stage_traffic(stage, kernel_substring):
    """HBM-side bytes per launch ((2 x FETCH_SIZE + WRITE_SIZE) KB, separate PMC passes) of a kernel of a committed stage
    profile (tools/profile_stages.sh → profiles/rNN_<stage>_summary.json), newest round first; None when there is none."""
    import glob
    import re

    def order(path):
        m = re.match(r"r(\d+)([a-z]*)_", os.path.basename(path))
        return (int(m.group(1)), m.group(2) == "", m.group(2)) if m else (0, False, "")
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_%s_summary.json" % stage)), key=order, reverse=True):
        try:
            prof = json.load(open(path))
        except Exception:  # noqa: BLE001
            continue
        for name, k in (prof.get("kernels") or {}).items():
            if kernel_substring in name and k.get("hbm_side_bytes_per_launch"):
                return {"bytes": k["hbm_side_bytes_per_launch"], "kernel_avg_us": k.get("avg_us"), "kernel": name.split("(")[0][:80],
                        "source": os.path.relpath(path, ROOT)}
    return None


def with_traffic(entry, stage, kernel_substring):
    t = stage_traffic(stage, kernel_substring)
    if t is not None:
        entry["roofline"]["traffic"] = t["bytes"]
        entry["traffic_source"] = ("NOT measured in this run: (2 x FETCH_SIZE + WRITE_SIZE) KB per launch of %s in the committed "
                                   "profile %s (separate --pmc passes; kernel average there %.1f us)"
                                   % (t["kernel"], t["source"], t["kernel_avg_us"] or 0.0))
    return entry


CPU_LEG_SECONDS = [0.0]  # wall time of every CPU baseline leg of this process, summed (bench.py reports it; the bar is 10 s)


def cpu_leg(fn):
    import functools

    @functools.wraps(fn)
    def wrapper(*args, **kwargs):
        t0 = time.perf_counter()
        try:
            return fn(*args, **kwargs)
        finally:
            CPU_LEG_SECONDS[0] += time.perf_counter() - t0
    return wrapper


def cpu_timed(fn, budget_s, min_passes=1):
    passes, t0 = 0, time.perf_counter()
    while True:
        fn()
        passes += 1
        el = time.perf_counter() - t0
        if el >= budget_s and passes >= min_passes:
            return passes, el


def host_threads():
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    return max(1, min(avail, int(os.environ.get("NOS_BENCH_CPU_THREADS", str(avail)))))


# ------------------------------------------------------------------------------------------------------------- matcher

def voxel_map_of(planes):
    """The synthetic scene's voxel table: first occurrence of every distinct mean → (means [V,3], S [V,9])."""
    _, first = np.unique(planes[3], return_index=True)
    return planes[3:6, first].T.copy(), planes[6:15, first].T.copy()


def stage_matcher(ctx, pkg, planes, want_cpu):
    """nos_ndt_match on the headline scene: 10 M scan points x 200 k voxels (MatchPointCloud,
    MDM/tests/simple_optimization_test.cc:296-342).  Algorithmic bytes per scan point: 24 in + 2 slots x 120 out (flat
    records, fp64) or 24 in + 2 x 4 out (voxel ids)."""
    api, synth = pkg["api"], pkg["synth"]
    n = planes.shape[1]
    means, S = voxel_map_of(planes)
    pts = planes[0:3].T.copy()
    Rt, tt = synth.true_pose("ndt")
    t0 = time.perf_counter()
    m = api.NdtMap(ctx, means, S, None, 1.0)
    t_map = 1e3 * (time.perf_counter() - t0)
    out = {}
    sc = api.Scan(ctx, pts)
    e = {"points": n, "voxels": len(m), "scan_order": "as generated (shuffled over the voxels)",
         "ms": timed_ms(ctx, lambda: m.match(sc, Rt, tt, 2, "f64"))}
    roof(e, n * (24 + 240), e["ms"]["median"])
    e["value"], e["unit"] = n / (e["ms"]["median"] * 1e-3), "points/s"
    out["matcher_10M_unsorted"] = e
    sc.close()
    t0 = time.perf_counter()
    sc2 = api.Scan(ctx, pts, sort_cell=1.0)
    t_sort = 1e3 * (time.perf_counter() - t0)
    ds, nm = m.match(sc2, Rt, tt, 2, "f64")
    ds.close()
    e = {"points": n, "voxels": len(m), "matches": nm, "scan_order": "sorted by grid cell once per scan (nos_scan_sort_by_cell)",
         "scan_upload_and_sort_ms": t_sort, "map_tables_ms": t_map,
         "ms": timed_ms(ctx, lambda: m.match(sc2, Rt, tt, 2, "f64"))}
    roof(e, n * (24 + 240), e["ms"]["median"])
    e["value"], e["unit"] = n / (e["ms"]["median"] * 1e-3), "points/s"
    e["bound_note"] = ("bytes = 24 B point + 2 x 120 B records per scan point; the search itself (≈ 54 candidate records of "
                       "32 B per point, served by L1 / L2) is what the kernel spends its time on: profiles/r04_match_summary.json")
    out["matcher_10M_cell_sorted"] = e
    e = {"points": n, "output": "voxel ids (nos_ndt_match_indexed, unsorted by voxel)",
         "ms": timed_ms(ctx, lambda: m.match_indexed(sc2, Rt, tt, 2, "f64", sort_by_voxel=False))}
    roof(e, n * (24 + 8 + 24 + 8), e["ms"]["median"])  # reads points, writes ids, then the dataset copy of both
    e["value"], e["unit"] = n / (e["ms"]["median"] * 1e-3), "points/s"
    out["matcher_10M_cell_sorted_ids"] = e
    sc2.close()
    m.close()
    if want_cpu:
        out["matcher_10M_cell_sorted"]["cpu_baseline"] = cpu_matcher(means, pts, Rt, tt)
    return out


@cpu_leg
def cpu_matcher(means, pts, Rt, tt, sample=400_000):
    from oracle import oracle_scene
    cores = host_threads()
    q = (Rt @ pts[:sample].T).T + tt
    idx, t_build, t_query = oracle_scene.match_point_cloud_kdtree(means, None, q, 1.0, 2, workers=cores)
    _, _, t_query1 = oracle_scene.match_point_cloud_kdtree(means, None, q[:sample // 8], 1.0, 2, workers=1)
    return {"value": sample / t_query, "unit": "points/s", "cores": cores, "kind": "port",
            "one_thread_value": (sample // 8) / t_query1, "tree_build_s": t_build, "matches_in_sample": int((idx >= 0).sum()),
            "sample": "first %d scan points against the same %d voxel means: k-d tree 2-nearest within the radius "
                      "(MatchPointCloud, MDM/tests/simple_optimization_test.cc:296-342; scipy.spatial.cKDTree standing in for "
                      "the un-vendored FLANN, queries spread over %d threads — the reference queries point by point on one)"
                      % (sample, means.shape[0], cores)}


# ----------------------------------------------------------------------------------------------------------- map build

def stage_mapbuild(ctx, pkg, want_cpu):
    """nos_ndt_map_build (UpdateNdtMap, MDM/tests/simple_optimization_test.cc:236-294): points [n][3] on the HOST → voxel
    statistics → matcher tables on the device.  The stage starts with host memory, so its bound is the host link:
    algorithmic bytes = 24 B per point over PCIe."""
    api, synth = pkg["api"], pkg["synth"]
    out = {}
    rng = np.random.default_rng(3)
    pts = rng.uniform(-0.5, 0.5, size=(10_000_000, 3)) * np.array([100.0, 100.0, 10.0])
    for label, res in (("mapbuild_10M_100k_voxels", 1.0), ("mapbuild_10M_796k_voxels", 0.5)):
        m, _ = api.NdtMap.build(ctx, pts, voxel_resolution=res, return_stats=False)
        nv = len(m)
        m.close()
        e = {"points": pts.shape[0], "voxel_resolution": res, "valid_voxels": nv,
             "ms": timed_ms(ctx, lambda: api.NdtMap.build(ctx, pts, voxel_resolution=res, return_stats=False)),
             "ms_with_statistics_download": timed_ms(ctx, lambda: api.NdtMap.build(ctx, pts, voxel_resolution=res), reps=2)}
        roof(e, pts.shape[0] * 24, e["ms"]["median"], PCIE_PEAK_GBPS, "pcie")
        e["value"], e["unit"] = pts.shape[0] / (e["ms"]["median"] * 1e-3), "points/s"
        out[label] = e
    room = synth.room_points()
    for label, exact in (("mapbuild_reference_scene_exact", True), ("mapbuild_reference_scene_wave_parallel", False)):
        e = {"points": room.shape[0], "voxel_resolution": 1.0,
             "mode": "NOS_MAP_REFERENCE_EXACT (sequential sums in point order, Eigen's solver restated)" if exact
                     else "wave-parallel sums + Jacobi",
             "ms": timed_ms(ctx, lambda: api.NdtMap.build(ctx, room, 1.0, 1.0, proper_sqrt_information=not exact,
                                                           reference_exact=exact), reps=5)}
        roof(e, room.shape[0] * 24, e["ms"]["median"], PCIE_PEAK_GBPS, "pcie")
        e["value"], e["unit"] = room.shape[0] / (e["ms"]["median"] * 1e-3), "points/s"
        out[label] = e
    if want_cpu:
        out["mapbuild_reference_scene_exact"]["cpu_baseline"] = cpu_mapbuild(room, 1.0, "the same 954 605 room points")
        out["mapbuild_10M_100k_voxels"]["cpu_baseline"] = cpu_mapbuild(pts[:1_000_000], 1.0,
                                                                        "the first 1 000 000 of the 10 M points")
    return out


@cpu_leg
def cpu_mapbuild(points, res, what):
    from oracle import oracle_scene
    oracle_scene.build_ndt_map_eigen(points[:20_000], res, max_voxels=1 << 18)
    t0 = time.perf_counter()
    m = oracle_scene.build_ndt_map_eigen(points, res, max_voxels=1 << 18)
    dt = time.perf_counter() - t0
    return {"value": points.shape[0] / dt, "unit": "points/s", "cores": 1, "kind": "port", "seconds": dt,
            "voxels": int(m["means"].shape[0]),
            "sample": "%s through oracle/scene_oracle.c (UpdateNdtMap restated: hash map of voxels, sums in point order, "
                      "Eigen::SelfAdjointEigenSolver<Matrix3d> restated; one thread, as the reference)" % what}


# ----------------------------------------------------------------------------------------------------------- ingestion

RECORD_STRIDE = 304  # sizeof(Correspondence), MDM/types.h:11-26
RECORD_OFFSETS = [0, 8, 16, 128, 136, 144] + [224 + 8 * (3 * j + i) for i in range(3) for j in range(3)]


def records_of(planes):
    """The reference's AoS correspondences: point at 0, ndt.mean at 128, ndt.sqrt_information (column-major 3x3) at 224."""
    n = planes.shape[1]
    rec = np.zeros((n, RECORD_STRIDE // 8))
    rec[:, 0:3] = planes[0:3].T
    rec[:, 16:19] = planes[3:6].T
    for i in range(3):
        for j in range(3):
            rec[:, 28 + 3 * j + i] = planes[6 + 3 * i + j]
    return rec


def stage_ingest(ctx, pkg, planes, want_cpu):
    """nos_ndt_dataset_create_from_records / nos_ndt_dataset_create: 10 M correspondences from HOST memory into the
    device-resident SoA dataset (the pack loop every Solve() of the reference opens with, MDM/..._analytic_simd.cc:19-28).
    Bound: the host link.  Algorithmic bytes: what has to cross it — 304 B per record shipped raw, 120 B packed."""
    NdtDataset = pkg["NdtDataset"]
    n = planes.shape[1]
    rec = records_of(planes)
    out = {}
    old = {k: ctx.get_option(k) for k in ("ingest",)}
    try:
        for label, mode, per in (("ingest_10M_records_raw", 2, RECORD_STRIDE), ("ingest_10M_records_host_pack", 1, 120)):
            ctx.set_option("ingest", mode)
            e = {"records": n, "record_bytes": RECORD_STRIDE,
                 "mode": "raw 304-byte records over PCIe, unpacked on the device" if mode == 2 else
                         "15 used doubles gathered by host threads into pinned planes, 120 B per record over PCIe",
                 "ms": timed_ms(ctx, lambda: NdtDataset.from_records(ctx, rec, RECORD_STRIDE, RECORD_OFFSETS, "f64"))}
            roof(e, n * per, e["ms"]["median"], PCIE_PEAK_GBPS, "pcie")
            e["value"], e["unit"] = n / (e["ms"]["median"] * 1e-3), "corr/s"
            out[label] = e
    finally:
        for k, v in old.items():
            ctx.set_option(k, v)
    e = {"records": n, "mode": "15 planar host planes (nos_ndt_dataset_create)",
         "ms": timed_ms(ctx, lambda: NdtDataset.from_planes(ctx, planes, "f64"))}
    roof(e, n * 120, e["ms"]["median"], PCIE_PEAK_GBPS, "pcie")
    e["value"], e["unit"] = n / (e["ms"]["median"] * 1e-3), "corr/s"
    out["ingest_10M_planes"] = e
    if want_cpu:
        out["ingest_10M_records_host_pack"]["cpu_baseline"] = cpu_pack(rec)
    return out


@cpu_leg
def cpu_pack(rec, sample=2_000_000):
    from oracle import loader as oracle
    sub = rec[:sample]
    oracle.pack_records_f32(sub[:50_000], RECORD_STRIDE, RECORD_OFFSETS)
    passes, el = cpu_timed(lambda: oracle.pack_records_f32(sub, RECORD_STRIDE, RECORD_OFFSETS), 0.4)
    return {"value": sample * passes / el, "unit": "corr/s", "cores": 1, "kind": "port",
            "sample": "%d passes over the first %d records: the AoS → 15 float planes loop of the SIMD classes' Solve() "
                      "(MDM/..._analytic_simd.cc:19-28) restated in C, one thread as there" % (passes, sample)}


# ------------------------------------------------------------------------------------------------------- pose graph

def stage_pgo(ctx, pkg, want_cpu):
    """BASELINE.json configs[4]: 1 M poses / ~4 M relative-pose constraints (PGO/ceres_cost_functor.h:17-98).  Three sweeps
    with their own lines — linearisation, the matrix-free product, one PCG iteration — plus the PCG solve and three LM
    iterations as round 3 reported them.  Algorithmic bytes per sweep: DESIGN.md §7 (records each sweep has to touch once)."""
    from nonlinear_optimizer_for_slam_amd import pgo
    synth = pkg["synth"]
    t0 = time.perf_counter()
    graph = synth.pose_graph(1_000_000, 3)
    t_gen = time.perf_counter() - t0
    N, M = 1_000_000, int(graph["ref"].size)
    t0 = time.perf_counter()
    g = pgo.PoseGraph(ctx, graph["init"], graph["ref"], graph["qry"], graph["meas"], None, None, graph["fixed"])
    t_create = time.perf_counter() - t0
    cost0, gnorm0 = g.linearize()
    lin_call = timed_ms(ctx, g.linearize, reps=5)
    lin_ms = g.time_sweep("linearize", repeats=10)
    mv_ms = g.time_sweep("matvec", 1e-3, repeats=20)
    # records a sweep touches once.  Linearisation (owner computes): poses 64 B, constraints 64 B, adjacency 2 x (4 + 4) B per
    # constraint, diagonal blocks (21 doubles) and gradient (6) written.  Product (block-local): one 80-byte entry per
    # constraint and block it touches, poses 64 B + halo poses again (pose 64 B + x 48 B), x read, y written, 6 diagonal entries
    # read; in a PCG iteration it also forms the direction (z read, p_new written).
    lay = g.layout_info()
    E, halo = lay["entries"], lay["halo_poses"]
    lin_bytes = 64 * N + 64 * M + 16 * M + (21 * 8 + 48) * N
    if E:
        mv_bytes = 80 * E + 64 * N + (64 + 48) * halo + (48 + 48 + 48) * N
    else:
        mv_bytes = 64 * N + 64 * M + 16 * M + (48 + 48 + 48) * N
    out = {}
    e = {"poses": N, "constraints": M, "ms": {"min": lin_ms, "median": lin_ms, "max": lin_ms, "n": 10},
         "ms_blocking_call": lin_call, "value": M / (lin_ms * 1e-3), "unit": "constraint linearisations/s",
         "timing": "hipEvent pair around 10 back-to-back sweeps (nos_pgo_time_sweep); ms_blocking_call = nos_pgo_linearize "
                   "with its two scalar readbacks"}
    out["pgo_linearize"] = with_traffic(roof(e, lin_bytes, lin_ms), "pgo", "pgo_linearize_kernel")
    e = {"poses": N, "constraints": M, "ms": {"min": mv_ms, "median": mv_ms, "max": mv_ms, "n": 20},
         "value": M / (mv_ms * 1e-3), "unit": "constraint products/s",
         "timing": "hipEvent pair around 20 back-to-back products of the PCG iteration's kind (in-launch p.Ap sum included)"}
    out["pgo_matvec"] = with_traffic(roof(e, mv_bytes, mv_ms), "pgo", "pgo_matvec_block_kernel" if E else "pgo_matvec_cg_kernel")
    # one PCG iteration, net of the set-up: two solves with exactly K1 and K2 iterations (tolerance 0)
    k1, k2 = 8, 56
    g.solve(1e-3, k1, 0.0)
    t1 = timed_ms(ctx, lambda: g.solve(1e-3, k1, 0.0), reps=3, warm=0)["median"]
    t2 = timed_ms(ctx, lambda: g.solve(1e-3, k2, 0.0), reps=3, warm=0)["median"]
    it_ms = (t2 - t1) / (k2 - k1)
    setup_ms = t1 - k1 * it_ms
    # the 6 launches of an iteration (vectors are 48 B per pose): product + direction (mv_bytes + z read + p_new written);
    # update + restriction (p, q, x, r read, x, r written, pose read); 3 PCR spans (2 x 36 doubles per aggregate and level, rhs
    # in and out per span); preconditioner + prolongation (21 factors + r + pose read, z written, coarse solution read)
    lay = g.layout_info()
    n_agg, levels = lay["aggregates"], lay["pcr_levels"]
    it_bytes = (mv_bytes + 2 * 48 * N) + (6 * 48 + 64) * N + (levels * 2 * 36 * 8 + 3 * 2 * 48) * n_agg + (21 * 8 + 48 + 48 + 64) * N
    e = {"poses": N, "constraints": M, "ms": {"min": it_ms, "median": it_ms, "max": it_ms, "n": k2 - k1},
         "launches_per_iteration": 6 if E else 7,
         "launches": "product (+ direction, + p.Ap tail) | update + restriction | 3 PCR spans | preconditioner + prolongation (+ r.z tail)",
         "layout": lay, "setup_ms_per_solve": setup_ms,
         "setup": "block-Jacobi factorisation + the coarse operator (3 assembly sweeps) + PCR elimination, once per solve",
         "value": 1e3 / it_ms, "unit": "PCG iterations/s",
         "timing": "(wall of a %d-iteration solve − wall of a %d-iteration solve) / %d, tolerance 0, median of 3 each" % (k2, k1, k2 - k1)}
    out["pgo_pcg_iteration"] = roof(e, it_bytes, it_ms)
    t0 = time.perf_counter()
    pcg_it, pcg_res, _ = g.solve(1e-3, 300, 1e-6)
    t_pcg = 1e3 * (time.perf_counter() - t0)
    t0 = time.perf_counter()
    lm_it, hist = g.optimize(max_iterations=3, gradient_tolerance=1e-6, parameter_tolerance=1e-6, pcg_iterations=300,
                             pcg_tolerance=1e-6)
    t_lm = 1e3 * (time.perf_counter() - t0)
    cost1, gnorm1 = g.linearize()
    out["pgo_1M_poses_4M_constraints (BASELINE.json configs[4])"] = {
        "poses": N, "constraints": M, "dtype": "f64",
        "linearize_ms": lin_call, "constraint_linearisations_per_s": M / (lin_call["min"] * 1e-3),
        "pcg": {"lambda": 1e-3, "iterations_to_1e-6": int(pcg_it), "relative_residual": float(pcg_res), "ms": t_pcg,
                "ms_per_iteration": t_pcg / max(1, int(pcg_it)), "ms_per_iteration_net_of_setup": it_ms, "setup_ms": setup_ms,
                "preconditioner": "two-level (block-Jacobi + rigid-motion coarse space)"},
        "lm": {"iterations": int(lm_it) + 1, "ms": t_lm, "cost_before": float(cost0), "cost_after": float(cost1),
               "gradient_norm_before": float(gnorm0), "gradient_norm_after": float(gnorm1),
               "pcg_iterations_per_solve": [int(h[3]) for h in hist]},
        "graph_generation_s": t_gen, "create_ms": 1e3 * t_create,
        "timing": "host wall clock around blocking calls (each call ends with a device synchronisation)",
        "parity": "unpinned against reference outputs (the reference has no analytic PGO and no captured run); "
                  "tests/test_pgo.py checks the linearisation against an explicit assembly"}
    g.close()
    if want_cpu:
        out["pgo_linearize"]["cpu_baseline"] = cpu_pgo(synth)
    return out


@cpu_leg
def cpu_pgo(synth, n=100_000):
    from oracle import loader
    d = synth.pose_graph(n, 3)
    loader.pgo_linearize(d["init"][:64], d["ref"][:8] % 64, d["qry"][:8] % 64, d["meas"][:8])  # library loaded, code paged in
    t0 = time.perf_counter()
    loader.pgo_linearize(d["init"], d["ref"], d["qry"], d["meas"], None, None, d["fixed"])
    dt = time.perf_counter() - t0
    m = int(d["ref"].size)
    return {"value": m / dt, "unit": "constraint linearisations/s", "cores": 1, "kind": "port", "seconds": dt,
            "sample": "one linearisation of a %d-pose / %d-constraint graph of the same generator through oracle/pgo_oracle.c "
                      "(plain C, one core: residual of PGO/ceres_cost_functor.h:17-98 with analytic Jacobians, diagonal blocks + "
                      "gradient + cost; pinned to the numpy restatement oracle_pgo.py by tests/test_oracle_golden.py — the "
                      "reference itself evaluates this through Ceres autodiff, which is absent here)" % (n, m)}


# ------------------------------------------------------------------------------------------- reference wrappers

def stage_reference_wrappers(ctx, pkg, want_cpu):
    """The only figures the reference publishes: wall times of two whole test wrappers on an amd64 desktop —
    results/maha_amd64_simple.txt:30,38 (126.1 ms scalar / 58.9 ms SIMD: 4 Solve() calls, 102 LM iterations, matching
    included) and results/reproj_amd64.txt:15,23 (1.327 / 0.400 ms).  The SAME work here, with the captured COST / iter
    lines checked inside the timed loop's result."""
    api, synth, solvers = pkg["api"], pkg["synth"], pkg["solvers"]
    from nonlinear_optimizer_for_slam_amd import pipeline
    out = {}
    known = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_known_answers.json")))
    room = synth.room_points()
    local, Rt, tt = synth.room_scan(room)
    gm, _ = api.NdtMap.build(ctx, room, 1.0, 1.0, reference_exact=True)   # built before the timed region, as the reference does
    loss = ("exponential", 1.0, 1.0)
    times, lines, outer = [], None, None
    for k in range(6):
        ctx.synchronize()
        t0 = time.perf_counter()
        sc = api.Scan(ctx, local)
        pose, rounds, outer = pipeline.scan_to_map(ctx, gm, sc, loss=loss, keep_multiple=4)
        dt = 1e3 * (time.perf_counter() - t0)
        sc.close()
        if k > 0:
            times.append(dt)
        lines = [["%.6g" % r["printed_cost"], r["iterations"]] for r in rounds]
    want = known["captured_ndt_runs"]["simple_6dof"]
    ok = lines == [list(x) for x in want["cost_lines"]] and outer == want["outer_iter"]
    out["reference_wrapper_ndt"] = {
        "ms": summarize(times), "scan_points": int(local.shape[0]), "solve_calls": len(lines),
        "lm_iterations": int(sum(l[1] for l in lines)), "cost_lines": lines, "outer_iter": outer,
        "cost_lines_equal_the_captured_run": bool(ok),
        "reference_published_ms": {"scalar_fp64": 126.1, "simd_fp32": 58.9, "where": "results/maha_amd64_simple.txt:30,38 (amd64 desktop)"},
        "work": "scan upload + rounds of {nos_ndt_match, floor(N/4)*4 tail drop, MahalanobisDistanceMinimizerHip::SolveDataset} "
                "until the pose stops moving; the map (954 605 points) is built before the timed region, as in the reference",
        "final_translation_error_m": float(np.max(np.abs(pose.t - tt)))}
    gm.close()
    planes, intr, Rtr, ttr = synth.reference_reprojection_scene()
    solver = solvers.ReprojectionErrorMinimizerHip()
    solver.SetLossFunction(loss)
    pose = solvers.Pose()
    solver.Solve(solvers.Options(), planes, intr, pose)
    times = []
    for _ in range(20):
        pose = solvers.Pose()
        t0 = time.perf_counter()
        okr = solver.Solve(solvers.Options(), planes, intr, pose)
        times.append(1e3 * (time.perf_counter() - t0))
    line = "COST: %.6g, iter: %d" % (solver.report.printed_cost, solver.report.iterations)
    out["reference_wrapper_reproj"] = {
        "ms": summarize(times), "points": int(planes.shape[1]), "cost_line": line,
        "cost_line_equals_the_captured_run": bool(okr and line == known["reprojection_analytic"]["cost_line"]),
        "reference_published_ms": {"scalar_fp64": 1.327, "simd_fp32": 0.400, "where": "results/reproj_amd64.txt:15,23 (amd64 desktop)"},
        "work": "cold ReprojectionErrorMinimizerHip::Solve(options, correspondences, intrinsics, &pose): records to the device, "
                "whole LM loop in one workgroup, pose back",
        "known_answer_source": known["reprojection_analytic"]["at"]}
    if want_cpu:
        out["reference_wrapper_ndt"]["cpu_baseline"] = cpu_wrapper_ndt(room, local, want)
        out["reference_wrapper_reproj"]["cpu_baseline"] = cpu_wrapper_reproj(planes, intr)
    return out


@cpu_leg
def cpu_wrapper_ndt(room, local, want):
    from oracle import loader as oracle
    from oracle import oracle_scene as scene
    ndt_map = scene.build_ndt_map_eigen(room, 1.0)
    loss = ("exponential", 1.0, 1.0)

    def solve_round(pl, R0, t0):
        r = oracle.ndt6_solve(pl, t0, R0, loss=loss)
        return r["R"], r["t"], r["printed_cost"], r["iterations"]

    t0 = time.perf_counter()
    _, _, rounds, outer = scene.captured_run_icp(solve_round, ndt_map, local, stride=4)
    dt = time.perf_counter() - t0
    lines = [[scene.printed(c), i] for c, i, _ in rounds]
    return {"value": 1e3 * dt, "unit": "ms", "cores": 1, "kind": "port",
            "cost_lines_equal_the_captured_run": lines == [list(x) for x in want["cost_lines"]] and outer == want["outer_iter"],
            "sample": "the same wrapper through the CPU oracle on this host, one pass: brute-force restatement of MatchPointCloud "
                      "(numpy, 9 356 points x 96 voxels) + oracle/nos_oracle.c's scalar fp64 LM loop per round"}


@cpu_leg
def cpu_wrapper_reproj(planes, intr):
    from oracle import loader as oracle
    intr4 = (1.0 / intr[0], 1.0 / intr[1], intr[2], intr[3])
    loss = ("exponential", 1.0, 1.0)
    oracle.reproj_solve(planes, intr4, np.zeros(3), np.eye(3), loss=loss)
    passes, el = cpu_timed(lambda: oracle.reproj_solve(planes, intr4, np.zeros(3), np.eye(3), loss=loss), 0.2, 5)
    r = oracle.reproj_solve(planes, intr4, np.zeros(3), np.eye(3), loss=loss)
    return {"value": 1e3 * el / passes, "unit": "ms", "cores": 1, "kind": "port", "iterations": int(r["iterations"]),
            "sample": "mean of %d solves of the same 630 correspondences through oracle/nos_oracle.c (scalar fp64 restatement "
                      "of REM/..._analytic.cc:12-162), one thread" % passes}
