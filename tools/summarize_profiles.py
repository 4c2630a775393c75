"""Turn the raw rocprofv3 output under gpurun_out/prof_<problem>_<dtype>_* (tools/profile_bench.sh) into the small,
tracked files under profiles/ that DESIGN.md and bench.py cite.

usage: python tools/summarize_profiles.py r02 [commit]
"""
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
commit = sys.argv[2] if len(sys.argv) > 2 else "unrecorded"
src = os.path.join(ROOT, "gpurun_out")
dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)


def newest(pattern):
    """gpurun merges new files into gpurun_out/ without deleting older runs' files: take the latest."""
    hits = glob.glob(pattern)
    return max(hits, key=os.path.getmtime) if hits else None


def counters(run, names):
    f = newest(os.path.join(src, run, "*", "*_counter_collection.csv"))
    if f is None:
        return None
    out = {}
    rows = [r for r in csv.DictReader(open(f)) if "assemble_kernel" in r["Kernel_Name"] or "solve_cluster" in r["Kernel_Name"]
            or "solve_resident" in r["Kernel_Name"]]
    for name in names:
        vals = [float(r["Counter_Value"]) for r in rows if r["Counter_Name"] == name]
        if vals:
            out[name] = {"launches": len(vals), "mean": sum(vals) / len(vals), "min": min(vals), "max": max(vals)}
    if rows:
        out["kernel"] = rows[0]["Kernel_Name"]
    return out


for stats_json in sorted(glob.glob(os.path.join(src, "prof_*_stats.json"))):
    base = os.path.basename(stats_json)[len("prof_"):-len("_stats.json")]
    problem, dtype = base.rsplit("_", 1)
    streaming = problem.endswith("stream")  # NOS_LM_CLUSTER=0 run: the launch-per-iteration kernel of the same problem
    stats = newest(os.path.join(src, "prof_%s_stats" % base, "*", "*_kernel_stats.csv"))
    if stats is None:
        continue
    try:
        bench = json.loads(open(stats_json).read().strip().splitlines()[-1])
    except Exception:  # noqa: BLE001
        continue
    shutil.copy(stats, os.path.join(dst, "%s_bench_%s_%s_kernel_stats.csv" % (tag, problem, dtype)))
    rows = list(csv.DictReader(open(stats)))
    hot = [r for r in rows if "assemble_kernel" in r["Name"] or "solve_cluster" in r["Name"] or "solve_resident" in r["Name"]]
    hot.sort(key=lambda r: -float(r["TotalDurationNs"]))
    k = hot[0]
    fetch = counters("prof_%s_fetch" % base, ["FETCH_SIZE"])
    write = counters("prof_%s_write" % base, ["WRITE_SIZE"])
    sq = counters("prof_%s_sq" % base, ["SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_ANY",
                                        "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_INSTS_VALU"])
    algo = bench["roofline"]["algorithmic_bytes_per_launch"]
    n = bench["config"]["points_per_gpu"]
    summary = {
        "commit": commit,
        "command": "rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --problem %s --dtype %s --steps 60 "
                   "--warmup 10 --repeats 3 --no-cpu-baseline --no-strong-baseline --no-cold" % (problem, dtype),
        "problem": problem[:-6] if streaming else problem, "dtype": dtype,
        "mode": "one launch per iteration (NOS_LM_CLUSTER=0)" if streaming else "product default", "workload": bench["config"]["workload"], "points_per_gpu": n,
        "kernel": k["Name"], "rocprof_calls": int(k["Calls"]), "rocprof_avg_ns": float(k["AverageNs"]),
        "rocprof_min_ns": float(k["MinNs"]), "rocprof_max_ns": float(k["MaxNs"]),
        "bench_kernel_ms_same_run": bench["roofline"]["kernel_ms"], "bench_ms_per_step_same_run": bench["ms_per_step"],
        "algorithmic_bytes_per_launch": algo,
        "achieved_GBps_from_rocprof_avg": algo / float(k["AverageNs"]),
        "frac_of_8000": algo / float(k["AverageNs"]) / 8000.0,
    }
    if "solve_cluster" in k["Name"]:
        # resident one-launch solve: one kernel call spans many LM iterations, its duration is not a per-iteration figure
        for key in ("achieved_GBps_from_rocprof_avg", "frac_of_8000"):
            summary.pop(key)
        summary["note"] = ("the hot kernel is the resident solve (whole LM loop in one launch, data on chip): per-iteration time = "
                           "bench_kernel_ms_same_run (bracket / iterations); the streaming kernel of this problem is profiled in "
                           "the *stream* summary")
    if fetch and write and "FETCH_SIZE" in fetch and "WRITE_SIZE" in write:
        # gfx950: FETCH_SIZE (KB) tallies the 128-B requests of a coalesced streaming read at 64 B (MI355X_MICROARCH.md §HBM):
        # x2; WRITE_SIZE is exact.  Datasets below 256 MiB are served by the Infinity Cache when warm; its hits are counted.
        traffic = 2.0 * fetch["FETCH_SIZE"]["mean"] * 1024.0 + write["WRITE_SIZE"]["mean"] * 1024.0
        summary.update({"FETCH_SIZE_KB": fetch["FETCH_SIZE"], "WRITE_SIZE_KB": write["WRITE_SIZE"],
                        "fetch_size_correction": "x2 (gfx950 FETCH_SIZE tallies 128-B requests at 64 B)",
                        "traffic_bytes_per_launch": traffic, "traffic_over_algorithmic": traffic / algo})
    if sq and "SQ_WAVE_CYCLES" in sq:
        wc = sq["SQ_WAVE_CYCLES"]["mean"]
        summary["sq"] = {kk: vv for kk, vv in sq.items() if kk != "kernel"}
        summary["sq_derived"] = {
            "valu_active_over_wave_cycles": sq["SQ_ACTIVE_INST_VALU"]["mean"] / wc,
            "any_inst_active_over_wave_cycles": sq["SQ_ACTIVE_INST_ANY"]["mean"] / wc,
            "wait_any_over_wave_cycles": sq["SQ_WAIT_ANY"]["mean"] / wc,
            "wait_inst_any_over_wave_cycles": sq["SQ_WAIT_INST_ANY"]["mean"] / wc,
            "valu_instructions_per_correspondence": sq["SQ_INSTS_VALU"]["mean"] * 64.0 / n if "SQ_INSTS_VALU" in sq else None,
            "note": "SQ_*_CYCLES / ACTIVE / WAIT count quad-cycles summed over waves; wave-level VALU-active fraction, the wave "
                    "parked on s_waitcnt / barrier (WAIT_ANY) and issue stalls (WAIT_INST_ANY) are disjoint shares of WAVE_CYCLES",
        }
    json.dump(summary, open(os.path.join(dst, "%s_bench_%s_%s_summary.json" % (tag, problem, dtype)), "w"), indent=1)
    print(json.dumps(summary, indent=1))
