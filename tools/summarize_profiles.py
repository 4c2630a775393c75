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


# iterations of every device-loop call of one bench.py run, in dispatch order: first the 50-step probe of the prewarm, then
# the warm-up, then the timed trains (tools/profile_bench.sh: --prewarm-ms 0 in the counter runs, so no further prewarm rounds)
COUNTER_RUN_CALLS = [50, 5, 20]
STATS_RUN_STEPS, STATS_RUN_REPEATS = 60, 3


def counters(run, names, hot_kernel=None, per_call_iterations=None):
    """Per-launch counter values of the hot kernel.  For the one-launch loop (a call spans many LM iterations) the values
    of call k are divided by per_call_iterations[k]: what one pass over the data costs."""
    f = newest(os.path.join(src, run, "*", "*_counter_collection.csv"))
    if f is None:
        return None
    out = {}
    rows = [r for r in csv.DictReader(open(f)) if "assemble_kernel" in r["Kernel_Name"] or "solve_cluster" in r["Kernel_Name"]
            or "solve_resident" in r["Kernel_Name"]]
    if hot_kernel is not None:
        rows = [r for r in rows if r["Kernel_Name"] == hot_kernel]
    if per_call_iterations is not None:
        ids = sorted({int(r["Dispatch_Id"]) for r in rows})
        if len(ids) != len(per_call_iterations):
            out["warning"] = "expected %d calls of the one-launch kernel, saw %d" % (len(per_call_iterations), len(ids))
            per_call_iterations = None
        else:
            scale = {d: float(per_call_iterations[k]) for k, d in enumerate(ids)}
            for r in rows:
                if r["Counter_Name"] not in ("SQ_WAVES",):
                    r["Counter_Value"] = float(r["Counter_Value"]) / scale[int(r["Dispatch_Id"])]
            out["normalisation"] = "per LM iteration: every call's value divided by its iteration count %s" % per_call_iterations
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
    host_loop = problem.endswith("host")    # --loop host run: plain nos_*_accumulate launches
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
    one_launch = "solve_cluster" in k["Name"]
    calls = COUNTER_RUN_CALLS if one_launch else None
    fetch = counters("prof_%s_fetch" % base, ["FETCH_SIZE"], k["Name"], calls)
    write = counters("prof_%s_write" % base, ["WRITE_SIZE"], k["Name"], calls)
    sq = counters("prof_%s_sq" % base, ["SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_ANY",
                                        "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_INSTS_VALU"], k["Name"], calls)
    algo = bench["roofline"]["algorithmic_bytes_per_launch"]
    n = bench["config"]["points_per_gpu"]
    summary = {
        "commit": commit,
        "command": "rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --problem %s --dtype %s --steps 60 "
                   "--warmup 10 --repeats 3 --no-cpu-baseline --no-strong-baseline --no-cold" % (problem, dtype),
        "problem": problem[:-6] if streaming else (problem[:-4] if host_loop else problem), "dtype": dtype,
        "mode": ("one launch per iteration (NOS_LM_CLUSTER=0)" if streaming else
                 ("host loop around nos_*_accumulate: the plain accumulate kernel (--loop host)" if host_loop else "product default")), "workload": bench["config"]["workload"], "points_per_gpu": n,
        "kernel": k["Name"], "rocprof_calls": int(k["Calls"]), "rocprof_avg_ns": float(k["AverageNs"]),
        "rocprof_min_ns": float(k["MinNs"]), "rocprof_max_ns": float(k["MaxNs"]),
        "bench_kernel_ms_same_run": bench["roofline"]["kernel_ms"], "bench_ms_per_step_same_run": bench["ms_per_step"],
        "algorithmic_bytes_per_launch": algo,
        "achieved_GBps_from_rocprof_avg": algo / float(k["AverageNs"]),
        "frac_of_8000": algo / float(k["AverageNs"]) / 8000.0,
    }
    if one_launch:
        # one-launch LM loop (data resident on chip, or streamed every iteration): one kernel call spans a whole train of LM
        # iterations, so the per-iteration duration is  call duration / iterations of that call.  The timed trains are the
        # STATS_RUN_REPEATS longest calls of the trace (STATS_RUN_STEPS iterations each).
        trace = newest(os.path.join(src, "prof_%s_stats" % base, "*", "*_kernel_trace.csv"))
        durs = sorted((int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(trace))
                       if r["Kernel_Name"] == k["Name"]), reverse=True)[:STATS_RUN_REPEATS]
        per_it = [d / float(STATS_RUN_STEPS) for d in durs]
        avg = sum(per_it) / len(per_it)
        summary.update({
            "one_launch_loop": True, "iterations_per_timed_call": STATS_RUN_STEPS, "timed_call_durations_ns": durs,
            "rocprof_ns_per_iteration": {"mean": avg, "min": min(per_it), "max": max(per_it)},
            "achieved_GBps_from_rocprof_avg": algo / avg, "frac_of_8000": algo / avg / 8000.0,
            "note": "the hot kernel runs the whole LM loop in one launch; rocprof_avg_ns is the mean over calls of different "
                    "lengths (prewarm 50, warm-up 10, timed 60 iterations) — the per-iteration figure is "
                    "rocprof_ns_per_iteration (timed calls / 60), which is what bench.py's bracket reports as kernel_ms; the "
                    "launch-per-iteration kernel of the same problem is profiled in the *stream* summary",
        })
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
