"""Turn the raw rocprofv3 output under gpurun_out/prof_* (tools/profile_bench.sh) into the small,
tracked files under profiles/ that DESIGN.md and bench.py cite.

usage: python tools/summarize_profiles.py r01
"""
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
src = os.path.join(ROOT, "gpurun_out")
dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)

def newest(pattern):
    """gpurun merges new files into gpurun_out/ without deleting older runs' files: take the latest."""
    return max(glob.glob(pattern), key=os.path.getmtime)


stats = newest(os.path.join(src, "prof_stats", "*", "*_kernel_stats.csv"))
shutil.copy(stats, os.path.join(dst, "%s_bench_kernel_stats.csv" % tag))
bench = json.load(open(os.path.join(src, "prof_stats.json")))
rows = list(csv.DictReader(open(stats)))
asm = [r for r in rows if "assemble_kernel" in r["Name"]][0]


def counter(run, name):
    f = newest(os.path.join(src, run, "*", "*_counter_collection.csv"))
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(f))
            if "assemble_kernel" in r["Kernel_Name"] and r["Counter_Name"] == name]
    kn = [r["Kernel_Name"] for r in csv.DictReader(open(f)) if "assemble_kernel" in r["Kernel_Name"]][0]
    return {"kernel": kn, "launches": len(vals), "mean": sum(vals) / len(vals), "min": min(vals), "max": max(vals)}


fetch_v0 = counter("prof_fetch_v0", "FETCH_SIZE")
fetch_v3 = counter("prof_fetch_v3", "FETCH_SIZE")
write_v0 = counter("prof_write_v0", "WRITE_SIZE")
algo = bench["roofline"]["algorithmic_bytes_per_launch"]
# gfx950: FETCH_SIZE (KB) reports exactly 1/2 of the bytes of a coalesced streaming read
# (/opt/skills/guides/MI355X_MICROARCH.md §HBM) — the 16 B/lane geometry (v3) is the calibrated
# pattern, and the 8 B/lane default geometry (v0) reads the same value on the same data, so the
# same x2 applies; WRITE_SIZE is exact.
traffic = 2.0 * fetch_v0["mean"] * 1024.0 + write_v0["mean"] * 1024.0
summary = {
    "command": "rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --steps 60 --warmup 10 --no-cpu-baseline",
    "pmc_commands": ["NOS_VARIANT=0 rocprofv3 --kernel-trace --pmc FETCH_SIZE ... -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline",
                     "NOS_VARIANT=3 rocprofv3 --kernel-trace --pmc FETCH_SIZE ... (16 B/lane calibration geometry)",
                     "NOS_VARIANT=0 rocprofv3 --kernel-trace --pmc WRITE_SIZE ..."],
    "workload": bench["config"]["workload"],
    "points_per_gpu": bench["config"]["points_per_gpu"],
    "dtype": bench["dtype"],
    "kernel": asm["Name"],
    "rocprof_calls": int(asm["Calls"]),
    "rocprof_avg_ns": float(asm["AverageNs"]),
    "rocprof_min_ns": float(asm["MinNs"]),
    "rocprof_max_ns": float(asm["MaxNs"]),
    "bench_hip_event_kernel_ms_mean_same_run": bench["roofline"]["kernel_ms_mean"],
    "bench_ms_per_step_same_run": bench["ms_per_step"],
    "algorithmic_bytes_per_launch": algo,
    "achieved_GBps_from_rocprof_avg": algo / float(asm["AverageNs"]),
    "FETCH_SIZE_KB_default_geometry_8B_per_lane": fetch_v0,
    "FETCH_SIZE_KB_calibration_geometry_16B_per_lane": fetch_v3,
    "WRITE_SIZE_KB_default_geometry": write_v0,
    "fetch_size_correction": "x2 (gfx950 FETCH_SIZE tallies 128-B requests at 64 B; MI355X_MICROARCH.md §HBM)",
    "traffic_bytes_per_launch": traffic,
    "traffic_over_algorithmic": traffic / algo,
}
json.dump(summary, open(os.path.join(dst, "%s_bench_summary.json" % tag), "w"), indent=1)
print(json.dumps(summary, indent=1))
