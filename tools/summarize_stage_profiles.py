"""Turn the raw rocprofv3 output of tools/profile_stages.sh (gpurun_out/prof_<tag>_<stage>_*) into one small tracked JSON
per stage under profiles/: per kernel — calls, average duration (kernel trace), and per-launch means of the counter
passes (FETCH_SIZE / WRITE_SIZE in KB as rocprofv3 reports them, HBM-side bytes = 2 x FETCH_SIZE + WRITE_SIZE with the
gfx950 correction of MI355X_MICROARCH.md §HBM, SQ shares, L2 hit rate).

usage: python tools/summarize_stage_profiles.py r04 [stage ...]
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
stages = sys.argv[2:] or ["pgo", "match", "mapbuild", "indexed"]
src = os.path.join(ROOT, "gpurun_out")
dst = os.environ.get("NOS_PROFILE_DST") or os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)


def newest(pattern):
    hits = glob.glob(pattern)
    return max(hits, key=os.path.getmtime) if hits else None


def short(name):
    """Kernel symbol without its argument list (templates kept)."""
    depth = 0
    for k, ch in enumerate(name):
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            return name[:k]
    return name


def counter_means(run):
    f = newest(os.path.join(src, run, "*", "*_counter_collection.csv"))
    if f is None:
        return {}
    acc = defaultdict(lambda: defaultdict(list))
    for r in csv.DictReader(open(f)):
        acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in acc.items()}


for stage in stages:
    base = "prof_%s_%s" % (tag, stage)
    stats = newest(os.path.join(src, base + "_stats", "*", "*_kernel_stats.csv"))
    if stats is None:
        print("no kernel stats for", stage)
        continue
    kernels = {}
    for r in csv.DictReader(open(stats)):
        kernels[short(r["Name"])] = {"calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3,
                                     "min_us": float(r["MinNs"]) / 1e3, "max_us": float(r["MaxNs"]) / 1e3,
                                     "total_ms": float(r["TotalDurationNs"]) / 1e6, "percent": float(r["Percentage"])}
    passes = {p: counter_means(base + "_" + p) for p in ("fetch", "write", "sq", "cache", "tcp")}
    for name, k in kernels.items():
        f = passes["fetch"].get(name, {}).get("FETCH_SIZE")
        w = passes["write"].get(name, {}).get("WRITE_SIZE")
        if f is not None:
            k["FETCH_SIZE_KB"] = f
        if w is not None:
            k["WRITE_SIZE_KB"] = w
        if f is not None and w is not None:
            k["hbm_side_bytes_per_launch"] = (2.0 * f + w) * 1024.0  # gfx950: FETCH_SIZE reads 1/2 of a wide stream
            k["hbm_side_GBps"] = k["hbm_side_bytes_per_launch"] / (k["avg_us"] * 1e-6) / 1e9
        sq = passes["sq"].get(name)
        if sq:
            wc = sq.get("SQ_WAVE_CYCLES", 0.0)
            k["sq"] = {c: sq[c] for c in sorted(sq)}
            if wc > 0:
                k["sq_shares"] = {"valu_active": sq.get("SQ_ACTIVE_INST_VALU", 0.0) / wc, "wait_any": sq.get("SQ_WAIT_ANY", 0.0) / wc,
                                  "issue_stall": sq.get("SQ_WAIT_INST_ANY", 0.0) / wc, "any_inst_active": sq.get("SQ_ACTIVE_INST_ANY", 0.0) / wc}
        ca = passes["cache"].get(name)
        if ca and (ca.get("TCC_HIT_sum", 0.0) + ca.get("TCC_MISS_sum", 0.0)) > 0:
            k["l2"] = {"hit": ca["TCC_HIT_sum"], "miss": ca["TCC_MISS_sum"],
                       "hit_rate": ca["TCC_HIT_sum"] / (ca["TCC_HIT_sum"] + ca["TCC_MISS_sum"])}
        tc = passes["tcp"].get(name)
        if tc:
            k["tcp"] = tc
    commit = "unrecorded"
    cf = os.path.join(src, "prof_%s_commit.txt" % tag)
    if os.path.exists(cf):
        commit = open(cf).read().strip()
    out = {"stage": stage, "tag": tag, "commit": commit,
           "how": "tools/profile_stages.sh: rocprofv3 --kernel-trace --stats, then FETCH_SIZE / WRITE_SIZE / SQ / TCC passes in separate "
                  "runs; counter values are means per launch; hbm_side_bytes = (2 x FETCH_SIZE + WRITE_SIZE) KB",
           "kernels": dict(sorted(kernels.items(), key=lambda kv: -kv[1]["total_ms"]))}
    path = os.path.join(dst, "%s_%s_summary.json" % (tag, stage))
    json.dump(out, open(path, "w"), indent=1)
    log = os.path.join(src, base + "_stats.log")
    if os.path.exists(log):
        with open(os.path.join(dst, "%s_%s_stats_run.txt" % (tag, stage)), "w") as fo:
            fo.write(open(log).read()[-6000:])
    print("wrote", path)
