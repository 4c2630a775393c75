#!/bin/bash
# Rehearsal of bench.py's N > 1 code path on a ONE-GPU box: ranks share device 0, launcher side over gloo,
# data path = the in-launch mailbox exchange.  (Real N > 1 numbers come from the driver's 8-GPU run.)
set -o pipefail
mkdir -p gpurun_out
# the same box without any communicator: what the exchanges below add per iteration is read against THIS line
python bench.py --no-cpu-baseline --no-other-configs --no-strong-baseline --no-cold --steps 100 > gpurun_out/rehearse_n1_plain.json 2> gpurun_out/rehearse_n1_plain.err || { tail -5 gpurun_out/rehearse_n1_plain.err; exit 1; }
NOS_BENCH_FORCE_DIST=1 python bench.py --no-cpu-baseline --steps 100 > gpurun_out/rehearse_n1_forced.json 2> gpurun_out/rehearse_n1_forced.err || { tail -5 gpurun_out/rehearse_n1_forced.err; exit 1; }
for n in 2 4; do
  NOS_BENCH_SHARED_GPU=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 29517 \
    bench.py --gpus $n --steps 60 --warmup 10 --points 2500000 --no-cpu-baseline > gpurun_out/rehearse_n$n.json 2> gpurun_out/rehearse_n$n.err || { tail -8 gpurun_out/rehearse_n$n.err; exit 1; }
done
python - <<'PY'
import json
r = json.loads(open("gpurun_out/rehearse_n1_plain.json").read().strip().splitlines()[-1])
print("%-20s no communicator, one-launch loop: step %.4f ms  (host loop %s)" % ("rehearse_n1_plain", r["ms_per_step"], (r.get("host_loop") or {}).get("ms_per_step")))
for f in ("rehearse_n1_forced", "rehearse_n2", "rehearse_n4"):
    r = json.loads(open("gpurun_out/%s.json" % f).read().strip().splitlines()[-1])
    print("%-20s n_gpus %d collective %-14s loop %-6s step %.4f ms  value %.2f G corr/s  err %.2e   timed: %s" % (
        f, r["n_gpus"], r["config"]["collective"], r["config"]["loop"], r["ms_per_step"], r["value"] / 1e9, r["final_translation_error_m"],
        {k: round(v["ms_per_step"]["median"], 4) for k, v in r["config"]["collectives_timed"].items()}))
PY
