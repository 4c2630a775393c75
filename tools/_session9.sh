#!/bin/bash
set -o pipefail
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out
timeout -k 10 900 python -m pytest tests -q -m gpu -x 2>&1 | tail -3
P='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"], d["ms_per_step_trains"], round(d["roofline"]["frac"],4), d["config"]["launches_per_train"], d.get("host_loop",{}).get("ms_per_step"))'
B="--no-cpu-baseline --no-strong-baseline --no-cold"
for c in "ndt6 f64" "ndt6 f64" "ndt6 f64" "ndt6 f32" "ndt6 f32"; do set -- $c; echo "$c"; python bench.py --problem $1 --dtype $2 $B 2>/dev/null | python -c "$P"; done
echo "ndt6 f64 lm_cluster=4"; NOS_LM_CLUSTER=4 python bench.py --problem ndt6 --dtype f64 $B 2>/dev/null | python -c "$P"
