#!/bin/bash
set -o pipefail
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out
timeout -k 10 900 python -m pytest tests -q -m gpu -x 2>&1 | tail -3
timeout -k 10 200 python tools/soak.py 90 2>&1 | tail -2 | tee $O/r02_soak.txt
P='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"], d["ms_per_step_trains"], round(d["roofline"]["frac"],4), d["config"]["launches_per_train"], d.get("host_loop",{}).get("ms_per_step"))'
B="--no-cpu-baseline --no-strong-baseline --no-cold"
echo "reproj f64"; python bench.py --problem reproj $B 2>/dev/null | python -c "$P"
echo "reproj f64 lm_cluster=5"; NOS_LM_CLUSTER=5 python bench.py --problem reproj $B 2>/dev/null | python -c "$P"
echo "100k"; python bench.py --problem ndt6 --points 100000 --steps 1000 --warmup 100 $B 2>/dev/null | python -c "$P"
echo "100k lm_cluster=5"; NOS_LM_CLUSTER=5 python bench.py --problem ndt6 --points 100000 --steps 1000 --warmup 100 $B 2>/dev/null | python -c "$P"
echo "ndt6 f64"; python bench.py $B 2>/dev/null | python -c "$P"
echo "ndt6 f64 lm_cluster=5"; NOS_LM_CLUSTER=5 python bench.py $B 2>/dev/null | python -c "$P"
