#!/bin/bash
set -o pipefail
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $O/r02_gputests_b.log 2>&1; echo "tests rc=$?"; tail -6 $O/r02_gputests_b.log
for c in "ndt6 f32" "ndt3 f32" "reproj f32" "reproj f64"; do set -- $c; python bench.py --problem $1 --dtype $2 --no-cpu-baseline > $O/r02_bench_$1_$2.json 2> $O/r02_bench_$1_$2.err; echo "$c rc=$?"; done
python bench.py --problem ndt6 --points 100000 --steps 1000 --warmup 100 --no-cpu-baseline --no-strong-baseline > $O/r02_bench_ndt6_100k.json 2> $O/r02_bench_ndt6_100k.err; echo "100k rc=$?"
