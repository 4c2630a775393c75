#!/bin/bash
set -o pipefail
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out
python bench.py > $O/r02_benchline_ndt6_f64.json 2> $O/r02_benchline_ndt6_f64.err; echo "ndt6 f64 rc=$?"
for c in "ndt6 f32" "ndt3 f64" "ndt3 f32" "reproj f64" "reproj f32"; do set -- $c; python bench.py --problem $1 --dtype $2 > $O/r02_benchline_$1_$2.json 2> $O/r02_benchline_$1_$2.err; echo "$c rc=$?"; done
python bench.py --problem ndt6 --points 100000 --steps 1000 --warmup 100 --no-strong-baseline > $O/r02_benchline_ndt6_100k.json 2> $O/r02_benchline_ndt6_100k.err; echo "100k rc=$?"
NOS_LM_CLUSTER=4 python bench.py --no-cpu-baseline --no-strong-baseline --no-cold > $O/r02_benchline_ndt6_f64_launchloop.json 2>/dev/null; echo "launch loop rc=$?"
timeout -k 10 300 python tools/measure_resident.py > $O/r02_measure_resident.jsonl 2>/dev/null; echo "resident rc=$?"
NOS_HIP_LIB=$PWD/tools/_bin/libnos_hip_timing.so timeout -k 10 200 python tools/resident_timing_probe.py 2>&1 | grep "resident-timing\|n =" > $O/r02_resident_timing.txt; echo "timing rc=$?"
