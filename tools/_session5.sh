#!/bin/bash
set -o pipefail
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out
for c in "reproj f64" "reproj f32"; do set -- $c; python bench.py --problem $1 --dtype $2 > $O/r02_bench_$1_$2.json 2> $O/r02_bench_$1_$2.err; echo "$c rc=$?"; done
python bench.py --problem ndt6 --points 100000 --steps 1000 --warmup 100 --no-strong-baseline > $O/r02_bench_ndt6_100k.json 2> $O/r02_bench_ndt6_100k.err; echo "100k rc=$?"
timeout -k 10 500 tools/profile_bench.sh "reproj:f64"; echo "profile rc=$?"
NOS_HIP_LIB=$PWD/tools/_bin/libnos_hip_timing.so timeout -k 10 200 python tools/resident_timing_probe.py 2>&1 | grep "resident-timing\|n =" > $O/r02_resident_timing.txt; cat $O/r02_resident_timing.txt | awk 'NR%3!=2'
