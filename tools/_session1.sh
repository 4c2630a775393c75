#!/bin/bash
set -o pipefail
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out
python bench.py --problem reproj --steps 500 --warmup 50 > $O/r02_bench_reproj_f64.json 2> $O/r02_bench_reproj_f64.err; echo "reproj rc=$?"; tail -c 600 $O/r02_bench_reproj_f64.err
NOS_BENCH_FORCE_DIST=1 timeout -k 10 300 python bench.py --steps 50 --repeats 2 --no-cpu-baseline --no-strong-baseline > $O/r02_bench_forcedist.json 2> $O/r02_bench_forcedist.err; echo "forcedist rc=$?"; tail -c 1500 $O/r02_bench_forcedist.err
rocprofv3 -L > $O/r02_counters_list.txt 2>&1; echo "counter list rc=$?"
tools/profile_bench.sh "reproj:f64" ; echo "profile rc=$?"
python bench.py > $O/r02_bench_ndt6_f64.json 2> $O/r02_bench_ndt6_f64.err; echo "ndt6 rc=$?"; tail -c 600 $O/r02_bench_ndt6_f64.err
python bench.py --problem ndt3 --no-cpu-baseline > $O/r02_bench_ndt3_f64.json 2> $O/r02_bench_ndt3_f64.err; echo "ndt3 rc=$?"
