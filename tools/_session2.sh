#!/bin/bash
set -o pipefail
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out
TUNE_REPEATS=20 timeout -k 10 400 python tools/tune_ndt6.py 10000000 f32 0 > $O/r02_tune_f32.txt 2>&1; echo "tune rc=$?"; sort -t'(' -k2 $O/r02_tune_f32.txt | head -3
timeout -k 10 900 tools/profile_bench.sh "reproj:f64 reproj:f64:stream ndt6:f32 ndt3:f64 ndt6:f64 reproj:f32:stream"; echo "profile rc=$?"
