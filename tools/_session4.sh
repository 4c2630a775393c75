#!/bin/bash
set -o pipefail
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out
( time python __graft_entry__.py smoke ) > $O/r02_smoke.log 2>&1; echo "smoke rc=$?"; tail -4 $O/r02_smoke.log
( time python bench.py > $O/r02_bench_default.json 2> $O/r02_bench_default.err ) 2> $O/r02_bench_default.time; echo "bench rc=$?"; cat $O/r02_bench_default.time
