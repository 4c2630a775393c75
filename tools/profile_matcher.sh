#!/bin/bash
set -o pipefail
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
rm -rf gpurun_out/prof_match
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_match -- python3 tools/measure_matcher.py 10000000 > gpurun_out/prof_match.log 2>&1 || exit 1
f=$(ls -t gpurun_out/prof_match/*/*_kernel_stats.csv | head -1); cut -c1-160 "$f" | head -12
grep -E "match|from_" gpurun_out/prof_match.log | head
