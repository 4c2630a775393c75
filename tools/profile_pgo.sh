#!/bin/bash
# rocprofv3 kernel trace of the config-4 pose-graph run (tools/measure_pgo.py) → gpurun_out/prof_pgo
set -o pipefail
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
rm -rf gpurun_out/prof_pgo
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_pgo -- python3 tools/measure_pgo.py 1000000 > gpurun_out/prof_pgo.log 2>&1 || exit 1
cat gpurun_out/prof_pgo/*/*_kernel_stats.csv | cut -c1-200
