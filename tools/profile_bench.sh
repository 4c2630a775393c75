#!/bin/bash
# Collects the rocprofv3 evidence for the headline bench on the GPU box (run via gpurun):
#   1. kernel trace + stats of `bench.py` (the same command the driver runs, fewer steps)
#   2. FETCH_SIZE and WRITE_SIZE in separate --pmc passes (TCC slots do not fit both), for the
#      default fp64 geometry (8 B/lane loads) and for the 16 B/lane geometry whose FETCH_SIZE
#      scale the microarch guide calibrates (reads exactly 1/2 of the bytes on gfx950).
# Outputs land in gpurun_out/prof_*/ ; summarise with tools/summarize_profiles.py.
set -o pipefail
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out
STEPS=${STEPS:-60}
rm -rf $OUT/prof_stats $OUT/prof_fetch_v0 $OUT/prof_fetch_v3 $OUT/prof_write_v0
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_stats -- python3 bench.py --steps $STEPS --warmup 10 --no-cpu-baseline > $OUT/prof_stats.json 2> $OUT/prof_stats.err || exit 1
echo "stats done"
NOS_VARIANT=0 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/prof_fetch_v0 -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/prof_fetch_v0.json 2> $OUT/prof_fetch_v0.err || exit 1
echo "fetch v0 done"
NOS_VARIANT=3 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/prof_fetch_v3 -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/prof_fetch_v3.json 2> $OUT/prof_fetch_v3.err || exit 1
echo "fetch v3 done"
NOS_VARIANT=0 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/prof_write_v0 -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/prof_write_v0.json 2> $OUT/prof_write_v0.err || exit 1
echo "write v0 done"
find $OUT/prof_stats $OUT/prof_fetch_v0 -name "*.csv" | head -20
