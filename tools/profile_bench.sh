#!/bin/bash
# Collects the rocprofv3 evidence for bench.py on the GPU box (run via gpurun), one set per problem / dtype:
#   (default mode = the one-launch LM loop; ":stream" = one launch per iteration, the kernel a single nos_*_accumulate call runs)
#   1. kernel trace + stats of `bench.py --problem P --dtype D` (the command the driver runs, fewer steps)
#   2. FETCH_SIZE and WRITE_SIZE in separate --pmc passes (the TCC slots do not fit both)
#   3. one SQ pass: wave cycles, VALU-active / wait / issue-stall quad-cycles, VALU instruction count
# usage: tools/profile_bench.sh "ndt6:f64 ndt6:f32 ndt3:f64 reproj:f64 reproj:f32"   (default: all five)
# Outputs land in gpurun_out/prof_<P>_<D>_{stats,fetch,write,sq}/ ; summarise with tools/summarize_profiles.py <tag>.
set -o pipefail
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out
CASES=${1:-"ndt6:f64 ndt6:f32 ndt3:f64 reproj:f64 ndt6:f64:stream ndt6:f32:stream reproj:f64:stream reproj:f32:stream"}
COMMON="--no-cpu-baseline --no-strong-baseline --no-cold --no-other-configs"
SQ="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU"
for c in $CASES; do
  # problem:dtype[:stream]  — "stream" = NOS_LM_CLUSTER=0: the launch-per-iteration (streaming) kernel instead of the resident solve
  P=$(echo $c | cut -d: -f1); D=$(echo $c | cut -d: -f2); M=$(echo $c | cut -d: -f3)
  B=$OUT/prof_${P}_${D}
  LOOP=""
  if [ "$M" = "stream" ]; then export NOS_LM_CLUSTER=0; B=$OUT/prof_${P}stream_${D}; else unset NOS_LM_CLUSTER; fi
  # "host" = the LM loop on the host around nos_*_accumulate: the plain accumulate kernel of the C ABI's inner cut (no pose
  # prologue, no in-launch LM step), one blocking call per iteration
  if [ "$M" = "host" ]; then LOOP="--loop host"; B=$OUT/prof_${P}host_${D}; fi
  rm -rf ${B}_stats ${B}_fetch ${B}_write ${B}_sq
  rocprofv3 --kernel-trace --stats --output-format csv -d ${B}_stats -- python3 bench.py --problem $P --dtype $D --steps 60 --warmup 10 --repeats 3 $LOOP $COMMON > ${B}_stats.json 2> ${B}_stats.err || exit 1
  echo "$c stats done"
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d ${B}_fetch -- python3 bench.py --problem $P --dtype $D --steps 20 --warmup 5 --repeats 1 --prewarm-ms 0 $LOOP $COMMON > ${B}_fetch.json 2> ${B}_fetch.err || exit 1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d ${B}_write -- python3 bench.py --problem $P --dtype $D --steps 20 --warmup 5 --repeats 1 --prewarm-ms 0 $LOOP $COMMON > ${B}_write.json 2> ${B}_write.err || exit 1
  rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d ${B}_sq -- python3 bench.py --problem $P --dtype $D --steps 20 --warmup 5 --repeats 1 --prewarm-ms 0 $LOOP $COMMON > ${B}_sq.json 2> ${B}_sq.err || exit 1
  echo "$c counters done"
done
git rev-parse HEAD > $OUT/prof_commit.txt 2>/dev/null || true
