#!/bin/bash
# rocprofv3 evidence for the SURVEY §8(f) stages and configs[4] (run on the GPU box via gpurun):
#   pgo      tools/measure_pgo.py 1000000        (linearise, PCG, LM loop)
#   match    tools/measure_matcher.py 10000000   (matcher, flat and id-only output; ingestion)
#   mapbuild tools/measure_mapbuild.py           (954 605 / 10 M points)
#   indexed  bench.py --layout indexed           (voxel-indexed assemble kernel)
# Per stage: one kernel trace with --stats, then counter passes in their OWN runs (FETCH_SIZE, WRITE_SIZE, SQ, cache
# hits): gpurun refuses --pmc together with the trace domains other than --kernel-trace.
# usage: tools/profile_stages.sh <tag> "pgo match mapbuild indexed"
# Outputs: gpurun_out/prof_<tag>_<stage>_{stats,fetch,write,sq,cache}/ ; summarise with tools/summarize_stage_profiles.py.
set -o pipefail
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
TAG=${1:-r04}
STAGES=${2:-"pgo match mapbuild indexed"}
OUT=gpurun_out
SQ="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU"
CACHE="TCC_HIT_sum TCC_MISS_sum"
TCP="TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum"
rocprofv3 -L > $OUT/rocprofv3_counters_${TAG}.txt 2>&1 || true
for s in $STAGES; do
  case $s in
    pgo) CMD="tools/measure_pgo.py 1000000" ;;
    match) CMD="tools/measure_matcher.py 10000000" ;;
    mapbuild) CMD="tools/measure_mapbuild.py" ;;
    indexed) CMD="bench.py --layout indexed --steps 40 --warmup 10 --repeats 3 --no-cpu-baseline --no-strong-baseline --no-cold --no-other-configs" ;;
    *) echo "unknown stage $s"; exit 2 ;;
  esac
  B=$OUT/prof_${TAG}_${s}
  rm -rf ${B}_stats ${B}_fetch ${B}_write ${B}_sq ${B}_cache ${B}_tcp
  rocprofv3 --kernel-trace --stats --output-format csv -d ${B}_stats -- python3 $CMD > ${B}_stats.log 2> ${B}_stats.err || exit 1
  echo "$s stats done"
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d ${B}_fetch -- python3 $CMD > ${B}_fetch.log 2> ${B}_fetch.err || exit 1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d ${B}_write -- python3 $CMD > ${B}_write.log 2> ${B}_write.err || exit 1
  rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d ${B}_sq -- python3 $CMD > ${B}_sq.log 2> ${B}_sq.err || exit 1
  rocprofv3 --kernel-trace --pmc $CACHE --output-format csv -d ${B}_cache -- python3 $CMD > ${B}_cache.log 2> ${B}_cache.err || exit 1
  rocprofv3 --kernel-trace --pmc $TCP --output-format csv -d ${B}_tcp -- python3 $CMD > ${B}_tcp.log 2> ${B}_tcp.err || echo "$s: TCP pass failed (counter names?)"
  echo "$s counters done"
  # gpurun copies back at most 64 MiB of gpurun_out/: summarise on the box, keep the small files, drop the raw traces
  NOS_PROFILE_DST=$OUT/stage_profiles python3 tools/summarize_stage_profiles.py $TAG $s || exit 1
  cp "$(ls -t ${B}_stats/*/*_kernel_stats.csv | head -1)" $OUT/stage_profiles/${TAG}_${s}_kernel_stats.csv
  rm -rf ${B}_stats ${B}_fetch ${B}_write ${B}_sq ${B}_cache ${B}_tcp
done
git rev-parse HEAD > $OUT/prof_${TAG}_commit.txt 2>/dev/null || true
