#!/bin/bash
# One pass over the measurement tools on the GPU box; text summaries land in gpurun_out/evidence/ (copy to profiles/).
set -o pipefail
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
E=gpurun_out/evidence
rm -rf $E && mkdir -p $E
python tools/measure_configs.py > $E/measure_configs.jsonl 2> $E/measure_configs.err || exit 1
echo "configs done"
python tools/measure_cold_solve.py > $E/measure_cold_solve.txt 2>&1 || exit 1
python tools/measure_mapbuild.py > $E/measure_mapbuild.txt 2>&1 || exit 1
python tools/measure_indexed.py 2>&1 | grep "bpc=1" > $E/measure_indexed.txt || exit 1
python tools/measure_indexed_k2.py > $E/measure_indexed_k2.txt 2>&1 || exit 1
echo "indexed done"
python tools/measure_pgo.py > $E/measure_pgo.txt 2>&1 || exit 1
bash tools/rehearse_multi.sh > $E/rehearse_multi.txt 2>&1 || exit 1
echo "rehearse done"
rm -rf gpurun_out/prof_match
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_match -- python3 tools/measure_matcher.py 10000000 > $E/measure_matcher.txt 2>&1 || exit 1
cp "$(ls -t gpurun_out/prof_match/*/*_kernel_stats.csv | head -1)" $E/matcher_kernel_stats.csv
rm -rf gpurun_out/prof_loop
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_loop -- python3 tools/loop_probe.py run > $E/loop_probe.log 2>&1 || exit 1
python tools/loop_probe.py report gpurun_out/prof_loop > $E/loop_probe.txt 2>&1 || exit 1
python bench.py --dtype f32 --no-cpu-baseline > $E/bench_f32.json 2>/dev/null || exit 1
python bench.py --loop host --no-cpu-baseline > $E/bench_host_loop.json 2>/dev/null || exit 1
python bench.py --layout indexed --no-cpu-baseline > $E/bench_indexed.json 2>/dev/null || exit 1
echo "all done"
