#!/bin/bash
set -o pipefail
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out
timeout -k 10 200 python tools/ab_bits.py > $O/ab_new.txt 2> $O/ab_new.err; echo "ab new rc=$?"
NOS_HIP_LIB=$PWD/tools/_bin/libnos_hip_bperm.so timeout -k 10 200 python tools/ab_bits.py > $O/ab_old.txt 2> $O/ab_old.err; echo "ab old rc=$?"
if cmp -s $O/ab_new.txt $O/ab_old.txt; then echo "A/B bits IDENTICAL ($(wc -l < $O/ab_new.txt) lines)"; else echo "A/B bits DIFFER"; diff $O/ab_new.txt $O/ab_old.txt | head -20; fi
timeout -k 10 500 python -m pytest tests -q -m gpu -x 2>&1 | tail -3
python bench.py --problem reproj --no-cpu-baseline > $O/r02_bench_reproj_f64_b.json 2>/dev/null; python -c "
import json; d=json.loads(open('$O/r02_bench_reproj_f64_b.json').read().strip().splitlines()[-1]); print('reproj', d['ms_per_step'], d['ms_per_step_trains'], d['host_loop']['ms_per_step'])"
python bench.py --problem ndt6 --points 100000 --steps 1000 --warmup 100 --no-strong-baseline --no-cpu-baseline > $O/r02_bench_ndt6_100k_b.json 2>/dev/null; python -c "
import json; d=json.loads(open('$O/r02_bench_ndt6_100k_b.json').read().strip().splitlines()[-1]); print('ndt6 100k', d['ms_per_step'], d['ms_per_step_trains'], d['host_loop']['ms_per_step'])"
NOS_HIP_LIB=$PWD/tools/_bin/libnos_hip_timing.so timeout -k 10 200 python tools/resident_timing_probe.py 2>&1 | grep "resident-timing\|n =" > $O/r02_resident_timing.txt; cat $O/r02_resident_timing.txt | awk 'NR%3!=2'
