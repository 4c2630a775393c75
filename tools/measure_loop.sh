#!/bin/bash
# Device-resident vs host LM loop, step time and in-loop kernel time (bench.py lines → gpurun_out/loop_*.json)
set -e
mkdir -p gpurun_out
for loop in device host; do
  python bench.py --no-cpu-baseline --loop $loop > gpurun_out/loop_${loop}_f64.json
  python bench.py --no-cpu-baseline --loop $loop --dtype f32 > gpurun_out/loop_${loop}_f32.json
done
NOS_LM_FUSED=0 python bench.py --no-cpu-baseline --loop device > gpurun_out/loop_device_sepstep_f64.json
NOS_LM_WINDOW=2 python bench.py --no-cpu-baseline --loop device > gpurun_out/loop_device_w2_f64.json
NOS_LM_WINDOW=8 python bench.py --no-cpu-baseline --loop device > gpurun_out/loop_device_w8_f64.json
NOS_BENCH_NO_EVENTS=1 python bench.py --no-cpu-baseline --loop device > gpurun_out/loop_device_noev_f64.json
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/loop_*.json')):
    r=json.loads(open(f).read().strip().splitlines()[-1])
    print('%-44s step %.5f ms  kernel %.5f ms  frac %.4f  %.2f Gcorr/s' % (f, r['ms_per_step'], r['roofline']['kernel_ms_mean'], r['roofline']['frac'], r['value']/1e9))
PY
