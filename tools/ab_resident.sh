#!/bin/bash
# Same-box A/B of two builds of libnos_hip.so on the one-launch solve forms (tools/measure_resident.py) and, with a
# -DNOS_LM_TIMING build, the phases of a resident iteration.  usage: tools/ab_resident.sh <libA.so> <libB.so> [timing.so]
cd "$(dirname "$0")/.."
for lib in "$1" "$2"; do
  echo "== $lib"
  NOS_HIP_LIB=$lib python tools/measure_resident.py 2>&1 | python -c "
import sys, json
for line in sys.stdin:
    try: r = json.loads(line)
    except Exception: print(line.rstrip()); continue
    print('%-6s %s n=%-8d %-16s one-launch %7.2f us   launch/iter %7.2f us' % (r['problem'], r['dtype'], r['n'], r['one_launch_form'], r['resident_us_per_iteration'], r['launch_per_iteration_us']))
"
done
if [ -n "$3" ]; then
  echo "== phases ($3)"
  NOS_HIP_LIB=$3 python tools/resident_timing_probe.py 2>&1 | grep -v "^$"
fi
