#!/bin/bash
# A/B on one box: back-to-back kernel + fused time of an older build (tools/_old) against the current tree.
for rep in 1 2; do
  (cd tools/_old && python tail_probe.py 2>&1 | grep "n=10000000" | sed 's/^/old  /')
  python tools/tail_probe.py 2>&1 | grep "n=10000000" | sed 's/^/new  /'
done
