#!/bin/bash
# round 3, GPU session 23: two-chain determinism probe with the decoder traced op by op
set -u
for rep in 1 2 3 4; do
  echo "== process $rep"
  timeout 200 python tools/determinism_probe.py small 32 12 -2 2>&1 | grep -v amdgpu | cut -c1-900
done
