#!/bin/bash
# round 3, GPU session 24: every decoder op alone on a side stream under the other chain's load
set -u
for rep in 1 2; do
  echo "== process $rep"
  PROBE_STRESS=1 timeout 300 python tools/determinism_probe.py small 32 8 -2 2>&1 | grep -v amdgpu | cut -c1-300
done
