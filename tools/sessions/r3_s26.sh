#!/bin/bash
# round 3, GPU session 26: the decoder's sampling op under different co-running loads
set -u
for rep in 1 2; do
  echo "== process $rep"
  PROBE_MSDA=loads timeout 300 python tools/determinism_probe.py small 32 24 -2 2>&1 | grep -v amdgpu | cut -c1-300 | head -40
done
