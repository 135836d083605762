#!/bin/bash
# round 3, GPU session 25: the decoder's sampling op under load - what the differing output elements hold
set -u
for mode in 1 2; do
  echo "== mode $mode"
  PROBE_MSDA=1 PROBE_STRESS=$mode timeout 300 python tools/determinism_probe.py small 32 24 -2 2>&1 | grep -v amdgpu | cut -c1-400 | head -40
done
