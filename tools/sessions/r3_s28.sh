#!/bin/bash
# round 3, GPU session 28: experimental forms of the sampling kernel beside the attention / GEMM kernels of the other chain
set -u
for rep in 1 2; do
  echo "== process $rep (attention without the LDS ring: the strongest disturbance)"
  LWDETR_ATTN_LDS=0 PROBE_MSDA=variants timeout 300 python tools/determinism_probe.py small 32 24 -2 2>&1 | grep -v amdgpu | cut -c1-300
done
echo "== default attention"
PROBE_MSDA=variants timeout 300 python tools/determinism_probe.py small 32 24 -2 2>&1 | grep -v amdgpu | cut -c1-300
