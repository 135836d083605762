#!/bin/bash
# round 3, GPU session 21: determinism probe (which stage differs between identical runs), with and without the new block kernel
set -u
for vb in 1 0; do
  for rep in 1 2 3; do echo "== LWDETR_VIT_BLOCK=$vb process $rep"; LWDETR_VIT_BLOCK=$vb timeout 200 python tools/determinism_probe.py small 32 5 2>&1 | grep -v amdgpu; done
done
