#!/bin/bash
# round 3, GPU session 33: two-chain determinism after the patch-resident convolution / in-place outputs
set -u
for rep in 1 2 3; do
  echo "== small 32 x40, 2 chains, process $rep"
  timeout 300 python tools/determinism_probe.py small 32 40 2 2>&1 | grep -v amdgpu | cut -c1-500 | tail -6
done
echo "== conv patch off"
LWDETR_CONV_PATCH=0 timeout 300 python tools/determinism_probe.py small 32 40 2 2>&1 | grep -v amdgpu | cut -c1-500 | tail -6
for i in 1 2 3 4; do timeout 300 python -m pytest tests/test_gpu_model.py -x -q -m gpu -k "full_size" 2>&1 | tail -1; done
