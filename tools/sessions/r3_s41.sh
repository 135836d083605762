#!/bin/bash
# round 3, GPU session 42: convolution k-loop per tap, stages unrolled, chunk offsets as immediates
set -u
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "conv3x3" 2>&1 | tail -1
timeout 300 python tools/conv_stress.py 20 2>&1 | grep -v amdgpu | grep "^load\|idle" | cut -c1-200
echo "== product"; python tools/conv_time.py 16 32 2>&1 | grep -v amdgpu
echo "== phase stamps"; CONV_TIMING=1 LWDETR_HIP_LIB=$PWD/tools/_timing/liblwdetr_conv_timing.so python tools/conv_time.py 16 32 2>&1 | grep -v amdgpu | grep -v "wave [123]"
echo "== C = 192, 80 x 80 and 20 x 20 (LWDETR_CONV_PATCH=2 | 0)"
for cp in 2 0; do LWDETR_CONV_PATCH=$cp CONV_C=192 CONV_HW=80 python tools/conv_time.py 8 32 2>&1 | grep -v amdgpu; LWDETR_CONV_PATCH=$cp CONV_C=192 CONV_HW=20 python tools/conv_time.py 32 2>&1 | grep -v amdgpu; done
