#!/bin/bash
# round 3, GPU session 39: software-pipelined k-loop of the patch-resident convolution: tests, stress, timing, ablations
set -u
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "conv3x3" 2>&1 | tail -1
timeout 300 python tools/conv_stress.py 20 2>&1 | grep -v amdgpu | grep "^load\|idle" | cut -c1-200
echo "== product"; python tools/conv_time.py 16 32 2>&1 | grep -v amdgpu
for b in 1 4 16; do echo "== ablation $b (1 no MFMA, 4 no epilogue, 16 no fragment reads)"; LWDETR_HIP_LIB=$PWD/tools/_timing/liblwdetr_conv_abl$b.so python tools/conv_time.py 16 32 2>&1 | grep -v amdgpu; done
