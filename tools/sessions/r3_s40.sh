#!/bin/bash
# round 3, GPU session 40: phase stamps of the patch-resident convolution (workgroup 37)
set -u
CONV_TIMING=1 LWDETR_HIP_LIB=$PWD/tools/_timing/liblwdetr_conv_timing.so python tools/conv_time.py 16 32 2>&1 | grep -v amdgpu
