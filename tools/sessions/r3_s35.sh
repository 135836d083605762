#!/bin/bash
# round 3, GPU session 35: the patch-resident convolution beside other kernels (tools/conv_stress.py)
set -u
for v in 0 1; do echo "== LWDETR_CONV_PATCH_VAR=$v"; LWDETR_CONV_PATCH_VAR=$v timeout 300 python tools/conv_stress.py 30 2>&1 | grep -v amdgpu | cut -c1-260; done
echo "== ring kernel (LWDETR_CONV_PATCH=0)"; LWDETR_CONV_PATCH=0 timeout 300 python tools/conv_stress.py 30 2>&1 | grep -v amdgpu | cut -c1-260
