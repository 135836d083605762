#!/bin/bash
# round 3, GPU session 37: ring depth of the patch-resident convolution (3: two workgroups per CU; 4+: one)
set -u
echo "== ring kernel (before)"; LWDETR_CONV_PATCH=0 python tools/conv_time.py 16 32 2>&1 | grep -v amdgpu
for n in 3 4 6 8; do echo "== patch kernel, NST $n"; LWDETR_CONV_PATCH_NST=$n python tools/conv_time.py 16 32 2>&1 | grep -v amdgpu; done
