#!/bin/bash
# round 3, GPU session 2: phase timing of lwdetr_vit_block (instrumented build)
set -u
OUT=gpurun_out/r3_s02
mkdir -p $OUT
export LWDETR_HIP_LIB=$(pwd)/tools/_timing/liblwdetr_hip_vbt.so
for cfg in "192 32 fp16" "384 32 fp16"; do
  timeout 200 python tools/vitblock_timing.py $cfg 2>&1 | grep -v amdgpu.ids | tee -a $OUT/vb_timing.txt
done
