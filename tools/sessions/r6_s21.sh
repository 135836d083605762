#!/bin/bash
# round 6, GPU session 21: phase stamps of the C = 192 block kernel as shipped (32 tokens per wave, two workgroups per CU, packed-f16 GELU), 16- and 32-image launches
set -u
O=$(pwd)/gpurun_out/r6s21; mkdir -p $O
for b in 16 32; do
  echo "== batch $b"; LWDETR_HIP_LIB=tools/_timing/liblwdetr_hip_vbt.so python tools/vitblock_timing.py 192 $b fp16 2>&1 | grep -v amdgpu.ids
done | tee $O/vitblock_phases.txt
