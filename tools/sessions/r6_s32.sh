#!/bin/bash
# round 6, GPU session 32: block kernel ablations, second set (4 = no weight DMA after the prologue, 16 = no s_barrier at the ring boundaries, 32 = no fc1 bias loads, 8 = no fragment reads)
set -u
O=$(pwd)/gpurun_out/r6s32; mkdir -p $O
for tag in "" _a4 _a20 _a36 _a52 _a60; do
  echo "== ablation '$tag' batch 16"; LWDETR_HIP_LIB=tools/_timing/liblwdetr_hip_vbt$tag.so python tools/vitblock_timing.py 192 16 fp16 2>&1 | grep -v amdgpu.ids | grep -v "wave [123]" | grep -v "workgroup last" | head -3
done | tee $O/vitblock_ablations2.txt
