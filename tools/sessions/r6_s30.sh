#!/bin/bash
# round 6, GPU session 30: ablations of the block kernel after the QKV / boundary work (1 = no GELU ticks, 2 = no MFMA, 4 = no weight DMA after the prologue, 8 = no fragment reads)
set -u
O=$(pwd)/gpurun_out/r6s30; mkdir -p $O
for tag in "" _a1 _a2 _a4 _a8 _a5 _a3; do
  echo "== ablation '$tag' batch 16"; LWDETR_HIP_LIB=tools/_timing/liblwdetr_hip_vbt$tag.so python tools/vitblock_timing.py 192 16 fp16 2>&1 | grep -v amdgpu.ids | grep -v "wave [123]" | grep -v "workgroup last" | head -3
done | tee $O/vitblock_ablations.txt
