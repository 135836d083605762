#!/bin/bash
# round 3, GPU session 7: ablations of the 8-wave block kernel (timing only; results wrong by construction)
set -u
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/r3_s07
mkdir -p $OUT
for a in "" _a1 _a2 _a3 _a4 _a16; do
  echo "== ablation '$a' (1 = no GELU, 2 = no MFMA, 4 = no DMA after the start, 16 = no barrier in the hidden loop)" | tee -a $OUT/vb8_ablate.txt
  ( export LWDETR_HIP_LIB=$ROOT/tools/_timing/liblwdetr_hip_vbt$a.so; timeout 200 python tools/vitblock_timing.py 192 32 fp16 2>&1 | grep -v amdgpu.ids | grep -v "workgroup last" | tee -a $OUT/vb8_ablate.txt )
done
