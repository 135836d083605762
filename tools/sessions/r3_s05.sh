#!/bin/bash
# round 3, GPU session 5: ablations of the hidden loop of lwdetr_vit_block (results wrong by construction, timing only)
set -u
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/r3_s05
mkdir -p $OUT
for a in "" _a1 _a2 _a3 _a4 _a8; do
  echo "== ablation '$a' (1 = no GELU, 2 = no MFMA, 4 = no DMA, 8 = no fragment reads)" | tee -a $OUT/vb_ablate.txt
  ( export LWDETR_HIP_LIB=$ROOT/tools/_timing/liblwdetr_hip_vbt$a.so; timeout 200 python tools/vitblock_timing.py 192 32 fp16 2>&1 | grep -v amdgpu.ids | grep -v "wave [123]" | grep -v "workgroup last" | tee -a $OUT/vb_ablate.txt )
done
