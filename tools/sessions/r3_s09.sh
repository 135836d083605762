#!/bin/bash
# round 3, GPU session 9: 4-wave block kernel with the two-term GELU, fused normalisation, Q-only scaling (C = 192 and 384)
set -u
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/r3_s09
mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "vit_block" > $OUT/t_vb.log 2>&1; tail -3 $OUT/t_vb.log
for cfg in "192 32 fp16" "384 32 fp16" "384 64 bf16"; do
  timeout 200 python tools/vitblock_bench.py $cfg 2>&1 | grep -v amdgpu.ids | tee -a $OUT/vb_bench.txt
done
( export LWDETR_HIP_LIB=$ROOT/tools/_timing/liblwdetr_hip_vbt.so; for cfg in "192 32 fp16" "384 32 fp16"; do timeout 200 python tools/vitblock_timing.py $cfg 2>&1 | grep -v amdgpu.ids | grep -v "wave [123]" | grep -v "workgroup" | tee -a $OUT/vb_timing.txt; done )
