#!/bin/bash
# round 3, GPU session 8: fragment read-ahead 8 (8-wave C = 192 kernel and the 4-wave C = 384 kernel)
set -u
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/r3_s08
mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "vit_block" > $OUT/t_vb.log 2>&1; tail -3 $OUT/t_vb.log
for cfg in "192 32 fp16" "384 32 fp16" "384 64 bf16"; do
  timeout 200 python tools/vitblock_bench.py $cfg 2>&1 | grep -v amdgpu.ids | tee -a $OUT/vb_bench.txt
done
for a in "" _a1; do
( export LWDETR_HIP_LIB=$ROOT/tools/_timing/liblwdetr_hip_vbt$a.so; timeout 200 python tools/vitblock_timing.py 192 32 fp16 2>&1 | grep -v amdgpu.ids | grep -v "workgroup last" | tee -a $OUT/vb_timing.txt )
done
