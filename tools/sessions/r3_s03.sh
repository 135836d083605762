#!/bin/bash
# round 3, GPU session 3: layer-tick GELU, 16-byte stores / loads through half-wave exchanges, exact store counts in the QKV ring waits
set -u
OUT=gpurun_out/r3_s03
mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "vit_block" > $OUT/t_vb.log 2>&1; tail -5 $OUT/t_vb.log
for cfg in "192 32 fp16" "384 32 fp16" "384 64 bf16"; do
  timeout 200 python tools/vitblock_bench.py $cfg 2>&1 | grep -v amdgpu.ids | tee -a $OUT/vb_bench.txt
done
export LWDETR_HIP_LIB=$(pwd)/tools/_timing/liblwdetr_hip_vbt.so
for cfg in "192 32 fp16" "384 32 fp16"; do
  timeout 200 python tools/vitblock_timing.py $cfg 2>&1 | grep -v amdgpu.ids | grep -v "wave [123]" | tee -a $OUT/vb_timing.txt
done
