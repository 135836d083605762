#!/bin/bash
# round 3, GPU session 6: first run of the 8-wave alternating-phase block kernel (C = 192)
set -u
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/r3_s06
mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "vit_block" > $OUT/t_vb.log 2>&1; tail -4 $OUT/t_vb.log
timeout 200 python tools/vitblock_bench.py 192 32 fp16 2>&1 | grep -v amdgpu.ids | tee -a $OUT/vb_bench.txt
( export LWDETR_HIP_LIB=$ROOT/tools/_timing/liblwdetr_hip_vbt.so; timeout 200 python tools/vitblock_timing.py 192 32 fp16 2>&1 | grep -v amdgpu.ids | tee -a $OUT/vb_timing.txt )
