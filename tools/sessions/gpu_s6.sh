#!/bin/bash
# GPU session 6: large-tile GEMM tuning variants (nt DMA, setprio, no sched_barrier), tightened golden tests
set -u
OUT=gpurun_out/s6
mkdir -p $OUT
for v in 0 1 2 4 3; do
  if [ $v = 0 ]; then unset LWDETR_HIP_LIB; else export LWDETR_HIP_LIB=$(pwd)/tools/_timing/liblwdetr_hip_big$v.so; fi
  echo "=== variant $v" >> $OUT/variants.txt
  timeout 300 python tools/gemm_big_bench.py xlarge >> $OUT/variants.txt 2>&1
done
unset LWDETR_HIP_LIB
cat $OUT/variants.txt | grep -v amdgpu.ids
timeout 900 python -m pytest tests/test_gpu_model.py -q -m gpu > $OUT/t_model.log 2>&1
tail -8 $OUT/t_model.log
cat gpurun_out/parity_float16_large_640_mlp*.json gpurun_out/parity_float16_small_640_mlp*.json gpurun_out/parity_bfloat16_medium_640_mlp*.json gpurun_out/parity_float16_xlarge_960_mlp*.json
