#!/bin/bash
# GPU session 20: large-tile GEMM with / without the L2 touch-prefetch
set -u
OUT=gpurun_out/s20
mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "gemm" > $OUT/t_gemm.log 2>&1
tail -3 $OUT/t_gemm.log
for v in 0 128; do
  echo "== variant $v"
  if [ $v = 0 ]; then unset LWDETR_HIP_LIB; else export LWDETR_HIP_LIB=$(pwd)/tools/_timing/libbig_v$v.so; fi
  timeout 200 python tools/gemm_big_bench.py xlarge large medium 2>&1 | grep -v amdgpu.ids | sed 's/(rel diff [^)]*)//g' | tee $OUT/big_v$v.txt
done
