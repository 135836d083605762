#!/bin/bash
# GPU session 23: software-pipelined large-tile GEMM loop vs the round-1 loop
set -u
OUT=gpurun_out/s23
mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "gemm or conv" > $OUT/t_gemm.log 2>&1
tail -3 $OUT/t_gemm.log
for v in 1 0; do
  echo "== pipe $v"
  if [ $v = 1 ]; then unset LWDETR_HIP_LIB; else export LWDETR_HIP_LIB=$(pwd)/tools/_timing/libbig_p$v.so; fi
  timeout 200 python tools/gemm_big_bench.py xlarge large 2>&1 | grep -v amdgpu.ids | sed 's/(rel diff [^)]*)//g' | tee $OUT/big_p$v.txt
done
unset LWDETR_HIP_LIB
timeout 300 python tools/big_timing.py 2>&1 | grep -v amdgpu.ids | tee $OUT/big_timing.txt
