#!/bin/bash
# GPU session 21: large-tile GEMM DMA issue schedules
set -u
OUT=gpurun_out/s21
mkdir -p $OUT
for v in 0 1 2; do
  echo "== sched $v"
  if [ $v = 0 ]; then unset LWDETR_HIP_LIB; else export LWDETR_HIP_LIB=$(pwd)/tools/_timing/libbig_s$v.so; fi
  timeout 200 python tools/gemm_big_bench.py xlarge large 2>&1 | grep -v amdgpu.ids | sed 's/(rel diff [^)]*)//g' | tee $OUT/big_s$v.txt
done
