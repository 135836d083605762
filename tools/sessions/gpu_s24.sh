#!/bin/bash
# GPU session 24: ablations of the pipelined large-tile GEMM loop
set -u
OUT=gpurun_out/s24
mkdir -p $OUT
for v in 8 32 4; do
  echo "== variant $v"
  export LWDETR_HIP_LIB=$(pwd)/tools/_timing/libbig_v$v.so
  timeout 200 python tools/gemm_big_bench.py xlarge 2>&1 | grep -v amdgpu.ids | sed 's/(rel diff [^)]*)//g' | tee $OUT/big_v$v.txt
done
