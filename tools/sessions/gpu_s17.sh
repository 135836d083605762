#!/bin/bash
# GPU session 17: large-tile GEMM ablations (what bounds it: DMA, LDS fragment reads or MFMA issue)
set -u
OUT=gpurun_out/s17
mkdir -p $OUT
for v in 0 8 16 32 24; do
  echo "== variant $v"
  if [ $v = 0 ]; then unset LWDETR_HIP_LIB; else export LWDETR_HIP_LIB=$(pwd)/tools/_timing/libbig_v$v.so; fi
  timeout 200 python tools/gemm_big_bench.py xlarge large 2>&1 | grep -v amdgpu.ids | sed 's/(rel diff [^)]*)//g' | tee $OUT/big_v$v.txt
done
