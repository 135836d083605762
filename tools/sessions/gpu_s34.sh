#!/bin/bash
# GPU session 34: issue priority of the two waves of a SIMD in the large-tile GEMM loop
set -u
OUT=gpurun_out/s34
mkdir -p $OUT
for v in 1 2 3; do
  echo "== prio $v"
  if [ $v = 0 ]; then unset LWDETR_HIP_LIB; else export LWDETR_HIP_LIB=$(pwd)/tools/_timing/libbig_prio$v.so; fi
  timeout 200 python tools/gemm_big_bench.py xlarge large 2>&1 | grep -v amdgpu.ids | sed 's/(rel diff [^)]*)//g; s/ring64\/128 *[0-9.]* us *[0-9.]* TF\/s//; s/big kb32.*//' | tee $OUT/big_prio$v.txt
done
