#!/bin/bash
# GPU session 29: fragment reads per MFMA in the pipelined loop
set -u
OUT=gpurun_out/s29
mkdir -p $OUT
for v in r2 r3; do
  echo "== $v"
  export LWDETR_HIP_LIB=$(pwd)/tools/_timing/libbig_$v.so
  timeout 200 python tools/gemm_big_bench.py xlarge large 2>&1 | grep -v amdgpu.ids | sed 's/(rel diff [^)]*)//g; s/ring64\/128 *[0-9.]* us *[0-9.]* TF\/s//' | tee $OUT/big_$v.txt
done
