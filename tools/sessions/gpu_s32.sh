#!/bin/bash
# GPU session 32: column tile of the large-tile GEMM per shape (256 / 192 / 128)
set -u
OUT=gpurun_out/s32
mkdir -p $OUT
for bn in 256 192 128; do
  echo "== BN $bn"
  LWDETR_GEMM_BIG_BN=$bn timeout 200 python tools/gemm_big_bench.py xlarge large medium 2>&1 | grep -v amdgpu.ids | sed 's/(rel diff [^)]*)//g; s/ring64\/128 *[0-9.]* us *[0-9.]* TF\/s//; s/big kb32.*//' | tee $OUT/big_bn$bn.txt
done
