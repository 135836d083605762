#!/bin/bash
# GPU session 18: large-tile GEMM feed probe
set -u
OUT=gpurun_out/s18
mkdir -p $OUT
for v in 0 32 48 64 8; do
  echo "== variant $v"
  if [ $v = 0 ]; then unset LWDETR_HIP_LIB; else export LWDETR_HIP_LIB=$(pwd)/tools/_timing/libbig_v$v.so; fi
  timeout 200 python tools/gemm_feed_probe.py 2>&1 | grep -v amdgpu.ids | tee $OUT/feed_v$v.txt
done
