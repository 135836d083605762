#!/bin/bash
# round 6, GPU session 23: what the block kernel's q / k / v^T stores cost - phase stamps with none / without v^T / without q, k; cache-policy bits on those stores (nt, sc0, sc0 sc1)
set -u
O=$(pwd)/gpurun_out/r6s23; mkdir -p $O
for tag in a0 a16 a32 a64 x2 x1 x17 a0; do for b in 16 32; do
  echo "== $tag batch $b"; LWDETR_HIP_LIB=tools/_timing/liblwdetr_hip_vbt_$tag.so python tools/vitblock_timing.py 192 $b fp16 2>&1 | grep -v amdgpu.ids | grep -v "wave [123]"
done; done | tee $O/vitblock_store_variants.txt
