#!/bin/bash
# round 3, GPU session 22: two-chain determinism probe - which buffer of which chain differs from the part run alone, under several plans
set -u
OUT=gpurun_out/r3_s22
mkdir -p $OUT
run() {  # label, env...
  local label=$1; shift
  for rep in 1 2; do
    echo "== $label process $rep"
    env "$@" timeout 200 python tools/determinism_probe.py small 32 12 2 2>&1 | grep -v amdgpu | cut -c1-600
  done
}
run default X=1
run vit_block_off LWDETR_VIT_BLOCK=0
run mlp_unfused LWDETR_MLP_FUSED=0
run ffn_unfused LWDETR_FFN_FUSED=0
run attn_lds_off LWDETR_ATTN_LDS=0
