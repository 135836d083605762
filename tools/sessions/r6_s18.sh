#!/bin/bash
# round 6, GPU session 18: lwdetr_gemm_few on the long-K plain launches of the unfused C = 384 single-image path (fc2: M = 1600, N = 384, K = 1536)?
set -u
O=$(pwd)/gpurun_out/r6s18; mkdir -p $O
for m in 1 2; do echo "## LWDETR_GEMM_FEW=$m"; LWDETR_GEMM_FEW=$m timeout 90 python tools/op_times.py --size large --batch 1 2>/dev/null | grep "Gemm" | awk '{print $4, $6, $7, $8}' | sort | uniq -c | sort -k2 -n | head -30; done | tee $O/op_times_large.txt
for rep in 1 2; do for m in 1 2; do echo "large few=$m: $(LWDETR_GEMM_FEW=$m python tools/lat_bs1.py --size large 2>/dev/null | tail -1)"; echo "medium few=$m: $(LWDETR_GEMM_FEW=$m python tools/lat_bs1.py --size medium --dtype bf16 2>/dev/null | tail -1)"; done; done | tee $O/lat.txt
