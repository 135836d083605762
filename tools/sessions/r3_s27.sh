#!/bin/bash
# round 3, GPU session 27: the sampling op under co-running GEMM / attention loads with their LDS-DMA paths switched off
set -u
run() { local label=$1; shift; echo "== $label"; env "$@" PROBE_MSDA=loads timeout 300 python tools/determinism_probe.py small 32 24 -2 2>&1 | grep -E "GemmOp|AttnOp|backbone|whole" | cut -c1-200; }
run "default" X=1
run "GEMM: no LDS-DMA (LWDETR_GEMM_DMA=0 LWDETR_GEMM_BIG=0)" LWDETR_GEMM_DMA=0 LWDETR_GEMM_BIG=0
run "attention: no LDS-DMA (LWDETR_ATTN_LDS=0)" LWDETR_ATTN_LDS=0
run "both off" LWDETR_GEMM_DMA=0 LWDETR_GEMM_BIG=0 LWDETR_ATTN_LDS=0
