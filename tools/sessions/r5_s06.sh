#!/bin/bash
# round 5, GPU session 6: single-image latency - which GEMM schedule knobs move it (64-deep stages, ring depth, tile shape), per-op listing
set -u
O=gpurun_out/r5s06; mkdir -p $O
( for rep in 1 2; do
  echo "default:            $(python tools/lat_bs1.py 2>&1 | grep -v amdgpu)"
  echo "LWDETR_GEMM_KB=64:  $(LWDETR_GEMM_KB=64 python tools/lat_bs1.py 2>&1 | grep -v amdgpu)"
  echo "LWDETR_GEMM_NST=2:  $(LWDETR_GEMM_NST=2 python tools/lat_bs1.py 2>&1 | grep -v amdgpu)"
  echo "LWDETR_GEMM_NST=4:  $(LWDETR_GEMM_NST=4 python tools/lat_bs1.py 2>&1 | grep -v amdgpu)"
  echo "LWDETR_VB_HALF=0:   $(LWDETR_VB_HALF=0 python tools/lat_bs1.py 2>&1 | grep -v amdgpu)"
  echo "LWDETR_CHAIN=1:     $(LWDETR_CHAIN=1 python tools/lat_bs1.py 2>&1 | grep -v amdgpu)"
  echo "LWDETR_MLP_SMALL_TT=2: $(LWDETR_MLP_SMALL_TT=2 python tools/lat_bs1.py 2>&1 | grep -v amdgpu)"
  echo "LWDETR_ATTN_LDS=3:  $(LWDETR_ATTN_LDS=3 python tools/lat_bs1.py 2>&1 | grep -v amdgpu)"
done ) | tee $O/lat_bs1_knobs.txt
python tools/op_times.py --size small --batch 1 2>&1 | grep -v amdgpu > $O/op_times_small_b1.txt; tail -3 $O/op_times_small_b1.txt
LWDETR_GEMM_KB=64 python tools/op_times.py --size small --batch 1 2>&1 | grep -v amdgpu > $O/op_times_small_b1_kb64.txt; tail -3 $O/op_times_small_b1_kb64.txt
