#!/bin/bash
# round 6, GPU session 16: lwdetr_gemm_few also for the plain single-segment Linear launches of one image (LWDETR_GEMM_FEW=2)? per-launch listing + latency
set -u
O=$(pwd)/gpurun_out/r6s16; mkdir -p $O
for m in 1 2; do
  echo "## LWDETR_GEMM_FEW=$m"; LWDETR_GEMM_FEW=$m timeout 60 python tools/op_times.py --batch 1 2>/dev/null | grep "Gemm\|sum" | cut -c1-150
done | tee $O/op_times_few_plain.txt
for rep in 1 2; do for m in 1 2; do echo "few=$m: $(LWDETR_GEMM_FEW=$m python tools/lat_bs1.py 2>/dev/null | tail -1)"; done; done | tee $O/lat.txt
LWDETR_GEMM_FEW=2 timeout 600 python -m pytest tests/test_gpu_model.py -x -q -m gpu -k "low_precision or golden" 2>&1 | tail -2 | tee $O/pytest.txt
