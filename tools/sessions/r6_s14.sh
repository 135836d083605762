#!/bin/bash
# round 6, GPU session 14: lwdetr_gemm_few (the projector's 3x3 convolutions at one or two images on fragment-major weights): kernel test, golden-batch model tests, latency A/B
set -u
O=$(pwd)/gpurun_out/r6s14; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "few_rows" 2>&1 | grep -v '^    ' | tail -4 | cut -c1-300 | tee $O/pytest_few.txt
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_default_plan.py -x -q -m gpu 2>&1 | grep -v '^    ' | tail -6 | cut -c1-400 | tee $O/pytest_model.txt
for rep in 1 2; do
  echo "few=0: $(LWDETR_GEMM_FEW=0 python tools/lat_bs1.py 2>/dev/null | tail -1)"
  echo "few=1: $(python tools/lat_bs1.py 2>/dev/null | tail -1)"
  echo "large few=0: $(LWDETR_GEMM_FEW=0 python tools/lat_bs1.py --size large 2>/dev/null | tail -1)"
  echo "large few=1: $(python tools/lat_bs1.py --size large 2>/dev/null | tail -1)"
done | tee $O/lat_few.txt
timeout 60 python tools/op_times.py --batch 1 2>/dev/null | grep -i "amode=1\|sum" | head -8 | cut -c1-150 | tee $O/op_times.txt
