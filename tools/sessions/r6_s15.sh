#!/bin/bash
# round 6, GPU session 15: lwdetr_gemm_few - waves per workgroup (how many CUs the weight stream is spread over)
set -u
O=$(pwd)/gpurun_out/r6s15; mkdir -p $O
for w in 8 4 2 1; do
  echo "waves=$w: $(LWDETR_GEMM_FEW_WAVES=$w python tools/lat_bs1.py 2>/dev/null | tail -1)   conv: $(LWDETR_GEMM_FEW_WAVES=$w timeout 60 python tools/op_times.py --batch 1 2>/dev/null | grep 'amode=1' | head -2 | awk '{print $3}' | tr '\n' ' ')"
done | tee $O/lat_few_waves.txt
echo "off: $(LWDETR_GEMM_FEW=0 python tools/lat_bs1.py 2>/dev/null | tail -1)" | tee -a $O/lat_few_waves.txt
timeout 300 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "few_rows_conv" 2>&1 | tail -2 | tee $O/pytest.txt
