#!/bin/bash
# round 6, GPU session 11: lwdetr_vit_block_few in the launch plan - kernel test (bit-identical to lwdetr_mlp_fused), golden-batch model tests, latency A/B incl. 32-token workgroups
set -u
O=$(pwd)/gpurun_out/r6s11; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "vit_block_few or mlp_fused" 2>&1 | grep -v '^    ' | tail -4 | cut -c1-300 | tee $O/pytest_few.txt
timeout 900 python -m pytest tests/test_gpu_model.py -x -q -m gpu 2>&1 | grep -v '^    ' | tail -4 | cut -c1-300 | tee $O/pytest_model.txt
for rep in 1 2; do
  echo "few=0:      $(LWDETR_VIT_BLOCK_FEW=0 python tools/lat_bs1.py 2>/dev/null | tail -1)"
  echo "few=1:      $(python tools/lat_bs1.py 2>/dev/null | tail -1)"
  echo "few=1 tt=2: $(LWDETR_MLP_SMALL_TT=2 python tools/lat_bs1.py 2>/dev/null | tail -1)"
  echo "tiny few=0: $(LWDETR_VIT_BLOCK_FEW=0 python tools/lat_bs1.py --size tiny 2>/dev/null | tail -1)"
  echo "tiny few=1: $(python tools/lat_bs1.py --size tiny 2>/dev/null | tail -1)"
done | tee $O/lat_few.txt
