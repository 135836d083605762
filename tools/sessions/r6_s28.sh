#!/bin/bash
# round 6, GPU session 28: a flaky parity failure of the block kernel at M = 64000 (two workgroups per CU) - which of the round's changes it belongs to
set -u
O=$(pwd)/gpurun_out/r6s28; mkdir -p $O
for lib in cur v1 v2; do
  L=tools/_timing/liblwdetr_$lib.so; [ $lib = cur ] && L=lw-detr_amd/liblwdetr_hip.so
  for i in 1 2 3 4 5 6; do echo "$lib run $i: $(LWDETR_HIP_LIB=$L python -m pytest tests/test_gpu_kernels.py -q -m gpu -k 'test_vit_block and 64000' 2>&1 | grep -E 'passed|failed' | tail -1)"; done
done | tee $O/flaky.txt
