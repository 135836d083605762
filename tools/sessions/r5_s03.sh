#!/bin/bash
# round 5, GPU session 3: block kernel at C = 192 with 32 tokens per wave (128 per workgroup), <= 256 registers, two workgroups per CU (LWDETR_VB_HALF=1)
set -u
O=gpurun_out/r5s03; mkdir -p $O
LWDETR_VB_HALF=1 timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "vit_block" 2>&1 | tail -5 | tee $O/pytest_half.txt
for h in 0 1 0 1; do
  for b in 32 16; do echo "half=$h batch=$b"; LWDETR_VB_HALF=$h ONLY=vit_block timeout 120 python tools/vitblock_bench.py 192 $b fp16 30 2>&1 | grep -v amdgpu; done
done | tee $O/vitblock_bench.txt
for h in 0 1 0 1; do
  echo "small half=$h"; LWDETR_VB_HALF=$h timeout 300 python bench.py --no-cpu-baseline --no-other-configs --no-latency --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
done 2>&1 | tee $O/bench_small.txt
for h in 0 1; do
  echo "tiny half=$h"; LWDETR_VB_HALF=$h timeout 300 python bench.py --size tiny --no-cpu-baseline --no-other-configs --no-latency --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
done 2>&1 | tee $O/bench_tiny.txt
