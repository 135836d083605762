#!/bin/bash
# round 5, GPU session 1: the 4-wave / 128-row form of the large-tile GEMM (two workgroups per CU) - tests, per-shape times, model level
set -u
O=gpurun_out/r5s01; mkdir -p $O
bash tools/box_info.sh > $O/box_info.txt 2>&1
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "large_tile" 2>&1 | tail -8 | tee $O/pytest.txt
timeout 300 python tools/gemm_big_bench.py xlarge large medium 2>&1 | tee $O/gemm_big_bench.txt
for wg2 in 0 2 0 2; do
  echo "xlarge wg2=$wg2"; LWDETR_GEMM_BIG_2WG=$wg2 timeout 300 python bench.py --size xlarge --batch 16 --res 960 --no-cpu-baseline --no-other-configs --no-latency --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
done 2>&1 | tee $O/bench_xlarge.txt
for wg2 in 0 2 0 2; do
  echo "large wg2=$wg2"; LWDETR_GEMM_BIG_2WG=$wg2 timeout 300 python bench.py --size large --batch 32 --no-cpu-baseline --no-other-configs --no-latency --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
done 2>&1 | tee $O/bench_large.txt
for wg2 in 0 2; do
  echo "medium wg2=$wg2"; LWDETR_GEMM_BIG_2WG=$wg2 timeout 300 python bench.py --size medium --batch 64 --dtype bf16 --no-cpu-baseline --no-other-configs --no-latency --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
done 2>&1 | tee $O/bench_medium.txt
