#!/bin/bash
# round 5, GPU session 2: start skew of the first round's workgroups in the large-tile GEMM (LWDETR_GEMM_BIG_STAGGER = groups * 1000 + tenths of a us per group)
set -u
O=gpurun_out/r5s02; mkdir -p $O
for sg in 0 2040 4020 4040 8010 8020 8035 0; do
  echo "## stagger $sg"; LWDETR_GEMM_BIG_STAGGER=$sg GEMM_BENCH_MODES=64 timeout 120 python tools/gemm_big_bench.py xlarge large 2>&1 | grep -v amdgpu
done | tee $O/stagger_shapes.txt
for sg in 0 4020 8020 0 4040 8035; do
  echo "xlarge stagger=$sg"; LWDETR_GEMM_BIG_STAGGER=$sg timeout 300 python bench.py --size xlarge --batch 16 --res 960 --no-cpu-baseline --no-other-configs --no-latency --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
done 2>&1 | tee $O/bench_xlarge.txt
