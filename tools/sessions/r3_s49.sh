#!/bin/bash
# round 3, GPU session 49: LWDETR.detect (PostProcess per launch chain): test, bench small / medium
set -u
OUT=gpurun_out/r3_s49; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_model.py -x -q -m gpu -k "detect or two_launch or full_size" 2>&1 | tail -2
for cfg in "--size small --batch 32 --dtype fp16" "--size medium --batch 64 --dtype bf16"; do
  timeout 600 python bench.py $cfg --no-cpu-baseline --no-latency > $OUT/bench_$(echo $cfg | cut -d' ' -f2).json 2> $OUT/bench.err
  python -c "
import json,sys;r=json.loads(open('$OUT/bench_$(echo $cfg | cut -d' ' -f2).json').read().strip().splitlines()[-1]);print(r['config']['workload'][:40], r['value'], r['ms_per_step'])"
done
