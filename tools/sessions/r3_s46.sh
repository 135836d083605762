#!/bin/bash
# round 3, GPU session 46: bench small / medium / large (window attention hd 32 on the one-wave kernel)
set -u
OUT=gpurun_out/r3_s46; mkdir -p $OUT
rocm-smi --showclocks 2>/dev/null | grep -i "sclk\|mclk" | head -4
for cfg in "--size small --batch 32 --dtype fp16" "--size medium --batch 64 --dtype bf16" "--size large --batch 32 --dtype fp16"; do
  timeout 600 python bench.py $cfg --no-cpu-baseline --no-latency > $OUT/bench_$(echo $cfg | cut -d' ' -f2).json 2> $OUT/bench.err
  python -c "
import json,sys;r=json.loads(open('$OUT/bench_$(echo $cfg | cut -d' ' -f2).json').read().strip().splitlines()[-1]);print(r['config']['workload'][:40], r['value'], r['ms_per_step']);print({k:round(v['ms_per_step'],3) for k,v in list(r['kernels'].items())[:9]})"
done
