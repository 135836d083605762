#!/bin/bash
# round 3, GPU session 45: window attention, one wave per (window, head): tests, timing, model
set -u
OUT=gpurun_out/r3_s45; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "attention" 2>&1 | tail -3
python tools/attn_bench.py small_b16_f16_win small_b32_f16_win medium_b64_bf16_win large_b32_f16_win --v=attn_kernel,win 2>&1 | grep -v amdgpu
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_chains.py -x -q -m gpu 2>&1 | tail -1
for cfg in "--size small --batch 32 --dtype fp16" "--size medium --batch 64 --dtype bf16"; do
  timeout 600 python bench.py $cfg --no-cpu-baseline --no-latency > $OUT/bench_$(echo $cfg | cut -d' ' -f2).json 2> $OUT/bench.err
  python -c "
import json,sys;r=json.loads(open('$OUT/bench_$(echo $cfg | cut -d' ' -f2).json').read().strip().splitlines()[-1]);print(r['config']['workload'][:40], r['value'], r['ms_per_step']);print({k:round(v['ms_per_step'],3) for k,v in list(r['kernels'].items())[:9]})"
done
