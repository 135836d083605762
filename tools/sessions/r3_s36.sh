#!/bin/bash
# round 3, GPU session 36: fragment reads drained before the stage barrier (convolution, attention ring) - stress, probe, tests, bench
set -u
OUT=gpurun_out/r3_s36
mkdir -p $OUT
for v in 0 4; do echo "== conv_stress LWDETR_CONV_PATCH_VAR=$v (4 = without the lgkmcnt wait)"; LWDETR_CONV_PATCH_VAR=$v timeout 300 python tools/conv_stress.py 30 2>&1 | grep -v amdgpu | grep "^load\|idle" | cut -c1-200; done
for rep in 1 2; do echo "== probe small 32 x40, 2 chains"; timeout 300 python tools/determinism_probe.py small 32 40 2 2>&1 | grep -v amdgpu | cut -c1-300 | tail -2; done
timeout 300 python tools/determinism_probe.py large 32 20 2 2>&1 | grep -v amdgpu | cut -c1-300 | tail -2
timeout 1200 python -m pytest tests/test_gpu_model.py tests/test_gpu_chains.py tests/test_gpu_kernels.py -x -q -m gpu > $OUT/t.log 2>&1; echo "tests: $(tail -1 $OUT/t.log)"
timeout 600 python bench.py --no-cpu-baseline --no-latency > $OUT/bench_small.json 2> $OUT/bench.err; python -c "
import json;r=json.loads(open('$OUT/bench_small.json').read().strip().splitlines()[-1]);print('small', r['value'], r['ms_per_step']);print({k:round(v['ms_per_step'],3) for k,v in r['kernels'].items()})"
