#!/bin/bash
# round 3, GPU session 62: plan buffers from 256 MB arenas (LWDETR_ARENA=1, default) vs one allocation per buffer (=0), same box; tests
set -u
OUT=gpurun_out/r3_s62; mkdir -p $OUT
for a in 1 0 1 0; do
  LWDETR_ARENA=$a timeout 200 python bench.py --no-cpu-baseline --no-latency 2>/dev/null | tail -1 > $OUT/b.json
  python -c "
import json;r=json.loads(open('$OUT/b.json').read());print('LWDETR_ARENA=$a', r['value'], r['ms_per_step'], {k:round(v['ms_per_step'],3) for k,v in list(r['kernels'].items())[:4]})"
done
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_chains.py -x -q -m gpu 2>&1 | tail -1
