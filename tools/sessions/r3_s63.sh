#!/bin/bash
# round 3, GPU session 63: is this a slow box, and does LWDETR_ARENA=1 change that? (bench small, alternating)
set -u
for a in 0 1; do
  LWDETR_ARENA=$a timeout 200 python bench.py --no-cpu-baseline --no-latency 2>/dev/null | tail -1 > gpurun_out/b63.json
  python -c "
import json;r=json.loads(open('gpurun_out/b63.json').read());print('LWDETR_ARENA=$a', r['value'], r['ms_per_step'], {k:round(v['ms_per_step'],3) for k,v in list(r['kernels'].items())[:3]})"
done
