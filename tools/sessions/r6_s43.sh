#!/bin/bash
# round 6, GPU session 43: final tree - the whole -m gpu suite, the default bench line (twice), box info, kernel statistics + counters of config 2
set -u
O=$(pwd)/gpurun_out/r6s43; mkdir -p $O
python -m pytest tests -q -x -m gpu 2>&1 | tail -4 | tee $O/pytest_gpu.txt
python bench.py 2>$O/bench_default.err | tail -1 > $O/bench_default.json
python bench.py 2>/dev/null | tail -1 > $O/bench_default_run2.json
bash tools/box_info.sh > $O/box_info.txt 2>&1
python -c "
import json
for f in ('bench_default.json','bench_default_run2.json'):
    d=json.load(open('$O/'+f)); print(f, d['value'], d['ms_per_step'], d['ms_per_step_passes'].get('after'), d.get('roofline',{}).get('frac'), d.get('roofline',{}).get('avg_launch_us'), d.get('cpu_baseline',{}).get('value'), d['latency_bs1_hipgraph_ms'], {k:v.get('img_s') for k,v in d['other_configs'].items()})
"
bash tools/profile_round.sh r6_small_b32_640_fp16 2>&1 | tail -2
