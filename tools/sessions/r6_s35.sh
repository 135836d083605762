#!/bin/bash
# round 6, GPU session 35: final tree - the whole -m gpu suite, the default bench line (twice), box info
set -u
O=$(pwd)/gpurun_out/r6s35; mkdir -p $O
python -m pytest tests -q -x -m gpu 2>&1 | tail -4 | tee $O/pytest_gpu.txt
python bench.py 2>$O/bench_default.err | tail -1 > $O/bench_default.json
python bench.py 2>/dev/null | tail -1 > $O/bench_default_run2.json
bash tools/box_info.sh > $O/box_info.txt 2>&1
python -c "
import json
for f in ('bench_default.json','bench_default_run2.json'):
    d=json.load(open('$O/'+f)); print(f, d['value'], d['ms_per_step'], d['ms_per_step_passes'].get('after'), d.get('roofline',{}).get('frac'), d.get('cpu_baseline',{}).get('value'), {k:(v.get('value') if isinstance(v,dict) else v) for k,v in d.get('other_configs',{}).items()} if isinstance(d.get('other_configs'),dict) else '')
"
