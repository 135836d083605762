#!/bin/bash
# round 6, GPU session 13: the whole GPU suite on the pruned default build with the pinned default-plan table; bench.py default line (telemetry, medians, cpu baseline)
set -u
O=$(pwd)/gpurun_out/r6s13; mkdir -p $O
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | grep -v '^    ' | tail -8 | cut -c1-300 | tee $O/pytest_gpu.txt
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -3 $O/bench_default.err
python -c "
import json; d=json.loads(open('$O/bench_default.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['ms_per_step_passes'])
print(json.dumps(d['config']['box']))
print(d.get('latency_bs1_hipgraph_ms'), {k: (v.get('img_s'), v.get('latency_bs1_hipgraph_ms_p50')) for k, v in d.get('other_configs', {}).items()})
print(json.dumps({k: d['cpu_baseline'].get(k) for k in ('value','cores','threads_started','runs_of_the_winner','spread','thread_budget','sweep')}))
"
