#!/bin/bash
# GPU session 38: full GPU suite + smoke + default bench on the final tree
set -u
OUT=gpurun_out/s38
mkdir -p $OUT
timeout 300 python -c 'import __graft_entry__ as g; g.smoke()' 2>&1 | tail -1
timeout 1500 python -m pytest tests -x -q -m gpu > $OUT/t_all.log 2>&1
tail -3 $OUT/t_all.log
timeout 600 python bench.py --latency > $OUT/bench_default.json 2> $OUT/bench_default.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/s38/bench_default.json").read().strip().splitlines()[-1])
print({k:d[k] for k in ("metric","value","unit","n_gpus","ms_per_step","dtype")}, d["roofline"]["frac"], d["latency_bs1_ms"], d["latency_bs1_hipgraph_ms"], d["cpu_baseline"]["value"])
PY
