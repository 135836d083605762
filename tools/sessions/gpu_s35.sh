#!/bin/bash
# GPU session 35: validation of the committed state: smoke, full GPU suite, default bench line, launch check
set -u
OUT=gpurun_out/s35
mkdir -p $OUT
timeout 300 python -c 'import __graft_entry__ as g; g.smoke()' 2>&1 | tail -2
timeout 1500 python -m pytest tests -x -q -m gpu > $OUT/t_all.log 2>&1
tail -4 $OUT/t_all.log
timeout 600 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/s35/bench_default.json").read().strip().splitlines()[-1])
print({k:d[k] for k in ("metric","value","unit","n_gpus","steps","warmup","ms_per_step","dtype","vs_baseline")}, d["roofline"], d["cpu_baseline"]["value"])
PY
timeout 300 python bench.py --launch-check --no-cpu-baseline 2>&1 | tail -3
