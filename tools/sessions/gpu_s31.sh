#!/bin/bash
# GPU session 31: HEADS_T epilogue with hoisted divisions; full GPU suite
set -u
OUT=gpurun_out/s31
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "gemm or conv or linear" > $OUT/t_gemm.log 2>&1
tail -2 $OUT/t_gemm.log
for cfg in "xlarge 16 fp16 960" "large 32 fp16 640"; do
  set -- $cfg
  timeout 400 python bench.py --size $1 --batch $2 --dtype $3 --res $4 --no-cpu-baseline --steps 10 --warmup 3 > $OUT/bench_$1.json 2> $OUT/bench_$1.err
  python - "$1" $OUT/bench_$1.json <<'PY'
import json, sys
tag, path = sys.argv[1:]
try:
    d = json.loads(open(path).read().strip().splitlines()[-1])
    print(tag, d["value"], d["ms_per_step"], {k: (v["ms_per_step"], v["launches_per_step"]) for k, v in list(d.get("kernels", {}).items())[:6]})
except Exception as e:
    print("ERR", tag, e); print(open(path.replace(".json", ".err")).read()[-800:])
PY
done
timeout 1500 python -m pytest tests -x -q -m gpu > $OUT/t_all.log 2>&1
tail -4 $OUT/t_all.log
