#!/bin/bash
# round 3, GPU session 15: two launch chains as the default for >= 32 images: equivalence test, the whole -m gpu suite, bench lines
set -u
OUT=gpurun_out/r3_s15
mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_model.py -x -q -m gpu -k "two_launch_chains" 2>&1 | tail -3
timeout 2400 python -m pytest tests -q -m gpu > $OUT/t_all.log 2>&1; tail -4 $OUT/t_all.log
for cfg in "small 32 fp16 640" "medium 64 bf16 640" "large 32 fp16 640" "xlarge 16 fp16 960" "tiny 32 fp16 640"; do
  set -- $cfg
  timeout 400 python bench.py --size $1 --batch $2 --dtype $3 --res $4 --no-cpu-baseline --steps 20 --warmup 5 > $OUT/bench_$1.json 2> $OUT/bench_$1.err
  LWDETR_STREAMS=1 timeout 400 python bench.py --size $1 --batch $2 --dtype $3 --res $4 --no-cpu-baseline --no-latency --no-roofline --steps 20 --warmup 5 > $OUT/bench_$1_one_chain.json 2> $OUT/bench_$1_one_chain.err
  python - "$1" $OUT/bench_$1.json $OUT/bench_$1_one_chain.json <<'PY'
import json, sys
tag, p2, p1 = sys.argv[1:]
try:
    d = json.loads(open(p2).read().strip().splitlines()[-1]); e = json.loads(open(p1).read().strip().splitlines()[-1])
    print(tag, "two chains", d["value"], d["ms_per_step"], "| one chain", e["value"], e["ms_per_step"], "|", d.get("roofline", {}).get("kernel"), d.get("roofline", {}).get("avg_launch_us"), d.get("latency_bs1_hipgraph_ms"))
except Exception as ex:
    print(tag, "FAILED", ex)
PY
done
