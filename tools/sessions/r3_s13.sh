#!/bin/bash
# round 3, GPU session 13: calibrated parity at the BASELINE configs, then bench + rocprofv3 (kernel stats + PMC passes) per config
set -u
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/r3_s13
mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_baseline_configs.py -q -m gpu > $OUT/t_parity.log 2>&1; tail -6 $OUT/t_parity.log
cp gpurun_out/parity_config_*.json $OUT/ 2>/dev/null
for cfg in "small 32 fp16 640" "medium 64 bf16 640" "large 32 fp16 640" "xlarge 16 fp16 960"; do
  set -- $cfg
  tag=r3_$1_b$2_$4_$3
  timeout 400 python bench.py --size $1 --batch $2 --dtype $3 --res $4 --no-cpu-baseline --steps 20 --warmup 5 > $OUT/bench_$tag.json 2> $OUT/bench_$tag.err
  timeout 900 bash tools/profile_round.sh $tag --size $1 --batch $2 --dtype $3 --res $4 > $OUT/profile_$tag.log 2>&1
  python - "$tag" $OUT/bench_$tag.json <<'PY'
import json, sys
tag, path = sys.argv[1:]
try:
    d = json.loads(open(path).read().strip().splitlines()[-1])
    print(tag, d["value"], d["ms_per_step"], d.get("roofline", {}).get("kernel"), d.get("roofline", {}).get("avg_launch_us"), d.get("roofline", {}).get("frac"), d.get("latency_bs1_hipgraph_ms"))
except Exception as e:
    print(tag, "FAILED", e)
PY
done
ls gpurun_out/ | grep keep_r3
