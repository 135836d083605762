#!/bin/bash
# round 3, GPU session 51: final tree - bench line + rocprofv3 (kernel stats + PMC passes, one launch chain) per BASELINE config, default bench line
set -u
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/r3_s51
mkdir -p $OUT
bash tools/box_info.sh > $OUT/box_info.log 2>&1
timeout 600 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
timeout 300 python bench.py --size tiny --no-cpu-baseline --no-latency > $OUT/bench_tiny.json 2> $OUT/bench_tiny.err
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
python - <<'PY'
import json
for f in ("default", "tiny"):
    try:
        d = json.loads(open(f"gpurun_out/r3_s51/bench_{f}.json").read().strip().splitlines()[-1])
        print(f, d["value"], d["ms_per_step"], d.get("roofline", {}).get("frac"), d.get("latency_bs1_hipgraph_ms"), (d.get("cpu_baseline") or {}).get("value"))
    except Exception as e:
        print(f, "FAILED", e)
PY
ls gpurun_out/ | grep keep_r3
