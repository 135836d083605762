#!/bin/bash
# round 3, GPU session 16: final default bench line (config 2, with cpu_baseline), the fixed consistency test, other configs' lines
set -u
OUT=gpurun_out/r3_s16
mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_model.py -x -q -m gpu -k "full_size_batch or two_launch" 2>&1 | tail -2
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench_default.json 2> $OUT/bench_default.err; tail -2 $OUT/bench_default.err
for cfg in "medium 64 bf16 640" "large 32 fp16 640"; do
  set -- $cfg
  timeout 400 python bench.py --size $1 --batch $2 --dtype $3 --res $4 --no-cpu-baseline --steps 20 --warmup 5 > $OUT/bench_$1.json 2> $OUT/bench_$1.err
done
python - <<'PY'
import json
for w in ("default", "medium", "large"):
    try:
        d = json.loads(open(f"gpurun_out/r3_s16/bench_{w}.json").read().strip().splitlines()[-1])
        print(w, d["value"], d["ms_per_step"], d["config"].get("launch_chains"), d["roofline"]["kernel"], d["roofline"]["avg_launch_us"], d["roofline"]["frac"], d["roofline"].get("mfma_busy_frac"), d.get("latency_bs1_hipgraph_ms"), (d.get("cpu_baseline") or {}).get("value"), (d.get("cpu_baseline") or {}).get("host"))
    except Exception as e:
        print(w, "FAILED", e)
PY
