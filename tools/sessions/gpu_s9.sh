#!/bin/bash
# GPU session 9: evidence - kernel stats + PMC summaries for BASELINE configs 2-5, plain bench lines, bs=1 latency
set -u
OUT=gpurun_out/s9
mkdir -p $OUT
bash tools/profile_round.sh small_b32_640_fp16 > $OUT/prof_small.log 2>&1
bash tools/profile_round.sh medium_b64_640_bf16 --size medium --batch 64 --dtype bf16 > $OUT/prof_medium.log 2>&1
bash tools/profile_round.sh large_b32_640_fp16 --size large --batch 32 --dtype fp16 > $OUT/prof_large.log 2>&1
bash tools/profile_round.sh xlarge_b16_960_fp16 --size xlarge --batch 16 --res 960 --dtype fp16 --steps 5 --warmup 2 > $OUT/prof_xlarge.log 2>&1
tail -4 $OUT/prof_*.log
python bench.py --latency > $OUT/bench_small.json 2> $OUT/bench_small.err
python bench.py --size tiny --no-cpu-baseline --latency > $OUT/bench_tiny.json 2> $OUT/bench_tiny.err
python bench.py --size medium --batch 64 --dtype bf16 --no-cpu-baseline --latency > $OUT/bench_medium.json 2> $OUT/bench_medium.err
python bench.py --size large --batch 32 --dtype fp16 --no-cpu-baseline > $OUT/bench_large.json 2> $OUT/bench_large.err
python bench.py --size xlarge --batch 16 --res 960 --dtype fp16 --no-cpu-baseline --steps 10 --warmup 3 > $OUT/bench_xlarge960.json 2> $OUT/bench_xlarge960.err
for f in small tiny medium large xlarge960; do python - <<PY
import json
try:
    d=json.loads(open("$OUT/bench_$f.json").read().strip().splitlines()[-1])
    print("$f", d["value"], d["ms_per_step"], d.get("model_mfma_frac"), d.get("latency_bs1_ms"), d.get("latency_bs1_hipgraph_ms"), d.get("cpu_baseline",{}).get("value"), d.get("cpu_baseline",{}).get("sweep"))
except Exception as e:
    print("ERR $f", e); print(open("$OUT/bench_$f.err").read()[-600:])
PY
done
ls gpurun_out/keep_*
