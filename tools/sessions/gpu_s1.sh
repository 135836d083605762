#!/bin/bash
# GPU session 1 (round 2): overlap ubench, attention tests, attention variant bench, per-config bench lines
set -u
OUT=gpurun_out/s1
mkdir -p $OUT
( timeout 120 tools/ubench/overlap > $OUT/overlap.txt 2>&1 ) 
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "attention" > $OUT/t_attn.log 2>&1
tail -5 $OUT/t_attn.log
timeout 900 python tools/attn_bench.py small_b32_f16 medium_b64_bf16 large_b32_f16 xlarge960_b16_f16 > $OUT/attn_bench.txt 2>&1
cat $OUT/attn_bench.txt
timeout 300 python bench.py --no-cpu-baseline > $OUT/bench_small.json 2> $OUT/bench_small.err
LWDETR_ATTN_LDS=0 timeout 300 python bench.py --no-cpu-baseline --no-roofline > $OUT/bench_small_oldattn.json 2> $OUT/bench_small_oldattn.err
timeout 300 python bench.py --size medium --batch 64 --dtype bf16 --no-cpu-baseline > $OUT/bench_medium.json 2> $OUT/bench_medium.err
timeout 300 python bench.py --size large --batch 32 --dtype fp16 --no-cpu-baseline > $OUT/bench_large.json 2> $OUT/bench_large.err
timeout 400 python bench.py --size xlarge --res 960 --batch 16 --dtype fp16 --no-cpu-baseline --steps 5 --warmup 2 > $OUT/bench_xlarge960.json 2> $OUT/bench_xlarge960.err
for f in small small_oldattn medium large xlarge960; do echo "== $f"; python - <<PY
import json
try:
    d=json.loads(open("$OUT/bench_$f.json").read().strip().splitlines()[-1])
    print(d["value"], d["ms_per_step"], {k:(v["ms_per_step"],v["launches_per_step"]) for k,v in d.get("kernels",{}).items()})
except Exception as e:
    print("ERR", e); print(open("$OUT/bench_$f.err").read()[-800:])
PY
done
cat $OUT/overlap.txt
