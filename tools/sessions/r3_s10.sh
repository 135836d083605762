#!/bin/bash
# round 3, GPU session 10: model level - bench.py with the rebuilt block kernel (LWDETR_VIT_BLOCK=1, default) vs lwdetr_mlp_fused (=0)
set -u
OUT=gpurun_out/r3_s10
mkdir -p $OUT
for cfg in "small 32 fp16" "medium 64 bf16" "tiny 32 fp16"; do
  set -- $cfg
  for vb in 1 0; do
    LWDETR_VIT_BLOCK=$vb timeout 400 python bench.py --size $1 --batch $2 --dtype $3 --no-cpu-baseline --no-latency --steps 20 --warmup 5 > $OUT/bench_$1_vb$vb.json 2> $OUT/bench_$1_vb$vb.err
    python - "$1 vb=$vb" $OUT/bench_$1_vb$vb.json <<'PY'
import json, sys
tag, path = sys.argv[1:]
try:
    d = json.loads(open(path).read().strip().splitlines()[-1])
    print(tag, d["value"], d["ms_per_step"], {k: (v["ms_per_step"], v["launches_per_step"]) for k, v in list(d.get("kernels", {}).items())[:4]})
except Exception as e:
    print(tag, "FAILED", e)
PY
  done
done
tail -3 $OUT/bench_small_vb1.err
