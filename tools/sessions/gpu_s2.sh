#!/bin/bash
# GPU session 2 (round 2): VALU ubench, full GPU test-suite, attention 1x10, GEMM tile experiments at C = 384 / 768
set -u
OUT=gpurun_out/s2
mkdir -p $OUT
( timeout 120 tools/ubench/valu > $OUT/valu.txt 2>&1 )
cat $OUT/valu.txt
timeout 1500 python -m pytest tests -q -m gpu --durations=12 > $OUT/t_gpu.log 2>&1
tail -25 $OUT/t_gpu.log
timeout 300 python tools/attn_bench.py small_b32_f16 large_b32_f16 --v=attn_kernel,1x8,1x10 > $OUT/attn_bench.txt 2>&1
cat $OUT/attn_bench.txt
for cfg in "medium 64 bf16 640" "large 32 fp16 640" "xlarge 16 fp16 960"; do
  set -- $cfg
  for tile in 0 3; do
    if [ $tile = 0 ]; then unset LWDETR_GEMM_TILE; else export LWDETR_GEMM_TILE=$tile; fi
    timeout 400 python bench.py --size $1 --batch $2 --dtype $3 --res $4 --no-cpu-baseline --steps 5 --warmup 2 > $OUT/bench_$1_tile$tile.json 2> $OUT/bench_$1_tile$tile.err
    python - <<PY
import json
try:
    d=json.loads(open("$OUT/bench_$1_tile$tile.json").read().strip().splitlines()[-1])
    print("$1 tile$tile", d["value"], d["ms_per_step"], {k:(v["ms_per_step"],v["launches_per_step"]) for k,v in d.get("kernels",{}).items()})
except Exception as e:
    print("ERR $1 tile$tile", e); print(open("$OUT/bench_$1_tile$tile.err").read()[-600:])
PY
  done
done
unset LWDETR_GEMM_TILE
cat gpurun_out/parity_config_*.json
