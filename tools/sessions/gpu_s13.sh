#!/bin/bash
# GPU session 13: C = 384 block kernel as 4 waves x 32 tokens (one wave per SIMD, 512 registers) vs 8 waves x 16 tokens
set -u
OUT=gpurun_out/s13
mkdir -p $OUT
export LWDETR_HIP_LIB=$(pwd)/tools/_timing/liblwdetr_hip_nw4.so
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "mlp_fused and 384" > $OUT/t_mlp.log 2>&1
tail -5 $OUT/t_mlp.log
for cfg in "medium 64 bf16 640" "large 32 fp16 640"; do
  set -- $cfg
  timeout 400 python bench.py --size $1 --batch $2 --dtype $3 --res $4 --no-cpu-baseline --steps 10 --warmup 3 > $OUT/bench_$1_nw4.json 2> $OUT/bench_$1_nw4.err
  python - <<PY
import json
try:
    d=json.loads(open("$OUT/bench_$1_nw4.json").read().strip().splitlines()[-1])
    print("$1 nw4", d["value"], d["ms_per_step"], {k:(v["ms_per_step"],v["launches_per_step"]) for k,v in list(d.get("kernels",{}).items())[:4]})
except Exception as e:
    print("ERR $1", e); print(open("$OUT/bench_$1_nw4.err").read()[-600:])
PY
done
timeout 600 python -m pytest tests/test_gpu_baseline_configs.py -x -q -m gpu -k "medium or large" > $OUT/t_cfg.log 2>&1
tail -4 $OUT/t_cfg.log
