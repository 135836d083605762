#!/bin/bash
# GPU session 3: large-tile GEMM correctness + speed, new tests
set -u
OUT=gpurun_out/s3
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "large_tile or qkv_head" > $OUT/t_big.log 2>&1
tail -15 $OUT/t_big.log
timeout 600 python tools/gemm_big_bench.py xlarge medium large > $OUT/gemm_big.txt 2>&1
cat $OUT/gemm_big.txt
timeout 900 python -m pytest tests/test_gpu_baseline_configs.py tests/test_gpu_model.py tests/test_gpu_msda.py tests/test_gpu_preprocess.py -q -m gpu > $OUT/t_model.log 2>&1
tail -15 $OUT/t_model.log
for cfg in "medium 64 bf16 640" "large 32 fp16 640" "xlarge 16 fp16 960"; do
  set -- $cfg
  timeout 400 python bench.py --size $1 --batch $2 --dtype $3 --res $4 --no-cpu-baseline --steps 5 --warmup 2 > $OUT/bench_$1.json 2> $OUT/bench_$1.err
  python - <<PY
import json
try:
    d=json.loads(open("$OUT/bench_$1.json").read().strip().splitlines()[-1])
    print("$1", d["value"], d["ms_per_step"], {k:(v["ms_per_step"],v["launches_per_step"]) for k,v in d.get("kernels",{}).items()})
except Exception as e:
    print("ERR $1", e); print(open("$OUT/bench_$1.err").read()[-600:])
PY
done
cat gpurun_out/parity_config_*.json | grep -E "logit_max|box_max|logit_mean|found|score|size|iou"
