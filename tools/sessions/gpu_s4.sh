#!/bin/bash
# GPU session 4: big GEMM v2 (interleaved DMA), parity-at-config tests v3, fused-vs-unfused ViT blocks at C = 384
set -u
OUT=gpurun_out/s4
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "large_tile or qkv_head or linear_bias" > $OUT/t_big.log 2>&1
tail -5 $OUT/t_big.log
timeout 600 python tools/gemm_big_bench.py xlarge medium large > $OUT/gemm_big.txt 2>&1
cat $OUT/gemm_big.txt
timeout 900 python -m pytest tests/test_gpu_baseline_configs.py -q -m gpu > $OUT/t_cfg.log 2>&1
tail -12 $OUT/t_cfg.log
for cfg in "medium 64 bf16 640" "large 32 fp16 640"; do
  set -- $cfg
  for fused in 1 0; do
    LWDETR_MLP_FUSED=$fused timeout 400 python bench.py --size $1 --batch $2 --dtype $3 --res $4 --no-cpu-baseline --steps 5 --warmup 2 > $OUT/bench_$1_fused$fused.json 2> $OUT/bench_$1_fused$fused.err
    python - <<PY
import json
try:
    d=json.loads(open("$OUT/bench_$1_fused$fused.json").read().strip().splitlines()[-1])
    print("$1 fused=$fused", d["value"], d["ms_per_step"], {k:(v["ms_per_step"],v["launches_per_step"]) for k,v in d.get("kernels",{}).items()})
except Exception as e:
    print("ERR $1", e); print(open("$OUT/bench_$1_fused$fused.err").read()[-600:])
PY
  done
done
timeout 400 python bench.py --size xlarge --batch 16 --dtype fp16 --res 960 --no-cpu-baseline --steps 5 --warmup 2 > $OUT/bench_xlarge.json 2> $OUT/bench_xlarge.err
python - <<PY
import json
d=json.loads(open("$OUT/bench_xlarge.json").read().strip().splitlines()[-1])
print("xlarge", d["value"], d["ms_per_step"], {k:(v["ms_per_step"],v["launches_per_step"]) for k,v in d.get("kernels",{}).items()})
PY
cat gpurun_out/parity_config_*.json | grep -E "logit_max|box_max|logit_mean|found|\"score|size|overlap|gap"
