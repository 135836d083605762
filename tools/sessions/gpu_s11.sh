#!/bin/bash
# GPU session 11: conv3x3 + BN=192 on the large-tile GEMM kernel
set -u
OUT=gpurun_out/s11
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "gemm" > $OUT/t_gemm.log 2>&1
tail -8 $OUT/t_gemm.log
python tools/op_times.py --size large --gemm-big 0 > $OUT/op_large.txt 2>&1
grep "Gemm" $OUT/op_large.txt | cut -c1-75,190-260 | awk '{k=$5" "$6" "$7" "$8; if(!seen[k]++) print}' | head -30
grep "sum of" $OUT/op_large.txt
for cfg in "medium 64 bf16 640" "large 32 fp16 640" "xlarge 16 fp16 960" "small 32 fp16 640"; do
  set -- $cfg
  timeout 400 python bench.py --size $1 --batch $2 --dtype $3 --res $4 --no-cpu-baseline --steps 10 --warmup 3 > $OUT/bench_$1.json 2> $OUT/bench_$1.err
  python - <<PY
import json
try:
    d=json.loads(open("$OUT/bench_$1.json").read().strip().splitlines()[-1])
    print("$1", d["value"], d["ms_per_step"], {k:(v["ms_per_step"],v["launches_per_step"]) for k,v in d.get("kernels",{}).items()})
except Exception as e:
    print("ERR $1", e); print(open("$OUT/bench_$1.err").read()[-600:])
PY
done
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_baseline_configs.py -x -q -m gpu > $OUT/t_model.log 2>&1
tail -5 $OUT/t_model.log
