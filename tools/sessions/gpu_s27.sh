#!/bin/bash
# GPU session 27: branch-free fast path of the shared GEMM epilogue
set -u
OUT=gpurun_out/s27
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -x -q -m gpu > $OUT/t_kernels_model.log 2>&1
tail -3 $OUT/t_kernels_model.log
timeout 200 python tools/gemm_big_bench.py xlarge large 2>&1 | grep -v amdgpu.ids | sed 's/(rel diff [^)]*)//g' | tee $OUT/big.txt
timeout 300 python tools/big_timing.py 2>&1 | grep -v amdgpu.ids | grep kb64 | sed 's/per step.*| epilogue/| epilogue/' | tee $OUT/big_timing.txt
for cfg in "xlarge 16 fp16 960" "large 32 fp16 640" "medium 64 bf16 640" "small 32 fp16 640"; do
  set -- $cfg
  timeout 400 python bench.py --size $1 --batch $2 --dtype $3 --res $4 --no-cpu-baseline --steps 10 --warmup 3 > $OUT/bench_$1.json 2> $OUT/bench_$1.err
  python - "$1" $OUT/bench_$1.json <<'PY'
import json, sys
tag, path = sys.argv[1:]
try:
    d = json.loads(open(path).read().strip().splitlines()[-1])
    print(tag, d["value"], d["ms_per_step"], {k: (v["ms_per_step"], v["launches_per_step"]) for k, v in list(d.get("kernels", {}).items())[:6]})
except Exception as e:
    print("ERR", tag, e); print(open(path.replace(".json", ".err")).read()[-800:])
PY
done
