#!/bin/bash
# GPU session 42: epilogue passes end on a raw barrier (no wait for store acknowledgements)
set -u
OUT=gpurun_out/s42
mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "gemm or conv or linear" > $OUT/t_gemm.log 2>&1; tail -1 $OUT/t_gemm.log
timeout 200 python tools/gemm_big_bench.py xlarge large 2>&1 | grep -v amdgpu.ids | sed 's/(rel diff [^)]*)//g; s/big kb32.*//' | tee $OUT/big.txt
for cfg in "small 32 fp16 640" "xlarge 16 fp16 960"; do
  set -- $cfg
  timeout 400 python bench.py --size $1 --batch $2 --dtype $3 --res $4 --no-cpu-baseline --steps 20 --warmup 5 > $OUT/bench_$1.json 2> $OUT/bench_$1.err
  python - "$1" $OUT/bench_$1.json <<'PY'
import json, sys
tag, path = sys.argv[1:]
d = json.loads(open(path).read().strip().splitlines()[-1])
print(tag, d["value"], d["ms_per_step"], {k: (v["ms_per_step"], v["launches_per_step"]) for k, v in list(d.get("kernels", {}).items())[:5]})
PY
done
