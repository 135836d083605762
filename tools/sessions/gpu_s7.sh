#!/bin/bash
# GPU session 7: LDS-ring attention on windows, in-place outputs; full GPU suite
set -u
OUT=gpurun_out/s7
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "attention" > $OUT/t_attn.log 2>&1
tail -6 $OUT/t_attn.log
timeout 600 python tools/attn_bench.py small_b32_f16_win medium_b64_bf16_win xlarge960_b16_f16_win --v=attn_kernel,1x8,2x8 > $OUT/attn_win.txt 2>&1
LWDETR_ATTN_LDS_CFG=104 timeout 200 python tools/attn_bench.py small_b32_f16_win medium_b64_bf16_win --v=1x8 >> $OUT/attn_win.txt 2>&1
cat $OUT/attn_win.txt
timeout 1500 python -m pytest tests -q -m gpu -x > $OUT/t_gpu.log 2>&1
tail -8 $OUT/t_gpu.log
for cfg in "small 32 fp16 640" "xlarge 16 fp16 960"; do
  set -- $cfg
  timeout 400 python bench.py --size $1 --batch $2 --dtype $3 --res $4 --no-cpu-baseline --latency > $OUT/bench_$1.json 2> $OUT/bench_$1.err
  python - <<PY
import json
try:
    d=json.loads(open("$OUT/bench_$1.json").read().strip().splitlines()[-1])
    print("$1", d["value"], d["ms_per_step"], d.get("latency_bs1_ms"), d.get("latency_bs1_hipgraph_ms"), {k:(v["ms_per_step"],v["launches_per_step"]) for k,v in d.get("kernels",{}).items()})
except Exception as e:
    print("ERR $1", e); print(open("$OUT/bench_$1.err").read()[-600:])
PY
done
