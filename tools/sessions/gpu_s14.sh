#!/bin/bash
# GPU session 14: decoder FFN as split-hidden block kernel + finishing LayerNorm chain (2 launches) vs the 3-launch plan
set -u
OUT=gpurun_out/s14
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "ffn_fused or mlp_fused or layernorm" > $OUT/t_ffn.log 2>&1
tail -5 $OUT/t_ffn.log
show() {
  python - "$1" "$2" <<'PY'
import json, sys
tag, path = sys.argv[1:]
try:
    d = json.loads(open(path).read().strip().splitlines()[-1])
    print(tag, d["value"], d["ms_per_step"], d.get("latency_bs1"), {k: (v["ms_per_step"], v["launches_per_step"]) for k, v in list(d.get("kernels", {}).items())[:8]})
except Exception as e:
    print("ERR", tag, e); print(open(path.replace(".json", ".err")).read()[-800:])
PY
}
for fused in 1 0; do
  for cfg in "small 32 fp16 640" "medium 64 bf16 640" "large 32 fp16 640"; do
    set -- $cfg
    LWDETR_FFN_FUSED=$fused timeout 400 python bench.py --size $1 --batch $2 --dtype $3 --res $4 --no-cpu-baseline --steps 20 --warmup 5 --latency > $OUT/bench_$1_f$fused.json 2> $OUT/bench_$1_f$fused.err
    show "$1 fused=$fused" $OUT/bench_$1_f$fused.json
  done
done
timeout 300 python tools/op_times.py --size small --batch 32 > $OUT/op_times_small.txt 2>&1; grep -i "ffn\|Ffn\|layernorm\|total" $OUT/op_times_small.txt | tail -12
timeout 300 python tools/op_times.py --size small --batch 1 > $OUT/op_times_small_b1.txt 2>&1; grep -i "ffn\|Ffn\|total" $OUT/op_times_small_b1.txt | tail -8
timeout 1500 python -m pytest tests -x -q -m gpu > $OUT/t_all.log 2>&1
tail -6 $OUT/t_all.log
