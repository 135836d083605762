#!/bin/bash
# round 3, GPU session 1: first run of the rebuilt fused ViT block kernel (lwdetr_vit_block): kernel tests, old vs new timing
set -u
OUT=gpurun_out/r3_s01
mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "vit_block" > $OUT/t_vb.log 2>&1; tail -25 $OUT/t_vb.log
for cfg in "192 32 fp16" "384 32 fp16" "384 64 bf16"; do
  timeout 200 python tools/vitblock_bench.py $cfg 2>&1 | grep -v amdgpu.ids | tee -a $OUT/vb_bench.txt
done
timeout 400 python bench.py --no-cpu-baseline --no-latency --steps 20 --warmup 5 > $OUT/bench_small.json 2> $OUT/bench_small.err; tail -3 $OUT/bench_small.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r3_s01/bench_small.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], {k: (v["ms_per_step"], v["launches_per_step"]) for k, v in list(d.get("kernels", {}).items())[:6]})
PY
