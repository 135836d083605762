#!/bin/bash
# round 3, GPU session 32: patch-resident 3x3 convolution (tests, bench with it on / off), outputs written in place by the launch chains
set -u
OUT=gpurun_out/r3_s32
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "conv3x3" > $OUT/t_conv.log 2>&1; echo "conv tests: $(tail -1 $OUT/t_conv.log)"
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_chains.py -x -q -m gpu > $OUT/t_model.log 2>&1; echo "model+chains: $(tail -1 $OUT/t_model.log)"
for cp in 1 0; do
  for cfg in "--size small --batch 32 --dtype fp16" "--size large --batch 32 --dtype fp16"; do
    LWDETR_CONV_PATCH=$cp timeout 600 python bench.py $cfg --no-cpu-baseline --no-latency > $OUT/bench_cp${cp}_$(echo $cfg | cut -d' ' -f2).json 2> $OUT/bench.err
    python - "$OUT/bench_cp${cp}_$(echo $cfg | cut -d' ' -f2).json" <<'PY'
import json, sys
r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], "img/s", r["value"], "ms", r["ms_per_step"])
for k in r.get("kernels", [])[:8]:
    print("   ", {a: (round(b, 4) if isinstance(b, float) else b) for a, b in k.items()})
PY
  done
done
