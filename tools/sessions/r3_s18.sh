#!/bin/bash
# round 3, GPU session 18: store-pattern micro-benchmark; 3 / 4 launch chains on medium / large; two-chain test after the n-chain change
set -u
OUT=gpurun_out/r3_s18
mkdir -p $OUT
./tools/ubench/store_pattern | tee $OUT/store_pattern.txt
timeout 300 python -m pytest tests/test_gpu_model.py -x -q -m gpu -k "two_launch or full_size" 2>&1 | tail -2
run() { tag=$1; shift; env "$@" timeout 400 python bench.py --no-cpu-baseline --no-latency --no-roofline --steps 30 --warmup 5 $BARGS > $OUT/bench_$tag.json 2> $OUT/bench_$tag.err; python - "$tag" $OUT/bench_$tag.json <<'PY'
import json, sys
tag, path = sys.argv[1:]
try:
    d = json.loads(open(path).read().strip().splitlines()[-1]); print(tag, d["value"], d["ms_per_step"])
except Exception as e:
    print(tag, "FAILED", e, open(path.replace(".json", ".err")).read()[-600:])
PY
}
BARGS="--size medium --batch 64 --dtype bf16"
run medium_2 LWDETR_STREAMS=2
run medium_4 LWDETR_STREAMS=4
run medium_8 LWDETR_STREAMS=8
BARGS="--size large --batch 32 --dtype fp16"
run large_2 LWDETR_STREAMS=2
run large_4 LWDETR_STREAMS=4
BARGS="--size small --batch 32 --dtype fp16"
run small_2 LWDETR_STREAMS=2
run small_4 LWDETR_STREAMS=4
BARGS="--size xlarge --batch 16 --dtype fp16 --res 960"
run xlarge_1 LWDETR_STREAMS=1
run xlarge_2 LWDETR_STREAMS=2
