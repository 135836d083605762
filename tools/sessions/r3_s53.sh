#!/bin/bash
# round 3, GPU session 53: every GEMM of the small / medium forward with the large-tile kernel off (gemm_tuning 0) beside the default choice
set -u
for cfg in "small 32 fp16" "small 16 fp16" "medium 32 bf16"; do
  set -- $cfg
  echo "== $cfg"
  python tools/op_times.py --size $1 --batch $2 --dtype $3 --gemm-big 0 2>&1 | grep "Gemm" | awk '{ if ($3+0 > 20) print }' | cut -c1-220
done
