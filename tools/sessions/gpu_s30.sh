#!/bin/bash
# GPU session 30: per-op times, every GEMM at default dispatch vs "large-tile whenever legal"
set -u
OUT=gpurun_out/s30
mkdir -p $OUT
for cfg in "small 32 fp16 640" "medium 64 bf16 640" "large 32 fp16 640" "xlarge 16 fp16 960"; do
  set -- $cfg
  timeout 300 python tools/op_times.py --size $1 --batch $2 --dtype $3 --res $4 --gemm-big 2 > $OUT/op_$1.txt 2>&1
  echo "== $1"; grep "GemmOp" $OUT/op_$1.txt | sort -k3 -n -r | head -14; tail -1 $OUT/op_$1.txt
done
