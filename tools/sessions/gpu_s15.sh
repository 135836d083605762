#!/bin/bash
# GPU session 15: decoder FFN split count sweep
set -u
OUT=gpurun_out/s15
mkdir -p $OUT
for s in 32 16 8 4 2; do
  echo "== LWDETR_FFN_SPLITS=$s"
  LWDETR_FFN_SPLITS=$s timeout 200 python tools/ffn_bench.py 2>&1 | tee -a $OUT/ffn_s$s.txt | tail -5
done
LWDETR_FFN_SPLITS=32 timeout 200 python tools/ffn_bench.py --c 384 --rows 300 4800 9600 2>&1 | tee $OUT/ffn_c384.txt | tail -4
