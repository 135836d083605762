#!/bin/bash
# round 3, GPU session 20: flakiness hunt - the two-chain / full-size consistency tests repeated, after the store micro-benchmark as in session 18
set -u
OUT=gpurun_out/r3_s20
mkdir -p $OUT
./tools/ubench/store_pattern > /dev/null
for i in 1 2 3 4 5 6 7 8; do
  timeout 300 python -m pytest tests/test_gpu_model.py -x -q -m gpu -k "two_launch or full_size" > $OUT/run_$i.log 2>&1
  echo "run $i: $(tail -1 $OUT/run_$i.log)"
done
grep -l "failed" $OUT/run_*.log | head -3 | while read f; do echo "== $f"; grep -E "^E |^tests/.*Error|def test_" $f | head -20; done
