#!/bin/bash
# round 3, GPU session 52: final tree - chain tests incl. large, two-chain determinism probes (medium, large), PMC re-run with the new kernel classes
set -u
OUT=gpurun_out/r3_s52; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_chains.py -x -q -m gpu 2>&1 | tail -2
for cfg in "medium 64 16" "large 32 16" "small 32 30"; do echo "== probe $cfg, 2 chains"; timeout 300 python tools/determinism_probe.py $cfg 2 2>&1 | grep -v amdgpu | cut -c1-300 | tail -2; done
for cfg in "small 32 fp16 640" "medium 64 bf16 640"; do
  set -- $cfg
  tag=r3_$1_b$2_$4_$3
  timeout 900 bash tools/profile_round.sh $tag --size $1 --batch $2 --dtype $3 --res $4 > $OUT/profile_$tag.log 2>&1
done
ls gpurun_out/ | grep keep_r3
