#!/bin/bash
# round 3, GPU session 31: msda.o built without packed-f32 - chain regression tests, MSDA tests, long two-chain determinism runs
set -u
OUT=gpurun_out/r3_s31
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_chains.py tests/test_gpu_msda.py -x -q -m gpu > $OUT/t_chains_msda.log 2>&1; echo "chains+msda: $(tail -1 $OUT/t_chains_msda.log)"
timeout 600 python -m pytest tests/test_gpu_model.py -x -q -m gpu -k "two_launch or full_size" > $OUT/t_model.log 2>&1; echo "model: $(tail -1 $OUT/t_model.log)"
for cfg in "small 32 40" "small 32 40" "medium 64 20" "large 32 20"; do
  echo "== $cfg, 2 chains"
  timeout 300 python tools/determinism_probe.py $cfg 2 2>&1 | grep -v amdgpu | cut -c1-400 | tail -4
done
echo "== per-op stress (probe), small"
PROBE_STRESS=1 timeout 300 python tools/determinism_probe.py small 32 8 -2 2>&1 | grep -v "identical of" | grep -v amdgpu | cut -c1-300
