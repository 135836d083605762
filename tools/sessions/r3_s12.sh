#!/bin/bash
# round 3, GPU session 12: rest of the -m gpu suite (after the baseline-config matcher fix) + the RCCL one-rank tests
set -u
OUT=gpurun_out/r3_s12
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_dist.py -x -q -m gpu > $OUT/t_dist.log 2>&1; tail -15 $OUT/t_dist.log
timeout 2400 python -m pytest tests -q -m gpu --deselect tests/test_gpu_dist.py > $OUT/t_all.log 2>&1; tail -12 $OUT/t_all.log
