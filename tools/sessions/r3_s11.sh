#!/bin/bash
# round 3, GPU session 11: the whole -m gpu suite on the new launch plan
set -u
OUT=gpurun_out/r3_s11
mkdir -p $OUT
timeout 2400 python -m pytest tests -x -q -m gpu > $OUT/t_all.log 2>&1; tail -8 $OUT/t_all.log
