#!/bin/bash
# round 6, GPU session 26: tools/microbench/filler_bench - the price of one vector instruction of each kind between 32x32x16 MFMAs of a wave alone on its SIMD
set -u
O=$(pwd)/gpurun_out/r6s26; mkdir -p $O
timeout 120 tools/_timing/filler_bench 256 2>&1 | tee $O/filler_bench_256wg.txt
timeout 120 tools/_timing/filler_bench 1 2>&1 | tee $O/filler_bench_1wg.txt
