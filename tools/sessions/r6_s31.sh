#!/bin/bash
# round 6, GPU session 31: tools/microbench/gelu_loop_bench - the block kernel's hidden loop rebuilt in isolation, ingredient by ingredient
set -u
O=$(pwd)/gpurun_out/r6s31; mkdir -p $O
timeout 300 tools/_timing/gelu_loop_bench 256 2>&1 | tee $O/gelu_loop_256wg.txt
timeout 300 tools/_timing/gelu_loop_bench 512 2>&1 | tee $O/gelu_loop_512wg.txt
