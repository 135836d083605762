#!/bin/bash
# round 6, GPU session 27: filler microbenchmark with the shader clock it ran at; the block kernel's phase stamps with its shader clock
set -u
O=$(pwd)/gpurun_out/r6s27; mkdir -p $O
timeout 300 tools/_timing/filler_bench 256 2>&1 | tee $O/filler_bench_256wg.txt
for b in 16 32; do echo "== batch $b"; LWDETR_HIP_LIB=tools/_timing/liblwdetr_hip_vbt.so python tools/vitblock_timing.py 192 $b fp16 2>&1 | grep -v amdgpu.ids | grep -v "wave [123]"; done | tee $O/vitblock_phases_clock.txt
