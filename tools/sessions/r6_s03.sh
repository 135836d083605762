#!/bin/bash
# round 6, GPU session 3: start-skew sweep of the persistent GEMM (window in us over the 32 workgroups of an XCD)
set -u
O=$(pwd)/gpurun_out/r6s03; mkdir -p $O
for rep in 1 2; do PT_SKEWS=4,8,12,16,24 timeout 300 python tools/gemm_big_bench.py xlarge 2>&1 | grep -v amdgpu.ids | cut -c1-420; done | tee $O/gemm_bench_pt_skew.txt
