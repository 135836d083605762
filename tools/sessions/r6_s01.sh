#!/bin/bash
# round 6, GPU session 1: the persistent large-tile GEMM (gemm_pt.hip) - parity against torch and bit-equality with gemm_big_kernel, then per-shape A/B
set -u
O=$(pwd)/gpurun_out/r6s01; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "persistent_tile" 2>&1 | grep -v '^    ' | tail -15 | cut -c1-300 | tee $O/pytest_pt.txt
for rep in 1 2; do timeout 300 python tools/gemm_big_bench.py xlarge 2>&1 | cut -c1-260; done | tee $O/gemm_bench_pt.txt
