#!/bin/bash
# round 6, GPU session 2: gemm_pt after the fixes (bit_cast trap, scalar bases, all 8 pieces in the last slot, counted wait behind an epilogue): tests, A/B, phase timing
set -u
O=$(pwd)/gpurun_out/r6s02; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "persistent_tile" 2>&1 | grep -v '^    ' | tail -8 | cut -c1-300 | tee $O/pytest_pt.txt
for rep in 1 2; do timeout 300 python tools/gemm_big_bench.py xlarge 2>&1 | grep -v amdgpu.ids | cut -c1-260; done | tee $O/gemm_bench_pt.txt
timeout 120 python tools/pt_timing.py 2>&1 | grep -v amdgpu.ids | tee $O/pt_timing.txt
