#!/bin/bash
# round 6, GPU session 7: gemm_pt + residual rows touched into L2 two steps ahead of the epilogue
set -u
O=$(pwd)/gpurun_out/r6s07; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "persistent_tile" 2>&1 | grep -v '^    ' | tail -8 | cut -c1-300 | tee $O/pytest_pt.txt
timeout 300 python tools/gemm_big_bench.py xlarge 2>&1 | grep -v amdgpu.ids | cut -c1-420 | tee $O/gemm_bench_pt.txt
timeout 120 python tools/pt_timing.py 2>&1 | grep -v amdgpu.ids | grep -B1 -A8 "ablation 0" | tee $O/pt_timing.txt
