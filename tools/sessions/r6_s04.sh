#!/bin/bash
# round 6, GPU session 5: gemm_pt with the lean epilogue (bias as the C operand of the first MFMAs, column scale from LDS, in-place swaps)
# tile start; schedule A/B: all 8 pieces in the last slot vs W in slot 0 / A in the last slot (pts = split)
set -u
O=$(pwd)/gpurun_out/r6s05; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "persistent_tile" 2>&1 | grep -v '^    ' | tail -8 | cut -c1-300 | tee $O/pytest_pt.txt
for rep in 1 2; do PT_SKEWS=0,16,s0,s16 timeout 300 python tools/gemm_big_bench.py xlarge 2>&1 | grep -v amdgpu.ids | cut -c1-420; done | tee $O/gemm_bench_pt.txt
timeout 120 python tools/pt_timing.py 2>&1 | grep -v amdgpu.ids | tee $O/pt_timing.txt
