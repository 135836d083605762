#!/bin/bash
# round 5, GPU session 17: bisect of the large-tile GEMM's regression against the round-4 tree on ONE box: (0) round-5 tree with the 8-wave entry point back on
# plain __launch_bounds__(512); (1) + no producer-statistics body in the epilogue fast path; (2) + the round-4 fast path; (r4) the round-4 tree
set -u
O=$(pwd)/gpurun_out/r5s17; mkdir -p $O
R5=$(pwd); R4=$(pwd)/tools/_timing/r4tree
for rep in 1 2; do
  echo "## round-4 tree"; (cd $R4 && timeout 200 python tools/gemm_big_bench.py xlarge large 2>&1 | grep -v amdgpu | sed 's/ring64.128 *[0-9.]* us *[0-9.]* TF.s (rel diff [0-9.e+-]*)//; s/  big kb32.*//' | cut -c1-110)
  echo "## round-5 tree, variant 0"; (cd $R5 && GEMM_BENCH_MODES=64 timeout 200 python tools/gemm_big_bench.py xlarge large 2>&1 | grep -v amdgpu | cut -c1-110)
  for v in 1 2; do echo "## round-5 tree, epilogue variant $v"; (cd $R5 && LWDETR_HIP_LIB=tools/_timing/libepi$v.so GEMM_BENCH_MODES=64 timeout 200 python tools/gemm_big_bench.py xlarge large 2>&1 | grep -v amdgpu | cut -c1-110); done
done | tee $O/gemm_big_bisect.txt
