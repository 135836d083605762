#!/bin/bash
# GPU session 39: persistent large-tile GEMM (next tile's first stage requested before the epilogue)
set -u
OUT=gpurun_out/s39
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "gemm or conv or linear" > $OUT/t_gemm.log 2>&1
tail -3 $OUT/t_gemm.log
timeout 200 python tools/gemm_big_bench.py xlarge large medium 2>&1 | grep -v amdgpu.ids | sed 's/(rel diff [^)]*)//g; s/ring64\/128 *[0-9.]* us *[0-9.]* TF\/s//' | tee $OUT/big.txt
