#!/bin/bash
set -u
O=gpurun_out/r5s08; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "layernorm_folded" 2>&1 | tail -30 | cut -c1-300 | tee $O/pytest_gemm.txt
timeout 900 python -m pytest tests/test_gpu_model.py -x -q -m gpu -k "layernorm_folded" 2>&1 | grep -v '^    ' | tail -40 | cut -c1-600 | tee $O/pytest_model.txt
LWDETR_LN_FOLD=1 timeout 900 python -m pytest tests/test_gpu_baseline_configs.py -x -q -m gpu -k "xlarge" 2>&1 | grep -v '^    ' | tail -30 | cut -c1-1500 | tee $O/pytest_baseline_xlarge_fold1.txt
LWDETR_LN_FOLD=0 timeout 900 python -m pytest tests/test_gpu_baseline_configs.py -x -q -m gpu -k "xlarge" 2>&1 | grep -v '^    ' | tail -30 | cut -c1-1500 | tee $O/pytest_baseline_xlarge_fold0.txt
