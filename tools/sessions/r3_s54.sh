#!/bin/bash
# round 3, GPU session 54: attn_kernel with all loads of a short sequence up front: tests, timing
set -u
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "attention" 2>&1 | tail -1
python tools/attn_bench.py small_b16_f16_win small_b32_f16_win medium_b64_bf16_win --v=attn_kernel,short,win 2>&1 | grep -v amdgpu
