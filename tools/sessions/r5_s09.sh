#!/bin/bash
set -u
O=gpurun_out/r5s09; mkdir -p $O
for f in 1 0; do
LWDETR_LN_FOLD=$f timeout 900 python -m pytest tests/test_gpu_baseline_configs.py -x -q -m gpu -k "xlarge" 2>&1 | tail -2 | cut -c1-200
cp gpurun_out/parity_config_xlarge960_b16_fp16.json $O/parity_xlarge_fold$f.json
done
timeout 900 python -m pytest tests/test_gpu_model.py -x -q -m gpu -k "layernorm_folded" 2>&1 | tail -3 | cut -c1-300
