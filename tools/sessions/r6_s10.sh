#!/bin/bash
# round 6, GPU session 10: few-token block kernel with fragment-major Wp / W1 / Wqkv (LWDETR_MLP_FRAG=1, experiment): correctness (golden tests at batch 1-2 run this kernel), latency A/B
set -u
O=$(pwd)/gpurun_out/r6s10; mkdir -p $O
LWDETR_MLP_FRAG=1 timeout 600 python -m pytest tests/test_gpu_model.py -x -q -m gpu -k "low_precision" 2>&1 | grep -v '^    ' | tail -4 | cut -c1-300 | tee $O/pytest_frag.txt
for rep in 1 2 3; do for f in 0 1; do echo "frag=$f: $(LWDETR_MLP_FRAG=$f python tools/lat_bs1.py 2>/dev/null | tail -1)"; done; done | tee $O/lat_frag.txt
LWDETR_MLP_FRAG=1 timeout 60 python tools/op_times.py --batch 1 2>/dev/null | grep -i "mlp\|sum" | head -14 | tee $O/op_times_frag.txt
