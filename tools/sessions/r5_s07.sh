#!/bin/bash
# round 5, GPU session 7: LayerNorm folded into the C = 768 GEMMs - kernel + model tests, xlarge / large bench with and without
set -u
O=gpurun_out/r5s07; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "layernorm or large_tile or gemm" 2>&1 | tail -6 | tee $O/pytest_gemm.txt
timeout 900 python -m pytest tests/test_gpu_model.py -x -q -m gpu -k "layernorm_folded or (teacher_forced and xlarge) or (fp32_matches and xlarge)" 2>&1 | tail -6 | tee $O/pytest_model.txt
for f in 0 1 0 1; do
  echo "xlarge ln_fold=$f"; LWDETR_LN_FOLD=$f timeout 300 python bench.py --size xlarge --batch 16 --res 960 --no-cpu-baseline --no-other-configs --no-latency --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
done 2>&1 | tee $O/bench_xlarge.txt
python tools/op_times.py --size xlarge --batch 16 --res 960 2>&1 | grep -v amdgpu | head -24 | cut -c1-110 | tee $O/op_times_xlarge_head.txt
timeout 900 python -m pytest tests/test_gpu_baseline_configs.py -x -q -m gpu -k "xlarge" 2>&1 | tail -4 | tee $O/pytest_baseline_xlarge.txt
