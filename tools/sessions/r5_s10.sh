#!/bin/bash
# round 5, GPU session 10: LayerNorm fold with the statistics out of the producing GEMM's epilogue
set -u
O=gpurun_out/r5s10; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "row_statistics or layernorm_folded or large_tile" 2>&1 | grep -v '^    ' | tail -25 | cut -c1-400 | tee $O/pytest_gemm.txt
timeout 900 python -m pytest tests/test_gpu_model.py -x -q -m gpu -k "layernorm_folded" 2>&1 | grep -v '^    ' | tail -12 | cut -c1-400 | tee $O/pytest_model.txt
for f in "0 1" "1 0" "1 1" "0 1" "1 0" "1 1"; do set -- $f
  echo "xlarge ln_fold=$1 prod_stats=$2"; LWDETR_LN_FOLD=$1 LWDETR_LN_FOLD_STATS=$2 timeout 300 python bench.py --size xlarge --batch 16 --res 960 --no-cpu-baseline --no-other-configs --no-latency --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
done 2>&1 | tee $O/bench_xlarge.txt
python tools/op_times.py --size xlarge --batch 16 --res 960 2>&1 | grep -v amdgpu | cut -c1-110 > $O/op_times_xlarge.txt; head -12 $O/op_times_xlarge.txt; tail -1 $O/op_times_xlarge.txt
timeout 900 python -m pytest tests/test_gpu_baseline_configs.py -x -q -m gpu -k "xlarge" 2>&1 | grep -v '^    ' | tail -6 | cut -c1-1200 | tee $O/pytest_baseline_xlarge.txt
cp gpurun_out/parity_config_xlarge960_b16_fp16.json $O/
