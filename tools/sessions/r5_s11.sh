#!/bin/bash
# round 5, GPU session 11: producer-epilogue statistics with the DPP all-reduce + pivot form; LDS-ring attention for one image at hd 16
set -u
O=gpurun_out/r5s11; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "row_statistics or layernorm_folded or attention" 2>&1 | grep -v '^    ' | tail -12 | cut -c1-400 | tee $O/pytest.txt
for f in "1 0" "1 1" "1 0" "1 1" "0 0"; do set -- $f
  echo "xlarge ln_fold=$1 prod_stats=$2"; LWDETR_LN_FOLD=$1 LWDETR_LN_FOLD_STATS=$2 timeout 300 python bench.py --size xlarge --batch 16 --res 960 --no-cpu-baseline --no-other-configs --no-latency --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
done 2>&1 | tee $O/bench_xlarge.txt
LWDETR_LN_FOLD_STATS=1 python tools/op_times.py --size xlarge --batch 16 --res 960 2>&1 | grep -v amdgpu | cut -c1-110 | head -9 | tee $O/op_times_xlarge_prod.txt
LWDETR_LN_FOLD_STATS=1 timeout 900 python -m pytest tests/test_gpu_model.py -x -q -m gpu -k "layernorm_folded" 2>&1 | tail -2
