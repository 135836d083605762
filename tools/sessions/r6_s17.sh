#!/bin/bash
# round 6, GPU session 17: default-plan table after lwdetr_gemm_few went into the plan; golden / low-precision model tests; latency of all five sizes, few-row kernels on / off
set -u
O=$(pwd)/gpurun_out/r6s17; mkdir -p $O
python tests/test_gpu_default_plan.py > $O/default_plan_stdout.json 2> $O/default_plan_err.txt; tail -2 $O/default_plan_err.txt
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_kernels.py -x -q -m gpu -k "low_precision or golden or few" 2>&1 | tail -2 | tee $O/pytest.txt
for sz in tiny small medium large; do
  dt=fp16; [ $sz = medium ] && dt=bf16
  echo "$sz r5-equivalent (LWDETR_VIT_BLOCK_FEW=0 LWDETR_GEMM_FEW=0): $(LWDETR_VIT_BLOCK_FEW=0 LWDETR_GEMM_FEW=0 python tools/lat_bs1.py --size $sz --dtype $dt 2>/dev/null | tail -1)"
  echo "$sz default: $(python tools/lat_bs1.py --size $sz --dtype $dt 2>/dev/null | tail -1)"
done | tee $O/lat_all_sizes.txt
echo "xlarge 960 default: $(python tools/lat_bs1.py --size xlarge --res 960 2>/dev/null | tail -1)" | tee -a $O/lat_all_sizes.txt
( echo "## small 640x640 fp16, batch 1 (the single-image path), round-6 tree"; timeout 60 python tools/op_times.py --batch 1 2>/dev/null ) | cut -c1-200 > $O/r6_op_times_small_b1.txt; tail -2 $O/r6_op_times_small_b1.txt
