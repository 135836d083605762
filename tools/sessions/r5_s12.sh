#!/bin/bash
# round 5, GPU session 12: split-K for the few-row GEMMs (single-image latency), regression check of the GEMM epilogue restructure
set -u
O=gpurun_out/r5s12; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "gemm or layernorm or conv or row_stat" 2>&1 | grep -v '^    ' | tail -15 | cut -c1-400 | tee $O/pytest_gemm.txt
( for rep in 1 2; do
  echo "split-K auto:  $(python tools/lat_bs1.py 2>&1 | grep -v amdgpu)"
  echo "split-K off:   $(LWDETR_GEMM_SPLITK=0 python tools/lat_bs1.py 2>&1 | grep -v amdgpu)"
  echo "split-K 2:     $(LWDETR_GEMM_SPLITK=2 python tools/lat_bs1.py 2>&1 | grep -v amdgpu)"
  echo "split-K 3:     $(LWDETR_GEMM_SPLITK=3 python tools/lat_bs1.py 2>&1 | grep -v amdgpu)"
done ) | tee $O/lat_bs1_splitk.txt
python tools/op_times.py --size small --batch 1 2>&1 | grep -v amdgpu | cut -c1-100 > $O/op_times_small_b1_splitk.txt; grep -c . $O/op_times_small_b1_splitk.txt; grep 'Gemm' $O/op_times_small_b1_splitk.txt | head -14; tail -1 $O/op_times_small_b1_splitk.txt
timeout 900 python -m pytest tests/test_gpu_model.py -x -q -m gpu -k "fp32_matches or teacher_forced or hip_graph" 2>&1 | tail -3 | tee $O/pytest_model.txt
for cfg in "--size xlarge --batch 16 --res 960" "--size large --batch 32" ""; do
  echo "bench $cfg"; timeout 300 python bench.py $cfg --no-cpu-baseline --no-other-configs --no-latency --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['ms_per_step_passes']['after'])"
done 2>&1 | tee $O/bench.txt
