#!/bin/bash
# round 5, GPU session 21: GELU on packed f16 pairs in the block kernel (LWDETR_VB_GELU16=1) - kernel tests, parity with the switch on
# (replicated goldens through the benchmarked plan, two-chain equality, BASELINE config 2 against the oracle), kernel and model A/B
set -u
O=$(pwd)/gpurun_out/r5s21; mkdir -p $O
timeout 240 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "test_vit_block" 2>&1 | grep -v '^    ' | tail -6 | cut -c1-300 | tee $O/pytest_vit_block.txt
export LWDETR_VB_GELU16=1
timeout 200 python -m pytest tests/test_gpu_model.py -x -q -m gpu -k "benchmarked_16bit_plan or two_launch_chains" 2>&1 | grep -v '^    ' | tail -6 | cut -c1-300 | tee $O/pytest_model_g16.txt
for f in gpurun_out/parity_replicated_float16_*.json; do cp $f $O/g16_$(basename $f); done
timeout 200 python -m pytest tests/test_gpu_baseline_configs.py -x -q -m gpu -k "small_b32" 2>&1 | grep -v '^    ' | tail -6 | cut -c1-300 | tee $O/pytest_config2_g16.txt
cp gpurun_out/parity_config_small_b32_fp16.json $O/g16_parity_config_small_b32_fp16.json 2>/dev/null
unset LWDETR_VB_GELU16
for rep in 1 2; do for g in 0 1; do
  echo "gelu16=$g C=192 B=32: $(LWDETR_VB_GELU16=$g ONLY=vit_block timeout 60 python tools/vitblock_bench.py 192 32 2>/dev/null | tail -1 | cut -c1-120)"
  echo "gelu16=$g C=192 B=16: $(LWDETR_VB_GELU16=$g ONLY=vit_block timeout 60 python tools/vitblock_bench.py 192 16 2>/dev/null | tail -1 | cut -c1-120)"
  echo "gelu16=$g C=384 B=32: $(LWDETR_VB_GELU16=$g ONLY=vit_block timeout 60 python tools/vitblock_bench.py 384 32 2>/dev/null | tail -1 | cut -c1-120)"
done; done | tee $O/vitblock_bench_g16.txt
run() { python bench.py "$@" --no-cpu-baseline --no-other-configs --no-latency --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['config']['launch_chains'])"; }
for rep in 1 2 3; do
  echo "small gelu16=0: $(LWDETR_VB_GELU16=0 run)"; echo "small gelu16=1: $(LWDETR_VB_GELU16=1 run)"
done | tee $O/bench_g16.txt
for rep in 1 2; do
  echo "large gelu16=0: $(LWDETR_VB_GELU16=0 run --size large --batch 32)"; echo "large gelu16=1: $(LWDETR_VB_GELU16=1 run --size large --batch 32)"
done | tee -a $O/bench_g16.txt
