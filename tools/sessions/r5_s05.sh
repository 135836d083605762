#!/bin/bash
# round 5, GPU session 5: window attention hd 16 - ablations of the one-wave kernel (what bounds it), the window-tile kernel (LDS-staged V^T + output tile)
set -u
O=gpurun_out/r5s05; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "attention" 2>&1 | tail -5 | tee $O/pytest_attention.txt
timeout 300 python tools/attn_bench.py small_b32_f16_win small_b16_f16_win --v=short,win,wtile 2>&1 | grep -v amdgpu | tee $O/attn_win_variants.txt
for abl in 1 2 4 8 7; do
  echo "## one-wave kernel, ablation $abl (1 = no stores, 2 = no V^T loads, 4 = no K / Q loads, 8 = no exp2)"
  LWDETR_HIP_LIB=tools/_timing/libaw_abl$abl.so timeout 120 python tools/attn_bench.py small_b32_f16_win small_b16_f16_win --v=win 2>&1 | grep -v amdgpu
done | tee $O/attn_win_ablations.txt
for wt in 0 2 0 2; do
  echo "small wtile=$wt"; LWDETR_ATTN_WTILE=$wt timeout 300 python bench.py --no-cpu-baseline --no-other-configs --no-latency --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
done 2>&1 | tee $O/bench_small.txt
timeout 900 python -m pytest tests/test_gpu_model.py -x -q -m gpu -k "benchmarked_16bit or two_launch or stem_launch or hip_graph or evaluate or export or detect" 2>&1 | tail -5 | tee $O/pytest_model_rest.txt
timeout 900 python -m pytest tests/test_gpu_msda.py tests/test_gpu_preprocess.py tests/test_gpu_topk.py -x -q -m gpu 2>&1 | tail -3 | tee $O/pytest_rest2.txt
