#!/bin/bash
# round 5, GPU session 18: after restoring round 4's shared epilogue byte for byte (the LayerNorm fold in its own copy + its own kernel, producer statistics removed):
# large-tile GEMM per shape and model level against the round-4 tree, one box; LayerNorm fold on / off
set -u
O=$(pwd)/gpurun_out/r5s18; mkdir -p $O
R5=$(pwd); R4=$(pwd)/tools/_timing/r4tree
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "gemm or layernorm or conv" 2>&1 | grep -v '^    ' | tail -8 | cut -c1-300 | tee $O/pytest_gemm.txt
for rep in 1 2; do
  echo "## round-4 tree"; (cd $R4 && timeout 200 python tools/gemm_big_bench.py xlarge large 2>&1 | grep -v amdgpu | sed 's/ring64.128 *[0-9.]* us *[0-9.]* TF.s (rel diff [0-9.e+-]*)//; s/  big kb32.*//' | cut -c1-110)
  echo "## round-5 tree"; (cd $R5 && GEMM_BENCH_MODES=64 timeout 200 python tools/gemm_big_bench.py xlarge large 2>&1 | grep -v amdgpu | cut -c1-110)
done | tee $O/gemm_big_r4_vs_r5.txt
run() { python bench.py "$@" --no-cpu-baseline --no-other-configs --no-latency --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['config']['launch_chains'])"; }
for rep in 1 2; do
  echo "r4 xlarge: $(cd $R4 && run --size xlarge --batch 16 --res 960)"
  echo "r5 xlarge fold=1: $(cd $R5 && LWDETR_LN_FOLD=1 run --size xlarge --batch 16 --res 960)"
  echo "r5 xlarge fold=0: $(cd $R5 && LWDETR_LN_FOLD=0 run --size xlarge --batch 16 --res 960)"
  echo "r5 xlarge fold=0 chains=1: $(cd $R5 && LWDETR_LN_FOLD=0 LWDETR_STREAMS=1 run --size xlarge --batch 16 --res 960)"
  echo "r4 large: $(cd $R4 && run --size large --batch 32)"; echo "r5 large: $(cd $R5 && run --size large --batch 32)"
  echo "r4 medium: $(cd $R4 && run --size medium --batch 64 --dtype bf16)"; echo "r5 medium: $(cd $R5 && run --size medium --batch 64 --dtype bf16)"
  echo "r4 small: $(cd $R4 && run)"; echo "r5 small: $(cd $R5 && run)"
done | tee $O/bench_r4_vs_r5.txt
