#!/bin/bash
# round 5, GPU session 16: the round-4 tree (git worktree of 3780fc1, built in tools/_timing/r4tree) against the round-5 tree on ONE box - did the epilogue
# restructure (fast path instantiated per feature) cost the large-tile GEMM anything?
set -u
O=$(pwd)/gpurun_out/r5s16; mkdir -p $O
R5=$(pwd); R4=$(pwd)/tools/_timing/r4tree
for rep in 1 2; do
  echo "## round-4 tree"; (cd $R4 && GEMM_BENCH_MODES=64 timeout 200 python tools/gemm_big_bench.py xlarge large 2>&1 | grep -v amdgpu | sed 's/ring64.128.*big kb64/big kb64/' | cut -c1-120)
  echo "## round-5 tree"; (cd $R5 && GEMM_BENCH_MODES=64 timeout 200 python tools/gemm_big_bench.py xlarge large 2>&1 | grep -v amdgpu | cut -c1-120)
done | tee $O/gemm_big_r4_vs_r5.txt
run() { python bench.py "$@" --no-cpu-baseline --no-other-configs --no-latency --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; }
for rep in 1 2; do
  for cfg in "--size xlarge --batch 16 --res 960" "--size large --batch 32" "--size medium --batch 64 --dtype bf16" ""; do
    echo "r4 $cfg: $(cd $R4 && run $cfg)"; echo "r5 $cfg: $(cd $R5 && run $cfg)"
  done
done | tee $O/bench_r4_vs_r5.txt
