#!/bin/bash
# round 6, GPU session 19: the round-5 TREE (git worktree of 772dd61 with its own library) beside the round-6 tree on one box - the four BASELINE configurations + tiny at their default
# plans, alternating, three times; the single-image latency of both trees (small, tiny, large); box info
set -u
O=$(pwd)/gpurun_out/r6s19; mkdir -p $O
R6=$(pwd); R5=$(pwd)/tools/_timing/r5tree
bash tools/box_info.sh 2>&1 | head -14 | tee $O/box_info.txt
run() { python bench.py "$@" --no-cpu-baseline --no-other-configs --no-latency --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('ms_per_step_passes',{}).get('after'))"; }
for rep in 1 2 3; do
  echo "r5 small: $(cd $R5 && run)"; echo "r6 small: $(cd $R6 && run)"
  echo "r5 medium: $(cd $R5 && run --size medium --batch 64 --dtype bf16)"; echo "r6 medium: $(cd $R6 && run --size medium --batch 64 --dtype bf16)"
  echo "r5 large: $(cd $R5 && run --size large --batch 32)"; echo "r6 large: $(cd $R6 && run --size large --batch 32)"
  echo "r5 xlarge: $(cd $R5 && run --size xlarge --batch 16 --res 960)"; echo "r6 xlarge: $(cd $R6 && run --size xlarge --batch 16 --res 960)"
  echo "r5 tiny: $(cd $R5 && run --size tiny)"; echo "r6 tiny: $(cd $R6 && run --size tiny)"
done | tee $O/bench_r5_vs_r6.txt
for rep in 1 2; do for sz in small tiny large; do
  echo "r5 $sz bs1: $(cd $R5 && python tools/lat_bs1.py --size $sz 2>/dev/null | tail -1)"; echo "r6 $sz bs1: $(cd $R6 && python tools/lat_bs1.py --size $sz 2>/dev/null | tail -1)"
done; done | tee $O/lat_r5_vs_r6.txt
