#!/bin/bash
# round 5, GPU session 20 (closing record): smoke() on the final library, then the round-4 tree beside the final round-5 tree on one box - the four BASELINE
# configurations at their default plans, alternating, three times; the single-image latency of both trees
set -u
O=$(pwd)/gpurun_out/r5s20; mkdir -p $O
R5=$(pwd); R4=$(pwd)/tools/_timing/r4tree
timeout 300 python -c 'import __graft_entry__ as g; g.smoke(); print("smoke ok")' 2>&1 | tail -2 | tee $O/smoke.txt
run() { python bench.py "$@" --no-cpu-baseline --no-other-configs --no-latency --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['config']['launch_chains'])"; }
lat() { python bench.py --no-cpu-baseline --no-other-configs --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['latency_bs1_hipgraph_ms'], d['latency_bs1_ms'])"; }
for rep in 1 2 3; do
  echo "r4 small: $(cd $R4 && run)"; echo "r5 small: $(cd $R5 && run)"
  echo "r4 medium: $(cd $R4 && run --size medium --batch 64 --dtype bf16)"; echo "r5 medium: $(cd $R5 && run --size medium --batch 64 --dtype bf16)"
  echo "r4 large: $(cd $R4 && run --size large --batch 32)"; echo "r5 large: $(cd $R5 && run --size large --batch 32)"
  echo "r4 xlarge: $(cd $R4 && run --size xlarge --batch 16 --res 960)"; echo "r5 xlarge: $(cd $R5 && run --size xlarge --batch 16 --res 960)"
done | tee $O/bench_r4_vs_r5_final.txt
for rep in 1 2; do echo "r4 bs1: $(cd $R4 && lat)"; echo "r5 bs1: $(cd $R5 && lat)"; done | tee $O/lat_r4_vs_r5_final.txt
