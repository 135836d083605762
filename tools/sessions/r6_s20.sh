#!/bin/bash
# round 6, GPU session 20: (1) config 2 and tiny, round-5 tree vs round-6 tree again after the telemetry moved out of the way of the timed region; (2) rocprofv3 kernel stats + PMC passes of
# the final tree: config 2 (the default bench command) and xlarge 960x960 B = 16
set -u
O=$(pwd)/gpurun_out/r6s20; mkdir -p $O
R6=$(pwd); R5=$(pwd)/tools/_timing/r5tree
run() { python bench.py "$@" --no-cpu-baseline --no-other-configs --no-latency --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('ms_per_step_passes',{}).get('after'))"; }
for rep in 1 2 3; do
  echo "r5 small: $(cd $R5 && run)"; echo "r6 small: $(cd $R6 && run)"
  echo "r5 tiny: $(cd $R5 && run --size tiny)"; echo "r6 tiny: $(cd $R6 && run --size tiny)"
done | tee $O/bench_r5_vs_r6_small_tiny.txt
bash tools/profile_round.sh r6_small_b32_640_fp16 2>&1 | tail -3
bash tools/profile_round.sh r6_xlarge_b16_960_fp16 --size xlarge --batch 16 --res 960 2>&1 | tail -3
