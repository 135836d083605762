#!/bin/bash
# round 5, final GPU session: the whole -m gpu suite, the default bench line (twice), rocprofv3 kernel stats + PMC summaries of every BASELINE configuration
set -u
O=gpurun_out/r5final; mkdir -p $O
bash tools/box_info.sh > $O/r5_final_box_info.txt 2>&1
timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -12 | tee $O/pytest_all.txt
timeout 600 python bench.py > $O/r5_bench_small_b32_640_fp16_default_with_cpu_baseline_and_other_configs.json 2> $O/bench_default.err; tail -c 600 $O/r5_bench_small_b32_640_fp16_default_with_cpu_baseline_and_other_configs.json
timeout 300 python bench.py --no-cpu-baseline --no-other-configs > $O/r5_bench_small_b32_640_fp16_run2.json 2>/dev/null
