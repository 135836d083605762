#!/bin/bash
# GPU session 33: round-2 final evidence: bench lines, single-image latency, rocprofv3 stats + PMC summaries per configuration
set -u
OUT=gpurun_out/s33
mkdir -p $OUT
timeout 600 python bench.py --latency > $OUT/bench_small_default.json 2> $OUT/bench_small_default.err
tail -c 1500 $OUT/bench_small_default.json
bash tools/profile_round.sh r2f_small_b32_640_fp16 > $OUT/prof_small.log 2>&1
bash tools/profile_round.sh r2f_xlarge_b16_960_fp16 --size xlarge --batch 16 --res 960 --dtype fp16 > $OUT/prof_xlarge.log 2>&1
bash tools/profile_round.sh r2f_large_b32_640_fp16 --size large --batch 32 --dtype fp16 > $OUT/prof_large.log 2>&1
bash tools/profile_round.sh r2f_medium_b64_640_bf16 --size medium --batch 64 --dtype bf16 > $OUT/prof_medium.log 2>&1
for cfg in "xlarge 16 fp16 960" "large 32 fp16 640" "medium 64 bf16 640" "tiny 32 fp16 640"; do
  set -- $cfg
  timeout 400 python bench.py --size $1 --batch $2 --dtype $3 --res $4 --no-cpu-baseline > $OUT/bench_$1.json 2> $OUT/bench_$1.err
  tail -c 400 $OUT/bench_$1.json | head -c 400; echo
done
ls gpurun_out/keep_r2f_*
