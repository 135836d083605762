#!/bin/bash
# round 5, final GPU session (profiles): tools/profile_round.sh for the four BASELINE configurations
set -u
bash tools/profile_round.sh r5_small_b32_640_fp16 > gpurun_out/r5_profile_small.log 2>&1
bash tools/profile_round.sh r5_medium_b64_640_bf16 --size medium --batch 64 --dtype bf16 > gpurun_out/r5_profile_medium.log 2>&1
bash tools/profile_round.sh r5_large_b32_640_fp16 --size large --batch 32 > gpurun_out/r5_profile_large.log 2>&1
bash tools/profile_round.sh r5_xlarge_b16_960_fp16 --size xlarge --batch 16 --res 960 > gpurun_out/r5_profile_xlarge.log 2>&1
ls gpurun_out/keep_r5_*
