#!/bin/bash
# round 6, GPU session 33: kernel statistics + counters of config 2 on the tree with the pipelined QKV phase / cheap ring boundaries (tools/profile_round.sh)
set -u
bash tools/profile_round.sh r6_small_b32_640_fp16 2>&1 | tail -3
