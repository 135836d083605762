#!/bin/bash
# round 6, GPU session 46: final tree - __graft_entry__.smoke(), kernel statistics + counters of xlarge 960x960 B = 16
set -u
O=$(pwd)/gpurun_out/r6s46; mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee $O/smoke.txt
bash tools/profile_round.sh r6_xlarge_b16_960_fp16 --size xlarge --batch 16 --res 960 2>&1 | tail -2
