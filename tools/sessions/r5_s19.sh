#!/bin/bash
# round 5, GPU session 19: xlarge 960x960 B = 16 on the repaired tree - LayerNorm fold x launch chains, alternating, against the round-4 tree
set -u
O=$(pwd)/gpurun_out/r5s19; mkdir -p $O
R5=$(pwd); R4=$(pwd)/tools/_timing/r4tree
run() { python bench.py "$@" --no-cpu-baseline --no-other-configs --no-latency --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['config']['launch_chains'])"; }
for rep in 1 2 3; do
  echo "r4 tree:           $(cd $R4 && run --size xlarge --batch 16 --res 960)"
  echo "r5 fold=0 chains=1: $(cd $R5 && LWDETR_LN_FOLD=0 LWDETR_STREAMS=1 run --size xlarge --batch 16 --res 960)"
  echo "r5 fold=1 chains=1: $(cd $R5 && LWDETR_LN_FOLD=1 LWDETR_STREAMS=1 run --size xlarge --batch 16 --res 960)"
  echo "r5 fold=1 chains=2: $(cd $R5 && LWDETR_LN_FOLD=1 LWDETR_STREAMS=2 run --size xlarge --batch 16 --res 960)"
done | tee $O/xlarge_fold_chains.txt
LWDETR_LN_FOLD=1 LWDETR_STREAMS=1 python tools/op_times.py --size xlarge --batch 16 --res 960 2>&1 | grep -v amdgpu | cut -c1-100 | head -9 | tee $O/op_times_fold1.txt
LWDETR_LN_FOLD=0 LWDETR_STREAMS=1 python tools/op_times.py --size xlarge --batch 16 --res 960 2>&1 | grep -v amdgpu | cut -c1-100 | head -9 | tee $O/op_times_fold0.txt
