#!/bin/bash
# round 5, GPU session 22: packed-f16 GELU on / off on a second box (f16 configurations that run the block kernel), alternating
set -u
O=$(pwd)/gpurun_out/r5s22; mkdir -p $O
run() { python bench.py "$@" --no-cpu-baseline --no-other-configs --no-latency --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; }
for rep in 1 2 3 4; do echo "small gelu16=0: $(LWDETR_VB_GELU16=0 run)"; echo "small gelu16=1: $(LWDETR_VB_GELU16=1 run)"; done | tee $O/bench_g16_box2.txt
for rep in 1 2 3; do echo "large gelu16=0: $(LWDETR_VB_GELU16=0 run --size large --batch 32)"; echo "large gelu16=1: $(LWDETR_VB_GELU16=1 run --size large --batch 32)"; done | tee -a $O/bench_g16_box2.txt
for rep in 1 2; do echo "tiny gelu16=0: $(LWDETR_VB_GELU16=0 run --size tiny --batch 32)"; echo "tiny gelu16=1: $(LWDETR_VB_GELU16=1 run --size tiny --batch 32)"; done | tee -a $O/bench_g16_box2.txt
