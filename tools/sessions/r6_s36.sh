#!/bin/bash
# round 6, GPU session 36: launch-chain counts on the final tree (config 2 and tiny, B = 32: 1 / 2 / 4 chains; B = 64: 2 / 4 / 8)
set -u
O=$(pwd)/gpurun_out/r6s36; mkdir -p $O
run() { python bench.py "$@" --no-cpu-baseline --no-other-configs --no-latency --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('ms_per_step_passes',{}).get('after'), d['config'].get('launch_chains'))"; }
for rep in 1 2; do
  for n in 1 2 4; do echo "small B=32 chains=$n: $(LWDETR_STREAMS=$n run)"; done
  for n in 2 4; do echo "tiny B=32 chains=$n: $(LWDETR_STREAMS=$n run --size tiny)"; done
  for n in 2 4 8; do echo "small B=64 chains=$n: $(LWDETR_STREAMS=$n run --batch 64)"; done
done | tee $O/chains.txt
