#!/bin/bash
# round 5, GPU session 14: launch-chain counts on the round-5 tree (half-tile block kernel, LayerNorm fold), decoder FFN split cap at 16-image chains
set -u
O=gpurun_out/r5s14; mkdir -p $O
run() { python bench.py "$@" --no-cpu-baseline --no-other-configs --no-latency --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['config']['launch_chains'])"; }
( for s in 2 1 4 2 3; do echo "small chains=$s: $(LWDETR_STREAMS=$s run)"; done
  for s in 1 2 1 2; do echo "xlarge chains=$s: $(LWDETR_STREAMS=$s run --size xlarge --batch 16 --res 960)"; done
  for s in 2 4; do echo "medium chains=$s: $(LWDETR_STREAMS=$s run --size medium --batch 64 --dtype bf16)"; done
  for c in 16 8 4 32; do echo "small FFN_SPLITS cap=$c: $(LWDETR_FFN_SPLITS=$c run)"; done
) 2>&1 | tee $O/chains_and_ffn.txt
