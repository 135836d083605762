#!/bin/bash
# round 6, GPU session 40: same-box A/B - one object at a time rebuilt with -fno-slp-vectorize (vitblock.o, chain.o, gemm_pt.o, gemm.o) against the library as it is (head)
set -u
O=$(pwd)/gpurun_out/r6s40; mkdir -p $O
run() { python bench.py "$@" --no-cpu-baseline --no-other-configs --no-latency --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('ms_per_step_passes',{}).get('after'))"; }
for rep in 1 2; do for lib in head ns_vitblock ns_chain ns_gemm_pt ns_gemm; do
  export LWDETR_HIP_LIB=tools/_timing/liblwdetr_$lib.so
  echo "$lib small: $(run)"; echo "$lib medium: $(run --size medium)"; echo "$lib large: $(run --size large)"; echo "$lib xlarge: $(run --size xlarge --batch 16 --res 960)"
done; done | tee $O/ab.txt
