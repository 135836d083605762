#!/bin/bash
# round 6, GPU session 37: same-box A/B - the committed block kernel (head) against the variant with the ring boundary in front of the last four MFMA slots (early)
set -u
O=$(pwd)/gpurun_out/r6s37; mkdir -p $O
run() { python bench.py "$@" --no-cpu-baseline --no-other-configs --no-latency --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('ms_per_step_passes',{}).get('after'))"; }
for rep in 1 2 3; do for lib in head early; do
  export LWDETR_HIP_LIB=tools/_timing/liblwdetr_$lib.so
  echo "$lib small: $(run)"; echo "$lib tiny: $(run --size tiny)"; echo "$lib medium: $(run --size medium)"; echo "$lib large: $(run --size large)"
done; done | tee $O/ab.txt
