#!/bin/bash
# round 6, GPU session 42: same-box A/B - mlp.o rebuilt with -fno-slp-vectorize (ns_mlp) against the library as it is (head): throughput of small / medium and the single-image latency
set -u
O=$(pwd)/gpurun_out/r6s42; mkdir -p $O
run() { python bench.py "$@" --no-cpu-baseline --no-other-configs --no-latency --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('ms_per_step_passes',{}).get('after'))"; }
for rep in 1 2 3; do for lib in head ns_mlp; do
  export LWDETR_HIP_LIB=tools/_timing/liblwdetr_$lib.so
  echo "$lib small: $(run)"; echo "$lib medium: $(run --size medium)"
  echo "$lib bs1 small: $(python tools/lat_bs1.py --size small 2>/dev/null | tail -1)"; echo "$lib bs1 large: $(python tools/lat_bs1.py --size large 2>/dev/null | tail -1)"
done; done | tee $O/ab.txt
