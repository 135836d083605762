#!/bin/bash
# round 6, GPU session 44: single-image latency - gemm.o / rowops.o / few.o rebuilt with -fno-slp-vectorize, one at a time, against the library as it is (head)
set -u
O=$(pwd)/gpurun_out/r6s44; mkdir -p $O
for rep in 1 2; do for lib in head ns_gemm ns_rowops ns_few; do
  export LWDETR_HIP_LIB=tools/_timing/liblwdetr_$lib.so
  for sz in small medium large; do echo "$lib bs1: $(python tools/lat_bs1.py --size $sz 2>/dev/null | tail -1)"; done
  echo "$lib bs1: $(python tools/lat_bs1.py --size xlarge --res 960 2>/dev/null | tail -1)"
done; done | tee $O/ab.txt
