#!/bin/bash
# round 3, GPU session 29: the sampling kernel compiled without packed-f32 / without SDWA, beside the other chain's attention / GEMM kernels
set -u
for v in nopk nosdwa both; do
  echo "== msda.hip build: $v"
  LWDETR_HIP_LIB=$PWD/tools/_timing/liblwdetr_msda_$v.so PROBE_VARS=0,4 PROBE_MSDA=variants timeout 300 python tools/determinism_probe.py small 32 24 -2 2>&1 | grep -v amdgpu | cut -c1-300
done
