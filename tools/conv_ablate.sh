#!/bin/bash
# Builds tools/_timing/liblwdetr_conv_abl<bits>.so: the product library with gemm.hip compiled with -DLWDETR_CONV_ABL=<bits> (tuning only);
# `timing` builds liblwdetr_conv_timing.so (-DLWDETR_CONV_TIMING=37: phase stamps of workgroup 37, read by CONV_TIMING=1 tools/conv_time.py).
#     tools/conv_ablate.sh 1 4 16 timing;  LWDETR_HIP_LIB=tools/_timing/liblwdetr_conv_abl4.so python tools/conv_time.py 16 32
set -eu
cd "$(dirname "$0")/../lw-detr_amd/csrc"
mkdir -p ../../tools/_timing
B=build
for bits in "$@"; do
  if [ "$bits" = timing ]; then
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -Wall -Wno-unused-function -DLWDETR_CONV_TIMING=37 -c gemm.hip -o /tmp/gemm_timing.o
    hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/_timing/liblwdetr_conv_timing.so /tmp/gemm_timing.o $B/msda.o $B/attention.o $B/rowops.o $B/topk.o $B/preproc.o $B/mlp.o $B/vitblock.o $B/prof.o
    continue
  fi
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -Wall -Wno-unused-function -DLWDETR_CONV_ABL=$bits -c gemm.hip -o /tmp/gemm_abl$bits.o
  hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/_timing/liblwdetr_conv_abl$bits.so /tmp/gemm_abl$bits.o $B/msda.o $B/attention.o $B/rowops.o $B/topk.o $B/preproc.o $B/mlp.o $B/vitblock.o $B/prof.o
done
