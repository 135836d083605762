#!/bin/bash
# Builds tools/_timing/liblwdetr_conv_abl<bits>.so: the product library with gemm.hip compiled with -DLWDETR_CONV_ABL=<bits> (tuning only)
set -eu
cd "$(dirname "$0")/../lw-detr_amd/csrc"
mkdir -p ../../tools/_timing
B=build
for bits in "$@"; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -Wall -Wno-unused-function -DLWDETR_CONV_ABL=$bits -c gemm.hip -o /tmp/gemm_abl$bits.o
  hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/_timing/liblwdetr_conv_abl$bits.so /tmp/gemm_abl$bits.o $B/msda.o $B/attention.o $B/rowops.o $B/topk.o $B/preproc.o $B/mlp.o $B/vitblock.o $B/prof.o
done
