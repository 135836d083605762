#!/bin/bash
# round 3, GPU session 34: which form of the patch-resident convolution stays reproducible beside the other chain
set -u
run() { local label=$1; shift; echo "== $label"; env "$@" timeout 300 python tools/determinism_probe.py small 32 30 2 2>&1 | grep -v amdgpu | cut -c1-200 | tail -2; }
run "default" X=1
run "drain the ring every stage (VAR 1)" LWDETR_CONV_PATCH_VAR=1
run "160 KB of LDS per workgroup (VAR 2)" LWDETR_CONV_PATCH_VAR=2
run "ring 4 deep (VAR 3)" LWDETR_CONV_PATCH_VAR=3
run "gemm.o without SLP vectorisation" LWDETR_HIP_LIB=$PWD/tools/_timing/liblwdetr_gemm_nopk.so
