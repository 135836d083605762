#!/bin/bash
# round 3, GPU session 38: ablation builds of the patch-resident convolution (tools/conv_ablate.sh): where do its 20 us go?
set -u
echo "== product"; python tools/conv_time.py 16 32 2>&1 | grep -v amdgpu
for b in 1 2 4 8 12 16 32; do echo "== ablation $b (1 no MFMA, 2 no patch DMA, 4 no epilogue, 8 no weight DMA, 16 no fragment reads, 32 no lgkmcnt wait)"; LWDETR_HIP_LIB=$PWD/tools/_timing/liblwdetr_conv_abl$b.so python tools/conv_time.py 16 32 2>&1 | grep -v amdgpu; done
