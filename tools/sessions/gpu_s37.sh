#!/bin/bash
# GPU session 37: single-image latency vs the 64x64 GEMM schedule knobs; kernel tests of the few-token block kernel
set -u
OUT=gpurun_out/s37
mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "mlp_fused" > $OUT/t_mlp.log 2>&1; tail -1 $OUT/t_mlp.log
run() {
  env "$@" timeout 200 python tools/op_times.py --size small --batch 1 2>/dev/null > $OUT/op_b1_$(echo "$@" | tr ' =' '__').txt
  f=$OUT/op_b1_$(echo "$@" | tr ' =' '__').txt
  echo "$@: $(tail -1 $f) | conv: $(grep 'N=128 K=1152' $f | awk '{s+=$3} END {printf "%.1f", s}') us | all Gemm: $(grep ' Gemm ' $f | awk '{s+=$3} END {printf "%.1f", s}') us"
}
run X=0
run LWDETR_GEMM_KB=64
run LWDETR_GEMM_NST=2
run LWDETR_GEMM_NST=4
run LWDETR_GEMM_TILE=1
