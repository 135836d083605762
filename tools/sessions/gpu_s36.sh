#!/bin/bash
# GPU session 36: few-token block kernel with 16-token workgroups + weight prefetch
set -u
OUT=gpurun_out/s36
mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "mlp_fused" > $OUT/t_mlp.log 2>&1; tail -2 $OUT/t_mlp.log
timeout 900 python -m pytest tests/test_gpu_model.py -x -q -m gpu > $OUT/t_model.log 2>&1; tail -2 $OUT/t_model.log
for tt in 1 2; do
  LWDETR_MLP_SMALL_TT=$tt timeout 300 python bench.py --latency --no-cpu-baseline --steps 5 --warmup 2 > $OUT/bench_lat_tt$tt.json 2> $OUT/bench_lat_tt$tt.err
  python - $tt <<'PY'
import json, sys
d=json.loads(open(f"gpurun_out/s36/bench_lat_tt{sys.argv[1]}.json").read().strip().splitlines()[-1])
print("TT", sys.argv[1], d["latency_bs1_ms"], d["latency_bs1_hipgraph_ms"])
PY
done
for b in 1 2 4 8; do
  for tt in 1 2; do
    LWDETR_MLP_SMALL_TT=$tt timeout 200 python tools/op_times.py --size small --batch $b 2>/dev/null | grep "MlpFusedOp" | awk -v b=$b -v tt=$tt '{s+=$3; n++} END {printf "batch %d TT %d: MlpFusedOp avg %.1f us (%d ops)\n", b, tt, s/n, n}'
  done
done
