#!/bin/bash
# round 3, GPU session 47: decoder with the query_pos products precomputed (one GEMM for all layers), q / k / v in one launch per layer
set -u
OUT=gpurun_out/r3_s47; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_model.py tests/test_gpu_chains.py tests/test_gpu_graph.py -x -q -m gpu 2>&1 | tail -2
for q in 1 0; do
  LWDETR_DEC_QPOS=$q timeout 600 python bench.py --no-cpu-baseline > $OUT/bench_small_q$q.json 2> $OUT/bench.err
  python -c "
import json,sys;r=json.loads(open('$OUT/bench_small_q$q.json').read().strip().splitlines()[-1]);print('LWDETR_DEC_QPOS=$q', r['value'], r['ms_per_step'], {k: v for k, v in r.items() if 'latency' in k})"
done
