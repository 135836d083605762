#!/bin/bash
# round 6, GPU session 9: the persistent GEMM at model level - xlarge 960x960 B = 16 and large B = 32 with LWDETR_GEMM_PT = 0 | 1 | 2 (2 = also the residual launches), alternating;
# kernel tests of gemm_pt, the BASELINE parity test of xlarge with the new default
set -u
O=$(pwd)/gpurun_out/r6s09; mkdir -p $O
run() { python bench.py "$@" --no-cpu-baseline --no-other-configs --no-latency --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('ms_per_step_passes',{}).get('after'))"; }
for rep in 1 2 3; do for m in 0 1 2; do echo "xlarge pt=$m: $(LWDETR_GEMM_PT=$m run --size xlarge --batch 16 --res 960)"; done; done | tee $O/bench_xlarge_pt.txt
for rep in 1 2; do for m in 0 1; do echo "large pt=$m: $(LWDETR_GEMM_PT=$m run --size large --batch 32)"; done; done | tee $O/bench_large_pt.txt
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "persistent_tile or large_tile" 2>&1 | grep -v '^    ' | tail -5 | cut -c1-300 | tee $O/pytest_pt.txt
timeout 900 python -m pytest tests/test_gpu_baseline_configs.py -x -q -m gpu -k "xlarge" 2>&1 | grep -v '^    ' | tail -8 | cut -c1-300 | tee $O/pytest_xlarge_parity.txt
