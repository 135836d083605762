#!/bin/bash
# round 6, GPU session 29: block kernel with the explicit LDS-read drain in front of every ring barrier - the M = 64000 parity cases six times, all block-kernel tests, stamps, bench
set -u
O=$(pwd)/gpurun_out/r6s29; mkdir -p $O
for i in 1 2 3 4 5 6; do echo "run $i: $(python -m pytest tests/test_gpu_kernels.py -q -m gpu -k 'test_vit_block and 64000' 2>&1 | grep -E 'passed|failed' | tail -1)"; done | tee $O/flaky.txt
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -q -x -m gpu 2>&1 | tail -3 | tee $O/pytest.txt
for b in 16 32; do
  echo "== batch $b"; LWDETR_HIP_LIB=tools/_timing/liblwdetr_hip_vbt.so python tools/vitblock_timing.py 192 $b fp16 2>&1 | grep -v amdgpu.ids | grep -v "wave [123]"
done | tee $O/vitblock_phases.txt
run() { python bench.py "$@" --no-cpu-baseline --no-other-configs --no-latency --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('ms_per_step_passes',{}).get('after'))"; }
for rep in 1 2 3; do echo "small: $(run)"; echo "tiny: $(run --size tiny)"; echo "medium: $(run --size medium)"; echo "large: $(run --size large)"; done | tee $O/bench.txt
