#!/bin/bash
# round 6, GPU session 41: vitblock.o without SLP vectorisation (tree) against the library before (head) - block-kernel parity incl. the M = 64000 cases six times, model tests, same-box A/B
set -u
O=$(pwd)/gpurun_out/r6s41; mkdir -p $O
for i in 1 2 3 4 5 6; do echo "run $i: $(python -m pytest tests/test_gpu_kernels.py -q -m gpu -k 'test_vit_block and 64000' 2>&1 | grep -E 'passed|failed' | tail -1)"; done | tee $O/flaky.txt
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -q -x -m gpu 2>&1 | tail -2 | tee $O/pytest.txt
run() { python bench.py "$@" --no-cpu-baseline --no-other-configs --no-latency --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('ms_per_step_passes',{}).get('after'))"; }
for rep in 1 2 3; do for lib in head tree; do
  if [ $lib = head ]; then export LWDETR_HIP_LIB=tools/_timing/liblwdetr_head.so; else unset LWDETR_HIP_LIB; fi
  echo "$lib small: $(run)"; echo "$lib tiny: $(run --size tiny)"; echo "$lib medium: $(run --size medium)"; echo "$lib large: $(run --size large)"
done; done | tee $O/ab.txt
