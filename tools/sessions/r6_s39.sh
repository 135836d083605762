#!/bin/bash
# round 6, GPU session 39: the hd 16 ring kernels of attention.hip in their own object without SLP vectorisation (tree) against the library before (head) - parity, same-box A/B
set -u
O=$(pwd)/gpurun_out/r6s39; mkdir -p $O
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -q -x -m gpu -k "attention or attn or golden or baseline or replicated" 2>&1 | tail -2 | tee $O/pytest.txt
run() { python bench.py "$@" --no-cpu-baseline --no-other-configs --no-latency --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('ms_per_step_passes',{}).get('after'))"; }
for rep in 1 2 3; do for lib in head tree; do
  if [ $lib = head ]; then export LWDETR_HIP_LIB=tools/_timing/liblwdetr_head.so; else unset LWDETR_HIP_LIB; fi
  echo "$lib small: $(run)"; echo "$lib tiny: $(run --size tiny)"; echo "$lib xlarge: $(run --size xlarge --batch 16 --res 960)"
done; done | tee $O/ab.txt
