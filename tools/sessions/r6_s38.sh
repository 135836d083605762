#!/bin/bash
# round 6, GPU session 38: same-box A/B - attention.o as shipped (head: hipcc SLP-packs the softmax's f32 adds / multiplies into v_pk_*_f32) against -fno-slp-vectorize (noslp)
set -u
O=$(pwd)/gpurun_out/r6s38; mkdir -p $O
run() { python bench.py "$@" --no-cpu-baseline --no-other-configs --no-latency --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('ms_per_step_passes',{}).get('after'))"; }
for lib in head noslp; do LWDETR_HIP_LIB=tools/_timing/liblwdetr_$lib.so python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "attention or attn" 2>&1 | tail -1; done | tee $O/pytest.txt
for rep in 1 2 3; do for lib in head noslp; do
  export LWDETR_HIP_LIB=tools/_timing/liblwdetr_$lib.so
  echo "$lib small: $(run)"; echo "$lib medium: $(run --size medium)"; echo "$lib xlarge: $(run --size xlarge --batch 16 --res 960)"
done; done | tee $O/ab.txt
for lib in head noslp; do echo "== $lib"; LWDETR_HIP_LIB=tools/_timing/liblwdetr_$lib.so python tools/op_times.py --size small --batch 16 2>/dev/null | grep "Attn" | head -4; done | tee $O/op_times_attn.txt
