#!/bin/bash
# round 6, GPU session 22: block kernel with the QKV phase's stores issued one boundary late - parity tests, phase stamps (and with the stores ablated), bench
set -u
O=$(pwd)/gpurun_out/r6s22; mkdir -p $O
python -m pytest tests/test_gpu_kernels.py -q -x -m gpu -k "vit_block or vit_qkv or vit_stem" 2>&1 | tail -3 | tee $O/pytest.txt
for lib in liblwdetr_hip_vbt.so liblwdetr_hip_vbt_a16.so; do for b in 16 32; do
  echo "== $lib batch $b"; LWDETR_HIP_LIB=tools/_timing/$lib python tools/vitblock_timing.py 192 $b fp16 2>&1 | grep -v amdgpu.ids | grep -v "wave [123]"
done; done | tee $O/vitblock_phases.txt
run() { python bench.py "$@" --no-cpu-baseline --no-other-configs --no-latency --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('ms_per_step_passes',{}).get('after'))"; }
for rep in 1 2 3; do echo "small: $(run)"; echo "tiny: $(run --size tiny)"; echo "medium: $(run --size medium)"; done | tee $O/bench.txt
