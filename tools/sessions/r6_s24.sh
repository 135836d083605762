#!/bin/bash
# round 6, GPU session 24: block kernel with the QKV phase software-pipelined (stores / conversions of step s - 1 between the MFMAs of step s) - parity tests, phase stamps, bench
set -u
O=$(pwd)/gpurun_out/r6s24; mkdir -p $O
python -m pytest tests/test_gpu_kernels.py -q -x -m gpu -k "vit_block or vit_qkv or vit_stem" 2>&1 | tail -3 | tee $O/pytest.txt
for b in 16 32; do
  echo "== batch $b"; LWDETR_HIP_LIB=tools/_timing/liblwdetr_hip_vbt.so python tools/vitblock_timing.py 192 $b fp16 2>&1 | grep -v amdgpu.ids | grep -v "wave [123]"
done | tee $O/vitblock_phases.txt
echo "== C=384 batch 16"; LWDETR_HIP_LIB=tools/_timing/liblwdetr_hip_vbt.so python tools/vitblock_timing.py 384 16 fp16 2>&1 | grep -v amdgpu.ids | grep -v "wave [123]" | tee -a $O/vitblock_phases.txt
run() { python bench.py "$@" --no-cpu-baseline --no-other-configs --no-latency --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('ms_per_step_passes',{}).get('after'))"; }
for rep in 1 2 3; do echo "small: $(run)"; echo "tiny: $(run --size tiny)"; echo "medium: $(run --size medium)"; done | tee $O/bench.txt
