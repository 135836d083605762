#!/bin/bash
# round 6, GPU session 34: row-chain kernels with the bounded LDS drain in front of the exact-form ring barrier - parity, config 2 / large bench
set -u
O=$(pwd)/gpurun_out/r6s34; mkdir -p $O
python -m pytest tests/test_gpu_chain.py tests/test_gpu_model.py -q -x -m gpu 2>&1 | tail -3 | tee $O/pytest.txt
run() { python bench.py "$@" --no-cpu-baseline --no-other-configs --no-latency --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('ms_per_step_passes',{}).get('after'))"; }
for rep in 1 2 3; do echo "small: $(run)"; echo "large: $(run --size large)"; done | tee $O/bench.txt
python tools/op_times.py --size small --batch 16 2>/dev/null | grep -i "enc_chain\|chain" | head -5 | tee $O/op_times_chain.txt
