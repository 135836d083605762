#!/bin/bash
# What kind of box is this? (partition modes, CU count, clocks under load, raw MFMA rate, the block kernel's time) - printed by GPU sessions so
# that run-to-run differences between gpurun boxes can be told from code changes
rocm-smi --showcomputepartition --showmemorypartition 2>/dev/null | grep -i "partition" | head -4
rocm-smi --showpower --showmaxpower 2>/dev/null | grep -i "power" | head -4
python - <<'PY' 2>&1 | grep -v amdgpu
import torch
p = torch.cuda.get_device_properties(0)
print("device:", p.name, "CUs", p.multi_processor_count, "mem GB", round(p.total_memory / 2**30), "clock MHz", getattr(p, "clock_rate", 0) // 1000)
PY
for u in mfma_peak whole_cu; do [ -x tools/ubench/$u ] || hipcc --offload-arch=gfx950 -O3 tools/ubench/$u.hip -o tools/ubench/$u 2>/dev/null; done
./tools/ubench/mfma_peak 2>/dev/null | head -6
./tools/ubench/whole_cu 2>/dev/null
python tools/vitblock_bench.py 2>&1 | grep -v amdgpu | head -8
