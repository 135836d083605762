#!/bin/bash
# round 3, GPU session 44: per-launch times of the small forward (batch 16 = one launch chain's part, and 32)
set -u
python tools/op_times.py --size small --batch 16 2>&1 | grep -v amdgpu > gpurun_out/r3_s44_b16.log
python tools/op_times.py --size small --batch 32 2>&1 | grep -v amdgpu > gpurun_out/r3_s44_b32.log
