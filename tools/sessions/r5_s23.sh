#!/bin/bash
# round 5, GPU session 23: per-launch listings of the final tree (tools/op_times.py): one image (the latency path) for small and large, one 16-image chain of config 2
set -u
O=$(pwd)/gpurun_out/r5s23; mkdir -p $O
( echo "## small 640x640 fp16, batch 1 (the single-image path)"; timeout 60 python tools/op_times.py --batch 1 2>/dev/null
  echo; echo "## small 640x640 fp16, batch 16 (one launch chain of BASELINE config 2)"; timeout 60 python tools/op_times.py --batch 16 2>/dev/null
  echo; echo "## large 640x640 fp16, batch 1"; timeout 60 python tools/op_times.py --size large --batch 1 2>/dev/null ) | cut -c1-200 > $O/r5_op_times_small_b1_b16_large_b1.txt
tail -3 $O/r5_op_times_small_b1_b16_large_b1.txt
