#!/bin/bash
# round 3, GPU session 50: whole GPU suite + smoke on the final tree
set -u
OUT=gpurun_out/r3_s50; mkdir -p $OUT
timeout 2400 python -m pytest tests -x -q -m gpu > $OUT/t_gpu.log 2>&1; echo "gpu suite: $(tail -1 $OUT/t_gpu.log)"
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
