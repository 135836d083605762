#!/bin/bash
# round 3, GPU session 17: final tree - the whole -m gpu suite and smoke()
set -u
OUT=gpurun_out/r3_s17
mkdir -p $OUT
timeout 2400 python -m pytest tests -q -m gpu > $OUT/t_all.log 2>&1; tail -4 $OUT/t_all.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
