#!/bin/bash
# round 5, GPU session 4: the whole -m gpu suite on the tree with the half-tile block kernel default, the widened 16-bit calibration, tiny B=32 parity,
# the replicated-golden test, two ranks on one GPU; then the default bench line
set -u
O=gpurun_out/r5s04; mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -25 | tee $O/pytest_all.txt
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 1500 $O/bench_default.json
