#!/bin/bash
# round 6, GPU session 12: after the knob table (no getenv on launch paths) and the pruned default build - the whole GPU suite, the same suite's experiment tests against
# the -DLWDETR_EXPERIMENTS build, the default-plan table, default vs experiments build at model level (must be equal)
set -u
O=$(pwd)/gpurun_out/r6s12; mkdir -p $O
python tests/test_gpu_default_plan.py > $O/default_plan_stdout.json 2> $O/default_plan_err.txt; tail -3 $O/default_plan_err.txt
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | grep -v '^    ' | tail -8 | cut -c1-300 | tee $O/pytest_gpu_default_build.txt
LWDETR_HIP_LIB=$(pwd)/tools/_timing/liblwdetr_exp.so timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -x -q -m gpu -k "large_tile or layernorm_folded or split_k or one_wave_per_window" 2>&1 | grep -v '^    ' | tail -5 | cut -c1-300 | tee $O/pytest_gpu_experiments_build.txt
run() { python bench.py "$@" --no-cpu-baseline --no-other-configs --no-latency --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('ms_per_step_passes',{}).get('after'))"; }
for rep in 1 2; do
  echo "small default build: $(run)"; echo "small experiments build: $(LWDETR_HIP_LIB=$(pwd)/tools/_timing/liblwdetr_exp.so run)"
  echo "xlarge default build: $(run --size xlarge --batch 16 --res 960)"; echo "xlarge experiments build: $(LWDETR_HIP_LIB=$(pwd)/tools/_timing/liblwdetr_exp.so run --size xlarge --batch 16 --res 960)"
done | tee $O/bench_builds.txt
