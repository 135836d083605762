#!/bin/bash
# Profile of one bench.py workload on the GPU box (run through gpurun from the repo root):
#     bash tools/profile_round.sh <tag> [bench.py workload flags, e.g. --size medium --batch 64 --dtype bf16]
# 1. rocprofv3 --kernel-trace --stats of the bench command            -> gpurun_out/keep_<tag>/<tag>_kernel_stats.csv
# 2. separate --pmc passes (counters are never combined with other trace domains): FETCH_SIZE | WRITE_SIZE |
#    MFMA busy + wave-cycle split | MFMA instruction counts + GRBM_GUI_ACTIVE | L2 hit / miss
# 3. tools/pmc_summary.py reduces them to gpurun_out/keep_<tag>/<tag>_pmc_summary.json (HBM bytes per launch, MFMA busy
#    fraction, wait split per kernel). Copy keep_<tag>/* to profiles/ to publish.
# All profiled runs use ONE launch chain (LWDETR_STREAMS=1): per-kernel durations and counters of a launch are only meaningful when
# no other kernel shares its CUs (the un-profiled bench line is the two-chain number).
set -u
export LWDETR_STREAMS=1
TAG=${1:-rX}; shift || true
ROOT=$(pwd)
OUT=$ROOT/gpurun_out
KEEP=$OUT/keep_$TAG
mkdir -p "$KEEP"
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_$TAG" -o p -- python "$ROOT/bench.py" --no-cpu-baseline --no-latency --no-other-configs "$@" > "$KEEP/${TAG}_bench_under_rocprofv3.json" 2> "$OUT/prof_${TAG}_bench.err"
find "$OUT/prof_$TAG" -name "*kernel_stats.csv" -exec cp {} "$KEEP/${TAG}_kernel_stats.csv" \;
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT" "SQ_INSTS_MFMA SQ_INSTS_VALU GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$OUT/pmc_${TAG}_$i" -o b -- python "$ROOT/bench.py" --steps 3 --warmup 2 --no-cpu-baseline --no-roofline --no-latency --no-other-configs "$@" > /dev/null 2> "$OUT/pmc_${TAG}_$i.err"
done
cd "$ROOT"
python tools/pmc_summary.py "$KEEP/${TAG}_kernel_stats.csv" "$OUT"/pmc_${TAG}_[1-5] > "$KEEP/${TAG}_pmc_summary.json"
rm -rf "$OUT/prof_$TAG" "$OUT"/pmc_${TAG}_[1-5]
ls -la "$KEEP"; tail -2 "$OUT/prof_${TAG}_bench.err"; tail -2 "$OUT"/pmc_${TAG}_1.err
