#!/bin/bash
# Round profile on the GPU box (run through gpurun from the repo root):  bash tools/profile_round.sh <tag>
# 1. rocprofv3 --kernel-trace --stats of the default bench command  -> gpurun_out/prof_<tag>/..._kernel_stats.csv
# 2. two separate --pmc passes (FETCH_SIZE, WRITE_SIZE; counters are never combined with other trace domains)
# 3. tools/pmc_summary.py reduces them to HBM bytes per launch per kernel -> gpurun_out/<tag>_hbm_traffic.json
set -u
TAG=${1:-rX}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_$TAG" -o small_b32_fp16 -- python "$ROOT/bench.py" --no-cpu-baseline > "$OUT/prof_${TAG}_bench.json" 2> "$OUT/prof_${TAG}_bench.err"
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$OUT/pmc_${TAG}_$c" -o b -- python "$ROOT/bench.py" --steps 3 --warmup 2 --no-cpu-baseline --no-roofline > /dev/null 2> "$OUT/pmc_${TAG}_$c.err"
done
cd "$ROOT"
du -sh "$OUT"/prof_$TAG "$OUT"/pmc_${TAG}_* 2>/dev/null
python tools/pmc_summary.py "$OUT/pmc_${TAG}_FETCH_SIZE" "$OUT/pmc_${TAG}_WRITE_SIZE" > "$OUT/${TAG}_hbm_traffic.json"
# keep the summaries only (the raw traces exceed what gpurun copies back)
mkdir -p "$OUT/keep_$TAG"
find "$OUT/prof_$TAG" \( -name "*kernel_stats.csv" -o -name "*domain_stats.csv" \) -exec cp {} "$OUT/keep_$TAG/" \;
rm -rf "$OUT/prof_$TAG" "$OUT"/pmc_${TAG}_FETCH_SIZE "$OUT"/pmc_${TAG}_WRITE_SIZE
ls -la "$OUT/keep_$TAG"; tail -3 "$OUT/prof_${TAG}_bench.err"; tail -2 "$OUT"/pmc_${TAG}_FETCH_SIZE.err
