#!/bin/bash
# GPU session 41: PMC profile of the rebuilt large-tile GEMM at the xlarge fc2 / fc1 shapes (separate --pmc passes, kernel-trace only)
set -u
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/s41
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for tag in fc2 fc1; do
  if [ $tag = fc2 ]; then shape="58368 768 3072"; else shape="58368 3072 768"; fi
  i=0
  for set in "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CU_CYCLES" "GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F16"; do
    i=$((i+1))
    timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/pmc_${tag}_$i -o p -- python $ROOT/tools/one_gemm.py $shape 64 > $OUT/pmc_${tag}_$i.log 2>&1
  done
  python $ROOT/tools/pmc_table.py $OUT/pmc_${tag}_[1-5] > $OUT/gemm_big_${tag}_xlarge_pmc.json
  rm -rf $OUT/pmc_${tag}_[1-5]
done
cd $ROOT
python - <<'PY'
import json
for tag in ("fc2", "fc1"):
    d = json.load(open(f"gpurun_out/s41/gemm_big_{tag}_xlarge_pmc.json"))
    for k, v in d.items():
        if "gemm_big" in k:
            busy = v["SQ_VALU_MFMA_BUSY_CYCLES"] / (v["GRBM_GUI_ACTIVE"] * 128)
            print(tag, k[:60], "launches", v["launches"], "mfma_busy", round(busy, 3), "L2 hit", round(v["TCC_HIT_sum"] / (v["TCC_HIT_sum"] + v["TCC_MISS_sum"]), 3),
                  "HBM MB", round((2 * v["FETCH_SIZE"] * 32 + v["WRITE_SIZE"] * 32) / 1e6 if False else 0, 1), "FETCH", v["FETCH_SIZE"], "WRITE", v["WRITE_SIZE"], "LDS conflict / active", round(v["SQ_LDS_BANK_CONFLICT"] / max(v["SQ_LDS_IDX_ACTIVE"], 1), 4))
PY
