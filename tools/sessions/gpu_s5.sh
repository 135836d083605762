#!/bin/bash
# GPU session 5: why does the large-tile GEMM stop at ~750 TFLOP/s?  L2 hit rate, HBM bytes, MFMA busy (separate PMC passes)
set -u
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/s5
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 -L > $OUT/counters_all.txt 2>&1 || true
grep -o "TCC_[A-Za-z0-9_]*" $OUT/counters_all.txt | sort -u | tr '\n' ' ' > $OUT/tcc_counters.txt
for shape in "58368 768 3072" "58368 3072 768"; do
  tag=$(echo $shape | tr ' ' 'x')
  i=0
  for set in "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/pmc_${tag}_$i -o p -- python $ROOT/tools/one_gemm.py $shape 64 > $OUT/pmc_${tag}_$i.log 2>&1
  done
  python $ROOT/tools/pmc_table.py $OUT/pmc_${tag}_1 $OUT/pmc_${tag}_2 $OUT/pmc_${tag}_3 $OUT/pmc_${tag}_4 $OUT/pmc_${tag}_5 > $OUT/pmc_$tag.json
  cat $OUT/pmc_$tag.json
  tail -3 $OUT/pmc_${tag}_1.log
  rm -rf $OUT/pmc_${tag}_[1-5]
done
cat $OUT/tcc_counters.txt | head -c 1500
