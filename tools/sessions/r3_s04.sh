#!/bin/bash
# round 3, GPU session 4: after the bit_cast fix - kernel tests, timing, and PMC passes on the block kernel alone (old vs new):
# effective clock (GRBM_GUI_ACTIVE / duration), wave-cycle split, MFMA busy, LDS stalls, instruction counts
set -u
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/r3_s04
mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "vit_block" > $OUT/t_vb.log 2>&1; tail -4 $OUT/t_vb.log
timeout 200 python tools/vitblock_bench.py 192 32 fp16 2>&1 | grep -v amdgpu.ids | tee -a $OUT/vb_bench.txt
( export LWDETR_HIP_LIB=$ROOT/tools/_timing/liblwdetr_hip_vbt.so; timeout 200 python tools/vitblock_timing.py 192 32 fp16 2>&1 | grep -v amdgpu.ids | grep -v "wave [123]" | tee -a $OUT/vb_timing.txt )
export TMPDIR=/tmp
cd /tmp
i=0
for set in "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_WAVES SQ_BUSY_CU_CYCLES"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/pmc_$i -o p -- python $ROOT/tools/vitblock_bench.py 192 32 fp16 5 > $OUT/pmc_$i.log 2>&1
done
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o p -- python $ROOT/tools/vitblock_bench.py 192 32 fp16 5 > $OUT/trace.log 2>&1
cd $ROOT
python tools/pmc_table.py $OUT/pmc_[1-3] > $OUT/vb_pmc.json
find $OUT/trace -name "*kernel_stats.csv" -exec cp {} $OUT/vb_kernel_stats.csv \;
rm -rf $OUT/pmc_[1-3] $OUT/trace
python - <<'PY'
import json
d = json.load(open("gpurun_out/r3_s04/vb_pmc.json"))
for k, v in d.items():
    if "vitblock" in k or "mlp_kernel" in k:
        print(k, json.dumps({a: round(b) for a, b in v.items()}))
PY
head -5 $OUT/vb_kernel_stats.csv
