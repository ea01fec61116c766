#!/bin/bash
# PMC passes over one encoder timing run (never combined with other trace domains): tools/pmc_enc.sh f32|bf16 KERNEL_SUBSTR
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
DT=${1:-f32}; PAT=${2:-stem_pool}
OUT=gpurun_out/pmc_enc_$DT
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_VALU --kernel-trace --output-format csv -d $OUT/pmc1 -- python3 tools/enc_kernel_times.py --dtype $DT --passes 1 > $OUT/pmc1.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc2 -- python3 tools/enc_kernel_times.py --dtype $DT --passes 1 > $OUT/pmc2.log 2>&1
python3 tools/summarize_pmc.py $OUT > $OUT/summary.txt 2>&1
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -delete; find $OUT -name "*agent_info.csv" -delete
grep -A12 "$PAT" $OUT/summary.txt | head -60
