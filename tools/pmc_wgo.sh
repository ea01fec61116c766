#!/bin/bash
# PMC passes over the fused-Winograd A/B tool (never combined with other trace domains): tools/pmc_wgo.sh [wgo_bench args]
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
OUT=gpurun_out/pmc_wgo
rm -rf $OUT; mkdir -p $OUT
ARGS="--rounds 1 $*"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_VALU --kernel-trace --output-format csv -d $OUT/pmc1 -- python3 tools/wgo_bench.py $ARGS > $OUT/pmc1.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc2 -- python3 tools/wgo_bench.py $ARGS > $OUT/pmc2.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -- python3 tools/wgo_bench.py $ARGS > $OUT/fetch.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr --kernel-trace --output-format csv -d $OUT/tcc -- python3 tools/wgo_bench.py $ARGS > $OUT/tcc.log 2>&1
python3 tools/summarize_pmc.py $OUT > $OUT/summary.txt 2>&1
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -delete; find $OUT -name "*agent_info.csv" -delete
grep -A12 "wino_gemm_out\|gemm_f32_kernel" $OUT/summary.txt | head -150
