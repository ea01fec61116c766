#!/bin/bash
# PMC passes over tools/gemm_bf16_bench.py (never combined with other trace domains).
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
OUT=gpurun_out/prof_bf16
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_VALU --kernel-trace --output-format csv -d $OUT/pmc1 -- python3 tools/gemm_bf16_bench.py --tiles 3,10 > $OUT/pmc1.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAVES --kernel-trace --output-format csv -d $OUT/pmc2 -- python3 tools/gemm_bf16_bench.py --tiles 3,10 > $OUT/pmc2.log 2>&1
rocprofv3 --pmc TA_BUSY_avr TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $OUT/pmc3 -- python3 tools/gemm_bf16_bench.py --tiles 3,10 > $OUT/pmc3.log 2>&1
python3 tools/summarize_pmc.py $OUT > $OUT/summary.txt 2>&1
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -delete; find $OUT -name "*agent_info.csv" -delete
grep -A12 "gemm_bf16_kernel<1, 1, 2" $OUT/summary.txt | head -80; tail -3 $OUT/pmc3.log
