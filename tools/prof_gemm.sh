#!/bin/bash
# rocprofv3 passes over tools/gemm_bench.py: kernel trace + stats, then PMC passes (never combined with
# other trace domains).  Output under gpurun_out/prof_gemm/.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
OUT=gpurun_out/prof_gemm
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 tools/gemm_bench.py --frames ${1:-128} > $OUT/trace.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_VALU --kernel-trace --output-format csv -d $OUT/pmc1 -- python3 tools/gemm_bench.py --frames ${1:-128} > $OUT/pmc1.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAVES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d $OUT/pmc2 -- python3 tools/gemm_bench.py --frames ${1:-128} > $OUT/pmc2.log 2>&1
rocprofv3 --pmc FETCH_SIZE TCC_HIT_sum --kernel-trace --output-format csv -d $OUT/pmc3 -- python3 tools/gemm_bench.py --frames ${1:-128} > $OUT/pmc3.log 2>&1
rocprofv3 --pmc WRITE_SIZE TCC_MISS_sum --kernel-trace --output-format csv -d $OUT/pmc4 -- python3 tools/gemm_bench.py --frames ${1:-128} > $OUT/pmc4.log 2>&1
find $OUT -name "*.csv" | head -30
python3 tools/summarize_pmc.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
