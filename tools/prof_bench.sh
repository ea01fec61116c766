#!/bin/bash
# rocprofv3 kernel trace + stats of the default bench.py run, then HBM-traffic PMC passes
# (FETCH_SIZE / WRITE_SIZE in separate passes, never combined with other trace domains).
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
OUT=gpurun_out/prof_bench
rm -rf $OUT; mkdir -p $OUT
ARGS="--steps 2 --warmup 2 --no-cpu-baseline --no-peaks $*"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py $ARGS > $OUT/trace.json 2> $OUT/trace.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -- python3 bench.py $ARGS > $OUT/fetch.json 2> $OUT/fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -- python3 bench.py $ARGS > $OUT/write.json 2> $OUT/write.err
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/mfma -- python3 bench.py $ARGS > $OUT/mfma.json 2> $OUT/mfma.err
python3 tools/summarize_pmc.py $OUT > $OUT/summary.txt 2>&1
python3 tools/mfma_util.py $OUT/mfma >> $OUT/summary.txt 2>&1
python3 tools/traffic_from_pmc.py $OUT > $OUT/traffic.json 2>> $OUT/summary.txt
# keep only the summaries (raw traces are large)
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -delete
find $OUT -name "*agent_info.csv" -delete
head -60 $OUT/summary.txt
