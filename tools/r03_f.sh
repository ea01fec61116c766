#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r03f; mkdir -p $O
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "lstm_dw" > $O/t_dw.log 2>&1; tail -3 $O/t_dw.log
CADRE_DW_TM=1 timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "lstm_dw" > $O/t_dw1.log 2>&1; tail -3 $O/t_dw1.log
timeout 300 python tools/lstm_step_bench.py 2>&1 | sed "s/^/TM=2 /" | tee $O/step_bench.txt
CADRE_DW_TM=1 timeout 300 python tools/lstm_step_bench.py 2>&1 | sed "s/^/TM=1 /" | tee -a $O/step_bench.txt
