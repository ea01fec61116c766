#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r03w; mkdir -p $O
timeout 300 python tools/attn_bench.py 2>&1 | grep "F=" | tee $O/attn.txt
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_encoder_gpu.py tests/test_timed_shapes_gpu.py -q -k "pam or encoder or golden" > $O/t.log 2>&1; tail -5 $O/t.log
