#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r04c18; mkdir -p $O
timeout 300 python tools/dbg/wino_err.py 2>&1 | grep -v amdgpu.ids | tee $O/wino_err.txt
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_encoder_gpu.py -q -m gpu -x -k "winograd or conv_algorithms or workspace" -s 2>&1 | grep -v amdgpu.ids | tail -8
timeout 800 python -m pytest tests/test_dp_gpu.py -q -m gpu -x -s -k two_ranks_one_gpu_match 2>&1 | grep -E "vs oracle|passed|failed"
