#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r04c30; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_encoder_gpu.py -q -m gpu -x -k "winograd or conv_algorithms or golden or invariance" 2>&1 | tail -4
timeout 300 python tools/wino_c64_ablate.py 2>&1 | tee $O/abl.txt
