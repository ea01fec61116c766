#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r04c22; mkdir -p $O
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -m gpu -x -k "winograd_c64" 2>&1 | tail -12
for v in 1 0; do echo "CADRE_WINOGRAD_C64=$v"; CADRE_WINOGRAD_C64=$v timeout 300 python tools/enc_kernel_times.py --frames 1024 --dtype f32 2>&1 | grep -E "forward|launch  [0-3] "; done | tee $O/enc_c64.txt
