#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r04c2; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu -x -k "ring" > $O/ring_tests.log 2>&1; tail -15 $O/ring_tests.log
timeout 600 python tools/enc_kernel_times.py --frames 2048 --dtype bf16 --passes 3 > $O/enc_bf16_pp2.txt 2>&1; grep -E "ring|forward" $O/enc_bf16_pp2.txt | head -30
CADRE_RING_G=1 timeout 600 python tools/enc_kernel_times.py --frames 2048 --dtype bf16 --passes 3 > $O/enc_bf16_pp1.txt 2>&1; grep -E "ring|forward" $O/enc_bf16_pp1.txt | head -30
