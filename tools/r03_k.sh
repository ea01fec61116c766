#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r03k; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_encoder_gpu.py -q -x -k "ring or encoder or bf16" > $O/t_ring.log 2>&1; tail -4 $O/t_ring.log
for i in 1 2; do timeout 600 python tools/enc_kernel_times.py --dtype bf16 --frames 2048 --passes 4 2>&1 | grep -E "forward|ring" | cut -c1-170; done | tee $O/enc_bf16.txt
