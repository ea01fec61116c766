#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r04c16; mkdir -p $O
CADRE_RING_C64S=0 timeout 600 python tools/ring_ablate.py --shape 2048 36 36 128 128 --shape 2048 72 72 64 64 --only 1 2 4 8 16 32 64 12 46 127 2>&1 | grep -v amdgpu.ids > $O/ring_ablate.txt; tail -45 $O/ring_ablate.txt
for s in "2048 36 36 128 128" "2048 72 72 64 64"; do for r in 0 1; do CADRE_RING_C64S=0 timeout 120 python tools/ring_trace.py --mode 4 --dtype bf16 --shape $s --resid $r 2>&1 | grep -v amdgpu.ids; done; done | tee $O/ring_clock.txt
timeout 300 python tools/enc_kernel_times.py --frames 2048 --dtype bf16 2>&1 | grep -v amdgpu.ids > $O/enc_layers_bf16.txt; head -3 $O/enc_layers_bf16.txt
timeout 300 python tools/enc_kernel_times.py --frames 1024 --dtype f32 2>&1 | grep -v amdgpu.ids > $O/enc_layers_f32.txt; head -3 $O/enc_layers_f32.txt
