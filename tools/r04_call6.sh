#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r04c6; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu -x -k "fused_stem_pool" > $O/tests.log 2>&1; tail -5 $O/tests.log
for ch in 3 1; do echo "CADRE_STEM_CH=$ch"; CADRE_STEM_CH=$ch timeout 600 python tools/stem_ablate.py --only 4 8 12 14 2>&1 | grep -v amdgpu.ids; done | tee $O/stem_ablate.txt
