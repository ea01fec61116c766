#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r03h; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_encoder_gpu.py -q -x -k "ring or encoder or bf16" > $O/t_ring.log 2>&1; tail -4 $O/t_ring.log
python - <<'PY' 2>&1 | tee gpurun_out/r03h/peaks.txt
import sys; sys.path.insert(0,'.')
from tools.peaks_bench import measure
r=measure()
for k,v in r.items(): print("%-36s %s"%(k,v))
PY
for m in 1 0 1 0; do
  echo "=== CADRE_RING_M16=$m"; CADRE_RING_M16=$m timeout 600 python tools/enc_kernel_times.py --dtype bf16 --frames 2048 --passes 4 2>&1 | grep -E "forward|ring" | tee -a $O/enc_m16_$m.txt
done
