#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r04c33; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_encoder_gpu.py tests/test_timed_shapes_gpu.py -q -m gpu -x -k "winograd or conv_algorithms or golden or invariance or timed or chunk" 2>&1 | tail -4
timeout 300 python tools/wino_c64_ab.py flr nobr 2>&1 | tail -7 | tee $O/ab.txt
timeout 300 python tools/wino_c64_ablate.py 2>&1 | tee $O/abl.txt
timeout 900 python bench.py --no-cpu-baseline --no-c3 --no-peaks > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/r04c33/bench.json').read().strip().splitlines()[-1])
print("C2", d['value'], d['ms_per_step'], d['t_encode_ms'], d['t_update_ms'], d['roofline']['kernel'], d['roofline']['frac'])
for k,v in list(d['roofline']['per_kernel'].items())[:6]: print("   ", k, v)
w=d['c2_direct_conv']; print("C2 direct", w['value'], w['ms_per_step'], w['t_encode_ms'], w['winograd_vs_direct'])
PY
