#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r04c37; mkdir -p $O
timeout 1200 python -m pytest tests/test_kernels_gpu.py tests/test_encoder_gpu.py tests/test_timed_shapes_gpu.py tests/test_dp_gpu.py -q -m gpu -x -k "winograd or conv_algorithms or golden or invariance or timed or chunk or dp or ranks" 2>&1 | tail -4
for v in 0 1 0 1; do echo "CADRE_WINOGRAD_M4=$v"; CADRE_WINOGRAD_M4=$v timeout 300 python tools/enc_kernel_times.py --frames 1024 --dtype f32 2>&1 | grep -E "forward|launch +(6|7|8) "; done | tee $O/m4.txt
timeout 900 python bench.py --no-cpu-baseline --no-c3 --no-peaks > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/r04c37/bench.json').read().strip().splitlines()[-1])
print("C2", d['value'], d['ms_per_step'], d['t_encode_ms'], d['t_update_ms'], d['roofline']['kernel'], d['roofline']['frac'])
for k,v in list(d['roofline']['per_kernel'].items())[:6]: print("   ", k, v)
w=d['c2_direct_conv']; print("C2 direct", w['value'], w['ms_per_step'], w['t_encode_ms'], w['winograd_vs_direct'])
PY
