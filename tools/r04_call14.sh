#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r04c14; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_encoder_gpu.py -q -m gpu -x -k "winograd or conv_algorithms or golden" -s 2>&1 | grep -v amdgpu.ids | tail -14 | tee $O/wino_tests.txt
for m in 2 3; do echo "CADRE_WINOGRAD_M=$m"; CADRE_WINOGRAD_M=$m timeout 300 python tools/enc_kernel_times.py --frames 1024 --dtype f32 2>&1 | grep -E "forward|launch (1[1-36-9]|22) "; done | tee $O/wino_m.txt
timeout 900 python bench.py --no-cpu-baseline --no-c3 --no-peaks > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/r04c14/bench.json').read().strip().splitlines()[-1])
print("C2", d['value'], d['ms_per_step'], d['t_encode_ms'], d['t_update_ms'], d['roofline']['kernel'], d['roofline']['frac'])
w=d['c2_direct_conv']; print("C2 direct", w['value'], w['ms_per_step'], w['t_encode_ms'], w['winograd_vs_direct'])
PY
