#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r04c13; mkdir -p $O
for mc in 256 128; do echo "CADRE_WINOGRAD_MIN_C=$mc"; CADRE_WINOGRAD_MIN_C=$mc timeout 300 python tools/enc_kernel_times.py --frames 1024 --dtype f32 2>&1 | grep -E "forward"; done | tee $O/wino_minc.txt
timeout 900 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/r04c13/bench.json').read().strip().splitlines()[-1])
print("C2", d['value'], d['ms_per_step'], d['t_encode_ms'], d['t_update_ms'], d['roofline']['kernel'], d['roofline']['frac'], d['config']['conv_algorithm'][:40])
w=d['c2_direct_conv']; print("C2 direct", w['value'], w['ms_per_step'], w['t_encode_ms'], w['winograd_vs_direct'])
c=d['c3']; print("C3", c['value'], c['ms_per_step'], c['t_encode_ms'], c['t_update_ms'])
PY
timeout 900 python -m pytest tests/test_encoder_gpu.py -q -m gpu -x 2>&1 | tail -3
