#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r04c4; mkdir -p $O
timeout 1500 python -m pytest tests/test_kernels_gpu.py tests/test_encoder_gpu.py tests/test_timed_shapes_gpu.py -q -m gpu -x > $O/tests.log 2>&1; tail -5 $O/tests.log
timeout 900 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/r04c4/bench.json').read().strip().splitlines()[-1])
print("C2", d['value'], d['ms_per_step'], d['t_encode_ms'], d['t_update_ms'], d['update_roofline']['ms_per_step'], d['roofline']['frac'])
c=d['c3']; print("C3", c['value'], c['ms_per_step'], c['t_encode_ms'], c['t_update_ms'], c['update_roofline']['ms_per_step'], c['roofline']['frac'], c['roofline']['kernel'])
for k,v in list(c['roofline']['per_kernel'].items())[:8]: print("   ", k, v)
PY
