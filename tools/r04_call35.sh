#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r04c35; mkdir -p $O
timeout 1200 python -m pytest tests/test_kernels_gpu.py tests/test_encoder_gpu.py tests/test_timed_shapes_gpu.py tests/test_bench_joint_gpu.py -q -m gpu -x 2>&1 | tail -3
timeout 900 python bench.py --no-cpu-baseline --no-peaks --no-direct-conv > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/r04c35/bench.json').read().strip().splitlines()[-1])
print("C2", d['value'], d['ms_per_step'], d['t_encode_ms'], d['t_update_ms'], d['roofline']['kernel'], d['roofline']['frac'])
c=d['c3']; print("C3", c['value'], c['ms_per_step'], c['ms_per_step_min_median_max'], c['t_encode_ms'], c['t_update_ms'], c['roofline']['kernel'], c['roofline']['frac'], c['encoder_fwd_hbm_frac'])
for k,v in list(c['roofline']['per_kernel'].items())[:9]: print("   ", k, v)
PY
