#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r04c1; mkdir -p $O
timeout 600 python tools/ring_ablate.py > $O/ablate.txt 2>&1; tail -40 $O/ablate.txt
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/r04c1/bench.json').read().strip().splitlines()[-1])
print("C2", d['value'], d['ms_per_step'], d['t_encode_ms'], d['t_update_ms'], d['update_roofline']['ms_per_step'], d['update_roofline']['hbm_frac'], d['roofline']['frac'])
c=d['c3']; print("C3", c['value'], c['ms_per_step'], c['t_encode_ms'], c['t_update_ms'], c['update_roofline']['ms_per_step'], c['update_roofline']['hbm_frac'], c['roofline']['frac'])
PY
timeout 1800 python -m pytest tests/ -q -m gpu -x > $O/gpu_tests.log 2>&1; tail -5 $O/gpu_tests.log
