#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r04c8; mkdir -p $O
timeout 2400 python -m pytest tests -q -m gpu -x > $O/tests.log 2>&1; tail -6 $O/tests.log
timeout 900 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/r04c8/bench.json').read().strip().splitlines()[-1])
print("C2", d['value'], d['ms_per_step'], d['t_encode_ms'], d['t_update_ms'], d['update_roofline']['ms_per_step'], d['roofline']['frac'])
c=d['c3']; print("C3", c['value'], c['ms_per_step'], c['t_encode_ms'], c['t_update_ms'], c['update_roofline']['ms_per_step'], c['roofline']['frac'], c['roofline']['kernel'])
PY
