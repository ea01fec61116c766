#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r03o; mkdir -p $O
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -k "lstm or ppo_loss" > $O/t_lstm.log 2>&1; tail -4 $O/t_lstm.log
timeout 1200 python -m pytest tests/test_learner_gpu.py tests/test_timed_shapes_gpu.py tests/test_dp_gpu.py tests/test_topology_gpu.py -q > $O/t_learner.log 2>&1; tail -5 $O/t_learner.log
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/r03o/bench.json').read().strip().splitlines()[-1])
print("C2", d['value'], d['ms_per_step'], d['t_encode_ms'], d['t_update_ms'], d['update_roofline']['ms_per_step'], d['update_roofline']['hbm_frac'])
c=d['c3']; print("C3", c['value'], c['ms_per_step'], c['t_encode_ms'], c['t_update_ms'], c['update_roofline']['ms_per_step'], c['update_roofline']['hbm_frac'])
PY
