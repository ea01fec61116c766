#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r03i; mkdir -p $O
timeout 900 python -m pytest tests/test_learner_gpu.py -q -x -s -k "act" > $O/t_act.log 2>&1; grep -E "passed|failed|act\(\)|Error|error" $O/t_act.log | tail -8
timeout 300 python tools/act_latency.py 2>&1 | tee $O/act_latency.txt
timeout 900 python -m pytest tests/test_learner_gpu.py tests/test_replay_snapshot_gpu.py tests/test_topology_gpu.py -q -x > $O/t_more.log 2>&1; tail -4 $O/t_more.log
python - <<'PY' 2>&1 | grep random
import sys; sys.path.insert(0,'.')
from tools.peaks_bench import measure
r=measure()
for k,v in r.items(): print("%-36s %s"%(k,v))
PY
