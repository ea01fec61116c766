#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r03g; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q > $O/tests.log 2>&1; tail -12 $O/tests.log
timeout 900 python bench.py --steps 5 --warmup 2 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.load(open("gpurun_out/r03g/bench.json"))
print("C2", d["value"], d["ms_per_step"], d["t_encode_ms"], d["t_update_ms"], d["update_roofline"]["ms_per_step"], d["update_roofline"]["kernels_per_update"], d["update_roofline"]["hbm_frac"])
c=d["c3"]; print("C3", c["value"], c["ms_per_step"], c["t_encode_ms"], c["t_update_ms"], c["update_roofline"]["ms_per_step"], c["update_roofline"]["hbm_frac"])
PY
