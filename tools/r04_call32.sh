#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r04c32; mkdir -p $O
timeout 600 python -m pytest tests/test_learner_gpu.py -q -m gpu -x 2>&1 | tail -3
for v in 0 1 0 1; do
  CADRE_UPDATE_OVERLAP=$v timeout 600 python bench.py --no-cpu-baseline --no-peaks --no-direct-conv --steps 3 --c3-steps 5 > $O/b$v.json 2> $O/b$v.err
  python3 - <<PY
import json
d=json.loads(open('$O/b$v.json').read().strip().splitlines()[-1])
print("overlap=$v  C2 %.2f ms (upd %.3f, step %.4f)   C3 %.2f ms (upd %.3f, step %.4f)" % (d['ms_per_step'], d['t_update_ms'], d['update_roofline']['ms_per_step'], d['c3']['ms_per_step'], d['c3']['t_update_ms'], d['c3']['update_roofline']['ms_per_step']))
PY
done
