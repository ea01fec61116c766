#!/bin/bash
# round-3 artefact pass: full -m gpu suite, the default bench line, rocprofv3 kernel stats + HBM-traffic PMC passes
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r03final; mkdir -p $O
timeout 1500 python -m pytest tests/ -q -m gpu > $O/gpu_tests.log 2>&1; tail -3 $O/gpu_tests.log
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/r03final/bench.json').read().strip().splitlines()[-1])
print("C2", d['value'], d['ms_per_step'], d['t_encode_ms'], d['t_update_ms'], d['update_roofline']['ms_per_step'], d['update_roofline']['hbm_frac'], d['roofline']['frac'])
c=d['c3']; print("C3", c['value'], c['ms_per_step'], c['t_encode_ms'], c['t_update_ms'], c['update_roofline']['ms_per_step'], c['update_roofline']['hbm_frac'], c['roofline']['frac'])
PY
timeout 1500 bash tools/prof_bench.sh > $O/prof_head.txt 2>&1
mkdir -p $O/prof; cp gpurun_out/prof_bench/summary.txt gpurun_out/prof_bench/traffic.json gpurun_out/prof_bench/trace.json $O/prof/ 2>/dev/null
find gpurun_out/prof_bench/trace -name "*kernel_stats.csv" -exec cp {} $O/prof/kernel_stats.csv \;
head -30 $O/prof_head.txt
