#!/bin/bash
# round-4 artefacts: the default bench line, rocprofv3 kernel stats + PMC passes of the same command (tools/prof_bench.sh)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r04final4; mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -15 > $O/gpu_tests.log; tail -3 $O/gpu_tests.log
timeout 1500 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
bash tools/prof_bench.sh > $O/prof_bench.log 2>&1; echo "prof rc=$?"
cp gpurun_out/prof_bench/summary.txt $O/rocprof_summary.txt; cp gpurun_out/prof_bench/traffic.json $O/hbm_traffic.json
f=$(find gpurun_out/prof_bench/trace -name "*kernel_stats.csv" | head -1); cp "$f" $O/kernel_stats.csv
cp gpurun_out/prof_bench/trace.json $O/bench_under_rocprof.json
timeout 300 python tools/peaks_bench.py > $O/peaks.txt 2>&1
python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/r04final4/bench.json').read().strip().splitlines()[-1])
print("C2", d['value'], d['ms_per_step'], d['t_encode_ms'], d['t_update_ms'], d['roofline']['kernel'], d['roofline']['frac'], d['cpu_baseline'])
w=d['c2_direct_conv']; print("C2 direct", w['value'], w['ms_per_step'], w['t_encode_ms'], w['winograd_vs_direct'])
c=d['c3']; print("C3", c['value'], c['ms_per_step'], c['t_encode_ms'], c['t_update_ms'], c['roofline']['kernel'], c['roofline']['frac'])
PY
