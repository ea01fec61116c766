#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r04c11; mkdir -p $O
timeout 900 python bench.py --no-cpu-baseline --no-c3 --no-peaks > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/r04c11/bench.json').read().strip().splitlines()[-1])
print("C2", d['value'], d['ms_per_step'], d['t_encode_ms'], d['t_update_ms'])
w=d['c2_winograd']; print("C2 winograd", w['value'], w['ms_per_step'], w['t_encode_ms'], w['vs_direct_conv'], w['encoder_tflops_executed'], w['encoder_tflops_direct_conv_equivalent'])
for k,v in list(w['per_kernel'].items())[:8]: print("   ", k, v)
PY
CADRE_WINOGRAD=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-peaks --no-c3 --no-winograd > $O/trace.json 2> $O/trace.err
f=$(find $O/trace -name "*kernel_stats.csv" | head -1); cp "$f" $O/wino_kernel_stats.csv; find $O/trace -name "*.csv" -size +1M -delete
head -14 $O/wino_kernel_stats.csv | cut -c1-150
