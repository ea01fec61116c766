#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r03s; mkdir -p $O
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -k "mlp_towers" > $O/t_mlp.log 2>&1; tail -12 $O/t_mlp.log; grep "mlp towers" $O/t_mlp.log
timeout 1200 python -m pytest tests/test_learner_gpu.py tests/test_timed_shapes_gpu.py -q -x > $O/t_learner.log 2>&1; tail -5 $O/t_learner.log
for cfg in C2 C3; do
  OUT=$O/ktrace_$cfg; rm -rf $OUT; mkdir -p $OUT
  timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 bench.py --config $cfg --steps 2 --warmup 2 --no-cpu-baseline --no-peaks --no-c3 > $OUT/trace.json 2> $OUT/trace.err
  python3 tools/update_step_kernels.py $OUT/trace > $OUT/step_census.txt 2>&1
  find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*agent_info.csv" -delete
  cat $OUT/step_census.txt
  python3 -c "
import json;d=json.loads(open('$OUT/trace.json').read().strip().splitlines()[-1]);print('   step ms', d['update_roofline']['ms_per_step'], 'value', d['value'])"
done
