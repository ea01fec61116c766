#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r03x; mkdir -p $O
for cfg in C2 C3; do
CADRE_BENCH_HOST_TRACE=1 timeout 600 python3 bench.py --config $cfg --steps 3 --warmup 2 --no-cpu-baseline --no-peaks --no-c3 > $O/b_$cfg.json 2> $O/b_$cfg.err
grep "host timeline" $O/b_$cfg.err | tail -3
done
