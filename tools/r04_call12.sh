#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r04c12; mkdir -p $O
for w in 0 1; do for sz in "144 256" "288 288"; do echo "CADRE_WINOGRAD=$w $sz"; CADRE_WINOGRAD=$w timeout 300 python tools/act_latency.py $sz 2>&1 | grep "act()"; done; done | tee $O/act_latency.txt
CADRE_WINOGRAD=1 timeout 2400 python -m pytest tests -q -m gpu -x --deselect tests/test_ab_ring_gpu.py > $O/tests_wino.log 2>&1; tail -4 $O/tests_wino.log
