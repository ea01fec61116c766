#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r04c5; mkdir -p $O
timeout 900 python -m pytest tests/test_learner_gpu.py tests/test_kernels_gpu.py -q -m gpu -x -k "weight_copies or weight_packing or row_sorted or full_size" > $O/tests.log 2>&1; tail -5 $O/tests.log
rocprofv3 --kernel-trace --stats --output-format csv -d $O/c3trace -- python3 bench.py --config C3 --steps 4 --warmup 2 --no-cpu-baseline --no-peaks > $O/c3trace.json 2> $O/c3trace.err
f=$(find $O/c3trace -name "*kernel_stats.csv" | head -1); cp "$f" $O/c3_kernel_stats.csv; find $O/c3trace -name "*.csv" -size +2M -delete
head -45 $O/c3_kernel_stats.csv | cut -c1-200
