#!/bin/bash
# The artefact pass of a round, one parameterised script (replaces the one-shot r0N_call*.sh / r0N_final*.sh files):
#   gpurun --timeout 3000 -- 'bash tools/round_artefacts.sh r05 [tests] [bench] [prof] [peaks] [layers]'
# Writes gpurun_out/<tag>_artefacts/: gpu_tests.log, bench.json (+ .err), rocprofv3 kernel stats + PMC summaries of the SAME
# bench command (tools/prof_bench.sh), hbm_traffic.json, peaks.txt, per-layer encoder timings.  Copy what is to be judged
# into profiles/ (tracked) afterwards.  No step names = all steps.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
TAG=${1:-r06}; shift
STEPS="${*:-tests bench prof peaks layers}"
O=gpurun_out/${TAG}_artefacts; mkdir -p $O
has() { [[ " $STEPS " == *" $1 "* ]]; }
if has tests; then
  timeout 2400 python -m pytest tests -q -m gpu > $O/gpu_tests.log 2>&1; echo "tests rc=$?"; tail -4 $O/gpu_tests.log
fi
if has bench; then
  timeout 1800 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
  python3 tools/bench_digest.py $O/bench.json
fi
if has prof; then
  bash tools/prof_bench.sh > $O/prof_bench.log 2>&1; echo "prof rc=$?"
  for S in headline c3; do
    cp gpurun_out/prof_bench/summary_$S.txt $O/rocprof_summary_$S.txt; cp gpurun_out/prof_bench/kernel_stats_$S.csv $O/kernel_stats_$S.csv
    cp gpurun_out/prof_bench/bench_under_rocprof_$S.json $O/bench_under_rocprof_$S.json
  done
  cp gpurun_out/prof_bench/traffic.json $O/hbm_traffic.json
  cp gpurun_out/prof_bench/update_step_census_headline.txt $O/update_step_census_C2.txt
  cp gpurun_out/prof_bench/update_step_census_c3.txt $O/update_step_census_C3.txt
fi
if has peaks; then
  timeout 300 python tools/peaks_bench.py > $O/peaks.txt 2>&1; tail -3 $O/peaks.txt
fi
if has layers; then
  timeout 300 python tools/enc_kernel_times.py --frames 2048 --dtype bf16 > $O/enc_layers_bf16_2048.txt 2>&1
  timeout 300 python tools/enc_kernel_times.py --frames 1024 --dtype f32 > $O/enc_layers_f32_1024.txt 2>&1
  head -3 $O/enc_layers_bf16_2048.txt $O/enc_layers_f32_1024.txt
fi
