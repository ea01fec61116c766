#!/bin/bash
# rocprofv3 kernel trace + stats and HBM-traffic / MFMA PMC passes of ONE bench section per command (`bench.py --section headline`,
# `bench.py --section c3`): the profiled launches are exactly those of the section whose roofline the bench line carries, so the
# per-launch averages under profiles/ and the line's HIP-event averages describe the same population.
# FETCH_SIZE / WRITE_SIZE in separate passes, never combined with other trace domains.
#   bash tools/prof_bench.sh [headline c3]      -> gpurun_out/prof_bench/{<section>/..., summary_<section>.txt, traffic.json}
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
OUT=gpurun_out/prof_bench
rm -rf $OUT; mkdir -p $OUT
SECTIONS="${*:-headline c3}"
STEPS=2; WARM=2
for S in $SECTIONS; do
  O=$OUT/$S; mkdir -p $O
  ARGS="--section $S --steps $STEPS --warmup $WARM"
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 bench.py $ARGS > $O/trace.json 2> $O/trace.err
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -- python3 bench.py $ARGS > $O/fetch.json 2> $O/fetch.err
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -- python3 bench.py $ARGS > $O/write.json 2> $O/write.err
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/mfma -- python3 bench.py $ARGS > $O/mfma.json 2> $O/mfma.err
  python3 tools/summarize_pmc.py $O > $OUT/summary_$S.txt 2>&1
  python3 tools/mfma_util.py $O/mfma >> $OUT/summary_$S.txt 2>&1
  # rounds of the profiled command: warm-up + timed + the three untimed split passes of run_config
  python3 tools/traffic_from_pmc.py $O $((STEPS + WARM + 3)) > $O/traffic.json 2>> $OUT/summary_$S.txt
  f=$(find $O/trace -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/kernel_stats_$S.csv
  cp $O/trace.json $OUT/bench_under_rocprof_$S.json
  # per-kernel census of one update step (between two Adam launches) of this section's trace
  python3 tools/update_step_kernels.py $O/trace > $OUT/update_step_census_$S.txt 2>&1
done
python3 - $OUT $SECTIONS > $OUT/traffic.json <<'PY'
import json, os, sys
root, secs = sys.argv[1], sys.argv[2:]
out = {"_note": "per section: rocprofv3 FETCH_SIZE (x2, gfx950) + WRITE_SIZE passes of `python bench.py --section <section> --steps 2 --warmup 2`; "
                "rounds = learner rounds of that command (2 warm-up + 2 timed + 3 untimed split passes), launches = dispatches of the kernel in it"}
for s in secs:
    out[s] = json.load(open(os.path.join(root, s, "traffic.json")))
json.dump(out, sys.stdout, indent=1, sort_keys=True)
PY
# keep only the summaries (raw traces are large)
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -delete
find $OUT -name "*agent_info.csv" -delete
for S in $SECTIONS; do head -40 $OUT/summary_$S.txt; done
