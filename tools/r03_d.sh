#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r03d; mkdir -p $O
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -k "lstm_dw" > $O/t_dw.log 2>&1; tail -5 $O/t_dw.log
timeout 300 python tools/lstm_step_bench.py 2>&1 | tee $O/step_bench.txt
for rt in 1; do CADRE_LSTM_RT=$rt timeout 300 python tools/lstm_step_bench.py 2>&1 | sed "s/^/RT=$rt /" | tee -a $O/step_bench.txt; done
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc1 -- python3 tools/lstm_step_bench.py > $O/pmc1.txt 2>&1
timeout 600 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum --kernel-trace --output-format csv -d $O/pmc2 -- python3 tools/lstm_step_bench.py > $O/pmc2.txt 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc3 -- python3 tools/lstm_step_bench.py > $O/pmc3.txt 2>&1
python3 - <<'PY'
import csv, glob, collections
for d in ("pmc1","pmc2","pmc3"):
    fs=glob.glob("gpurun_out/r03d/%s/**/*counter_collection.csv"%d, recursive=True)
    if not fs: print(d,"no csv"); continue
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        k=r["Kernel_Name"][:60]
        if "lstm_" in k or "transpose" in k:
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in agg.items():
        print(d, k)
        for c,vals in v.items():
            print("     %-24s mean %.4g  n=%d  (last quarter mean %.4g)"%(c, sum(vals)/len(vals), len(vals), sum(vals[-len(vals)//4:])/max(1,len(vals[-len(vals)//4:]))))
PY
find $O -name "*.csv" -size +2M -delete
timeout 900 python -m pytest tests/test_learner_gpu.py tests/test_timed_shapes_gpu.py -q -x > $O/t_learner.log 2>&1; tail -5 $O/t_learner.log
