#!/bin/bash
# Ablation of the fp32 GEMM k-loop (diagnostic builds: results are wrong by construction, only the
# timing matters) — which part of the loop keeps the matrix pipe below the 'nolds' ceiling.
#   nostore: global loads kept, LDS writes dropped     nogload: LDS writes kept, global loads dropped
#   noload : neither                                   nobar  : noload without the k-loop barrier
#   nolds  : nobar without the fragment reads (pure MFMA + epilogue)
cd $GRAFT_REPO_ROOT
bash tools/ab_build.sh noload -DABL_NOLOAD &
bash tools/ab_build.sh nobar -DABL_NOLOAD -DABL_NOBAR &
bash tools/ab_build.sh nolds -DABL_NOLOAD -DABL_NOBAR -DABL_NOLDSREAD &
bash tools/ab_build.sh nostore -DABL_NOSTORE &
bash tools/ab_build.sh nogload -DABL_NOGLOAD &
bash tools/ab_build.sh noepi -DABL_NOEPI &
wait
python tools/ab_gemm.py base=cadre_amd/csrc/libcadre_hip.so nostore=/tmp/v_nostore.so nogload=/tmp/v_nogload.so noload=/tmp/v_noload.so nobar=/tmp/v_nobar.so nolds=/tmp/v_nolds.so noepi=/tmp/v_noepi.so 2>&1 | tail -14
