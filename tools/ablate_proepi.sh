#!/bin/bash
# Ablation of the per-tile overheads of the fp32 GEMM: im2col decode (nopro), epilogue (noepi).
cd $GRAFT_REPO_ROOT
bash tools/ab_build.sh nopro -DABL_NOPRO &
bash tools/ab_build.sh noepi -DABL_NOEPI &
bash tools/ab_build.sh noboth -DABL_NOPRO -DABL_NOEPI &
bash tools/ab_build.sh nogload -DABL_NOGLOAD &
bash tools/ab_build.sh nolds -DABL_NOLOAD -DABL_NOBAR -DABL_NOLDSREAD &
wait
python tools/ab_gemm.py base=cadre_amd/csrc/libcadre_hip.so nopro=/tmp/v_nopro.so noepi=/tmp/v_noepi.so noboth=/tmp/v_noboth.so nogload=/tmp/v_nogload.so nolds=/tmp/v_nolds.so 2>&1 | tail -14
