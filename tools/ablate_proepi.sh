cd $GRAFT_REPO_ROOT
bash tools/ab_build.sh nopro -DABL_NOPRO &
bash tools/ab_build.sh noepi -DABL_NOEPI &
bash tools/ab_build.sh noboth -DABL_NOPRO -DABL_NOEPI &
wait
python tools/ab_gemm.py base=cadre_amd/csrc/libcadre_hip.so nopro=/tmp/v_nopro.so noepi=/tmp/v_noepi.so noboth=/tmp/v_noboth.so 2>&1 | tail -12
