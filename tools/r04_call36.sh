#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
for rep in 1 2; do for v in $LIBS; do echo "== $v"; CADRE_HIP_LIB=$PWD/tools/_trace/libcadre_$v.so timeout 300 python tools/enc_kernel_times.py --frames 2048 --dtype bf16 2>&1 | grep -E "forward|launch +(6|8|11|13|16|18|23) "; done; done
