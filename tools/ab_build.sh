#!/bin/bash
# build variant libs with extra -D flags: tools/ab_build.sh name "-DFLAG" ...
cd "$GRAFT_REPO_ROOT" || cd /root/repo
name=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC "$@" -o /tmp/v_$name.so cadre_amd/csrc/gemm_f32.hip cadre_amd/csrc/conv_stream_f32.hip cadre_amd/csrc/gemm_bf16.hip cadre_amd/csrc/conv_stream_bf16.hip cadre_amd/csrc/cadre_kernels.hip 2>/dev/null
