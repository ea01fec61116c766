#!/usr/bin/env python3
"""What bounds the dense 128 x 128-wave-tile kernel on the inter-task shape: builds gemm_bf16_w128.hip with -DGW_ABL=<bits> (one .so
per ablation, into tools/_trace/) and times each, interleaved rounds in one process.  Ablations skip work (results wrong by
construction): 1 MFMAs, 2 A-fragment reads, 4 A DMA, 8 B loads, 16 stores.
    python tools/gw128_ablate.py --build-only      # build container
    python tools/gw128_ablate.py                   # GPU box"""
import argparse
import ctypes
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ABLS = [0, 16, 8, 4, 2, 12, 30, 31]
NAMES = {0: "full kernel", 1: "no MFMA", 2: "no A-fragment reads", 4: "no A DMA", 8: "no B loads", 16: "no stores",
         12: "no A DMA, no B loads", 30: "schedule + MFMA only", 31: "empty schedule (barriers + waits)"}
FLAGS = ["-mllvm", "-enable-misched=0", "-mllvm", "-pragma-unroll-threshold=262144"]


def so_path(abl):
    return os.path.join(ROOT, "tools", "_trace", "libgw128_abl%d.so" % abl)


def build():
    os.makedirs(os.path.join(ROOT, "tools", "_trace"), exist_ok=True)
    srcs = [os.path.join(ROOT, "cadre_amd", "csrc", f) for f in ("gemm_bf16_w128.hip", "cadre_kernels.hip")]
    procs = [subprocess.Popen(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC",
                               "-DGW_ABL=%d" % abl] + FLAGS + ["-o", so_path(abl)] + srcs, stderr=subprocess.DEVNULL) for abl in ABLS]
    for p in procs:
        assert p.wait() == 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=2048)
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--build-only", action="store_true")
    args = ap.parse_args()
    if args.build_only:
        return build()
    import torch
    from cadre_amd.encoder import _w128_dense_b
    vp = ctypes.c_void_p
    libs = {}
    for abl in ABLS:
        if os.path.exists(so_path(abl)):
            L = ctypes.CDLL(so_path(abl))
            L.cadre_gemm_bf16_w128.argtypes = [vp, vp, vp] + [ctypes.c_int32] * 6 + [vp]
            libs[abl] = L
    M, N, K, split = args.frames, 1536, 41472, 16
    A = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    B = _w128_dense_b(torch.randn(N, K) * 0.05).to(torch.bfloat16).cuda()
    C = torch.empty(split, M, N, device="cuda")
    a = (A.data_ptr(), B.data_ptr(), C.data_ptr(), M, N, K, K, N, split, None)
    t = {k: [] for k in libs}
    for L in libs.values():
        for _ in range(2):
            assert L.cadre_gemm_bf16_w128(*a) == 0
    torch.cuda.synchronize()
    for _ in range(args.rounds):
        for k, L in libs.items():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                assert L.cadre_gemm_bf16_w128(*a) == 0
            e1.record()
            torch.cuda.synchronize()
            t[k].append(e0.elapsed_time(e1) / 3)
    fl = 2.0 * M * N * K
    print("dense bf16 [%d][%d] x [%d][%d]^T, %d slices (%.0f GFLOP; A %.0f MB, B %.0f MB, slabs %.0f MB)" % (M, K, N, K, split, fl / 1e9, M * K * 2 / 1e6, N * K * 2 / 1e6, split * M * N * 4 / 1e6))
    base = np.median(t[0])
    for k in libs:
        m = np.median(t[k])
        print("  abl %3d %-40s median %7.1f us  min %7.1f  (%5.1f %% of full; %6.0f TF)" % (k, NAMES[k], 1e3 * m, 1e3 * min(t[k]), 100 * m / base, fl / m / 1e9), flush=True)


if __name__ == "__main__":
    main()
