#!/usr/bin/env python3
"""A/B of libcadre_hip.so build variants in ONE process (interleaved rounds, median):
    python tools/ab_gemm.py base=cadre_amd/csrc/libcadre_hip.so prio=/tmp/v_prio.so ..."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cadre_amd import hip  # noqa: E402

libs = {}
for a in sys.argv[1:]:
    name, path = a.split("=")
    L = C.CDLL(path)
    L.cadre_gemm_f32.argtypes = [C.POINTER(hip.GemmDesc), C.c_void_p]
    L.cadre_gemm_f32.restype = C.c_int
    libs[name] = L


def desc(x, w, out, M, N, K, conv, a_mode, tile):
    d = hip.GemmDesc()
    d.A, d.B, d.C = x.data_ptr(), w.data_ptr(), out.data_ptr()
    d.lda, d.ldb, d.ldc = (0 if a_mode >= 2 else K), K, N
    d.M, d.N, d.K, d.a_mode, d.b_mode, d.act, d.batch = M, N, K, a_mode, 0, 1, 1
    for f in ("a_div", "b_div", "c_div", "s_div", "r_div"):
        setattr(d, f, 1)
    if conv:
        d.H, d.W, d.Cin, d.Ho, d.Wo, d.KH, d.KW, d.stride, d.pad = conv
    d.split_k, d.tile, d.flags = 1, tile, 0
    return d


F = 512
cases = [("stem 7x7s2 4->64 @288 t3", 288, 288, 4, 64, 7, 2, 3, 3), ("layer1 3x3 64->64 @72 t3", 72, 72, 64, 64, 3, 1, 1, 3), ("layer2 3x3 128 @36 t3", 36, 36, 128, 128, 3, 1, 1, 3),
         ("layer2 3x3 128 @36 t1", 36, 36, 128, 128, 3, 1, 1, 1), ("layer2 3x3 128 @36 t8", 36, 36, 128, 128, 3, 1, 1, 8),
         ("layer3 3x3 256 @18 t3", 18, 18, 256, 256, 3, 1, 1, 3), ("layer3 3x3 256 @18 t8", 18, 18, 256, 256, 3, 1, 1, 8),
         ("layer4 3x3 512 @9 t3", 9, 9, 512, 512, 3, 1, 1, 3), ("layer4 3x3 512 @9 t8", 9, 9, 512, 512, 3, 1, 1, 8),
         ("dense 4096^3 t1", 0, 0, 0, 0, 0, 0, 0, 1), ("dense 4096^3 t8", 0, 0, 0, 0, 0, 0, 0, 8),
         ("lstm step 64x2120x544 x8 t3", -1, 64, 2120, 544, 8, 0, 0, 3), ("lstm step 64x2120x544 x8 t9", -1, 64, 2120, 544, 8, 0, 0, 9),
         ("lstm step 64x2120x544 x8 t2", -1, 64, 2120, 544, 8, 0, 0, 2),
         ("lstm step 64x2120x544 x8 t9 seg16", -1, 64, 2120, 544, 8, 16, 0, 9), ("lstm step 64x2120x544 x8 t3 seg16", -1, 64, 2120, 544, 8, 16, 0, 3)]
if os.environ.get("AB_ONLY"):
    cases = [c for c in cases if os.environ["AB_ONLY"] in c[0]]
st = torch.cuda.current_stream().cuda_stream
for name, H, W, ci, co, k, s, p, tile in cases:
    if H == -1:        # batched skinny GEMM: (name, -1, M, N, K, batch, ...)
        M, N, K, Z = W, ci, co, k
        x = torch.randn(Z, M, K, device="cuda"); w = torch.randn(Z, N, K, device="cuda") * 0.05; out = torch.empty(Z, M, N, device="cuda")
        d = desc(x, w, out, M, N, K, None, 0, tile)
        d.batch = Z
        d.a_str, d.b_str, d.c_str = M * K, N * K, M * N
        d.a_mod = d.b_mod = d.c_mod = 1 << 30
        if s:          # row-sorted: batch entry z owns rows [s*(z%4), s*(z%4)+s) of the 64-row period
            segt = torch.tensor([[s * (z % 4), s] for z in range(Z)], dtype=torch.int32, device="cuda")
            d.seg_mode, d.seg_period, d.seg_div, d.row_seg = 1, M, 1, segt.data_ptr()
    elif H:
        Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
        Kc = k * 32 if ci == 4 else k * k * ci
        x = torch.randn(F, H, W, ci, device="cuda"); w = torch.randn(co, Kc, device="cuda") * 0.05
        out = torch.empty(F, Ho, Wo, co, device="cuda")
        M, N, K = F * Ho * Wo, co, Kc
        d = desc(x, w, out, M, N, K, (H, W, ci, Ho, Wo, k, k, s, p), 3 if ci == 4 else 2, tile)
    else:
        M = N = K = 4096
        x = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda"); out = torch.empty(M, N, device="cuda")
        d = desc(x, w, out, M, N, K, None, 0, tile)
    times = {n: [] for n in libs}
    for rnd in range(12):
        for n, L in libs.items():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); rc = L.cadre_gemm_f32(C.byref(d), st); e1.record()
            torch.cuda.synchronize()
            assert rc == 0
            if rnd >= 2:
                times[n].append(e0.elapsed_time(e1) * 1e-3)
    Zb = max(1, d.batch)
    row = "  ".join("%s %6.1f (%6.1f us)" % (n, 2.0 * M * N * K * Zb / sorted(t)[len(t) // 2] / 1e12, sorted(t)[len(t) // 2] * 1e6) for n, t in times.items())
    print("%-28s TFLOP/s: %s" % (name, row), flush=True)
