#!/usr/bin/env python3
"""Rounding error of one 3x3 conv layer against float64: direct (implicit GEMM), Winograd F(2x2) / F(3x3) / F(4x4) / F(6x6) (three
launches: cadre_winograd_in -> batched cadre_gemm_f32 -> cadre_winograd_out), torch-CPU fp32."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cadre_amd import hip
from cadre_amd.encoder import _winograd_u, _khwc
hip.lib()
g = torch.Generator().manual_seed(0)
for (F, H, W, C, N) in ((16, 9, 9, 512, 512), (8, 18, 18, 256, 256), (4, 36, 36, 128, 128), (32, 3, 3, 512, 512)):
    x = torch.relu(torch.randn(F, H, W, C, generator=g))
    w = torch.randn(N, C, 3, 3, generator=g) / (9 * C) ** 0.5
    ref = torch.nn.functional.conv2d(x.double().permute(0, 3, 1, 2), w.double(), padding=1).permute(0, 2, 3, 1)
    cpu32 = torch.nn.functional.conv2d(x.permute(0, 3, 1, 2), w, padding=1).permute(0, 2, 3, 1)
    xd = x.cuda()
    out = torch.empty(F, H, W, N, device="cuda")
    K = 9 * C
    hip.gemm(xd, _khwc(w).cuda(), out, F * H * W, N, K, 0, K, N, a_mode=2, conv=(H, W, C, H, W, 3, 3, 1, 1))
    res = {"torch-CPU fp32": cpu32, "direct (MFMA implicit GEMM)": out.cpu()}
    for m in (2, 3, 4, 6):
        P, T = (m + 2) ** 2, F * -(-H // m) * -(-W // m)
        V = torch.empty(P, T, C, device="cuda"); Mx = torch.empty(P, T, N, device="cuda"); o = torch.empty(F, H, W, N, device="cuda")
        L = hip.lib()
        hip.check(L.cadre_winograd_in(hip.ptr(xd), hip.ptr(V), F, H, W, C, m, hip.stream()), "in")
        hip.gemm(V, _winograd_u(w, m).cuda(), Mx, T, N, C, C, C, N, batch=P, a_z=(1, P, T * C), b_z=(1, P, N * C), c_z=(1, P, T * N))
        hip.check(L.cadre_winograd_out(hip.ptr(Mx), None, None, None, hip.ptr(o), F, H, W, N, 0, m, hip.stream()), "out")
        res["Winograd F(%dx%d)" % (m, m)] = o.cpu()
        del V, Mx, o
    s = ref.abs().max()
    print("%dx%d C=%d N=%d:" % (H, W, C, N), "  ".join("%s max %.2e rms %.2e" % (k, float((v.double() - ref).abs().max() / s), float(((v.double() - ref) ** 2).mean().sqrt() / s)) for k, v in res.items()))
