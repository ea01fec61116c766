#!/usr/bin/env python3
"""Micro-benchmark of cadre_gemm_bf16 on the encoder's conv shapes and a dense 8192^3."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cadre_amd import hip  # noqa: E402
from tools.gemm_bench import timeit  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=512)
ap.add_argument("--tiles", default="0,1,4,7")
a = ap.parse_args()
F = a.frames
tiles = [int(t) for t in a.tiles.split(",")]
hip.lib()
cases = [("layer1 3x3 64->64 @72", 72, 72, 64, 64, 3, 1, 1), ("layer2.0 3x3s2 64->128 @72", 72, 72, 64, 128, 3, 2, 1), ("layer2 3x3 128->128 @36", 36, 36, 128, 128, 3, 1, 1),
         ("layer3 3x3 256->256 @18", 18, 18, 256, 256, 3, 1, 1), ("layer4 3x3 512->512 @9", 9, 9, 512, 512, 3, 1, 1)]
for name, H, W, ci, co, k, s, p in cases:
    x = torch.randn(F, H, W, ci, device="cuda").bfloat16(); w = (torch.randn(co, k * k * ci, device="cuda") * 0.05).bfloat16()
    Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    out = torch.empty(F, Ho, Wo, co, device="cuda", dtype=torch.bfloat16)
    sc = torch.rand(co, device="cuda"); sh = torch.randn(co, device="cuda")
    K, M = k * k * ci, F * Ho * Wo
    row = []
    for tl in tiles:
        if (tl in (1, 4, 7) and co <= 64) or (tl == 7 and co < 256) or (tl in (10, 11) and co > 128):
            row.append("   --  "); continue
        t = timeit(lambda: hip.gemm(x, w, out, M, co, K, 0, K, co, a_mode=2, scale=sc, shift=sh, act=1,
                                    conv=(H, W, ci, Ho, Wo, k, k, s, p), bf16=True, flags=2, tile=tl))
        row.append("%6.0f " % (2.0 * M * co * K / t / 1e12))
    print("%-28s F=%d TFLOP/s by tile %s: %s" % (name, F, tiles, " ".join(row)), flush=True)
for tl in tiles:
    if tl == 0:
        continue
    M = N = K = 8192
    A = torch.randn(M, K, device="cuda").bfloat16(); B = torch.randn(N, K, device="cuda").bfloat16()
    Cc = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    t = timeit(lambda: hip.gemm(A, B, Cc, M, N, K, K, K, N, bf16=True, flags=2, tile=tl))
    print("dense 8192^3 bf16 tile %d: %7.0f TFLOP/s" % (tl, 2.0 * M * N * K / t / 1e12), flush=True)
