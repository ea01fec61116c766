#!/usr/bin/env python3
"""cadre_gemm_bf16 on the C3 encoder's tile-GEMM launches (stride-2 3x3 convs, 1x1 convs) at 2048 frames, by tile."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cadre_amd import hip  # noqa: E402
from tools.gemm_bench import timeit  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=2048)
ap.add_argument("--tiles", default="0,1,4,7,13")
a = ap.parse_args()
F = a.frames
tiles = [int(t) for t in a.tiles.split(",")]
hip.lib()
cases = [("layer2.0 3x3/s2 64->128 @72", 72, 72, 64, 128, 3, 2, 1), ("layer2.0 1x1/s2 64->128 @72", 72, 72, 64, 128, 1, 2, 0),
         ("layer3.0 3x3/s2 128->256 @36", 36, 36, 128, 256, 3, 2, 1), ("layer3.0 1x1/s2 128->256 @36", 36, 36, 128, 256, 1, 2, 0),
         ("layer4.0 3x3/s2 256->512 @18", 18, 18, 256, 512, 3, 2, 1), ("layer4.0 1x1/s2 256->512 @18", 18, 18, 256, 512, 1, 2, 0),
         ("head 1x1 512->512 @9", 9, 9, 512, 512, 1, 1, 0), ("head 1x1 128->512 @9", 9, 9, 128, 512, 1, 1, 0)]
for name, H, W, ci, co, k, s, p in cases:
    x = torch.randn(F, H, W, ci, device="cuda").bfloat16(); w = (torch.randn(co, k * k * ci, device="cuda") * 0.05).bfloat16()
    Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    out = torch.empty(F, Ho, Wo, co, device="cuda", dtype=torch.bfloat16)
    sc = torch.rand(co, device="cuda"); sh = torch.randn(co, device="cuda")
    K, M = k * k * ci, F * Ho * Wo
    row = []; ref = None
    for tl in tiles:
        if (tl in (7, 14) and co < 256):
            row.append("     --     "); continue
        t = timeit(lambda: hip.gemm(x, w, out, M, co, K, 0, K, co, a_mode=2, scale=sc, shift=sh, act=1,
                                    conv=(H, W, ci, Ho, Wo, k, k, s, p), bf16=True, flags=2, tile=tl))
        if ref is None:
            ref = out.clone()
        same = "" if bool((out == ref).all()) else " DIFF %.3g" % float((out.float() - ref.float()).abs().max())
        row.append("%5.0f us %4.0f%s" % (t * 1e6, 2.0 * M * co * K / t / 1e12, same))
    print("%-30s M=%-8d us / TFLOP/s by tile %s: %s" % (name, M, tiles, " | ".join(row)), flush=True)
