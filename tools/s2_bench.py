#!/usr/bin/env python3
"""Stride-2 3x3 convs of the bf16 trunk (layer2.0 / 3.0 / 4.0 conv1) at 2048 frames: the plane-window kernel
(cadre_conv3x3_s2, csrc/conv3x3_s2.hip) against the implicit-GEMM tile kernel it replaces (cadre_gemm_bf16 a_mode 2),
interleaved rounds in one process, HIP events.  python tools/s2_bench.py [frames]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cadre_amd import hip                      # noqa: E402
from cadre_amd.encoder import _s2_w            # noqa: E402


def main():
    F = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
    g = torch.Generator(device="cuda").manual_seed(1)
    for name, H, Cin, Cout in (("layer2.0", 72, 64, 128), ("layer3.0", 36, 128, 256), ("layer4.0", 18, 256, 512)):
        x = torch.randn(F, H, H, Cin, device="cuda", generator=g).to(torch.bfloat16)
        w = (torch.randn(Cout, Cin, 3, 3, device="cuda", generator=g) / np.sqrt(9 * Cin)).to(torch.bfloat16)
        sh = torch.randn(Cout, device="cuda", generator=g)
        w2 = _s2_w(w.float().cpu()).to(torch.bfloat16).cuda()
        wk = w.permute(0, 2, 3, 1).reshape(Cout, -1).contiguous()
        Ho = H // 2
        M, K = F * Ho * Ho, 9 * Cin
        o1 = torch.empty(F, Ho, Ho, Cout, device="cuda", dtype=torch.bfloat16)
        o2 = torch.empty_like(o1)

        def run_s2():
            hip.conv3x3_s2(x, w2, None, sh, o1, F, H, H, Cin, Cout, 1)

        def run_tile():
            hip.gemm(x, wk, o2, M, Cout, K, 0, K, Cout, a_mode=2, shift=sh, act=1, conv=(H, H, Cin, Ho, Ho, 3, 3, 2, 1), bf16=True, flags=2)
        ts = {"s2": [], "tile": []}
        for rnd in range(6):
            for k, fn in (("s2", run_s2), ("tile", run_tile)):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                fn(); torch.cuda.synchronize()
                e0.record()
                for _ in range(5):
                    fn()
                e1.record(); torch.cuda.synchronize()
                if rnd:
                    ts[k].append(e0.elapsed_time(e1) / 5)
        d = float((o1.float() - o2.float()).abs().max() / o2.float().abs().max())
        fl = 2.0 * M * Cout * K
        nb = (F * H * H * Cin + M * Cout) * 2
        for k in ("s2", "tile"):
            t = np.median(ts[k])
            print("%s 3x3/s2 %d->%d @%d F=%d  %-4s %.3f ms (min %.3f)  %7.1f TFLOP/s  %6.0f GB/s" % (
                name, Cin, Cout, H, F, k, t, min(ts[k]), fl / t / 1e9, nb / t / 1e6), flush=True)
        print("   max |s2 - tile| / max |tile| = %.2e" % d, flush=True)


if __name__ == "__main__":
    main()
