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
from cadre_amd.encoder import _ring_w, _s1x_w, _s2_w            # noqa: E402


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
    # conv2 of the same blocks: window conv with the shortcut tensor as residual + the shortcut's own launch on the tile kernel,
    # against cadre_conv3x3_s1x (the shortcut as K-extension)
    for name, H, C1, Cd in (("layer2.0", 36, 128, 64), ("layer3.0", 18, 256, 128), ("layer4.0", 9, 512, 256)):
        t = torch.randn(F, H, H, C1, device="cuda", generator=g).to(torch.bfloat16)
        x = torch.randn(F, 2 * H, 2 * H, Cd, device="cuda", generator=g).to(torch.bfloat16)
        w2 = (torch.randn(C1, C1, 3, 3, device="cuda", generator=g) / np.sqrt(9 * C1)).to(torch.bfloat16)
        wd = (torch.randn(C1, Cd, 1, 1, device="cuda", generator=g) / np.sqrt(Cd)).to(torch.bfloat16)
        sh2, shd = torch.randn(C1, device="cuda", generator=g), torch.randn(C1, device="cuda", generator=g)
        wf = _s1x_w(w2.float().cpu(), wd.float().cpu()).to(torch.bfloat16).cuda()
        wr = _ring_w(w2.float().cpu(), 64).to(torch.bfloat16).cuda()
        wdk = wd.reshape(C1, Cd).contiguous()
        M = F * H * H
        idt = torch.empty(F, H, H, C1, device="cuda", dtype=torch.bfloat16)
        o1 = torch.empty_like(idt); o2 = torch.empty_like(idt)
        shs = (sh2 + shd).contiguous()

        def run_fused():
            hip.conv3x3_s1x(t, x, wf, shs, o1, F, H, H, C1, Cd, C1, 1)

        def run_down():
            hip.gemm(x, wdk, idt, M, C1, Cd, 0, Cd, C1, a_mode=2, shift=shd, act=0, conv=(2 * H, 2 * H, Cd, H, H, 1, 1, 2, 0), bf16=True, flags=2)

        def run_ring():
            hip.conv3x3_ring(t, wr, None, sh2, idt, o2, F, H, H, C1, C1, 1)
        ts = {"fused": [], "down": [], "ring": []}
        for rnd in range(6):
            for k, fn in (("fused", run_fused), ("down", run_down), ("ring", run_ring)):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                fn(); torch.cuda.synchronize()
                e0.record()
                for _ in range(5):
                    fn()
                e1.record(); torch.cuda.synchronize()
                if rnd:
                    ts[k].append(e0.elapsed_time(e1) / 5)
        d = float((o1.float() - o2.float()).abs().max() / o2.float().abs().max())
        fl = 2.0 * M * C1 * (9 * C1 + Cd)
        tf, td_, tr = np.median(ts["fused"]), np.median(ts["down"]), np.median(ts["ring"])
        print("%s conv2 %d->%d @%d + shortcut %d->%d  fused %.3f ms (%.1f TFLOP/s)  |  window conv + residual %.3f + shortcut launch %.3f = %.3f ms   max diff %.2e"
              % (name, C1, C1, H, Cd, C1, tf, fl / tf / 1e9, tr, td_, tr + td_, d), flush=True)


def plain():
    """The new kernel's structure against conv3x3_ring_pp_kernel at equal work: cadre_conv3x3_s1x with NO shortcut (Cd = 0) on the
    stride-1 convs without residual (conv1 of the second blocks)."""
    F = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
    g = torch.Generator(device="cuda").manual_seed(2)
    for name, H, C in (("layer2", 36, 128), ("layer3", 18, 256), ("layer4", 9, 512)):
        t = torch.randn(F, H, H, C, device="cuda", generator=g).to(torch.bfloat16)
        w = (torch.randn(C, C, 3, 3, device="cuda", generator=g) / np.sqrt(9 * C)).to(torch.bfloat16)
        sh = torch.randn(C, device="cuda", generator=g)
        wr = _ring_w(w.float().cpu(), 64).to(torch.bfloat16).cuda()
        wf = wr.reshape(C, -1, 64).contiguous()
        o1 = torch.empty(F, H, H, C, device="cuda", dtype=torch.bfloat16); o2 = torch.empty_like(o1)
        ts = {"s1x": [], "ring": []}
        fns = {"s1x": lambda: hip.conv3x3_s1x(t, None, wf, sh, o1, F, H, H, C, 0, C, 1),
               "ring": lambda: hip.conv3x3_ring(t, wr, None, sh, None, o2, F, H, H, C, C, 1)}
        for rnd in range(6):
            for k, fn in fns.items():
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                fn(); torch.cuda.synchronize()
                e0.record()
                for _ in range(5):
                    fn()
                e1.record(); torch.cuda.synchronize()
                if rnd:
                    ts[k].append(e0.elapsed_time(e1) / 5)
        fl = 2.0 * F * H * H * C * 9 * C
        d = float((o1.float() - o2.float()).abs().max() / o2.float().abs().max())
        print("%s 3x3/s1 %d->%d @%d no residual: s1x (no shortcut) %.3f ms (%.1f TFLOP/s) | ring_pp %.3f ms (%.1f TFLOP/s)   max diff %.2e"
              % (name, C, C, H, np.median(ts["s1x"]), fl / np.median(ts["s1x"]) / 1e9, np.median(ts["ring"]), fl / np.median(ts["ring"]) / 1e9, d), flush=True)


if __name__ == "__main__":
    main()
    plain()
