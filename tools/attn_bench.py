#!/usr/bin/env python3
"""Timing of the position / channel attention kernels (cadre_pam, cadre_cam; da_att.py:32-83) alone: F frames of an
Np-position map, 20 launches back to back per kernel (HIP events on the launch stream)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cadre_amd import hip  # noqa: E402


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def main():
    L = hip.lib()
    g = torch.Generator(device="cuda").manual_seed(3)
    for F, Np in ((256, 81), (1024, 81), (2048, 81), (1024, 40), (8, 40)):
        x = torch.randn(F * Np, 128, device="cuda", generator=g)
        qkv = torch.randn(F * Np, 160, device="cuda", generator=g) * 0.3
        y = torch.empty_like(x)
        yb = torch.empty(F * Np, 128, device="cuda", dtype=torch.bfloat16)
        st = hip.stream()
        t = [timeit(lambda: hip.check(L.cadre_pam(x.data_ptr(), qkv.data_ptr(), 0.5, y.data_ptr(), F, Np, st), "pam")),
             timeit(lambda: hip.check(L.cadre_cam(x.data_ptr(), 0.5, y.data_ptr(), F, Np, st), "cam")),
             timeit(lambda: hip.check(L.cadre_pam_bf16out(x.data_ptr(), qkv.data_ptr(), 0.5, yb.data_ptr(), F, Np, st), "pamb")),
             timeit(lambda: hip.check(L.cadre_cam_bf16out(x.data_ptr(), 0.5, yb.data_ptr(), F, Np, st), "camb"))]
        print("F=%d Np=%d: pam %.1f us, cam %.1f us, pam (bf16 out) %.1f us, cam (bf16 out) %.1f us" % ((F, Np) + tuple(t)), flush=True)


if __name__ == "__main__":
    main()
