#!/usr/bin/env python3
"""PAM / CAM timing at the bench's shape (F frames of 9 x 9 x 128): python tools/dbg/pam_cam_time.py [F]   (CADRE_PAM_LARGE=1 / CADRE_CAM_LARGE=1: the
row-block kernels at this size)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cadre_amd import hip
L = hip.lib()
F = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
Np = 81
x = torch.randn(F, Np, 128, device="cuda") * 0.4
qkv = torch.randn(F * Np, 160, device="cuda") * 0.3
y = torch.empty_like(x)
def t(fn):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 10 * 1e3
tp = t(lambda: hip.check(L.cadre_pam(hip.ptr(x), hip.ptr(qkv), 0.5, hip.ptr(y), F, Np, hip.stream()), "pam"))
yp = y.clone()
tc = t(lambda: hip.check(L.cadre_cam(hip.ptr(x), 0.7, hip.ptr(y), F, Np, hip.stream()), "cam"))
print("F=%d Np=%d: pam %.1f us, cam %.1f us  (checksums %.6f %.6f)" % (F, Np, tp, tc, float(yp.double().sum()), float(y.double().sum())))
