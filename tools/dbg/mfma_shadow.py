#!/usr/bin/env python3
"""Cycles per v_mfma_f32_16x16x4_f32 (32 alone) with K independent instructions behind each, ONE wave per SIMD (tools/dbg/mfma_shadow.hip).
    python tools/dbg/mfma_shadow.py --build-only     # build container
    python tools/dbg/mfma_shadow.py                  # GPU box"""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SO = os.path.join(ROOT, "tools", "_trace", "libmfma_shadow.so")
if "--build-only" in sys.argv:
    os.makedirs(os.path.dirname(SO), exist_ok=True)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-o", SO,
                           os.path.join(ROOT, "tools", "dbg", "mfma_shadow.hip")])
    sys.exit(0)
import torch
L = ctypes.CDLL(SO)
L.shadow_run.argtypes = [ctypes.c_int] * 4 + [ctypes.c_void_p, ctypes.c_void_p]
L.shadow_run_bf16.argtypes = [ctypes.c_int] * 4 + [ctypes.c_void_p, ctypes.c_void_p]
sink = torch.zeros(4, device="cuda")
iters, wgs = 4000, 256
names = {0: "v_pk_add_f32", 1: "v_add_f32", 2: "s_nop 0", 3: "ds_read_b64", 4: "v_mov_b32"}
for wgs, label in ((256, "one wave per SIMD"), (512, "two waves per SIMD (two workgroups per CU)")):
    print(label)
    for op in range(5):
        row = []
        for k in (0, 1, 2, 3, 4, 6, 8):
            best = 1e9
            for _ in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                assert L.shadow_run(op, k, wgs, iters, sink.data_ptr(), None) == 0
                e1.record(); torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1))
            waves_per_simd = wgs // 256
            row.append(best * 1e-3 * 2.4e9 / (iters * 32 * waves_per_simd))
        print("  %-14s cycles per MFMA (2.4 GHz) at K = 0,1,2,3,4,6,8: %s" % (names[op], "  ".join("%5.1f" % v for v in row)))

print("v_mfma_f32_32x32x16_bf16 (8 per iteration; alone: 16 passes)")
for wgs, label in ((256, "one wave per SIMD"), (512, "two waves per SIMD (two workgroups per CU)")):
    print(label)
    for op in range(5):
        row = []
        for k in (0, 1, 2, 3, 4, 6, 8):
            best = 1e9
            for _ in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                assert L.shadow_run_bf16(op, k, wgs, iters, sink.data_ptr(), None) == 0
                e1.record(); torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1))
            row.append(best * 1e-3 * 2.4e9 / (iters * 8 * (wgs // 256)))
        print("  %-14s cycles per MFMA (2.4 GHz) at K = 0,1,2,3,4,6,8: %s" % (names[op], "  ".join("%5.1f" % v for v in row)))
