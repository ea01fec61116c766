#!/usr/bin/env python3
"""What bounds the 128 x 128-wave-tile window conv: builds conv3x3_w128.hip with -DW1_ABL=<bits> (one .so per ablation, into
tools/_trace/) and times each on the trunk shapes, interleaved rounds in one process.  Ablations skip work (results wrong by
construction): 1 MFMAs, 2 pixel-fragment reads, 4 window DMA, 8 weight loads, 16 stores, 32 epilogue arithmetic, 64 residual loads.
    python tools/w128_ablate.py --build-only [--define W1_D=3]      # build container (hipcc cross-compiles)
    python tools/w128_ablate.py [--resid]                            # GPU box"""
import argparse
import ctypes
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ABLS = [0, 1, 2, 8, 4, 48, 14, 62, 63]
NAMES = {128: "zero bias fragment (no table read)", 384: "no bias MFMAs", 432: "no bias MFMAs, no epilogue", 0: "full kernel", 1: "no MFMA", 2: "no pixel-fragment reads", 4: "no window DMA", 8: "no weight loads", 16: "no stores",
         48: "no epilogue (arithmetic, stores)", 14: "MFMA + epilogue only (no reads, no DMA, no weight loads)",
         62: "schedule + MFMA only", 63: "empty schedule (barriers + waits)", 64: "no residual loads", 112: "no epilogue, no residual loads"}
FLAGS = ["-mllvm", "-enable-misched=0", "-mllvm", "-pragma-unroll-threshold=262144"]


def so_path(abl, tag):
    return os.path.join(ROOT, "tools", "_trace", "libw128_%sabl%d.so" % (tag, abl))


def build(extra, tag, abls):
    os.makedirs(os.path.join(ROOT, "tools", "_trace"), exist_ok=True)
    srcs = [os.path.join(ROOT, "cadre_amd", "csrc", f) for f in ("ab/conv3x3_w128.hip", "cadre_kernels.hip")]
    for i in range(0, len(abls), 4):
        procs = [subprocess.Popen(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC",
                                   "-DW1_ABL=%d" % abl] + FLAGS + extra + ["-o", so_path(abl, tag)] + srcs,
                                  stderr=subprocess.DEVNULL) for abl in abls[i:i + 4]]
        for p in procs:
            assert p.wait() == 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=2048)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--build-only", action="store_true")
    ap.add_argument("--resid", action="store_true")
    ap.add_argument("--tag", default="")
    ap.add_argument("--abls", default=None)
    ap.add_argument("--define", action="append", default=[], help="extra -D for the builds, e.g. W1_D=3")
    args = ap.parse_args()
    abls = [int(v) for v in args.abls.split(",")] if args.abls else ABLS
    if args.build_only:
        return build(["-D" + d for d in args.define], args.tag, abls)
    import torch
    from cadre_amd.encoder import _w128_w
    vp = ctypes.c_void_p
    libs = {}
    for abl in abls:
        if os.path.exists(so_path(abl, args.tag)):
            L = ctypes.CDLL(so_path(abl, args.tag))
            L.cadre_conv3x3_w128.argtypes = [vp] * 6 + [ctypes.c_int32] * 6 + [vp]
            libs[abl] = L
    F = args.frames
    for H, C in ((36, 128), (18, 256), (9, 512)):
        x = torch.randn(F, H, H, C, device="cuda").to(torch.bfloat16)
        r = torch.randn(F, H, H, C, device="cuda").to(torch.bfloat16)
        w = _w128_w(torch.randn(C, C, 3, 3) * 0.05).to(torch.bfloat16).cuda()
        sh = torch.zeros(C, device="cuda")
        out = torch.empty(F, H, H, C, device="cuda", dtype=torch.bfloat16)
        a = (x.data_ptr(), w.data_ptr(), None, sh.data_ptr(), r.data_ptr() if args.resid else None, out.data_ptr(), F, H, H, C, C, 1, None)
        t = {k: [] for k in libs}
        for L in libs.values():
            for _ in range(2):
                assert L.cadre_conv3x3_w128(*a) == 0
        torch.cuda.synchronize()
        for _ in range(args.rounds):
            for k, L in libs.items():
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(3):
                    assert L.cadre_conv3x3_w128(*a) == 0
                e1.record()
                torch.cuda.synchronize()
                t[k].append(e0.elapsed_time(e1) / 3)
        M = F * H * H
        fl = 2.0 * M * C * 9 * C
        nb = (M * C * (3 if args.resid else 2)) * 2
        print("bf16 3x3/s1 F=%d %dx%d %d->%d %s (%.0f GFLOP, %.2f GB algorithmic)" % (F, H, H, C, C, "resid" if args.resid else "", fl / 1e9, nb / 1e9))
        base = np.median(t[abls[0]])
        for k in libs:
            m = np.median(t[k])
            print("  abl %3d %-58s median %7.1f us  min %7.1f  (%5.1f %% of full; %6.0f TF, %5.2f TB/s)" % (
                k, NAMES.get(k, "?"), 1e3 * m, 1e3 * min(t[k]), 100 * m / base, fl / m / 1e9, nb / m / 1e9), flush=True)


if __name__ == "__main__":
    main()
