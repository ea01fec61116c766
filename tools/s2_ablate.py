#!/usr/bin/env python3
"""What bounds the stride-2 plane-window conv: builds conv3x3_s2.hip with -DS2_ABL=<bits> (one .so per ablation, into
tools/_trace/) and times each on the three trunk shapes, interleaved rounds in one process.  Ablations skip work (results
wrong by construction): 1 MFMAs, 2 fragment reads, 4 window DMA, 8 weight DMA, 16 epilogue.
    python tools/s2_ablate.py --build-only        # build container (hipcc cross-compiles)
    python tools/s2_ablate.py                     # GPU box"""
import argparse
import ctypes
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ABLS = [0, 16]
NAMES = {0: "full kernel", 1: "no MFMA", 2: "no fragment reads", 4: "no window DMA", 8: "no weight DMA", 16: "no epilogue",
         12: "no DMA at all", 14: "MFMA + epilogue only (no reads, no DMA)", 30: "schedule + MFMA only", 31: "empty schedule (barriers + waits)",
         32: "stores fully coalesced (wrong places)"}


def so_path(abl):
    return os.path.join(ROOT, "tools", "_trace", "libs2_abl%d.so" % abl)


def build(extra):
    os.makedirs(os.path.join(ROOT, "tools", "_trace"), exist_ok=True)
    srcs = [os.path.join(ROOT, "cadre_amd", "csrc", f) for f in ("conv3x3_s2.hip", "cadre_kernels.hip")]
    procs = [subprocess.Popen(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC",
                               "-DS2_ABL=%d" % abl] + extra + ["-o", so_path(abl)] + srcs,
                              stderr=subprocess.DEVNULL) for abl in ABLS]
    for p in procs:
        assert p.wait() == 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=2048)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--build-only", action="store_true")
    ap.add_argument("--define", action="append", default=[], help="extra -D for the builds, e.g. S2_CAP=3")
    args = ap.parse_args()
    if args.build_only:
        return build(["-D" + d for d in args.define])
    import torch
    from cadre_amd.encoder import _s2_w
    vp = ctypes.c_void_p
    libs = {}
    for abl in ABLS:
        if os.path.exists(so_path(abl)):
            L = ctypes.CDLL(so_path(abl))
            L.cadre_conv3x3_s2.argtypes = [vp] * 5 + [ctypes.c_int32] * 6 + [vp]
            libs[abl] = L
    F = args.frames
    for H, Cin, N in ((72, 64, 128), (36, 128, 256), (18, 256, 512)):
        x = torch.randn(F, H, H, Cin, device="cuda").to(torch.bfloat16)
        w = _s2_w(torch.randn(N, Cin, 3, 3) * 0.05).to(torch.bfloat16).cuda()
        sh = torch.zeros(N, device="cuda")
        out = torch.empty(F, H // 2, H // 2, N, device="cuda", dtype=torch.bfloat16)
        a = (x.data_ptr(), w.data_ptr(), None, sh.data_ptr(), out.data_ptr(), F, H, H, Cin, N, 1, None)
        t = {k: [] for k in libs}
        for L in libs.values():
            for _ in range(2):
                assert L.cadre_conv3x3_s2(*a) == 0
        torch.cuda.synchronize()
        for _ in range(args.rounds):
            for k, L in libs.items():
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(3):
                    assert L.cadre_conv3x3_s2(*a) == 0
                e1.record()
                torch.cuda.synchronize()
                t[k].append(e0.elapsed_time(e1) / 3)
        M = F * (H // 2) ** 2
        fl = 2.0 * M * N * 9 * Cin
        nb = (F * H * H * Cin + M * N) * 2
        print("bf16 3x3/s2 F=%d %dx%d %d->%d  (%.0f GFLOP, %.2f GB algorithmic)" % (F, H, H, Cin, N, fl / 1e9, nb / 1e9))
        base = np.median(t[0])
        for k in libs:
            m = np.median(t[k])
            print("  abl %3d %-42s median %7.1f us  min %7.1f  (%5.1f %% of full; %6.0f TF, %5.2f TB/s)" % (
                k, NAMES[k], 1e3 * m, 1e3 * min(t[k]), 100 * m / base, fl / m / 1e9, nb / m / 1e9), flush=True)


if __name__ == "__main__":
    main()
