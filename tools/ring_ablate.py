#!/usr/bin/env python3
"""What bounds a window-conv launch: builds conv3x3_ring.hip with -DRING_ABL=<bits> (one .so per ablation, into
tools/_trace/) and times each on the same shape, interleaved rounds in one process (cdna_hip_programming.md 5.4 rule 24).
Ablations skip work (results are wrong by construction): 1 MFMAs, 2 fragment reads, 4 window DMA, 8 weight DMA,
16 global stores, 32 the whole epilogue, 64 residual loads.
    python tools/ring_ablate.py --build-only                 # in the build container (hipcc cross-compiles)
    python tools/ring_ablate.py --shape 2048 72 72 64 64     # on the GPU box"""
import argparse
import ctypes
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ABLS = [0, 1, 2, 4, 8, 16, 32, 64, 3, 12, 48, 15, 63, 127, 46, 44, 36, 34, 126]
NAMES = {128: "no halo masks, no position arithmetic (padded-layout emulation)", 160: "no halo masks, no epilogue", 0: "full kernel", 1: "no MFMA", 2: "no fragment reads", 4: "no window DMA", 8: "no weight DMA", 16: "no stores",
         32: "no epilogue", 64: "no residual loads", 3: "no MFMA, no reads", 12: "no DMA at all", 48: "no epilogue, no stores",
         46: "schedule + MFMA only (no reads, DMA, epilogue)", 44: "MFMA + reads only (no DMA, no epilogue)",
         36: "no window DMA, no epilogue", 34: "no reads, no epilogue", 126: "schedule + MFMA only, residual loads off too",
         15: "DMA + reads + MFMA off (epilogue only)", 63: "everything off but residual loads", 127: "empty schedule (barriers + waits)"}


def so_path(abl):
    return os.path.join(ROOT, "tools", "_trace", "libring_abl%d.so" % abl)


def build():
    os.makedirs(os.path.join(ROOT, "tools", "_trace"), exist_ok=True)
    srcs = [os.path.join(ROOT, "cadre_amd", "csrc", f) for f in ("conv3x3_ring.hip", "cadre_kernels.hip")]
    procs = []
    for abl in ABLS:
        procs.append(subprocess.Popen(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC",
                                       "-DRING_ABL=%d" % abl, "-o", so_path(abl)] + srcs, stderr=subprocess.DEVNULL))
        if len(procs) >= 7:
            for p in procs:
                assert p.wait() == 0
            procs = []
    for p in procs:
        assert p.wait() == 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", type=int, nargs=5, action="append", metavar=("F", "H", "W", "CIN", "N"))
    ap.add_argument("--resid", type=int, nargs="*", default=[0, 1])
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--build-only", action="store_true")
    ap.add_argument("--only", type=int, nargs="*", default=None, help="ablation codes to build / run (default: all)")
    args = ap.parse_args()
    global ABLS
    if args.only is not None:
        ABLS = [0] + [a for a in args.only if a != 0]
    if args.build_only:
        return build()
    import torch
    from cadre_amd.encoder import _ring_w
    vp = ctypes.c_void_p
    libs = {}
    for abl in ABLS:
        if os.path.exists(so_path(abl)):
            L = ctypes.CDLL(so_path(abl))
            L.cadre_conv3x3_ring.argtypes = [vp] * 6 + [ctypes.c_int32] * 7 + [vp]
            libs[abl] = L
    for (F, H, W, Cin, N) in (args.shape or [(2048, 72, 72, 64, 64), (2048, 36, 36, 128, 128)]):
        for resid in args.resid:
            x = torch.randn(F, H, W, Cin, device="cuda").to(torch.bfloat16)
            wr = _ring_w(torch.randn(N, Cin, 3, 3) * 0.05, 64).to(torch.bfloat16).cuda()
            sc, sh = torch.ones(N, device="cuda"), torch.zeros(N, device="cuda")
            res = torch.randn(F, H, W, N, device="cuda").to(torch.bfloat16) if resid else None
            out = torch.empty(F, H, W, N, device="cuda", dtype=torch.bfloat16)
            a = (x.data_ptr(), wr.data_ptr(), sc.data_ptr(), sh.data_ptr(), res.data_ptr() if resid else None, out.data_ptr(),
                 F, H, W, Cin, N, 1, 1 | 2 | (4 if resid else 0), None)
            t = {k: [] for k in libs}
            for L in libs.values():
                for _ in range(2):
                    assert L.cadre_conv3x3_ring(*a) == 0
            torch.cuda.synchronize()
            for _ in range(args.rounds):
                for k, L in libs.items():
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(3):
                        assert L.cadre_conv3x3_ring(*a) == 0
                    e1.record()
                    torch.cuda.synchronize()
                    t[k].append(e0.elapsed_time(e1) / 3)
            fl = 2.0 * F * H * W * N * 9 * Cin
            nb = F * H * W * 2 * (Cin + N * (2 if resid else 1))
            print("bf16 F=%d %dx%d %d->%d resid=%d  (%.0f GFLOP, %.2f GB algorithmic)" % (F, H, W, Cin, N, resid, fl / 1e9, nb / 1e9))
            base = np.median(t[0])
            for k in libs:
                m = np.median(t[k])
                print("  abl %3d %-44s median %7.1f us  min %7.1f  (%5.1f %% of full; %6.0f TF, %5.2f TB/s)" % (
                    k, NAMES[k], 1e3 * m, 1e3 * min(t[k]), 100 * m / base, fl / m / 1e9, nb / m / 1e9))


if __name__ == "__main__":
    main()
