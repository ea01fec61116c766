#!/usr/bin/env python3
"""What bounds the fused Winograd kernel (csrc/winograd_fused.hip): builds it with -DWGO_ABL=<bits> (one .so per ablation, into
tools/_trace/) and times cadre_winograd_gemm_out on the trunk shapes, interleaved rounds in one process.  Ablations skip work
(results wrong by construction): 1 MFMAs, 2 fragment reads, 4 DMA after the prologue, 8 epilogue.
    python tools/wgo_ablate.py --build-only        # build container (hipcc cross-compiles)
    python tools/wgo_ablate.py                     # GPU box"""
import argparse
import ctypes
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ABLS = [0, 8, 16, 32, 48, 64, 1, 2, 4, 9, 6, 14, 11, 15]
NAMES = {32: "epilogue without inverse transform / transposes", 48: "epilogue: residual loads + waits only", 64: "no drain before the epilogue", 16: "no stores", 0: "full kernel", 1: "no MFMA", 2: "no fragment reads", 4: "no DMA", 8: "no epilogue", 9: "no MFMA, no epilogue",
         6: "no reads, no DMA", 14: "schedule + MFMA only", 11: "schedule + DMA only", 15: "empty schedule (barriers + waits)"}


def so_path(abl, tag=""):
    return os.path.join(ROOT, "tools", "_trace", "libwgo_abl%d%s.so" % (abl, tag))


def build(extra, tag=""):
    os.makedirs(os.path.join(ROOT, "tools", "_trace"), exist_ok=True)
    srcs = [os.path.join(ROOT, "cadre_amd", "csrc", f) for f in ("winograd_fused.hip", "cadre_kernels.hip")]
    procs = [subprocess.Popen(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-mllvm", "-enable-misched=0",
                               "-mllvm", "-pragma-unroll-threshold=262144", "-DWGO_ABL=%d" % abl] + extra + ["-o", so_path(abl, tag)] + srcs,
                              stderr=subprocess.DEVNULL) for abl in ABLS]
    for p in procs:
        assert p.wait() == 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=1024)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--build-only", action="store_true")
    ap.add_argument("--tag", default="")
    ap.add_argument("--define", action="append", default=[], help="extra -D for the builds")
    ap.add_argument("--shapes", default="36,128,128;18,256,256;9,512,512")
    ap.add_argument("--variants", default="", help="A/B builds instead of the ablations: 'name:DEF=1,DEF2=3;name2:...' (name 'base': no defines)")
    args = ap.parse_args()
    global ABLS
    if args.variants:
        vs = [v.split(":") for v in args.variants.split(";")]
        ABLS = list(range(100, 100 + len(vs)))
        for k, v in zip(ABLS, vs):
            NAMES[k] = v[0]
        if args.build_only:
            os.makedirs(os.path.join(ROOT, "tools", "_trace"), exist_ok=True)
            srcs = [os.path.join(ROOT, "cadre_amd", "csrc", f) for f in ("winograd_fused.hip", "cadre_kernels.hip")]
            procs = [subprocess.Popen(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-mllvm", "-enable-misched=0",
                                       "-mllvm", "-pragma-unroll-threshold=262144"] + ["-D" + d for d in (v[1].split(",") if len(v) > 1 and v[1] else [])]
                                      + ["-o", so_path(k, args.tag)] + srcs, stderr=subprocess.DEVNULL) for k, v in zip(ABLS, vs)]
            for p in procs:
                assert p.wait() == 0
            return
    if args.build_only:
        return build(["-D" + d for d in args.define], args.tag)
    import torch
    from cadre_amd.encoder import _winograd_m, _winograd_u_frag
    vp, i32 = ctypes.c_void_p, ctypes.c_int32
    libs = {}
    for abl in ABLS:
        if os.path.exists(so_path(abl, args.tag)):
            L = ctypes.CDLL(so_path(abl, args.tag))
            L.cadre_winograd_gemm_out.argtypes = [vp] * 6 + [i32] * 7 + [vp]
            L.cadre_winograd_in_frag.argtypes = [vp, vp] + [i32] * 5 + [vp]
            L.cadre_winograd_frag_elems.restype = ctypes.c_int64
            L.cadre_winograd_frag_elems.argtypes = [i32] * 5
            libs[abl] = L
    F = args.frames
    for shp in args.shapes.split(";"):
        H, Cin, N = (int(v) for v in shp.split(","))
        m = 4 if H % 4 == 0 else 3                         # (the tile edges the fused kernels are built for: F(4x4) on 36 x 36, F(3x3) on 18 x 18 / 9 x 9)
        P, T = (m + 2) ** 2, F * (-(-H // m)) ** 2
        x = torch.randn(F, H, H, Cin, device="cuda")
        uf = _winograd_u_frag(torch.randn(N, Cin, 3, 3) * 0.05, m).cuda()
        sc, sh = torch.ones(N, device="cuda"), torch.zeros(N, device="cuda")
        res = torch.randn(F, H, H, N, device="cuda")
        out = torch.empty(F, H, H, N, device="cuda")
        L0 = libs[ABLS[0]]
        Vf = torch.empty(int(L0.cadre_winograd_frag_elems(F, H, H, Cin, m)), device="cuda")
        assert L0.cadre_winograd_in_frag(x.data_ptr(), Vf.data_ptr(), F, H, H, Cin, m, None) == 0
        a = (Vf.data_ptr(), uf.data_ptr(), sc.data_ptr(), sh.data_ptr(), res.data_ptr(), out.data_ptr(), F, H, H, Cin, N, 1, m, None)
        t = {k: [] for k in libs}
        for L in libs.values():
            for _ in range(2):
                assert L.cadre_winograd_gemm_out(*a) == 0
        torch.cuda.synchronize()
        for _ in range(args.rounds):
            for k, L in libs.items():
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(3):
                    assert L.cadre_winograd_gemm_out(*a) == 0
                e1.record()
                torch.cuda.synchronize()
                t[k].append(e0.elapsed_time(e1) / 3)
        # the input transform into fragment order of every build (same process, interleaved)
        tin = {k: [] for k in libs}
        for _ in range(args.rounds):
            for k, L in libs.items():
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(3):
                    assert L.cadre_winograd_in_frag(x.data_ptr(), Vf.data_ptr(), F, H, H, Cin, m, None) == 0
                e1.record()
                torch.cuda.synchronize()
                tin[k].append(e0.elapsed_time(e1) / 3)
        print("   cadre_winograd_in_frag: " + "  ".join("%s %.3f ms" % (NAMES.get(k, str(k)), np.median(tin[k])) for k in libs))
        fl = 2.0 * P * T * Cin * N
        print("fp32 Winograd F(%dx%d) gemm_out F=%d %dx%d %d->%d  (%.0f GFLOP executed, MFMA floor %.3f ms)" % (m, m, F, H, H, Cin, N, fl / 1e9, fl / 157.3e9))
        base = np.median(t[ABLS[0]])
        for k in libs:
            md = np.median(t[k])
            print("   %-40s %.3f ms  (%+.0f %%)  %.1f TFLOP/s-equivalent" % (NAMES.get(k, str(k)), md, 100 * (md / base - 1), fl / md / 1e9), flush=True)


if __name__ == "__main__":
    main()
