#!/usr/bin/env python3
"""A/B builds and ablations of the fused 64-channel Winograd kernel (csrc/winograd_c64.hip): one .so per variant into tools/_trace/,
timed interleaved in one process at the trunk's layer-1 shape (F frames of 72 x 72 x 64, without / with residual).
    python tools/w2_ablate.py --build-only --variants 'base:;notransform:W2_ABL=64'     # build container
    python tools/w2_ablate.py --variants 'base:;notransform:W2_ABL=64'                  # GPU box"""
import argparse
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def so_path(name):
    return os.path.join(ROOT, "tools", "_trace", "libw2_%s.so" % name)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=1024)
    ap.add_argument("--hw", type=int, default=72)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--build-only", action="store_true")
    ap.add_argument("--variants", default="base:;notransform:W2_ABL=64;mfma_only:W2_ABL=30;mfma_only_notransform:W2_ABL=94")
    args = ap.parse_args()
    vs = [v.split(":") for v in args.variants.split(";")]
    if args.build_only:
        os.makedirs(os.path.join(ROOT, "tools", "_trace"), exist_ok=True)
        srcs = [os.path.join(ROOT, "cadre_amd", "csrc", f) for f in ("winograd_c64.hip", "cadre_kernels.hip")]
        procs = [subprocess.Popen(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-mllvm", "-enable-misched=0",
                                   "-mllvm", "-pragma-unroll-threshold=262144"] + ["-D" + d for d in (v[1].split(",") if len(v) > 1 and v[1] else [])]
                                  + ["-o", so_path(v[0])] + srcs, stderr=subprocess.DEVNULL) for v in vs]
        for p in procs:
            assert p.wait() == 0
        return
    import torch
    from cadre_amd.encoder import _winograd_u_c64
    vp, i32 = ctypes.c_void_p, ctypes.c_int32
    libs = {}
    for v in vs:
        L = ctypes.CDLL(so_path(v[0]))
        L.cadre_winograd_c64.argtypes = [vp] * 6 + [i32] * 4 + [vp]
        libs[v[0]] = L
    F, H = args.frames, args.hw
    x = torch.randn(F, H, H, 64, device="cuda")
    res = torch.randn(F, H, H, 64, device="cuda")
    w = torch.randn(64, 64, 3, 3) * 0.05
    U = _winograd_u_c64(w).cuda()
    sc, sh = torch.rand(64, device="cuda") + 0.5, torch.randn(64, device="cuda")
    out = torch.empty_like(x)
    ref = {}
    for with_res in (False, True):
        times = {k: [] for k in libs}
        for rnd in range(args.rounds + 1):
            for k, L in libs.items():
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(4):
                    rc = L.cadre_winograd_c64(x.data_ptr(), U.data_ptr(), sc.data_ptr(), sh.data_ptr(), res.data_ptr() if with_res else None,
                                              out.data_ptr(), F, H, H, 1, None)
                    assert rc == 0
                e1.record(); torch.cuda.synchronize()
                if rnd:
                    times[k].append(e0.elapsed_time(e1) / 4)
                if rnd == 0:
                    if k == vs[0][0]:
                        ref[with_res] = out.clone()
                    else:
                        print("  %-28s %s the first variant" % (k, "bit-identical to" if torch.equal(out, ref[with_res]) else "DIFFERS from"))
        for k in libs:
            t = sorted(times[k])
            print("%s %-28s median %.3f ms  min %.3f" % ("res  " if with_res else "nores", k, t[len(t) // 2], t[0]))


if __name__ == "__main__":
    main()
