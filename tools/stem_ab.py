#!/usr/bin/env python3
"""A/B builds of the fused front (csrc/stem_pool.hip) by -D defines: one .so per variant into tools/_trace/, timed interleaved in one
process at F frames of 288 x 288 (fp32).
    python tools/stem_ab.py --build-only --variants 'div255_a:STEM_DIV255_A=1;scale:'      # build container
    python tools/stem_ab.py --variants 'div255_a:STEM_DIV255_A=1;scale:'                   # GPU box"""
import argparse
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def so_path(name):
    return os.path.join(ROOT, "tools", "_trace", "libstem_%s.so" % name)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=1024)
    ap.add_argument("--hw", type=int, default=288)
    ap.add_argument("--rounds", type=int, default=6)
    ap.add_argument("--build-only", action="store_true")
    ap.add_argument("--variants", default="div255_a:STEM_DIV255_A=1;scale:")
    args = ap.parse_args()
    vs = [v.split(":") for v in args.variants.split(";")]
    if args.build_only:
        os.makedirs(os.path.join(ROOT, "tools", "_trace"), exist_ok=True)
        srcs = [os.path.join(ROOT, "cadre_amd", "csrc", f) for f in ("stem_pool.hip", "cadre_kernels.hip")]
        procs = [subprocess.Popen(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC"]
                                  + ["-D" + d for d in (v[1].split(",") if len(v) > 1 and v[1] else [])] + ["-o", so_path(v[0])] + srcs,
                                  stderr=subprocess.DEVNULL) for v in vs]
        for p in procs:
            assert p.wait() == 0
        return
    import torch
    vp, i32, i64 = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64
    libs = {}
    for v in vs:
        L = ctypes.CDLL(so_path(v[0]))
        L.cadre_stem_pool.argtypes = [vp] * 5 + [i32] * 4 + [i64, i64, i32, i64, vp]
        libs[v[0]] = L
    F, H = args.frames, args.hw
    Hp = ((H + 6 - 7) // 2 + 1 + 2 - 3) // 2 + 1
    img = torch.randint(0, 2 ** 31 - 1, (F, H, H), dtype=torch.int32, device="cuda")
    w = torch.randn(64, 50, 4, device="cuda") * 0.05
    w[:, 49] = 0
    sc, sh = torch.rand(64, device="cuda") + 0.5, torch.randn(64, device="cuda") * 0.1
    out = torch.empty(F, Hp, Hp, 64, device="cuda")
    ref = None
    times = {k: [] for k in libs}
    for rnd in range(args.rounds + 1):
        for k, L in libs.items():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                rc = L.cadre_stem_pool(img.data_ptr(), w.data_ptr(), sc.data_ptr(), sh.data_ptr(), out.data_ptr(), F, H, H, 0,
                                       Hp * Hp * 64, Hp * 64, 64, 0, None)
                assert rc == 0, rc
            e1.record(); torch.cuda.synchronize()
            if rnd:
                times[k].append(e0.elapsed_time(e1) / 3)
            elif ref is None:
                ref = out.clone()
            else:
                print("  %-20s max |diff| / max |ref| vs the first variant: %.2e" % (k, float((out - ref).abs().max() / ref.abs().max())))
    for k in libs:
        t = sorted(times[k])
        print("%-20s median %.3f ms  min %.3f" % (k, t[len(t) // 2], t[0]))


if __name__ == "__main__":
    main()
