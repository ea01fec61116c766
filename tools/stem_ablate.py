#!/usr/bin/env python3
"""What bounds the fused front: builds stem_pool.hip with -DSTEM_ABL=<bits> (one .so per ablation, tools/_trace/) and
times each on 1024 frames of 288 x 288, interleaved rounds in one process.  Ablations skip work (wrong results by
construction): 1 MFMAs, 2 byte -> float conversion (fp32), 4 staging of the next rows, 8 epilogue, 16 ring reads,
32 weight reads, 64 the per-iteration barrier.
    python tools/stem_ablate.py --build-only      # build container
    python tools/stem_ablate.py                   # GPU box"""
import argparse
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ABLS = [0, 1, 2, 4, 8, 16, 32, 64, 12, 14, 30, 62, 126]
NAMES = {0: "full kernel", 1: "no MFMA", 2: "no conversion", 4: "no staging", 8: "no epilogue", 16: "no ring reads", 32: "no weight reads",
         64: "no barrier", 12: "no staging, no epilogue", 14: "MFMA + reads (no conversion, staging, epilogue)",
         30: "MFMA + weight reads only", 62: "MFMA only (+ barrier)", 126: "MFMA only, no barrier"}


def so_path(abl):
    return os.path.join(ROOT, "tools", "_trace", "libstem_abl%d.so" % abl)


def build(abls):
    os.makedirs(os.path.join(ROOT, "tools", "_trace"), exist_ok=True)
    srcs = [os.path.join(ROOT, "cadre_amd", "csrc", f) for f in ("stem_pool.hip", "cadre_kernels.hip")]
    procs = []
    for abl in abls:
        procs.append(subprocess.Popen(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC",
                                       "-DSTEM_ABL=%d" % abl, "-o", so_path(abl)] + srcs, stderr=subprocess.DEVNULL))
        if len(procs) >= 7:
            for p in procs:
                assert p.wait() == 0
            procs = []
    for p in procs:
        assert p.wait() == 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--build-only", action="store_true")
    ap.add_argument("--only", type=int, nargs="*", default=None)
    ap.add_argument("--frames", type=int, default=1024)
    ap.add_argument("--rounds", type=int, default=5)
    args = ap.parse_args()
    abls = ABLS if args.only is None else [0] + [a for a in args.only if a]
    if args.build_only:
        return build(abls)
    import torch
    from cadre_amd.encoder import _stem_taps
    vp, i32, i64 = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64
    F, H, W = args.frames, 288, 288
    libs = {}
    for key in abls:
        path = so_path(key)
        if not os.path.exists(path):
            continue
        L = ctypes.CDLL(path)
        L.cadre_stem_pool.argtypes = [vp, vp, vp, vp, vp, i32, i32, i32, i32, i64, i64, i32, i64, vp]
        libs[key] = L
    x = torch.randint(0, 2 ** 31 - 1, (F, H, W), dtype=torch.int32, device="cuda")
    w = torch.randn(64, 4, 7, 7) * 0.05
    sc, sh = (0.5 + torch.rand(64)).cuda(), torch.randn(64).cuda()
    for bf in (0, 1):
        wt = _stem_taps(w, 56 if bf else 50, row8=bool(bf)).cuda().to(torch.bfloat16 if bf else torch.float32)
        out = torch.empty(F, 72, 72, 64, device="cuda", dtype=torch.bfloat16 if bf else torch.float32)
        a = (x.data_ptr(), wt.data_ptr(), None if bf else sc.data_ptr(), sh.data_ptr(), out.data_ptr(), F, H, W, bf, 72 * 72 * 64, 72 * 64, 64, 0, None)
        t = {k: [] for k in libs}
        for k, L in libs.items():
            for _ in range(2):
                assert L.cadre_stem_pool(*a) == 0
        torch.cuda.synchronize()
        for _ in range(args.rounds):
            for k, L in libs.items():
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(3):
                    L.cadre_stem_pool(*a)
                e1.record()
                torch.cuda.synchronize()
                t[k].append(e0.elapsed_time(e1) / 3)
        base = sorted(t[0])[len(t[0]) // 2]
        print("%s, %d frames of %dx%d (median of %d rounds x 3 launches):" % ("bf16" if bf else "fp32", F, H, W, args.rounds))
        for k in libs:
            m = sorted(t[k])[len(t[k]) // 2]
            print("  %-4s %-52s %7.3f ms  (%+6.1f %%)" % (k, NAMES.get(k, ""), m, 100.0 * (m - base) / base))


if __name__ == "__main__":
    main()
