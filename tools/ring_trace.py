#!/usr/bin/env python3
"""Where a conv3x3_ring work item spends its time: builds the library with -DRING_TRACE into /tmp (the kernel then
stamps the shader clock at six points of every item, wave 0 of each workgroup) and prints the mean cycles of
  wait  item start -> first publish wait done      barrier0 -> through the first barrier
  kloop first barrier -> last k-tile done          barrier1 -> through the pre-epilogue barrier
  epilogue -> its last store issued
for one layer shape.  Run on the GPU box: python tools/ring_trace.py --dtype bf16 --shape 1024 72 72 64 64 --resid 1"""
import argparse
import ctypes
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--shape", type=int, nargs=5, default=(1024, 72, 72, 64, 64), metavar=("F", "H", "W", "CIN", "N"))
    ap.add_argument("--resid", type=int, default=0)
    ap.add_argument("--build-only", action="store_true")
    ap.add_argument("--mode", type=int, default=1, help="1: sections of an item; 2: inside k-tiles 4 and 5 of the first chunk; 3: inside a ping-pong staging slot")
    args = ap.parse_args()
    so = os.path.join(ROOT, "tools", "_trace", "libcadre_trace%d.so" % args.mode)    # (git-ignored; build before gpurun: --build-only)
    if args.build_only or not os.path.exists(so):
        os.makedirs(os.path.dirname(so), exist_ok=True)
        srcs = [os.path.join(ROOT, "cadre_amd", "csrc", f) for f in ("conv3x3_ring.hip", "cadre_kernels.hip")]
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-DRING_TRACE=%d" % args.mode,
                               "-o", so] + srcs)
        if args.build_only:
            return
    import torch
    lib = ctypes.CDLL(so)
    from cadre_amd.encoder import _ring_w
    F, H, W, Cin, N = args.shape
    bf = args.dtype == "bf16"
    td = torch.bfloat16 if bf else torch.float32
    x = torch.randn(F, H, W, Cin, device="cuda").to(td)
    w = (torch.randn(N, Cin, 3, 3) * 0.05)
    wr = _ring_w(w, 64 if bf else 32).to(td).cuda()
    sc, sh = torch.ones(N, device="cuda"), torch.zeros(N, device="cuda")
    res = torch.randn(F, H, W, N, device="cuda").to(td) if args.resid else None
    out = torch.empty(F, H, W, N, device="cuda", dtype=td)
    trace = torch.zeros(512 * 64 * 8, dtype=torch.int64, device="cuda")
    vp = ctypes.c_void_p
    lib.cadre_conv3x3_ring.argtypes = [vp] * 6 + [ctypes.c_int32] * 7 + [vp]
    flags = (1 | 2 | (4 if res is not None else 0)) if bf else 0
    a = (x.data_ptr(), wr.data_ptr(), sc.data_ptr(), sh.data_ptr(), res.data_ptr() if res is not None else None, out.data_ptr(),
         F, H, W, Cin, N, 1, flags, None)
    for _ in range(3 if args.mode != 4 else 400):            # (mode 4: the clock the chip holds after ~0.5 s of back-to-back launches)
        assert lib.cadre_conv3x3_ring(*a) == 0
    torch.cuda.synchronize()
    lib.cadre_ring_set_trace(vp(trace.data_ptr()))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    assert lib.cadre_conv3x3_ring(*a) == 0
    e1.record()
    torch.cuda.synchronize()
    if args.mode == 4:                                       # in-kernel clock of the untouched schedule
        c = trace.cpu().numpy()[:1024].reshape(256, 4)
        c = c[c[:, 3] != 0]
        clk = (c[:, 2] - c[:, 0]) / np.maximum(1, (c[:, 3] - c[:, 1])) * 100e6
        print("%s F=%d %dx%d %d->%d resid=%d G=%s: %.3f ms; in-kernel clock median %.3f GHz (min %.3f, max %.3f) over %d workgroups; "
              "cycles per workgroup median %.0f" % (args.dtype, F, H, W, Cin, N, args.resid, os.environ.get("CADRE_RING_G", "auto"), e0.elapsed_time(e1),
                                                    np.median(clk) / 1e9, clk.min() / 1e9, clk.max() / 1e9, len(clk), np.median(c[:, 2] - c[:, 0])))
        return
    t = trace.cpu().numpy().reshape(512, 64, 8)
    if args.mode == 1 and t[:, :, 7].any():                  # G-k-tiles-per-slot kernel (conv3x3_ring_pp2_kernel): 8 stamps per item
        for gname, sel in (("group 0 (wave 0)", slice(0, None, 2)), ("group 1 (wave 4)", slice(1, None, 2))):
            tt = t[sel]
            live = tt[:, :, 7] != 0
            live[:, 0] = False
            d = np.diff(tt, axis=2)[live]
            item = (tt[:, 1:, 0] - tt[:, :-1, 0])[live[:, 1:] & (tt[:, :-1, 0] != 0)]
            print("%s %s F=%d %dx%d %d->%d resid=%d: %.3f ms, %d items (pp2)" % (gname, args.dtype, F, H, W, Cin, N, args.resid, e0.elapsed_time(e1), d.shape[0]))
            for i, n in enumerate(["epilogue+clear", "rest of R(0), M(0)", "R(1): reads + DMA issue", "R(1): lgkm wait + barrier", "M(1): MFMAs + reads",
                                   "M(1): vmcnt wait", "M(1): barrier"]):
                print("  %-28s mean %7.0f  median %7.0f  p90 %7.0f" % (n, d[:, i].mean(), np.median(d[:, i]), np.percentile(d[:, i], 90)))
            print("  %-28s mean %7.0f" % ("item", item.mean()))
        return
    if args.mode == 1 and t[:, :, 6].any():                  # ping-pong kernel: wave 0 (group 0) of even workgroups, wave 4 (group 1) of odd
        for gname, sel in (("group 0 (wave 0)", slice(0, None, 2)), ("group 1 (wave 4)", slice(1, None, 2))):
            tt = t[sel]
            live = tt[:, :, 6] != 0
            live[:, 0] = False
            d = np.diff(tt[:, :, :7], axis=2)[live]
            item = (tt[:, 1:, 0] - tt[:, :-1, 0])[live[:, 1:] & (tt[:, :-1, 0] != 0)]
            print("%s %s F=%d %dx%d %d->%d resid=%d: %.3f ms, %d items" % (gname, args.dtype, F, H, W, Cin, N, args.resid, e0.elapsed_time(e1), d.shape[0]))
            for i, n in enumerate(["epilogue+clear", "k-tiles 0..3", "R(4): issue+reads+wait", "barrier A", "M(4)", "barrier B"]):
                print("  %-24s mean %7.0f  median %7.0f  p90 %7.0f" % (n, d[:, i].mean(), np.median(d[:, i]), np.percentile(d[:, i], 90)))
            print("  %-24s mean %7.0f" % ("item", item.mean()))
        return
    if args.mode == 3:
        for gname, sel in (("group 0 (wave 0)", slice(0, None, 2)), ("group 1 (wave 4)", slice(1, None, 2))):
            tt = t[sel]
            live = tt[:, :, 3] != 0
            live[:, 0] = False
            d = np.diff(tt[:, :, :4], axis=2)[live]
            print("%s %s F=%d %dx%d %d->%d resid=%d: staging slot of k-tile 4, %d items" % (gname, args.dtype, F, H, W, Cin, N, args.resid, d.shape[0]))
            for i, n in enumerate(["issue reads + DMA", "wait vmcnt", "wait lgkmcnt(0)"]):
                print("  %-20s mean %7.0f  median %7.0f  p90 %7.0f" % (n, d[:, i].mean(), np.median(d[:, i]), np.percentile(d[:, i], 90)))
        return
    if args.mode == 2:
        live = (t[:, :, 6] != 0)
        live[:, 0] = False
        d = np.diff(t[:, :, :7], axis=2)[live]
        print("%s F=%d %dx%d %d->%d resid=%d: %.3f ms, %d items traced (wave 0, k-tile 4 then 5 of chunk 0)" % (
            args.dtype, F, H, W, Cin, N, args.resid, e0.elapsed_time(e1), d.shape[0]))
        for i, n in enumerate(["wait vmcnt", "barrier", "issue DMA", "reads+MFMA", "wait vmcnt'", "barrier'"]):
            print("  %-12s mean %7.0f  median %7.0f  p90 %7.0f" % (n, d[:, i].mean(), np.median(d[:, i]), np.percentile(d[:, i], 90)))
        return
    live = t[:, :, 5] != 0
    live[:, 0] = False                                    # first item of a workgroup: cold start
    d = np.diff(t[:, :, :6], axis=2)[live]                # [n, 5]
    nxt = (t[:, 1:, 0] - t[:, :-1, 5])[live[:, 1:] & (t[:, :-1, 5] != 0)]
    per_item = (t[:, 1:, 0] - t[:, :-1, 0])[live[:, 1:] & (t[:, :-1, 0] != 0)]
    names = ["setup->wait", "barrier0", "kloop", "barrier1", "epilogue"]
    print("%s F=%d %dx%d %d->%d resid=%d: %.3f ms, %d items traced" % (args.dtype, F, H, W, Cin, N, args.resid, e0.elapsed_time(e1), d.shape[0]))
    for i, n in enumerate(names):
        print("  %-12s mean %8.0f  median %8.0f  p90 %8.0f  (s_memtime ticks)" % (n, d[:, i].mean(), np.median(d[:, i]), np.percentile(d[:, i], 90)))
    print("  %-12s mean %8.0f" % ("tail->next", nxt.mean()))
    print("  %-12s mean %8.0f  median %8.0f" % ("item", per_item.mean(), np.median(per_item)))


if __name__ == "__main__":
    main()
