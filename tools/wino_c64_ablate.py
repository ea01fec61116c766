#!/usr/bin/env python3
"""What bounds the fused layer-1 Winograd kernel: builds winograd_c64.hip with -DW2_ABL=<bits> (tools/_trace/) and times each on
1024 frames of 72 x 72 x 64.  1 no MFMA, 2 no patch loads, 4 no weight DMA, 8 no epilogue, 16 no wait + barrier."""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ABLS = [0, 2, 4, 8, 16, 6, 14, 30, 32, 40, 44]
NAMES = {0: "full kernel", 1: "no MFMA", 2: "no patch loads", 4: "no weight DMA", 8: "no epilogue", 16: "no wait + barrier", 6: "no loads at all",
         14: "MFMA + LDS reads + sync only", 32: "patch loads from a fixed 64 KB (cache hits)", 40: "cache-hit patches, no epilogue",
         44: "cache-hit patches, no epilogue, no weight DMA", 30: "MFMA + LDS reads only", 31: "LDS reads + transforms only"}
so = lambda a: os.path.join(ROOT, "tools", "_trace", "libw64_abl%d.so" % a)
if "--build-only" in sys.argv:
    os.makedirs(os.path.dirname(so(0)), exist_ok=True)
    srcs = [os.path.join(ROOT, "cadre_amd", "csrc", f) for f in ("winograd_c64.hip", "cadre_kernels.hip")]
    ps = [subprocess.Popen(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-mllvm", "-enable-misched=0", "-mllvm", "-pragma-unroll-threshold=262144", "-DW2_ABL=%d" % a, "-o", so(a)] + srcs,
                           stderr=subprocess.DEVNULL) for a in ABLS]
    assert all(p.wait() == 0 for p in ps)
    sys.exit(0)
import torch
from cadre_amd.encoder import _winograd_u_c64
vp, i32 = ctypes.c_void_p, ctypes.c_int32
F, H, W = 1024, 72, 72
x = torch.randn(F, H, W, 64, device="cuda"); res = torch.randn(F, H, W, 64, device="cuda"); out = torch.empty_like(x)
u = _winograd_u_c64(torch.randn(64, 64, 3, 3) / 24).cuda(); sc = torch.rand(64, device="cuda") + 0.5; sh = torch.randn(64, device="cuda")
libs = {}
for a in ABLS:
    if os.path.exists(so(a)):
        L = ctypes.CDLL(so(a)); L.cadre_winograd_c64.argtypes = [vp] * 6 + [i32] * 4 + [vp]; libs[a] = L
for use_res in (0, 1):
    args = (x.data_ptr(), u.data_ptr(), sc.data_ptr(), sh.data_ptr(), res.data_ptr() if use_res else None, out.data_ptr(), F, H, W, 1, None)
    t = {a: [] for a in libs}
    for L in libs.values():
        assert L.cadre_winograd_c64(*args) == 0
    torch.cuda.synchronize()
    for _ in range(5):
        for a, L in libs.items():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(2):
                L.cadre_winograd_c64(*args)
            e1.record(); torch.cuda.synchronize()
            t[a].append(e0.elapsed_time(e1) / 2)
    print("fused F(2x2) 64->64, %d frames of %dx%d, resid=%d (direct conv: 2.97 ms; MFMA floor 1.15 ms at 2.3 GHz):" % (F, H, W, use_res))
    for a in libs:
        print("  %2d %-34s %7.3f ms" % (a, NAMES[a], sorted(t[a])[2]))
