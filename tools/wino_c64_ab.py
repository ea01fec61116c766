#!/usr/bin/env python3
"""Same-box A/B of builds of the fused layer-1 Winograd kernel: python tools/wino_c64_ab.py old new ... times
tools/_trace/libw64_<name>.so interleaved on 1024 frames of 72 x 72 x 64 and checks that the outputs are bit-identical."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from cadre_amd.encoder import _winograd_u_c64
names = sys.argv[1:] or ["old", "new"]
vp, i32 = ctypes.c_void_p, ctypes.c_int32
F, H, W = 1024, 72, 72
torch.manual_seed(0)
x = torch.randn(F, H, W, 64, device="cuda"); res = torch.randn(F, H, W, 64, device="cuda")
u = _winograd_u_c64(torch.randn(64, 64, 3, 3) / 24).cuda(); sc = torch.rand(64, device="cuda") + 0.5; sh = torch.randn(64, device="cuda")
libs = {}
for n in names:
    L = ctypes.CDLL(os.path.join(ROOT, "tools", "_trace", "libw64_%s.so" % n)); L.cadre_winograd_c64.argtypes = [vp] * 6 + [i32] * 4 + [vp]; libs[n] = L
for use_res in (0, 1):
    outs = {n: torch.empty_like(x) for n in names}
    args = {n: (x.data_ptr(), u.data_ptr(), sc.data_ptr(), sh.data_ptr(), res.data_ptr() if use_res else None, outs[n].data_ptr(), F, H, W, 1, None) for n in names}
    for n, L in libs.items():
        assert L.cadre_winograd_c64(*args[n]) == 0
    torch.cuda.synchronize()
    t = {n: [] for n in names}
    for _ in range(7):
        for n, L in libs.items():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(2):
                L.cadre_winograd_c64(*args[n])
            e1.record(); torch.cuda.synchronize()
            t[n].append(e0.elapsed_time(e1) / 2)
    print("fused F(2x2) 64->64, %d frames of %dx%d, resid=%d:" % (F, H, W, use_res))
    for n in names:
        same = bool((outs[n] == outs[names[0]]).all())
        print("  %-12s %7.3f ms (min %7.3f)  bit-identical to %s: %s" % (n, sorted(t[n])[3], min(t[n]), names[0], same))
