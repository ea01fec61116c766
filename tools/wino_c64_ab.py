#!/usr/bin/env python3
"""Same-box A/B of builds of the fused layer-1 Winograd kernel: python tools/wino_c64_ab.py old new ... times
tools/_trace/libw64_<name>.so interleaved on 1024 frames of 72 x 72 x 64, checks the outputs bit-identical to the first
and reports the error against torch's conv2d.  name:c16 = a build older than ABI 10 (cout and cin axes of U in natural order), name:c8 = an ABI-10 build (cin axis natural)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from cadre_amd.encoder import _winograd_u_c64
specs = sys.argv[1:] or ["old", "new"]
names = [n.split(":")[0] for n in specs]
natural = {n.split(":")[0] for n in specs if n.endswith(":c16")}
nopairs = {n.split(":")[0] for n in specs if n.endswith(":c8")}
vp, i32 = ctypes.c_void_p, ctypes.c_int32
F, H, W = 1024, 72, 72
torch.manual_seed(0)
x = torch.randn(F, H, W, 64, device="cuda"); res = torch.randn(F, H, W, 64, device="cuda")
w = torch.randn(64, 64, 3, 3) / 24
u_new = _winograd_u_c64(w).cuda()
u_c8 = _winograd_u_c64(w, cin_pairs=False).cuda()
pos = torch.arange(64); inv = torch.empty(64, dtype=torch.long); inv[4 * (pos % 16) + pos // 16] = pos
u_old = u_c8[:, :, inv.cuda(), :].contiguous()
ref_conv = torch.nn.functional.conv2d(x[:8].permute(0, 3, 1, 2), w.cuda(), padding=1).permute(0, 2, 3, 1)
sc = torch.rand(64, device="cuda") + 0.5; sh = torch.randn(64, device="cuda")
libs = {}
for n in names:
    L = ctypes.CDLL(os.path.join(ROOT, "tools", "_trace", "libw64_%s.so" % n)); L.cadre_winograd_c64.argtypes = [vp] * 6 + [i32] * 4 + [vp]; libs[n] = L
for use_res in (0, 1):
    outs = {n: torch.empty_like(x) for n in names}
    args = {n: (x.data_ptr(), (u_old if n in natural else u_c8 if n in nopairs else u_new).data_ptr(), sc.data_ptr(), sh.data_ptr(), res.data_ptr() if use_res else None, outs[n].data_ptr(), F, H, W, 1, None) for n in names}
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
        ref = torch.relu(ref_conv * sc + sh + (res[:8] if use_res else 0))
        err = float((outs[n][:8] - ref).abs().max())
        print("  %-12s %7.3f ms (min %7.3f)  bit-identical to %s: %s   max |err| vs torch conv %.2e" % (n, sorted(t[n])[3], min(t[n]), names[0], same, err))
