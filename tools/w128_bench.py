#!/usr/bin/env python3
"""cadre_conv3x3_w128 (128 x 128 wave tile, weights streamed to registers) against the ping-pong window kernel on the stride-1
3x3 layers of the bf16 trunk and head, same box, interleaved rounds (A/B library: CADRE_BUILD_AB=1 python -m cadre_amd.build).   python tools/w128_bench.py [frames]"""
import sys
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("CADRE_HIP_LIB", os.path.join(ROOT, "cadre_amd", "csrc", "libcadre_hip_ab.so"))      # (the kernel lives in the A/B build)
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cadre_amd import hip
from cadre_amd.encoder import _w128_w, _ring_w


def main():
    Fr = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
    L = hip.lib()
    shapes = [("layer2 128->128 @36", 36, 128, 128), ("layer3 256->256 @18", 18, 256, 256), ("layer4 512->512 @9", 9, 512, 512),
              ("conv5a 512->128 @9", 9, 512, 128), ("conv51 128->128 @9", 9, 128, 128)]
    for name, H, Cin, N in shapes:
        g = torch.Generator(device="cuda").manual_seed(H)
        x = torch.randn(Fr, H, H, Cin, device="cuda", generator=g).to(torch.bfloat16)
        r = torch.randn(Fr, H, H, N, device="cuda", generator=g).to(torch.bfloat16)
        w = (torch.randn(N, Cin, 3, 3, device="cuda", generator=g) / np.sqrt(9 * Cin))
        sh = torch.randn(N, device="cuda", generator=g)
        wf = _w128_w(w.float().cpu()).to(torch.bfloat16).cuda()
        wr = _ring_w(w.float().cpu(), 64).to(torch.bfloat16).cuda()
        o1 = torch.empty(Fr, H, H, N, device="cuda", dtype=torch.bfloat16)
        o2 = torch.empty_like(o1)
        flops = 2.0 * Fr * H * H * N * 9 * Cin
        for res in (False, True):
            def f_w():
                hip.conv3x3_w128(x, wf, sh, r if res else None, o1, Fr, H, H, Cin, N, 1)

            def f_r():
                hip.conv3x3_ring(x, wr, None, sh, r if res else None, o2, Fr, H, H, Cin, N, 1)
            t = {"w128": [], "ring_pp": []}
            for f in (f_w, f_r):
                f()
            torch.cuda.synchronize()
            for rnd in range(6):
                for key, f in (("w128", f_w), ("ring_pp", f_r)):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(5):
                        f()
                    e1.record()
                    torch.cuda.synchronize()
                    t[key].append(e0.elapsed_time(e1) / 5)
            d = float((o1.float() - o2.float()).abs().max() / o2.float().abs().max())
            tw, tr = np.median(t["w128"]), np.median(t["ring_pp"])
            print("%-22s F=%d %s  w128 %.3f ms (%.1f TFLOP/s) | ring_pp %.3f ms (%.1f TFLOP/s)   max diff %.2e"
                  % (name, Fr, "resid" if res else "     ", tw, flops / tw / 1e9, tr, flops / tr / 1e9, d), flush=True)


if __name__ == "__main__":
    main()
