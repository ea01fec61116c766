"""Same-box A/B of the fp32 Winograd convs: three launches (cadre_winograd_in -> batched cadre_gemm_f32 -> cadre_winograd_out)
against the fused form (cadre_winograd_in_frag -> cadre_winograd_gemm_out, csrc/winograd_fused.hip), interleaved rounds in one
process, on the encoder's layer shapes at `--frames` frames of 288 x 288.  Prints ms per conv and TFLOP/s of EXECUTED FLOPs."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cadre_amd import hip  # noqa: E402
from cadre_amd.encoder import _winograd_m, _winograd_u, _winograd_u_frag  # noqa: E402

SHAPES = [("layer2", 36, 128, 128, True), ("layer2_nores", 36, 128, 128, False), ("layer3", 18, 256, 256, True), ("layer4", 9, 512, 512, True),
          ("conv5a", 9, 512, 128, False), ("conv51", 9, 128, 128, False)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=1024)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    L = hip.lib()
    F = a.frames
    for name, H, Cin, N, use_res in SHAPES:
        if a.only and a.only not in name:
            continue
        m = 4 if H % 4 == 0 else 3                         # (the tile edges the fused kernels are built for: F(4x4) on 36 x 36, F(3x3) on 18 x 18 / 9 x 9)
        P, T = (m + 2) ** 2, F * (-(-H // m)) ** 2
        g = torch.Generator().manual_seed(1)
        x = torch.randn(F, H, H, Cin, generator=g).cuda()
        w = torch.randn(N, Cin, 3, 3, generator=g) / (9 * Cin) ** 0.5
        sc, sh = (0.5 + torch.rand(N, generator=g)).cuda(), torch.randn(N, generator=g).cuda()
        res = torch.randn(F, H, H, N, generator=g).cuda() if use_res else None
        u, uf = _winograd_u(w, m).cuda(), _winograd_u_frag(w, m).cuda()
        V = torch.empty(P, T, Cin, device="cuda")
        Mx = torch.empty(P, T, N, device="cuda")
        Vf = torch.empty(int(L.cadre_winograd_frag_elems(F, H, H, Cin, m)), device="cuda")
        o1, o2 = torch.empty(F, H, H, N, device="cuda"), torch.empty(F, H, H, N, device="cuda")
        st = hip.stream()

        def unfused():
            hip.check(L.cadre_winograd_in(hip.ptr(x), hip.ptr(V), F, H, H, Cin, m, st), "in")
            hip.gemm(V, u, Mx, T, N, Cin, Cin, Cin, N, batch=P, a_z=(1, P, T * Cin), b_z=(1, P, N * Cin), c_z=(1, P, T * N))
            hip.check(L.cadre_winograd_out(hip.ptr(Mx), hip.ptr(sc), hip.ptr(sh), hip.ptr(res), hip.ptr(o1), F, H, H, N, 1, m, st), "out")

        def fused_in():
            hip.check(L.cadre_winograd_in_frag(hip.ptr(x), hip.ptr(Vf), F, H, H, Cin, m, st), "in_frag")

        def fused_go():
            hip.check(L.cadre_winograd_gemm_out(hip.ptr(Vf), hip.ptr(uf), hip.ptr(sc), hip.ptr(sh), hip.ptr(res), hip.ptr(o2), F, H, H, Cin, N, 1, m, st), "gemm_out")

        def timed(fn, n=3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(n):
                fn()
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / n
        unfused(); fused_in(); fused_go()
        torch.cuda.synchronize()
        err = float((o1 - o2).abs().max() / o1.abs().max())
        tu, ti, tg = [], [], []
        for _ in range(a.rounds):
            tu.append(timed(unfused)); ti.append(timed(fused_in)); tg.append(timed(fused_go))
        fl = 2.0 * P * T * Cin * N
        med = lambda v: sorted(v)[len(v) // 2]
        print("%-16s m=%d T=%d K=%d N=%d: three launches %.3f ms | fused in %.3f + gemm_out %.3f = %.3f ms (%.1f TFLOP/s executed on gemm_out) | max rel diff %.2e"
              % (name, m, T, Cin, N, med(tu), med(ti), med(tg), med(ti) + med(tg), fl / med(tg) / 1e9, err), flush=True)


if __name__ == "__main__":
    main()
