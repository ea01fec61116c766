#!/usr/bin/env python3
"""Where a fused LSTM step launch (ppo_update.hip) spends its time: builds ppo_update.hip with -DLSTM_TRACE into
tools/_trace/ (wave 0 of every workgroup then stamps the shader clock at section boundaries and the 100 MHz wall clock
at entry and exit) and prints, for the forward and the backward step kernel in the update's layout (8 nets, rows sorted
by command), the per-workgroup section cycles and the launch's wall-clock picture: when workgroups start and end
relative to the first start.  Build here (no GPU needed): python tools/lstm_trace.py --build-only; run on the GPU box."""
import argparse
import ctypes
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SO = os.path.join(ROOT, "tools", "_trace", "libcadre_lstm_trace.so")      # (git-ignored; travels with gpurun)


def build():
    os.makedirs(os.path.dirname(SO), exist_ok=True)
    srcs = [os.path.join(ROOT, "cadre_amd", "csrc", f) for f in ("ppo_update.hip", "cadre_kernels.hip")]
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-DLSTM_TRACE", "-o", SO] + srcs)


def report(name, t, ms):
    live = t[:, 10] == 1
    t = t[live]
    d = np.diff(t[:, :7], axis=1)
    names = ["issue loads (+ LDS store fwd)", "barrier (operands landed)", "MFMA loop", "gate tiles -> LDS + barrier", "cell math + stores issued",
             "stores drained"]
    print("%s: %d workgroups, launch %.1f us" % (name, t.shape[0], ms))
    for i, n in enumerate(names):
        print("  %-32s median %6.0f  p90 %6.0f  max %6.0f cycles" % (n, np.median(d[:, i]), np.percentile(d[:, i], 90), d[:, i].max()))
    tot = t[:, 6] - t[:, 0]
    print("  %-32s median %6.0f  p90 %6.0f  max %6.0f cycles" % ("workgroup total", np.median(tot), np.percentile(tot, 90), tot.max()))
    r0 = t[:, 8].min()
    beg, end = (t[:, 8] - r0) / 100.0, (t[:, 9] - r0) / 100.0           # us since the first workgroup started
    print("  wall clock (us since first start): starts median %.2f p90 %.2f max %.2f; ends median %.2f p90 %.2f max %.2f; "
          "lifetime median %.2f max %.2f" % (np.median(beg), np.percentile(beg, 90), beg.max(), np.median(end), np.percentile(end, 90),
                                               end.max(), np.median(end - beg), (end - beg).max()))
    clk = np.median(tot / np.maximum((t[:, 9] - t[:, 8]) / 100.0, 1e-3)) / 1e3
    print("  shader clock during the workgroups: %.2f GHz" % clk)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--build-only", action="store_true")
    ap.add_argument("--B", type=int, nargs="*", default=[64, 256])
    args = ap.parse_args()
    if args.build_only or not os.path.exists(SO):
        build()
        if args.build_only:
            return
    import torch
    L = ctypes.CDLL(SO)
    vp, i32, i64 = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64
    L.cadre_lstm_step_fwd.argtypes = [vp, i64, vp, i64, vp, i32, i64, vp, vp, vp, vp, vp, i32, i64, i32, i32, i32, vp, i32, vp]
    L.cadre_lstm_step_bwd.argtypes = [vp, i64, vp, vp, i64, vp, vp, i32, i64, vp, vp, i64, vp, vp, i32, i64, i32, i32, i32, vp, i32, vp, i32, vp]
    L.cadre_pack_lstm_weights.argtypes = [vp, i64, i32, i32, i32, vp, vp, i64, vp]
    L.cadre_lstm_set_trace.argtypes = [vp]
    Z, S, C, D, DP, H4, H4P = 8, 8, 4, 530, 544, 2120, 2176
    g = torch.Generator(device="cuda").manual_seed(1)
    sL = 2 * H4 * DP + 2 * H4
    params = torch.randn(Z * sL, device="cuda", generator=g) * 0.04
    NP = 34 * 4 * 34 * 256
    packed = torch.zeros(2, Z, NP, device="cuda")
    assert L.cadre_pack_lstm_weights(params.data_ptr() + 4 * H4 * DP, sL, DP, D, Z, packed[0].data_ptr(), packed[1].data_ptr(), NP, None) == 0
    trace = torch.zeros(4096 * 16, dtype=torch.int64, device="cuda")
    for B in args.B:
        # uneven runs, as a sampled minibatch has them (not aligned to 16- or 32-row tiles)
        runs = {64: ([13, 22, 11, 18], [19, 9, 21, 15]), 256: ([70, 58, 49, 79], [61, 66, 72, 57])}[B]
        seg_l, cmd_l = [], []
        for head in runs:
            b0, cm = 0, []
            for c, n in enumerate(head):
                seg_l.append([b0, n]); cm += [c] * n; b0 += n
            cmd_l.append(cm)
        seg = torch.tensor(seg_l, dtype=torch.int32, device="cuda")
        cmds = torch.tensor(cmd_l, dtype=torch.int32, device="cuda")
        G = torch.randn(Z, S, B, H4P, device="cuda", generator=g) * 0.3
        dG = torch.randn(Z, S, B, H4P, device="cuda", generator=g) * 0.3
        Hs = torch.randn(Z, S + 1, B, DP, device="cuda", generator=g) * 0.3
        Cs = torch.randn(Z, S + 1, B, DP, device="cuda", generator=g) * 0.3
        TC = torch.tanh(Cs)
        dC = torch.zeros(Z, B, DP, device="cuda")
        dGp = torch.randn(2, Z, (B + 15) // 16, 16 * H4P, device="cuda", generator=g) * 0.3

        def fwd(k):
            assert L.cadre_lstm_step_fwd(packed[0].data_ptr(), NP, params.data_ptr() + 4 * (2 * H4 * DP + H4), sL,
                                         G[:, k].data_ptr(), H4P, S * B * H4P, Hs[:, k].data_ptr(), Cs[:, k].data_ptr(),
                                         Hs[:, k + 1].data_ptr(), Cs[:, k + 1].data_ptr(), TC[:, k + 1].data_ptr(), DP,
                                         (S + 1) * B * DP, B, D, Z, seg.data_ptr(), k & 1, None) == 0

        def bwd(k):
            assert L.cadre_lstm_step_bwd(packed[1].data_ptr(), NP, dGp[k & 1].data_ptr(), dGp[(k - 1) & 1].data_ptr(), dGp.stride(1),
                                         dG[:, k - 1].data_ptr(), G[:, k - 1].data_ptr(), H4P, S * B * H4P, None, dC.data_ptr(), B * DP,
                                         TC[:, k].data_ptr(), Cs[:, k - 1].data_ptr(), DP, (S + 1) * B * DP, B, D, Z, cmds.data_ptr(), C,
                                         seg.data_ptr(), k & 1, None) == 0

        for name, fn, ks in (("lstm_step_fwd B=%d" % B, fwd, list(range(S))), ("lstm_step_bwd B=%d" % B, bwd, list(range(S - 1, 0, -1)))):
            assert L.cadre_lstm_set_trace(None) == 0
            for _ in range(3):
                for k in ks:
                    fn(k)
            torch.cuda.synchronize()
            # the traced launch is the LAST of a back-to-back sequence (the state the update's graph runs it in)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            for k in ks[:-1]:
                fn(k)
            trace.zero_()
            assert L.cadre_lstm_set_trace(trace.data_ptr()) == 0
            e0.record()
            fn(ks[-1])
            e1.record()
            torch.cuda.synchronize()
            report(name, trace.cpu().numpy().reshape(4096, 16), e0.elapsed_time(e1) * 1e3)


if __name__ == "__main__":
    main()
