#!/usr/bin/env python3
"""Timing of the fused LSTM step kernels (ppo_update.hip) alone, in the update's layout: 8 nets, rows sorted by command
(B = 64: 9-22 rows per net, B = 256: 49-79; uneven runs), 20 launches back to back per kernel (HIP events on the launch stream)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cadre_amd import hip  # noqa: E402


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def main():
    L = hip.lib()
    Z, S, C, D, DP, H4, H4P = 8, 8, 4, 530, 544, 2120, 2176
    g = torch.Generator(device="cuda").manual_seed(1)
    sL = 2 * H4 * DP + 2 * H4
    params = torch.randn(Z * sL, device="cuda", generator=g) * 0.04
    grads = torch.zeros(Z * sL, device="cuda")
    NP = 34 * 4 * 34 * 256
    packed = torch.zeros(2, Z, NP, device="cuda")
    st = hip.stream()
    for B in (64, 256):
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
        X = torch.randn(2, S, B, DP, device="cuda", generator=g)
        dC = torch.zeros(Z, B, DP, device="cuda")
        dGp = torch.randn(2, Z, (B + 15) // 16, 16 * H4P, device="cuda", generator=g) * 0.3
        t = [0]

        def fwd():
            k = t[0] % S; t[0] += 1
            hip.check(L.cadre_lstm_step_fwd(packed[0].data_ptr(), NP, params.data_ptr() + 4 * (2 * H4 * DP + H4), sL,
                                            G[:, k].data_ptr(), H4P, S * B * H4P, Hs[:, k].data_ptr(), Cs[:, k].data_ptr(),
                                            Hs[:, k + 1].data_ptr(), Cs[:, k + 1].data_ptr(), TC[:, k + 1].data_ptr(), DP,
                                            (S + 1) * B * DP, B, D, Z, seg.data_ptr(), k & 1, st), "f")

        sync = torch.zeros(Z * S + 1, dtype=torch.int32, device="cuda")

        def seq():
            hip.check(L.cadre_lstm_seq_fwd(packed[0].data_ptr(), NP, params.data_ptr() + 4 * (2 * H4 * DP + H4), sL, G.data_ptr(), H4P,
                                           S * B * H4P, Hs.data_ptr(), Cs.data_ptr(), TC.data_ptr(), DP, (S + 1) * B * DP, B, D, S, Z,
                                           seg.data_ptr(), sync.data_ptr(), st), "q")

        def bwd():
            k = 1 + t[0] % (S - 1); t[0] += 1
            hip.check(L.cadre_lstm_step_bwd(packed[1].data_ptr(), NP, dGp[k & 1].data_ptr(), dGp[(k - 1) & 1].data_ptr(), dGp.stride(1),
                                            dG[:, k - 1].data_ptr(), G[:, k - 1].data_ptr(), H4P, S * B * H4P, None, dC.data_ptr(), B * DP, TC[:, k].data_ptr(),
                                            Cs[:, k - 1].data_ptr(), DP, (S + 1) * B * DP, B, D, Z, cmds.data_ptr(), C,
                                            seg.data_ptr(), k & 1, st), "b")

        def dw():
            hip.check(L.cadre_lstm_dw(dG.data_ptr(), H4P, S * B * H4P, Hs.data_ptr(), X.data_ptr(), DP, (S + 1) * B * DP, S * B * DP,
                                      C, grads.data_ptr() + 4 * H4 * DP, grads.data_ptr(), grads.data_ptr() + 4 * 2 * H4 * DP,
                                      grads.data_ptr() + 4 * (2 * H4 * DP + H4), DP, sL, B, S, H4, DP, Z, seg.data_ptr(), st), "d")

        def tr():
            hip.check(L.cadre_pack_lstm_weights(params.data_ptr() + 4 * H4 * DP, sL, DP, D, Z, packed[0].data_ptr(), packed[1].data_ptr(), NP, st), "t")
        tr()
        has_seq = hasattr(L, "cadre_lstm_seq_fwd")          # (the persistent recurrence is an A/B-build kernel: CADRE_HIP_LIB=.../libcadre_hip_ab.so)
        print("B=%d: lstm_step_fwd %.1f us, lstm_seq_fwd (8 steps) %.1f us, lstm_step_bwd %.1f us, lstm_dw %.1f us, pack_weights %.1f us, "
              "seq status %d" % (B, timeit(fwd), timeit(seq, 10) if has_seq else float("nan"), timeit(bwd), timeit(dw, 10), timeit(tr, 10),
                                 int(sync[-1])), flush=True)


if __name__ == "__main__":
    main()
