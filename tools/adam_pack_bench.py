#!/usr/bin/env python3
"""cadre_clip_adam_graph + cadre_pack_lstm_weights (two steps of the optimiser hand-off) against cadre_clip_adam_pack_graph (the
optimiser writes the fragment-order W_hh copies itself), interleaved rounds in one process, on the real arena (19.4 M parameters)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cadre_amd import hip  # noqa: E402
from cadre_amd.arena import PPOArena  # noqa: E402


def main():
    L = hip.lib()
    a = PPOArena("cuda:0", 530, {"steer": 33, "throttle": 3}, 4)
    a.params.normal_(0, 0.05)
    a.grads.normal_(0, 0.3)
    a.ensure_adam()
    n_pack = ((a.D + 15) // 16) * 4 * (a.DP // 16) * 256
    wp = torch.zeros(2, a.Z, n_pack, device="cuda")
    st = hip.stream()

    def adam():
        hip.check(L.cadre_clip_adam_graph(hip.ptr(a.params), hip.ptr(a.grads), hip.ptr(a.exp_avg), hip.ptr(a.exp_avg_sq), hip.ptr(a.seg_off),
                                          2 * a.Z, hip.ptr(a.norms2), 250.0, 3e-4, 0.9, 0.999, 1e-8, hip.ptr(a.step_dev), st), "adam")

    def pack():
        hip.check(L.cadre_pack_lstm_weights(hip.ptr(a.params[a.o_whh:]), a.size_L, a.DP, a.D, a.Z, hip.ptr(wp[0]), hip.ptr(wp[1]), wp.stride(1), st), "pack")

    def fused():
        hip.check(L.cadre_clip_adam_pack_graph(hip.ptr(a.params), hip.ptr(a.grads), hip.ptr(a.exp_avg), hip.ptr(a.exp_avg_sq), hip.ptr(a.seg_off),
                                               2 * a.Z, hip.ptr(a.norms2), 250.0, 3e-4, 0.9, 0.999, 1e-8, hip.ptr(a.step_dev), a.Z, a.size_L,
                                               a.o_whh, a.H4, a.DP, a.D, hip.ptr(wp[0]), hip.ptr(wp[1]), wp.stride(1), st), "fused")

    def timed(fn, n=10):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e3
    for f in (adam, pack, fused):
        f()
    torch.cuda.synchronize()
    ta, tp, tf = [], [], []
    for _ in range(5):
        ta.append(timed(adam)); tp.append(timed(pack)); tf.append(timed(fused))
    med = lambda v: sorted(v)[len(v) // 2]
    print("prep + sqnorm + adam %.1f us | pack %.1f us | sum %.1f us || prep + sqnorm + adam (skip W_hh) + adam_whh_pack %.1f us (launches back to back, HIP events)"
          % (med(ta), med(tp), med(ta) + med(tp), med(tf)))


if __name__ == "__main__":
    main()
