#!/usr/bin/env python3
"""Per-GEMM-launch times of one encoder pass (C2 shapes, F frames), back-to-back vs after an idle gap:
separates kernel quality from clock/power-state ramp effects seen inside bench.py."""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cadre_amd import hip, synth  # noqa: E402
from cadre_amd.encoder import DANetEncoderHIP  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=1024)
    ap.add_argument("--gap-ms", type=float, default=0.0)
    ap.add_argument("--passes", type=int, default=4)
    args = ap.parse_args()
    H, W, F = 144, 256, args.frames
    enc = DANetEncoderHIP(synth.encoder_state(*synth.feat_hw(H, W), 7), H, W, "cuda:0", max_frames=F)
    x = torch.rand(F, H, W, 4, device="cuda")
    x[..., 3] = 0
    for _ in range(2):
        enc.forward_nhwc(x)
    torch.cuda.synchronize()
    rows = []
    for it in range(args.passes):
        if args.gap_ms:
            time.sleep(args.gap_ms * 1e-3)
        hip.PROFILE = prof = []
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        enc.forward_nhwc(x)
        e1.record()
        torch.cuda.synchronize()
        hip.PROFILE = None
        rows.append(([a.elapsed_time(b) for _, _, a, b, _ in prof], [f for _, f, _, _, _ in prof], [k for k, _, _, _, _ in prof], e0.elapsed_time(e1)))
    t = np.array([r[0] for r in rows])
    fl = np.array(rows[0][1])
    print("pass totals ms:", ["%.2f" % r[3] for r in rows])
    for i in range(t.shape[1]):
        print("launch %2d key %s  %8.3f ms (min %.3f max %.3f)  %6.1f TF" % (i, rows[0][2][i], t[:, i].mean(), t[:, i].min(), t[:, i].max(),
                                                                       fl[i] / t[:, i].mean() / 1e9))


if __name__ == "__main__":
    main()
