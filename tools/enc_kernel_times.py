#!/usr/bin/env python3
"""Per-launch times of one encoder pass (F frames, fp32 or bf16), HIP events around every GEMM/conv launch
plus the whole pass: which layer costs what, at which TFLOP/s and GB/s (algorithmic bytes)."""
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
    ap.add_argument("--size", type=int, nargs=2, default=(288, 288))
    ap.add_argument("--dtype", default="f32", choices=["f32", "bf16"])
    ap.add_argument("--gap-ms", type=float, default=0.0)
    ap.add_argument("--passes", type=int, default=4)
    args = ap.parse_args()
    (H, W), F = args.size, args.frames
    enc = DANetEncoderHIP(synth.encoder_state(*synth.feat_hw(H, W), 7), H, W, "cuda:0", max_frames=F, dtype=args.dtype)
    gen = torch.Generator(device="cuda").manual_seed(1)
    rgb = torch.randint(0, 256, (F, H, W, 3), dtype=torch.uint8, device="cuda", generator=gen)
    route = ((torch.rand(F, W, H, device="cuda", generator=gen) < 0.15) * 255).to(torch.uint8)
    out = torch.zeros(F, 512, device="cuda")
    for _ in range(2):
        enc.forward_nhwc(enc.preprocess(rgb, route), out)
    torch.cuda.synchronize()
    rows = []
    for it in range(args.passes):
        if args.gap_ms:
            time.sleep(args.gap_ms * 1e-3)
        hip.PROFILE = prof = []
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        e0.record()
        x = enc.preprocess(rgb, route)
        e1.record()
        enc.forward_nhwc(x, out)
        e2.record()
        torch.cuda.synchronize()
        hip.PROFILE = None
        rows.append(([a.elapsed_time(b) for _, _, a, b, _, _ in prof], prof, e0.elapsed_time(e1), e1.elapsed_time(e2)))
    t = np.array([r[0] for r in rows])
    prof = rows[0][1]
    tot = np.mean([r[3] for r in rows])
    print("%s %dx%d F=%d: preprocess %.3f ms, forward %.3f ms (%.1f TFLOP/s over the encoder), GEMM/conv launches %.3f ms"
          % (args.dtype, H, W, F, np.mean([r[2] for r in rows]), tot, F * enc.flops_per_frame() / tot / 1e9, t.mean(0).sum()))
    for i, (key, fl, _a, _b, shape, nb) in enumerate(prof):
        ms = t[:, i].mean()
        print("launch %2d key %-16s M,N,K=%-22s %8.3f ms (min %.3f)  %7.1f TF  %6.0f GB/s" % (
            i, key, shape[:3], ms, t[:, i].min(), fl / ms / 1e9, nb / ms / 1e6))


if __name__ == "__main__":
    main()
