#!/usr/bin/env python3
"""Encoder pass (preprocess + forward) as eager launches vs one hipGraph replay: are the ~10 us between consecutive
large kernels (kernel trace) a property of eager launches?  python tools/enc_graph_ab.py --dtype bf16 --frames 2048"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cadre_amd import synth  # noqa: E402
from cadre_amd.encoder import DANetEncoderHIP  # noqa: E402
from cadre_amd.learner import PPOLearnerHIP  # noqa: E402


def timed(fn, reps):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=1024)
    ap.add_argument("--dtype", default="f32", choices=["f32", "bf16"])
    args = ap.parse_args()
    H = W = 288
    F = args.frames
    enc = DANetEncoderHIP(synth.encoder_state(*synth.feat_hw(H, W), 7), H, W, "cuda:0", max_frames=F, dtype=args.dtype)
    gen = torch.Generator(device="cuda").manual_seed(1)
    rgb = torch.randint(0, 256, (F, H, W, 3), dtype=torch.uint8, device="cuda", generator=gen)
    route = ((torch.rand(F, W, H, device="cuda", generator=gen) < 0.15) * 255).to(torch.uint8)
    out = torch.zeros(F, 512, device="cuda")

    def run():
        enc.forward_nhwc(enc.preprocess(rgb, route), out)
    for _ in range(3):
        run()
    ref = out.clone()
    t_e = [timed(run, 5) for _ in range(3)]
    g = PPOLearnerHIP._capture(run)
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, ref)
    t_g = [timed(g.replay, 5) for _ in range(3)]
    t_e2 = [timed(run, 5) for _ in range(3)]
    print("%s F=%d: eager %s ms, hipGraph %s ms, eager again %s ms" % (args.dtype, F, ["%.3f" % t for t in t_e], ["%.3f" % t for t in t_g],
                                                                      ["%.3f" % t for t in t_e2]))


if __name__ == "__main__":
    main()
