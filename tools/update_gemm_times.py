#!/usr/bin/env python3
"""Per-launch GEMM times of ONE PPO minibatch update (minibatch 64, graphs off so that every launch
is bracketed by HIP events): which shapes carry the update's GEMM time."""
import os
import sys

os.environ["CADRE_HIP_GRAPHS"] = "0"
import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cadre_amd import hip, synth  # noqa: E402
from tests.test_learner_gpu import make_agent  # noqa: E402


def main():
    B, S = 64, 8
    agent = make_agent(84, 84)
    r = np.random.RandomState(5)
    ds = []
    for hd, K in (("steer", 33), ("throttle", 3)):
        tup = (torch.from_numpy((r.standard_normal((S * B, 530)) * 0.5).astype(np.float32)),
               torch.from_numpy(r.randint(0, K, (B, 1)).astype(np.int64)),
               torch.from_numpy((0.3 * r.standard_normal((B, 1))).astype(np.float32)),
               torch.from_numpy(r.standard_normal((B, 1)).astype(np.float32)), torch.ones(B, 1),
               torch.from_numpy((-np.log(K) + 0.2 * r.standard_normal((B, 1))).astype(np.float32)),
               torch.from_numpy(r.standard_normal((B, 1)).astype(np.float32)),
               [torch.from_numpy((0.1 * r.standard_normal((B, 530))).astype(np.float32)),
                torch.from_numpy((0.1 * r.standard_normal((B, 530))).astype(np.float32))],
               torch.from_numpy(r.randint(0, 4, (B, 1)).astype(np.int32)))
        ds.append(tuple(x.cuda() if not isinstance(x, list) else [y.cuda() for y in x] for x in tup))
    for _ in range(3):
        agent.update_policy(ds[0], ds[1])
    torch.cuda.synchronize()
    agg = {}
    for rep in range(5):
        hip.PROFILE = prof = []
        agent.update_policy(ds[0], ds[1])
        torch.cuda.synchronize()
        hip.PROFILE = None
        for key, flops, e0, e1, shape in prof:
            a = agg.setdefault((key, shape), [0, 0.0, flops])
            a[0] += 1; a[1] += e0.elapsed_time(e1) * 1e3
    tot = 0.0
    print("%-14s %-34s %6s %9s %9s %8s" % ("tile/am/bm", "M,N,K,batch,splitk,seg", "calls", "us/call", "us/step", "TF"))
    for (key, shape), (n, us, fl) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print("%-14s %-34s %6d %9.1f %9.1f %8.1f" % (key, shape, n // 5, us / n, us / 5, fl / (us / n) / 1e6))
        tot += us / 5
    print("GEMM total per update: %.1f us" % tot)


if __name__ == "__main__":
    main()
