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
    from ppo_agent.storage import RolloutStorage
    from tests.helpers import fill_storages
    B = int(os.environ.get("B", "64"))
    T = 2 * B
    agent = make_agent(84, 84)
    if os.environ.get("UNSORTED"):
        agent.learner.use_sorted = False
    data = fill_storages(T, 700)
    pair = []
    for hd in ("steer", "throttle"):
        st = RolloutStorage(T, 2, 530, 8, 530, True, 0.99, 0.95)
        for k, v in data[hd].items():
            getattr(st, k).copy_(torch.from_numpy(v))
        st.to("cuda:0")
        st.compute_returns(torch.tensor([0.05]))
        pair.append(st)
    g = torch.Generator().manual_seed(1)
    idx = [torch.randperm(T, generator=g)[:B] for _ in range(2)]
    batches = [(pair[0], idx[0], pair[0].advantages, pair[1], idx[1], pair[1].advantages)]
    print("sorted rows:", agent.learner.sorted_rows(B))

    def run():
        agent.update_policy_from_storages(batches)
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    agg = {}
    for rep in range(5):
        hip.PROFILE = prof = []
        run()
        torch.cuda.synchronize()
        hip.PROFILE = None
        for key, flops, e0, e1, shape, _nbytes in prof:
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
