#!/usr/bin/env python3
"""RCCL on the real device at world_size 1 (one GPU per box here): init the `nccl` backend, run the
gradient hand-off `Shared_grad_buffers.add_gradient` -> `chief_step` (which all-reduces the gradient arena
once per optimiser step) and compare with the same steps without torch.distributed.  Own process so that
the process group never leaks into the pytest process (tests/test_topology_gpu.py launches it)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def run(steps, use_dist):
    import numpy as np
    import torch
    from ppo_agent.chief import chief_step
    from ppo_agent.models import Shared_grad_buffers
    from tests.test_learner_gpu import make_agent
    agent = make_agent(84, 84)
    shared = Shared_grad_buffers(agent.model_dict, agent.device)
    r = np.random.RandomState(3)
    B = 16
    samp = []
    for K in (33, 3):
        samp.append((torch.from_numpy((r.standard_normal((8 * B, 530)) * 0.5).astype(np.float32)).cuda(),
                     torch.from_numpy(r.randint(0, K, (B, 1))).cuda(),
                     torch.from_numpy((0.3 * r.standard_normal((B, 1))).astype(np.float32)).cuda(),
                     torch.from_numpy(r.standard_normal((B, 1)).astype(np.float32)).cuda(), torch.ones(B, 1).cuda(),
                     torch.from_numpy((-np.log(K) + 0.2 * r.standard_normal((B, 1))).astype(np.float32)).cuda(),
                     torch.from_numpy(r.standard_normal((B, 1)).astype(np.float32)).cuda(),
                     [torch.zeros(B, 530).cuda(), torch.zeros(B, 530).cuda()],
                     torch.from_numpy(r.randint(0, 4, (B, 1)).astype(np.int32)).cuda()))
    losses = []
    for _ in range(steps):
        losses.append(agent.update_policy(samp[0], samp[1]))
        shared.add_gradient(agent.model_dict)
        chief_step(shared, None, 250.0)
    torch.cuda.synchronize()
    return agent.arena.params.clone(), losses, getattr(shared, "n_allreduce", 0)


def main():
    import torch
    import torch.distributed as dist
    steps = 3
    p0, l0, n0 = run(steps, False)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    os.environ["CADRE_BENCH_FORCE_DIST"] = "1"          # world_size 1: still issue the RCCL all-reduce
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        t = torch.ones(4, device="cuda")
        dist.all_reduce(t)
        p1, l1, n1 = run(steps, True)
        bt = p1.clone()
        dist.broadcast(bt, 0)                            # the startup broadcast of bench.py
        res = dict(backend=dist.get_backend(), world=dist.get_world_size(), allreduce_calls=n1, allreduce_calls_nodist=n0,
                   params_equal=bool(torch.equal(p0, p1)), losses_equal=l0 == l1, warmup_sum=float(t.sum()),
                   broadcast_equal=bool(torch.equal(bt, p1)), moved=float((p1 - p0).abs().max()),
                   rccl_version=".".join(str(v) for v in torch.cuda.nccl.version()))
    finally:
        dist.destroy_process_group()
    print("RCCL_RESULT " + json.dumps(res), flush=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
