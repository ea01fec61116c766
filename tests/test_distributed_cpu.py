"""CPU, world_size 2 over gloo: the data-parallel gradient hand-off of the N>1 path.
`Shared_grad_buffers.add_gradient` + `all_reduce` must SUM (not average — chief.py:18, models.py:237) the flat
gradient arena across ranks, leave every rank with identical sums, and the reference key scheme
('<model>_<param>_grad') must alias the reduced buffer.  (On the GPU box the same call runs
over RCCL; the arena/views/bookkeeping exercised here are device-independent host logic.)"""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from cadre_amd.arena import PPOArena
        from ppo_agent.models import LSTM, Model, Shared_grad_buffers, _no_orthogonal_init
        arena = PPOArena("cpu", 530, {"steer": 33, "throttle": 3}, 4)
        torch.manual_seed(0)                                   # identical init on every rank
        md = {}
        with _no_orthogonal_init():                            # 32 QR factorisations are not under test
            for c in range(4):
                for head, k in (("steer", 33), ("throttle", 3)):
                    md["%s_ppo_%d" % (head, c)] = arena.bind("%s_ppo_%d" % (head, c), Model(530, k))
                    md["%s_lstm_%d" % (head, c)] = arena.bind("%s_lstm_%d" % (head, c), LSTM(530, hid_size=530))
        p0 = arena.params.clone()
        # rank-dependent gradients written through the per-parameter views
        for name, m in md.items():
            for pn, p in m.named_parameters():
                p.grad.fill_(float(rank + 1))
        shared = Shared_grad_buffers(md, torch.device("cpu"))
        shared.add_gradient(md)
        # a second worker agent of this process with its own nets (reference topology, main.py:63-68):
        # accumulated locally; the cross-rank SUM then runs ONCE per optimiser step (chief_step), not per
        # add_gradient — per-call all-reduces would count the first worker's gradients `world` times over
        arena2 = PPOArena("cpu", 530, {"steer": 33, "throttle": 3}, 4)
        md2 = {}
        with _no_orthogonal_init():
            md2["steer_ppo_0"] = arena2.bind("steer_ppo_0", Model(530, 33))
        arena2.grads.fill_(10.0 * (rank + 1))
        arena2.grads.view(-1)[arena.o_whh:arena.o_whh + 2120 * 544].view(2120, 544)[:, 530:] = 0
        shared.add_gradient(md2)
        ok = shared.counter.get() == 2
        shared.all_reduce()
        shared.all_reduce()                                    # idempotent until the next hand-in
        want = float(sum(11 * (r + 1) for r in range(world)))
        for key, g in shared.grads.items():
            ok &= bool((g == want).all())
        w = md["steer_lstm_2"].rnn.weight_hh
        ok &= shared.grads["steer_lstm_2_rnn.weight_hh_grad"].data_ptr() == w.grad.data_ptr()
        ok &= tuple(w.shape) == (2120, 530) and w.stride(0) == 544          # padded arena view
        pad = arena.grads.view(-1)[arena.o_whh:arena.o_whh + 2120 * 544].view(2120, 544)[:, 530:]
        ok &= float(pad.abs().max()) == 0.0                                 # padding never touched
        gathered = [torch.zeros_like(p0) for _ in range(world)]
        dist.all_gather(gathered, p0)
        ok &= all(torch.equal(gathered[0], t) for t in gathered)
        shared.reset()
        ok &= float(arena.grads.abs().max()) == 0.0 and shared.counter.get() == 0
        q.put((rank, ok))
    finally:
        dist.destroy_process_group()


def test_gradient_allreduce_sum_gloo_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(60)
    assert res == [(0, True), (1, True)]
