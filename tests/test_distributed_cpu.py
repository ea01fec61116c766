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


def _sharded_worker(rank, world, port, q):
    """reduce-scatter / all-gather form of the exchange (CADRE_GRAD_EXCHANGE=sharded) and the MLP-bucket form, host
    bookkeeping over gloo on CPU tensors: shard bounds, SUM inside the shard, parameters gathered from their owners,
    bucketed all-reduce == one all-reduce."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), CADRE_GRAD_EXCHANGE="sharded")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from cadre_amd.arena import PPOArena
        from ppo_agent.models import Model, Shared_grad_buffers, _no_orthogonal_init
        arena = PPOArena("cpu", 530, {"steer": 33, "throttle": 3}, 4)
        with _no_orthogonal_init():
            md = {"steer_ppo_0": arena.bind("steer_ppo_0", Model(530, 33))}
        shared = Shared_grad_buffers(md, torch.device("cpu"))
        ok = shared.exchange_mode() == "sharded" and shared.dist_world() == world
        lo, hi = shared.shard()
        n = arena.total // world
        ok &= (lo, hi) == (rank * n, (rank + 1) * n) and lo % 4 == 0
        g = torch.Generator().manual_seed(5)
        base = torch.randn(arena.total, generator=g)
        arena.grads.copy_(base * (rank + 1))
        ok &= shared.reduce_scatter() is None                               # nothing handed in yet
        shared.add_gradient(md)
        ok &= shared.pending() and shared.reduce_scatter() == (lo, hi)
        want = base * float(sum(r + 1 for r in range(world)))
        ok &= bool(torch.equal(arena.grads[lo:hi], want[lo:hi]))             # the shard holds the SUM over ranks
        ok &= (not shared.pending()) and shared.reduce_scatter() is None     # once per hand-in
        arena.params.fill_(-1.0)
        arena.params[lo:hi] = float(rank + 10)                               # "updated" shard
        shared.all_gather_params()
        for r in range(world):
            ok &= bool((arena.params[r * n:(r + 1) * n] == float(r + 10)).all())
        nrm = torch.tensor([1.0 + rank, 2.0], dtype=torch.float64)
        shared.all_reduce_norms(nrm)
        ok &= nrm.tolist() == [float(sum(1 + r for r in range(world))), 2.0 * world]
        shared.reset()
        # bucketed all-reduce: grads[P0:] first (async), grads[:P0] + wait at the optimiser step
        os.environ["CADRE_GRAD_EXCHANGE"] = "allreduce"
        arena.grads.copy_(base * (rank + 1))
        shared.reduce_bucket_async(arena.P0)
        shared.add_gradient(md)
        shared.all_reduce()
        ok &= bool(torch.equal(arena.grads, want)) and shared.n_allreduce == 2
        # three buckets, as the learner cuts them: MLP towers, then the steer nets' LSTM block; all_reduce takes the rest
        arena.grads.copy_(base * (rank + 1))
        half = (arena.Z // 2) * arena.size_L
        shared.reduce_bucket_async(arena.P0, arena.total)
        shared.reduce_bucket_async(0, half)
        shared.add_gradient(md)
        shared.all_reduce()
        ok &= bool(torch.equal(arena.grads, want)) and shared.n_allreduce == 3
        # a bucket nobody collects (no hand-in afterwards) is waited for by reset()
        shared.reduce_bucket_async(arena.P0)
        shared.reset()
        ok &= not shared._buckets
        # bucketing is OPT-IN (no multi-GPU RCCL run of it exists yet) ...
        os.environ.pop("CADRE_GRAD_BUCKETS", None)
        ok &= shared.overlap_hook() is None
        os.environ["CADRE_GRAD_BUCKETS"] = "1"
        ok &= shared.overlap_hook() is not None and shared.overlap_hook(arena) is not None
        # ... and never handed to an agent whose nets live in ANOTHER arena (reference topology: the worker's gradients are
        # ADDED to the shared arena after its backward — a bucket of the shared arena reduced during that backward would
        # leave before the add and never be summed): such a worker gets the one blocking all-reduce
        arena_w = PPOArena("cpu", 530, {"steer": 33, "throttle": 3}, 4)
        with _no_orthogonal_init():
            md_w = {"steer_ppo_0": arena_w.bind("steer_ppo_0", Model(530, 33))}
        from ppo_agent.models import arena_of
        hook = shared.overlap_hook(arena_of(md_w))
        ok &= hook is None
        arena.grads.zero_()
        arena_w.grads.copy_(base * (rank + 1))                  # the worker's backward wrote its own arena
        shared.add_gradient(md_w)                               # models.py:231-239: += into the shared arena
        shared.all_reduce()
        ok &= bool(torch.equal(arena.grads, want))              # every range summed over the ranks, MLP towers and steer LSTMs included
        os.environ.pop("CADRE_GRAD_BUCKETS")
        shared.reset()
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


def test_sharded_and_bucketed_exchange_gloo_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_sharded_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(60)
    assert res == [(0, True), (1, True)]


def _handin_child(shared, q):
    """A worker process of the reference topology (main.py:63-68): its pickled copy of the buffers."""
    shared.counter.increment()           # what add_gradient does to the shared state (no arena on this side)
    q.put(shared.pending())


def test_pending_exchange_is_shared_state_across_processes():
    """reference main.py:57-70: workers and chief hold separately pickled copies of Shared_grad_buffers.  A hand-in
    by a worker process must make the exchange due in the chief's copy (a process-local flag would not).  (The
    bookkeeping alone: the arena itself only travels between processes as HIP-IPC handles — the spawned chief +
    worker run on the device is tests/test_topology_gpu.py.)"""
    from ppo_agent.models import Shared_grad_buffers
    from ppo_agent.utils import Counter
    mp.set_start_method("spawn", force=True)                   # reference main.py:24 (locks must be born under it)
    shared = Shared_grad_buffers.__new__(Shared_grad_buffers)
    shared.counter, shared._reduced_at, shared._n_exchange = Counter(), Counter(), Counter()
    assert not shared.pending()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_handin_child, args=(shared, q))
    p.start()
    assert q.get(timeout=120) is True
    p.join(60)
    assert shared.pending() and shared.counter.get() == 1      # seen by this (the chief's) copy
    shared.all_reduce()                                        # no process group: marks the hand-ins as covered
    assert not shared.pending() and shared.n_allreduce == 0
    shared.counter.reset(); shared._reduced_at.reset()
    assert not shared.pending()


def test_bench_starts_one_fresh_process_per_rank():
    """`python bench.py --gpus N` as the driver calls it (no torchrun): the parent never touches the GPU and starts N
    children with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*; rank 0's line is the only stdout."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "4", "--spawn-selftest"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["rank"] == 0 and d["world"] == 4 and d["master"].startswith("127.0.0.1:")
    seen = sorted(json.loads(ln.split("selftest ", 1)[1])["local_rank"] for ln in p.stderr.splitlines() if "selftest" in ln)
    assert seen == [0, 1, 2, 3]


def test_bench_parent_ends_the_peers_of_a_dead_rank():
    """A rank that dies (OOM, ABI mismatch) leaves its peers waiting in a collective: the parent polls every child,
    terminates the others and returns the dead rank's status instead of hanging until an outer timeout."""
    import subprocess
    import sys
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["CADRE_BENCH_SELFTEST_FAIL_RANK"] = "1"
    t0 = time.monotonic()
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "3", "--spawn-selftest"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert p.returncode == 3, (p.returncode, p.stderr[-2000:])
    assert time.monotonic() - t0 < 120
    assert "rank 1 failed with status 3" in p.stderr


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
