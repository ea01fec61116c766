#!/usr/bin/env python3
"""Plays reference main.py:24-72 on one GPU with a synthetic env: the parent builds the shared nets on the
device, `Shared_grad_buffers`, `optim.Adam`, TrafficLight/Counter, then SPAWNS one `chief` process and one
`train` worker, handing everything over by pickling (HIP-IPC tensor handles) exactly like the reference
launcher.  Afterwards the same two episodes are run with the in-process hand-off and the two final snapshots
are compared.  Run as its own process (tests/test_topology_gpu.py does):

    python -m tests.spawn_topology_driver OUT_DIR
"""
import copy
import functools
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def chief_in_process_group(*args):
    """The chief process of main.py:57-60 as a member of a (world-size-1, forced) process group: its `chief_step` must
    run the cross-rank gradient exchange although the hand-ins happened in OTHER processes (the workers' pickled copies
    of Shared_grad_buffers) — the exchange is due by shared state, not by a process-local flag."""
    import torch
    import torch.distributed as dist
    from ppo_agent.chief import chief
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29547")
    os.environ["CADRE_BENCH_FORCE_DIST"] = "1"
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        chief(*args)
    finally:
        dist.destroy_process_group()


def main(out_dir):
    import torch
    import torch.multiprocessing as mp
    import torch.optim as optim
    from cadre_amd import synth
    from ppo_agent.chief import chief
    from ppo_agent.models import Shared_grad_buffers, arena_of, create_model
    from ppo_agent.train import train
    from ppo_agent.utils import Counter, TrafficLight
    from tests.helpers import SyntheticEnv, topology_cfgs
    try:
        mp.set_start_method("spawn")
    except RuntimeError:
        pass
    res = {}
    snaps = {}
    keep = []          # earlier modes' shared tensors stay allocated: a block that was exported over HIP IPC is never handed out
                       # again inside this process (re-exporting a reused block has failed with hipIpcGetMemHandle: invalid
                       # argument on some boxes of the pool, once in ~10 runs)
    for mode in ("spawned", "spawned_dist", "in_process"):
        torch.cuda.ipc_collect()
        work = os.path.join(out_dir, mode)
        os.makedirs(work, exist_ok=True)
        train_cfg, agent_cfg, env_cfg, rollout_cfg = topology_cfgs(work)
        _, shared = create_model(agent_cfg.model_cfg, load_vae=False)          # main.py:38
        arena_of(shared).load_numpy_state(synth.ppo_state(11))
        plist = []
        for name in shared:                                                    # main.py:40-43
            shared[name] = shared[name].share_memory()
            plist += list(shared[name].parameters())
        device = torch.device("cuda:" + str(agent_cfg.model_cfg.device_num))
        bufs = Shared_grad_buffers(shared, device)
        keep.append((shared, bufs))
        if mode in ("spawned", "spawned_dist"):
            light, counter, sons = TrafficLight(), Counter(), Counter()
            opt = optim.Adam(plist, lr=train_cfg.lr)
            procs = [mp.Process(target=chief if mode == "spawned" else chief_in_process_group,
                                args=(1, light, counter, shared, bufs, opt, sons, train_cfg.max_grad_norm, 1))]
            procs.append(mp.Process(target=functools.partial(train, env_cls=SyntheticEnv), args=(
                0, train_cfg, copy.deepcopy(agent_cfg), copy.deepcopy(env_cfg), rollout_cfg, light, counter, shared,
                bufs, sons)))
            for p in procs:
                p.start()
            for p in procs:
                p.join(540)
            res["exitcodes" if mode == "spawned" else "exitcodes_dist"] = [p.exitcode for p in procs]
            for p in procs:
                if p.is_alive():
                    p.terminate()
            res[mode + "_exchanges"] = bufs.n_allreduce          # shared counter: exchanges the chief process ran
        else:
            sons = Counter()
            train(0, train_cfg, copy.deepcopy(agent_cfg), copy.deepcopy(env_cfg), rollout_cfg, None, None, shared,
                  bufs, sons, env_cls=SyntheticEnv)
        torch.cuda.synchronize()
        snap = os.path.join(work, "models", "ppo_model_%d.pt" % (train_cfg.max_episode - 1))
        res[mode + "_snapshot"] = os.path.exists(snap)
        if os.path.exists(snap):
            snaps[mode] = torch.load(snap, map_location="cpu", weights_only=False)
        res[mode + "_shared_param_sum"] = float(arena_of(shared).params.double().sum())
    if len(snaps) == 3:
        for tag, key in (("spawned", "max_abs_param_diff"), ("spawned_dist", "max_abs_param_diff_dist")):
            worst, n = 0.0, 0
            for name in snaps[tag]:
                a, b = snaps[tag][name].state_dict(), snaps["in_process"][name].state_dict()
                for k in a:
                    worst = max(worst, float((a[k] - b[k]).abs().max()))
                    n += 1
            res["tensors_compared"] = n
            res[key] = worst
        init = synth.ppo_state(11)
        moved = max(float((snaps["spawned"][m].state_dict()[k] - torch.from_numpy(init[m][k])).abs().max())
                    for m in snaps["spawned"] for k in snaps["spawned"][m].state_dict())
        res["max_abs_update"] = moved
    print("TOPOLOGY_RESULT " + json.dumps(res), flush=True)
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1]))
