#!/usr/bin/env python3
"""Data-parallel learner round with SEVERAL ranks on ONE GPU (one GPU per box here): each rank is a fresh process
on cuda:0, the process group runs over `gloo` with device tensors, and every rank plays `bench.learner_round`
(joint encode -> get_values -> GAE -> 8 x [batched update_policy + gradient exchange(SUM) + per-model clip + Adam],
reference ppo_agent/train.py:76-110, models.py:231-239, chief.py:13-21) on a C1-sized config with W = 2 workers and
its own worker seeds.  The parent (no GPU call) starts the ranks and collects what they wrote:

    python -m tests.dp_ranks_driver OUT_DIR MODE [WORLD]      MODE in {allreduce, buckets, sharded}

Rank r writes OUT_DIR/rank<r>.npz: the parameter arena after round 1 and after round 2 (graph replays), its losses, the number of
exchanges it ran and its storages' feature rows (for the oracle comparison in tests/test_dp_gpu.py)."""
import json
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

CFG = dict(workers=2, T=32, H=84, W=84, chunk_windows=256, dedup=False)


def agent_for(rank, device_num=0):
    from cadre_amd import synth
    from ppo_agent.agent import CadreAgent
    H, W = CFG["H"], CFG["W"]
    fh, fw = synth.feat_hw(H, W)
    mcfg = dict(use_lstm=True, vae_device=device_num, device_num=device_num, vae_params="CoPM", measurement_dim=18,
                num_output=dict(steer=33, throttle=3), command_num=4, obs_hw=(H, W), weights_init="none",
                vae_state_dict=synth.encoder_state(fh, fw, 7), encoder_max_frames=CFG["chunk_windows"] * 8)
    agent = CadreAgent(rank=rank, model_cfg=mcfg, frame=8, STEER_CONTROL={i: (i - 16) / 16.0 for i in range(33)},
                       THROTTLE_CONTROL={0: [0, 0], 1: [0, 1], 2: [0.6, 0]}, ent_coeff=0.01, value_coeff=0.1,
                       clip_coeff=1.0, clip=0.1)
    agent.arena.load_numpy_state(synth.ppo_state(11))
    return agent


def rank_main(out_dir, mode):
    import numpy as np
    import torch
    import torch.distributed as dist
    import bench
    from ppo_agent.models import Shared_grad_buffers
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    os.environ["CADRE_GRAD_EXCHANGE"] = "sharded" if mode == "sharded" else "allreduce"
    if world == 1:
        os.environ["CADRE_BENCH_FORCE_DIST"] = "1"           # still run the collectives
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        os.environ["CADRE_GRAD_BUCKETS"] = "1" if mode == "buckets" else "0"      # (opt-in; "allreduce" = the one blocking exchange, the default)
        cfg = dict(CFG)
        agent = agent_for(rank)
        dist.broadcast(agent.arena.params, 0)
        workers = [bench.Worker(cfg, 1234 + 1000 * rank + w, agent.device) for w in range(cfg["workers"])]
        joint = bench.JointFrames(workers)
        shared = Shared_grad_buffers(agent.model_dict, agent.device)
        torch.manual_seed(100 + rank)
        losses = bench.learner_round(agent, workers, cfg, shared, joint=joint)
        torch.cuda.synchronize()
        params1 = agent.arena.params.cpu().numpy()
        feats = np.stack([wk.stor[0].obs.cpu().numpy() for wk in workers])
        adv = np.stack([np.stack([wk.stor[j].advantages.cpu().numpy() for j in (0, 1)]) for wk in workers])
        # a second round on the captured hipGraphs (the first one ran the eager warm-up and the captures)
        losses2 = bench.learner_round(agent, workers, cfg, shared, joint=joint)
        torch.cuda.synchronize()
        np.savez(os.path.join(out_dir, "rank%d.npz" % rank), params1=params1, params2=agent.arena.params.cpu().numpy(),
                 losses=np.array(losses), losses2=np.array(losses2), n_exchange=shared.n_allreduce,
                 mode=shared.exchange_mode(), feats=feats, adv=adv)
    finally:
        dist.destroy_process_group()
    return 0


def main(out_dir, mode, world=2):
    if "RANK" in os.environ:
        return rank_main(out_dir, mode)
    os.makedirs(out_dir, exist_ok=True)
    sock = socket.socket(); sock.bind(("127.0.0.1", 0)); port = sock.getsockname()[1]; sock.close()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK="0", WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""),
                   HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, "-m", "tests.dp_ranks_driver", out_dir, mode, str(world)], cwd=ROOT, env=env))
    rcs = [p.wait(timeout=1500) for p in procs]
    print("DP_RESULT " + json.dumps(dict(exitcodes=rcs, mode=mode, world=world)), flush=True)
    return 0 if all(rc == 0 for rc in rcs) else 1


if __name__ == "__main__":
    sys.exit(main(sys.argv[1], sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 2))
