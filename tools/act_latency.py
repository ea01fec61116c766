#!/usr/bin/env python3
"""Per-env-step latency of CadreAgent.act (reference probe: 82-106 ms on 8 CPU cores at 144x256)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cadre_amd import synth  # noqa: E402
from ppo_agent.agent import CadreAgent  # noqa: E402

H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (144, 256)
fh, fw = synth.feat_hw(H, W)
cfg = dict(use_lstm=True, vae_device=0, device_num=0, vae_params="CoPM", measurement_dim=18,
           num_output=dict(steer=33, throttle=3), command_num=4, obs_hw=(H, W), weights_init="none",
           vae_state_dict=synth.encoder_state(fh, fw, 7))
agent = CadreAgent(rank=0, model_cfg=cfg, frame=8, STEER_CONTROL={i: (i - 16) / 16.0 for i in range(33)},
                   THROTTLE_CONTROL={0: [0, 0], 1: [0, 1], 2: [0.6, 0]}, ent_coeff=0.01, value_coeff=0.1,
                   clip_coeff=1.0, clip=0.1)
agent.arena.load_numpy_state(synth.ppo_state(11))
steps = synth.synth_rollout(60, H, W, seed=3)
for cache, graph in ((True, True), (True, False), (False, False)):
    agent.latent_cache = cache
    agent.act_graph = graph
    agent._ag = None
    agent._cache = None
    ts = []
    for td in steps:
        obs = dict(rgb=td["rgb"], route_fig=td["route_fig"].copy(), measurements=td["measurements"], command=td["command"])
        torch.cuda.synchronize(); t0 = time.perf_counter()
        feat, a, lp, v, _ = agent.act(obs)
        ctl = agent.convert_action(a)          # .item() sync, like the reference loop
        ts.append(time.perf_counter() - t0)
    ts = np.array(ts[20:]) * 1e3
    print("act() %dx%d latent_cache=%s hipGraph=%s: median %.2f ms  p90 %.2f ms" % (H, W, cache, graph, np.median(ts), np.percentile(ts, 90)))
