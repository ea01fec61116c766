"""GPU: SURVEY.md §8 rows f2 (rollout record / replay, BASELINE config C5) and f3 (checkpoint files).

  * f2: an episode recorded by `RolloutRecorder` during a live `train()`-style loop and read back with
    `replay.load_episode` must drive the HIP learner to exactly the numbers the live observations produced:
    features bit-identical, losses and parameters after the learner section bit-identical — and `bench.py --replay`'s
    Worker builds the same storages from the file.
  * f3: the encoder is loaded from a `net_epoch<N>` checkpoint FILE (`{'autoencoder': state_dict}`,
    experiments_builder.py:446-462) under $CHALLENGE_DIR exactly where `create_model(load_vae=True)` (models.py:54-70)
    looks for it; a snapshot written by one agent is loaded by another and reproduces its actions.
"""
import os

import numpy as np
import pytest
import torch

from cadre_amd import replay, synth
from tests.test_learner_gpu import make_agent

pytestmark = pytest.mark.gpu


def _run_episode(agent, observations, T, recorder=None):
    """train.py:50-75 with given observations: act, record, insert."""
    from ppo_agent.storage import RolloutStorage
    st = [RolloutStorage(T, 2, 530, 8, 530, True, 0.99, 0.95) for _ in range(2)]
    for s in st:
        s.to("cuda:0")
    feats = []
    for t in range(T):
        obs = observations(t)
        raw = dict(obs, rgb=obs["rgb"].copy(), route_fig=obs["route_fig"].copy())
        feat, action, alp, values, hidden = agent.act(obs)
        feats.append(feat.clone())
        rew, done = obs["_reward"], obs["_done"]
        if recorder is not None:
            recorder.step(raw, action, alp, values, rew, done)
        for j in range(2):
            st[j].insert(feat, action[j], alp[j], values[j], float(rew[j]), torch.tensor([[0.0] if done[j] else [1.0]]),
                         hidden, obs["command"])
    return st, torch.stack(feats)


def _learner(agent, st):
    from ppo_agent.models import Shared_grad_buffers
    from ppo_agent.train import learner_section
    cfg = dict(use_adv_norm=True, ppo_epoch=2, max_grad_norm=250.0, lr=3e-4)
    shared = Shared_grad_buffers(agent.model_dict, agent.device)
    torch.manual_seed(5)
    out = learner_section(agent, st[0], st[1], False, cfg, shared)
    torch.cuda.synchronize()
    return out, agent.arena.params.clone()


def test_recorded_episode_replays_bit_identically(tmp_path):
    H = W = 84
    T = 16
    steps = synth.synth_rollout(T, H, W, seed=31)

    def live(t):
        td = steps[t]
        return dict(rgb=td["rgb"], route_fig=td["route_fig"].copy(), measurements=td["measurements"], command=td["command"],
                    _reward=td["reward"], _done=td["done"])
    a1 = make_agent(H, W)
    rec = replay.RolloutRecorder(str(tmp_path), worker=0)
    torch.manual_seed(9)
    st1, f1 = _run_episode(a1, live, T, rec)
    path = rec.end_episode()

    ep = replay.load_episode(path)
    assert ep["rgb"].shape[0] == T + 7 and ep["window"].shape == (T, 8)          # distinct frames stored once

    def replayed(t):
        o = replay.windows(ep, t)
        o["_reward"], o["_done"] = ep["reward"][t], ep["done"][t].astype(bool)
        return o
    a2 = make_agent(H, W)
    torch.manual_seed(9)
    st2, f2 = _run_episode(a2, replayed, T)
    assert torch.equal(f1, f2)                                                    # features: same bits
    for s1, s2 in zip(st1, st2):
        for k in ("action", "action_log_probs", "value_preds", "rewards", "masks", "command"):
            assert torch.equal(getattr(s1, k), getattr(s2, k)), k
        assert ep["action"][:, 0].tolist() == st1[0].action[:T, 0].tolist()      # what was recorded is what was stored
    out1, p1 = _learner(a1, st1)
    out2, p2 = _learner(a2, st2)
    assert out1 == out2 and torch.equal(p1, p2)                                   # losses and parameters: same bits
    assert float((p1 - make_agent(H, W).arena.params).abs().max()) > 0

    # bench.py --replay builds its worker from the same file (C5 plumbing): storages equal the recorded scalars
    import bench
    cfg = dict(T=T, H=H, W=W)
    wk = bench.Worker(cfg, 0, torch.device("cuda:0"), ep)
    assert torch.equal(wk.stor[0].action[:T], st1[0].action[:T]) and torch.equal(wk.stor[1].rewards[:T], st1[1].rewards[:T])
    assert wk.rgb.shape[0] == T + 7 and wk.win.numel() == T * 8


def test_encoder_loads_from_net_epoch_file_and_snapshot_transfers(tmp_path, monkeypatch):
    from ppo_agent.agent import CadreAgent
    H = W = 84
    sd = synth.encoder_state(3, 3, 7)
    ck = tmp_path / "carla_perception" / "Experiments34" / "danet912_nocrash_IL_n10_k1234_r40"
    ck.mkdir(parents=True)
    torch.save({"autoencoder": {k: torch.as_tensor(v) for k, v in sd.items()}, "epoch": 90}, str(ck / "net_epoch90"))
    monkeypatch.setenv("CHALLENGE_DIR", str(tmp_path))
    cfg = dict(use_lstm=True, vae_device=0, device_num=0, vae_params="CoPM", measurement_dim=18,
               num_output=dict(steer=33, throttle=3), command_num=4, obs_hw=(H, W), weights_init="none")   # no vae_state_dict
    kw = dict(rank=0, frame=8, STEER_CONTROL={i: (i - 16) / 16.0 for i in range(33)},
              THROTTLE_CONTROL={0: [0, 0], 1: [0, 1], 2: [0.6, 0]}, ent_coeff=0.01, value_coeff=0.1, clip_coeff=1.0, clip=0.1)
    from_file = CadreAgent(model_cfg=cfg, **kw)
    in_memory = make_agent(H, W)
    assert from_file.vae_model.fingerprint == in_memory.vae_model.fingerprint
    # snapshot written by one agent (reference dict-of-pickled-modules format, agent.py:245-260) drives another
    in_memory.arena.params.mul_(1.25)
    snap = str(tmp_path / "ppo_model_7.pt")
    in_memory.save_snapshot(snap, fix_missing_lstm=True)
    from_file.load_snapshot(snap, None)
    assert torch.equal(from_file.arena.params, in_memory.arena.params)
    td = synth.synth_rollout(1, H, W, seed=3)[0]
    outs = []
    for ag in (from_file, in_memory):
        torch.manual_seed(1)
        f, a, lp, v, _ = ag.act(dict(rgb=td["rgb"], route_fig=td["route_fig"].copy(), measurements=td["measurements"],
                                     command=td["command"]))
        outs.append((f.clone(), [int(a[0]), int(a[1])], lp[0].item(), v[1].item()))
    assert torch.equal(outs[0][0], outs[1][0]) and outs[0][1:] == outs[1][1:]
