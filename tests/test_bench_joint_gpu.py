"""bench.py encodes the W workers of one GPU as one stream of windows (chunks cut across workers): the storages it
fills must be the ones the per-worker encode fills, bit for bit (frames are independent; SURVEY.md 8d round definition)."""
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

pytestmark = pytest.mark.gpu


def test_joint_encode_equals_per_worker_encode():
    import bench
    from cadre_amd import synth
    from ppo_agent.agent import CadreAgent
    cfg = dict(T=32, H=84, W=84, workers=4, chunk_windows=16, dedup=False)
    fh, fw = synth.feat_hw(84, 84)
    mcfg = dict(use_lstm=True, vae_device=0, device_num=0, vae_params="CoPM", measurement_dim=18,
                num_output=dict(steer=33, throttle=3), command_num=4, obs_hw=(84, 84), weights_init="none",
                vae_state_dict=synth.encoder_state(fh, fw, 7), encoder_max_frames=64 * 8, encoder_dtype="bf16")
    agent = CadreAgent(rank=0, model_cfg=mcfg, frame=8, STEER_CONTROL={i: (i - 16) / 16.0 for i in range(33)},
                       THROTTLE_CONTROL={0: [0, 0], 1: [0, 1], 2: [0.6, 0]}, ent_coeff=0.01, value_coeff=0.1,
                       clip_coeff=1.0, clip=0.1)
    ws = [bench.Worker(cfg, 1234 + w, agent.device) for w in range(4)]
    for wk in ws:
        bench.encode_worker(agent, wk, cfg, 16)
    want = [wk.stor[0]._obs.clone() for wk in ws]
    for wk in ws:
        wk.stor[0]._obs.zero_(); wk.stor[1]._obs.zero_()
    bench.encode_joint(agent, ws, bench.JointFrames(ws), cfg, 48)         # chunks that straddle workers
    torch.cuda.synchronize()
    for w, wk in enumerate(ws):
        assert float(want[w].abs().sum()) > 0
        assert torch.equal(want[w], wk.stor[0]._obs)
        assert torch.equal(wk.stor[1]._obs, wk.stor[0]._obs)
