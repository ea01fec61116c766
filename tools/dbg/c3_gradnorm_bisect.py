#!/usr/bin/env python3
"""Root cause of the bf16 gradient-norm deviation in tests/test_timed_shapes_gpu.py::test_c3_bf16_contract_end_to_end (VERDICT r5
item 4).  One process per kernel selection (the library reads its switches when it loads):
    python tools/dbg/c3_gradnorm_bisect.py            # default kernels
    CADRE_S1X_CONV=0 python tools/dbg/c3_gradnorm_bisect.py ; CADRE_S2_CONV=0 ... ; CADRE_RING_CONV=0 ...
Prints, for the test's 12 windows at 84 x 84: the bf16 feature error, the losses, the GLOBAL and PER-MODEL gradient norms of
update_policy on (a) the device's bf16-encoder features, (b) the oracle's fp32 features pushed through the SAME device update
(isolates the update kernels), against the oracle on fp32 features; and (c) what UNBIASED feature noise of the same size does to
the oracle's own gradient norm (20 draws): the bound a backward-pass sanity check can hold."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from cadre_amd import synth  # noqa: E402
from oracle import encoder_ref, ppo_ref  # noqa: E402
from ppo_agent.agent import CadreAgent  # noqa: E402


def samples(feat_stack, B, to):
    out = []
    for K in (33, 3):
        rr = np.random.RandomState(40 + K)
        obs = feat_stack.permute(1, 0, 2).reshape(8 * B, 530)
        t = (obs, torch.from_numpy(rr.randint(0, K, (B, 1)).astype(np.int64)),
             torch.from_numpy((0.3 * rr.standard_normal((B, 1))).astype(np.float32)),
             torch.from_numpy(rr.standard_normal((B, 1)).astype(np.float32)), torch.ones(B, 1),
             torch.from_numpy((-np.log(K) + 0.2 * rr.standard_normal((B, 1))).astype(np.float32)),
             torch.from_numpy(rr.standard_normal((B, 1)).astype(np.float32)),
             [torch.zeros(B, 530), torch.zeros(B, 530)],
             torch.from_numpy(rr.randint(0, 4, (B, 1)).astype(np.int32)))
        out.append(tuple(to(x) if not isinstance(x, list) else [to(y) for y in x] for x in t))
    return out


def oracle_norms(st0, feats, B):
    p = ppo_ref.to_torch_params(st0, requires_grad=True)
    s = samples(feats, B, lambda x: x.contiguous())
    losses = ppo_ref.update_policy(p, s[0], s[1])
    per = {m: float(torch.sqrt(sum((q.grad.double() ** 2).sum() for q in d.values()))) for m, d in p.items()}
    return losses, per, float(np.sqrt(sum(v * v for v in per.values())))


def device_norms(agent, feats, B):
    s = samples(feats, B, lambda x: x.contiguous().cuda())
    agent.arena.grads.zero_()
    losses = agent.update_policy(s[0], s[1])
    per = {}
    for name, mod in agent.model_dict.items():
        per[name] = float(torch.sqrt(sum((q.grad.double() ** 2).sum() for q in mod.parameters())))
    return losses, per, float(agent.arena.grads.double().norm())


def main():
    H = W = 84
    n = 12
    sd = synth.encoder_state(3, 3, 7)
    cfg = dict(use_lstm=True, vae_device=0, device_num=0, vae_params="CoPM", measurement_dim=18, num_output=dict(steer=33, throttle=3),
               command_num=4, obs_hw=(H, W), weights_init="none", vae_state_dict=sd, encoder_dtype="bf16", latent_cache=False)
    agent = CadreAgent(rank=0, model_cfg=cfg, frame=8, STEER_CONTROL={i: (i - 16) / 16.0 for i in range(33)},
                       THROTTLE_CONTROL={0: [0, 0], 1: [0, 1], 2: [0.6, 0]}, ent_coeff=0.01, value_coeff=0.1, clip_coeff=1.0, clip=0.1)
    st0 = synth.ppo_state(11)
    agent.arena.load_numpy_state(st0)
    steps = synth.synth_rollout(n, H, W, seed=77)
    fr, fb = [], []
    for i, td in enumerate(steps):
        fr.append(encoder_ref.latent_feature(td["rgb"], td["route_fig"], td["measurements"], sd))
        torch.manual_seed(1000 + i)
        feat, *_ = agent.act(dict(rgb=td["rgb"], route_fig=td["route_fig"].copy(), measurements=td["measurements"], command=td["command"]))
        fb.append(feat.cpu().clone())
    fr, fb = torch.stack(fr), torch.stack(fb)
    d = fb - fr
    scale = float(fr.abs().max())
    print("switches: S1X=%s S2=%s RING=%s | feature error: max %.3e rms %.3e of max |f| = %.3f; mean signed error %.3e (bias / rms = %.3f)"
          % (os.environ.get("CADRE_S1X_CONV", "1"), os.environ.get("CADRE_S2_CONV", "1"), os.environ.get("CADRE_RING_CONV", "1"),
             float(d.abs().max()) / scale, float(d.pow(2).mean().sqrt()) / scale, scale, float(d.mean()) / scale,
             float(d.mean()) / float(d.pow(2).mean().sqrt())))
    l_ref, per_ref, gn_ref = oracle_norms(st0, fr, n)
    l_dev, per_dev, gn_dev = device_norms(agent, fb, n)
    l_d32, per_d32, gn_d32 = device_norms(agent, fr, n)
    l_ob, per_ob, gn_ob = oracle_norms(st0, fb, n)
    print("global |grad|: oracle(fp32 feats) %.5f | device(bf16 feats) %.5f (%+.2f %%) | device(fp32 feats) %.5f (%+.4f %%) | oracle(bf16 feats) %.5f (%+.2f %%)"
          % (gn_ref, gn_dev, 100 * (gn_dev / gn_ref - 1), gn_d32, 100 * (gn_d32 / gn_ref - 1), gn_ob, 100 * (gn_ob / gn_ref - 1)))
    print("losses: oracle %s | device bf16 %s" % ([round(x, 6) for x in l_ref], [round(x, 6) for x in l_dev]))
    for m in sorted(per_ref):
        print("   %-18s oracle %.5f  device(bf16) %+.2f %%  oracle(bf16 feats) %+.2f %%  device(fp32) %+.4f %%"
              % (m, per_ref[m], 100 * (per_dev[m] / per_ref[m] - 1), 100 * (per_ob[m] / per_ref[m] - 1), 100 * (per_d32[m] / per_ref[m] - 1)))
    # unbiased noise of the same rms on the first 512 feature columns (the encoder's part; measurements are exact)
    rms = float(d[..., :512].pow(2).mean().sqrt())
    devs = []
    g = torch.Generator().manual_seed(5)
    for _ in range(20):
        fn = fr.clone()
        fn[..., :512] += rms * torch.randn(fr[..., :512].shape, generator=g)
        _l, _p, gn = oracle_norms(st0, fn, n)
        devs.append(gn / gn_ref - 1)
    devs = np.array(devs)
    print("oracle |grad| under UNBIASED gaussian feature noise of the same rms (%.3e), 20 draws: mean %+.2f %%, std %.2f %%, min %+.2f %%, max %+.2f %%"
          % (rms, 100 * devs.mean(), 100 * devs.std(), 100 * devs.min(), 100 * devs.max()))


if __name__ == "__main__":
    main()
