"""smoke(): one small hot-path invocation on cuda:0 checked against the oracle — test infrastructure, called by
__graft_entry__.smoke() (moved out of the product package in round 6: nothing under cadre_amd/ imports oracle/)."""
import numpy as np
import torch


def smoke():
    from cadre_amd import synth
    from oracle import encoder_ref, ppo_ref                     # checker only
    from ppo_agent.agent import CadreAgent
    H = W = 84
    sd = synth.encoder_state(3, 3, 7)
    cfg = dict(use_lstm=True, vae_device=0, device_num=0, vae_params="CoPM", measurement_dim=18,
               num_output=dict(steer=33, throttle=3), command_num=4, obs_hw=(H, W), weights_init="none", vae_state_dict=sd)
    agent = CadreAgent(rank=0, model_cfg=cfg, frame=8, STEER_CONTROL={i: (i - 16) / 16.0 for i in range(33)},
                       THROTTLE_CONTROL={0: [0, 0], 1: [0, 1], 2: [0.6, 0]}, ent_coeff=0.01, value_coeff=0.1,
                       clip_coeff=1.0, clip=0.1)
    st0 = synth.ppo_state(11)
    agent.arena.load_numpy_state(st0)
    td = synth.synth_rollout(1, H, W, seed=1)[0]
    torch.manual_seed(0)
    feat, a, lp, v, _ = agent.act(dict(rgb=td["rgb"], route_fig=td["route_fig"].copy(),
                                       measurements=td["measurements"], command=td["command"]))
    want = encoder_ref.latent_feature(td["rgb"], td["route_fig"], td["measurements"], sd)
    err = float((feat.cpu() - want).abs().max() / want.abs().max())
    assert err < 2e-4, err
    # one PPO update vs oracle autograd
    r = np.random.RandomState(0)
    B = 8
    samp, dsamp = [], []
    for K in (33, 3):
        t = (torch.from_numpy((r.standard_normal((8 * B, 530)) * 0.5).astype(np.float32)),
             torch.from_numpy(r.randint(0, K, (B, 1)).astype(np.int64)),
             torch.from_numpy((0.3 * r.standard_normal((B, 1))).astype(np.float32)),
             torch.from_numpy(r.standard_normal((B, 1)).astype(np.float32)), torch.ones(B, 1),
             torch.from_numpy((-np.log(K) + 0.2 * r.standard_normal((B, 1))).astype(np.float32)),
             torch.from_numpy(r.standard_normal((B, 1)).astype(np.float32)),
             [torch.zeros(B, 530), torch.zeros(B, 530)],
             torch.from_numpy(r.randint(0, 4, (B, 1)).astype(np.int32)))
        samp.append(t)
        dsamp.append(tuple(x.cuda() if not isinstance(x, list) else [y.cuda() for y in x] for x in t))
    params = ppo_ref.to_torch_params(st0, requires_grad=True)
    want_l = ppo_ref.update_policy(params, samp[0], samp[1])
    got_l = agent.update_policy(dsamp[0], dsamp[1])
    lerr = max(abs(g - w) / max(abs(w), 1e-12) for g, w in zip(got_l, want_l))
    assert lerr < 1e-4, (got_l, want_l)
    print("smoke ok: encoder feature rel-err %.2e, PPO losses rel-err %.2e, actions %s" %
          (err, lerr, [int(a[0]), int(a[1])]))
