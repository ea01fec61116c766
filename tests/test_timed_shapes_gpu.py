"""GPU: parity AT THE SHAPES bench.py TIMES (BASELINE configs C2 and C3), not only at the small
golden shapes.

  * C2 encoder: the two 288x288 golden frames (tests/golden/enc_288.npz, imported reference) placed
    at positions 0 and 1023 of a 1024-frame chunk — the tile choices of the timed run (streamed stem,
    8-wave 128x128 conv tile, 2 GiB buffer windows) are the ones under test, and the result must be
    bit-identical to the 2-frame run (batch invariance at full size).
  * C3 update: W = 4 workers x minibatch 64 = 256 rows in one launch chain must equal the SUM of four
    reference `update_policy` gradient sets (reference ppo_agent/models.py:231-239, train.py:93-102),
    computed by the oracle per worker.
  * C3 bf16 contract: bf16 encoder features -> fp32 heads/losses against the fp32
    oracle end to end, tolerances written below.
"""
import numpy as np
import pytest
import torch

from cadre_amd import synth
from tests.helpers import fill_storages
from tests.test_learner_gpu import make_agent, rel

pytestmark = pytest.mark.gpu

ENC_TOL = 2e-4            # fp32 encoder vs reference goldens (same bar as tests/test_encoder_gpu.py)
# ---- C3 contract ("bf16 encoder / fp32 losses"): bf16 storage of activations and conv weights (8
# significand bits, ~4e-3 per rounding, 20 layers deep), fp32 accumulation, fp32 heads and losses.
C3_FEAT_TOL = 1.5e-2      # max |feat_bf16 - feat_fp32| / max |feat_fp32|   (measured 8e-3)
C3_LOSS_TOL = 2e-2        # each of the three update_policy losses, relative, vs oracle-on-fp32 (measured 3e-3)
C3_VALUE_TOL = 2e-2       # critic values / log-probs of act(): abs error relative to max(1, |ref|) (measured 3e-3)
C3_MARGIN = 0.25          # action indices must agree wherever the fp32 top-2 gap of log(p/q) exceeds this
C3_GRAD_NORM_TOL = 5e-2   # ceiling on |grad| (bf16 features) vs |grad| (fp32 features); the operative bound is 4 sigma of the measured noise spread (sigma ~ 1.3 %)


def test_c2_encoder_goldens_inside_1024_frame_chunk(golden):
    from cadre_amd import hip
    from cadre_amd.encoder import DANetEncoderHIP
    g = golden("enc_288")
    H, W, n = int(g["H"]), int(g["W"]), int(g["n"])
    assert (H, W, n) == (288, 288, 2)
    sd = synth.encoder_state(*synth.feat_hw(H, W), int(g["seed"]))
    r = np.random.RandomState(int(g["frame_seed"]))
    rgb2 = torch.from_numpy(r.randint(0, 256, (n, H, W, 3)).astype(np.uint8)).cuda()
    route2 = torch.from_numpy(((r.rand(n, W, H) < 0.15) * 255).astype(np.uint8)).cuda()
    small = DANetEncoderHIP(sd, H, W, "cuda:0", max_frames=2)
    lat2 = small.latent(rgb2, route2).clone()
    del small
    F = 1024
    gen = torch.Generator(device="cuda").manual_seed(5)
    rgb = torch.randint(0, 256, (F, H, W, 3), dtype=torch.uint8, device="cuda", generator=gen)
    route = ((torch.rand(F, W, H, device="cuda", generator=gen) < 0.15) * 255).to(torch.uint8)
    rgb[0], rgb[F - 1] = rgb2[0], rgb2[1]
    route[0], route[F - 1] = route2[0], route2[1]
    enc = DANetEncoderHIP(sd, H, W, "cuda:0", max_frames=F)
    hip.PROFILE = prof = []
    try:
        lat = enc.latent(rgb, route)
        torch.cuda.synchronize()
    finally:
        hip.PROFILE = None
    code = lambda k: {"ring": 65, "wino_c64": 66, "wgo": 67}.get(k[0], k[0])
    tiles = {code(k) for k, *_ in prof}
    launched = {(code(k), shape[0], shape[1]) for k, _f, _a, _b, shape, _nb in prof}
    # the timed run's kernels: fused front (stem_pool.hip, no GEMM launch for the stem), the four stage-1 convs as the fused
    # Winograd kernel (66) on all 1024 x 36 x 36 tiles of the chunk, 8-wave 128x128 tile (8) for layer2.0 and for the
    # Winograd convs' batched GEMMs
    assert enc.fused_stem and 8 in tiles, sorted(tiles)
    assert sum(1 for k, *_ in prof if k[0] == "wino_c64") == 4 and (66, F * 36 * 36, 64) in launched, sorted(launched)[:8]
    # layer2's three stride-1 convs: F(4x4) with the plane products and the inverse transform in one kernel (winograd_fused.hip)
    assert sum(1 for k, *_ in prof if k[0] == "wgo") == 3 and (67, F * 81, 128) in launched, sorted(launched)[:8]
    got = torch.stack([lat[0], lat[F - 1]])
    e = rel(got.cpu().numpy(), g["latent"])
    print("288x288 goldens inside a 1024-frame chunk: latent rel-max-err %.2e, tiles %s" % (e, sorted(tiles)))
    assert e < ENC_TOL
    assert torch.equal(got, lat2), "per-frame results must not depend on the batch they were computed in"
    assert bool(torch.isfinite(lat).all())


def test_c3_bf16_goldens_inside_2048_frame_joint_chunk(golden):
    """BASELINE C3 at the shape bench.py times: bf16 encoder, ONE 2048-frame chunk of 288x288 windows assembled the way
    `bench.encode_joint` does (window ids into the concatenated frames of the workers of a GPU, the chunk straddling two
    workers; the sliding-window gather rides on the packing pass).  The two reference golden frames
    (tests/golden/enc_288.npz, danet.py:216-238) sit at positions 0 and 2047: latent within the bf16 bar of the
    golden AND bit-identical to the 2-frame bf16 run; every stride-1 3x3 conv must have run on the ping-pong window
    kernels conv3x3_c64s_kernel / conv3x3_ring_pp_kernel<true, ...> (asserted from the launch profile) — 1.36 GB bf16 activations per
    layer-1 tensor, next to the 2 GiB buffer window, 5-6 persistent items per CU."""
    from cadre_amd import hip
    from cadre_amd.encoder import DANetEncoderHIP
    g = golden("enc_288")
    H, W, n = int(g["H"]), int(g["W"]), int(g["n"])
    assert (H, W, n) == (288, 288, 2)
    sd = synth.encoder_state(*synth.feat_hw(H, W), int(g["seed"]))
    r = np.random.RandomState(int(g["frame_seed"]))
    rgb2 = torch.from_numpy(r.randint(0, 256, (n, H, W, 3)).astype(np.uint8)).cuda()
    route2 = torch.from_numpy(((r.rand(n, W, H) < 0.15) * 255).astype(np.uint8)).cuda()
    small = DANetEncoderHIP(sd, H, W, "cuda:0", max_frames=2, dtype="bf16")
    lat2 = small.latent(rgb2, route2).clone()
    del small
    # two workers of T = 128 steps: T + 7 source frames each, windows t .. t+7 (bench.Worker / JointFrames)
    T, S, nW = 128, 8, 2
    nf = T + S - 1
    gen = torch.Generator(device="cuda").manual_seed(9)
    rgb = torch.randint(0, 256, (nW * nf, H, W, 3), dtype=torch.uint8, device="cuda", generator=gen)
    route = ((torch.rand(nW * nf, W, H, device="cuda", generator=gen) < 0.15) * 255).to(torch.uint8)
    win = (torch.arange(T).view(T, 1) + torch.arange(S).view(1, S)).reshape(-1)
    ids = torch.cat([win + w * nf for w in range(nW)]).cuda()            # 2048 window-frame ids across both workers
    F = ids.numel()
    assert F == 2048 and int(ids[0]) == 0 and int(ids[F - 1]) == nW * nf - 1
    rgb[0], route[0] = rgb2[0], route2[0]                                # first frame of worker 0's first window
    rgb[nW * nf - 1], route[nW * nf - 1] = rgb2[1], route2[1]            # last frame of worker 1's last window
    enc = DANetEncoderHIP(sd, H, W, "cuda:0", max_frames=F, dtype="bf16")
    lat = torch.zeros(F, 544, device="cuda")
    hip.PROFILE = prof = []
    try:
        x = enc.preprocess(rgb, route, frame_idx=ids)
        enc.forward_nhwc(x, lat, ldo=544)
        torch.cuda.synchronize()
    finally:
        hip.PROFILE = None
    ring = [k for k, *_ in prof if k[0] == "ring"]
    # 4 (layer1) + 2 + 2 + 2 (layer2-4: conv1 of the first block is stride 2, its conv2 carries the shortcut) + conv5a/5c/51/52
    # = 14 stride-1 3x3 convs on conv3x3_c64s_kernel (layer1) / conv3x3_ring_pp_kernel
    assert len(ring) == 14 and all(k[1] and k[6] == 1 and k[7] == (9 if k[2] == 64 else 0) for k in ring), ring
    # round 5: the three stride-2 3x3 convs on the plane-window kernel, the three conv2 + shortcut pairs on the K-extension kernel;
    # no implicit-GEMM conv is left on the tile kernel (the 1x1 / s2 shortcut launches are gone)
    assert [k for k, *_ in prof if k[0] == "s2"] == [("s2", 10), ("s2", 9), ("s2", 9)], [k for k, *_ in prof if k[0] in ("s2", "s1x")]
    assert [k for k, *_ in prof if k[0] == "s1x"] == [("s1x", 11, 2), ("s1x", 10, 2), ("s1x", 9, 2)]
    conv3 = [(k, shp) for k, _f, _a, _b, shp, _nb in prof if k[0] == "bf16" and k[2] == 2]
    assert len(conv3) == 0, conv3
    got = torch.stack([lat[0, :512], lat[F - 1, :512]])
    e = rel(got.cpu().numpy(), g["latent"])
    print("C3 joint chunk (2048 frames, bf16): latent rel-max-err vs reference golden %.2e" % e)
    assert e < C3_FEAT_TOL
    assert torch.equal(got, lat2), "per-frame bf16 results must not depend on the chunk they were computed in"
    assert bool(torch.isfinite(lat).all())
    # windows repeat frames: the 8 copies of one source frame inside the chunk carry identical latents
    assert torch.equal(lat[7, :512], lat[8 + 6, :512])                   # frame 7 = window 0 slot 7 = window 1 slot 6


def _worker_storages(nW, T, seed0, device):
    from ppo_agent.storage import RolloutStorage
    cpu, dev = [], []
    for w in range(nW):
        data = fill_storages(T, seed0 + w)
        r = np.random.RandomState(900 + w)
        pair_c, pair_d = [], []
        for hd in ("steer", "throttle"):
            d = dict(data[hd])
            d["returns"] = r.standard_normal((T + 1, 1)).astype(np.float32)
            adv = r.standard_normal((T, 1)).astype(np.float32)
            s = RolloutStorage(T, 2, 530, 8, 530, True, 0.99, 0.95)
            for k, v in d.items():
                getattr(s, k).copy_(torch.from_numpy(v))
            s.to(device)
            s.advantages.copy_(torch.from_numpy(adv))
            pair_d.append(s)
            pair_c.append(({k: torch.from_numpy(v) for k, v in d.items()}, torch.from_numpy(adv)))
        cpu.append(pair_c)
        dev.append(pair_d)
    return cpu, dev


@pytest.mark.parametrize("sorted_rows", [True, False])
def test_c3_update_is_sum_of_per_worker_reference_updates(sorted_rows, monkeypatch):
    """num_processes = 4, minibatch 256 per GPU: one batched update == sum over the 4 workers of the
    reference's per-worker update_policy (each worker normalises by ITS minibatch: sum of means)."""
    from oracle import ppo_ref
    monkeypatch.setenv("CADRE_SORTED_UPDATE", "1" if sorted_rows else "0")
    nW, T, Bw = 4, 128, 64
    agent = make_agent(84, 84)
    assert agent.learner.sorted_rows(nW * Bw) == sorted_rows
    cpu, dev = _worker_storages(nW, T, 700, "cuda:0")
    r = np.random.RandomState(17)
    idx = [(torch.from_numpy(r.permutation(T)[:Bw]), torch.from_numpy(r.permutation(T)[:Bw])) for _ in range(nW)]
    st0 = synth.ppo_state(11)
    params = ppo_ref.to_torch_params(st0, requires_grad=True)
    gsum = {m: {k: torch.zeros_like(p) for k, p in d.items()} for m, d in params.items()}
    lsum = np.zeros(3)
    for w in range(nW):
        (ss, sa), (ts, ta) = cpu[w]
        l3 = ppo_ref.update_policy(params, ppo_ref.gather_minibatch(ss, idx[w][0], sa),
                                   ppo_ref.gather_minibatch(ts, idx[w][1], ta))
        lsum += np.array(l3)
        for m, d in params.items():
            for k, p in d.items():
                gsum[m][k] += p.grad
    batches = [(dev[w][0], idx[w][0], dev[w][0].advantages, dev[w][1], idx[w][1], dev[w][1].advantages)
               for w in range(nW)]
    for rep in range(3):                               # eager, capture, hipGraph replay
        got = agent.update_policy_from_storages(batches, sync=True)
        assert rel(got, lsum) < 1e-4, (rep, got, lsum)
        worst = 0.0
        for mn, d in gsum.items():
            gv = agent.arena.views(agent.arena.grads, mn)
            scale = max(float(t.abs().max()) for t in d.values())
            for k, t in d.items():
                err = float((gv[k].cpu() - t).abs().max()) / max(scale, 1e-12)
                worst = max(worst, err)
                assert err < 2e-4, (rep, mn, k, err)
    print("W=4 x 64 (%s): losses rel %.2e, worst per-parameter gradient error %.2e"
          % ("row-sorted" if sorted_rows else "unsorted", rel(got, lsum), worst))


def test_c3_bf16_contract_end_to_end():
    """bf16 encoder -> fp32 LSTM/heads/losses vs the fp32 oracle on the same observations."""
    from oracle import encoder_ref, ppo_ref
    from ppo_agent.agent import CadreAgent
    H = W = 84
    n = 12
    sd = synth.encoder_state(3, 3, 7)
    cfg = dict(use_lstm=True, vae_device=0, device_num=0, vae_params="CoPM", measurement_dim=18,
               num_output=dict(steer=33, throttle=3), command_num=4, obs_hw=(H, W), weights_init="none",
               vae_state_dict=sd, encoder_dtype="bf16", latent_cache=False)
    agent = CadreAgent(rank=0, model_cfg=cfg, frame=8, STEER_CONTROL={i: (i - 16) / 16.0 for i in range(33)},
                       THROTTLE_CONTROL={0: [0, 0], 1: [0, 1], 2: [0.6, 0]}, ent_coeff=0.01, value_coeff=0.1,
                       clip_coeff=1.0, clip=0.1)
    st0 = synth.ppo_state(11)
    agent.arena.load_numpy_state(st0)
    params = ppo_ref.to_torch_params(st0)
    steps = synth.synth_rollout(n, H, W, seed=77)
    feats_ref, feats, agree, decided = [], [], 0, 0
    worst_v = 0.0
    for i, td in enumerate(steps):
        want = encoder_ref.latent_feature(td["rgb"], td["route_fig"], td["measurements"], sd)
        feats_ref.append(want)
        torch.manual_seed(1000 + i)
        feat, a, lp, v, _h = agent.act(dict(rgb=td["rgb"], route_fig=td["route_fig"].copy(),
                                             measurements=td["measurements"], command=td["command"]))
        feats.append(feat.cpu().clone())
        # oracle act on the fp32 features with the same exponential draws
        torch.manual_seed(1000 + i)
        c = td["command"]
        for hd, K, j in (("steer", 33, 0), ("throttle", 3, 1)):
            with torch.no_grad():
                x, _ = ppo_ref.lstm_forward(want, (torch.zeros(1, 530), torch.zeros(1, 530)), params["%s_lstm_%d" % (hd, c)])
                logits = ppo_ref.categorical_logits(x, params["%s_ppo_%d" % (hd, c)])
                val = ppo_ref.mlp3(x, params["%s_ppo_%d" % (hd, c)], "critic")
            q = torch.empty(1, K).exponential_(1)
            score = (logits - torch.log(q))[0]
            top = torch.topk(score, 2).values
            a_ref = int(torch.argmax(score))
            assert a_ref == int(ppo_ref.sample_from_logits(logits, q))
            if float(top[0] - top[1]) > C3_MARGIN:
                decided += 1
                assert int(a[j]) == a_ref, (i, hd, float(top[0] - top[1]))
            agree += int(int(a[j]) == a_ref)
            ev = abs(v[j].item() - val.item()) / max(1.0, abs(val.item()))
            el = abs(lp[j].item() - logits[0, a_ref].item()) / max(1.0, abs(logits[0, a_ref].item())) if int(a[j]) == a_ref else 0.0
            worst_v = max(worst_v, ev, el)
    fr, fb = torch.stack(feats_ref), torch.stack(feats)
    e_feat = float((fb - fr).abs().max() / fr.abs().max())
    print("C3 contract: feature rel-max-err %.2e, value/log-prob err %.2e, actions agree %d/%d (%d above margin %.2f)"
          % (e_feat, worst_v, agree, 2 * n, decided, C3_MARGIN))
    assert e_feat < C3_FEAT_TOL and worst_v < C3_VALUE_TOL
    assert decided >= n                                   # the margin rule really decides most draws
    # one update_policy on the 12 windows: HIP on bf16 features vs oracle on fp32 features
    r = np.random.RandomState(3)
    B = n

    def samples(feat_stack, to):
        out = []
        for K in (33, 3):
            rr = np.random.RandomState(40 + K)
            obs = feat_stack.permute(1, 0, 2).reshape(8 * B, 530)          # time-major [S*B, D]
            t = (obs, torch.from_numpy(rr.randint(0, K, (B, 1)).astype(np.int64)),
                 torch.from_numpy((0.3 * rr.standard_normal((B, 1))).astype(np.float32)),
                 torch.from_numpy(rr.standard_normal((B, 1)).astype(np.float32)), torch.ones(B, 1),
                 torch.from_numpy((-np.log(K) + 0.2 * rr.standard_normal((B, 1))).astype(np.float32)),
                 torch.from_numpy(rr.standard_normal((B, 1)).astype(np.float32)),
                 [torch.zeros(B, 530), torch.zeros(B, 530)],
                 torch.from_numpy(rr.randint(0, 4, (B, 1)).astype(np.int32)))
            out.append(tuple(to(x) if not isinstance(x, list) else [to(y) for y in x] for x in t))
        return out
    p2 = ppo_ref.to_torch_params(st0, requires_grad=True)
    s_ref = samples(fr, lambda x: x.contiguous())
    want_l = ppo_ref.update_policy(p2, s_ref[0], s_ref[1])
    s_dev = samples(fb, lambda x: x.contiguous().cuda())
    got_l = agent.update_policy(s_dev[0], s_dev[1])
    e_loss = max(abs(a_ - b_) / max(abs(b_), 1e-12) for a_, b_ in zip(got_l, want_l))
    gn_ref = float(torch.sqrt(sum((p.grad.double() ** 2).sum() for d in p2.values() for p in d.values())))
    gn = float(agent.arena.grads.double().norm())
    print("C3 contract: losses rel err %.2e (got %s want %s), |grad| %.4f vs %.4f" % (e_loss, got_l, want_l, gn, gn_ref))
    assert e_loss < C3_LOSS_TOL
    # ---- the backward pass, model by model (VERDICT r5 item 4 / ADVICE r5: one global norm with a loosened bound could hide a
    # numeric change in a kernel).  Root cause of the 0.4-0.9 % -> 2.8 % jump of round 5, measured with
    # tools/dbg/c3_gradnorm_bisect.py (profiles/r06_c3_gradnorm_bisect.txt): NOT the update kernels — fed the oracle's fp32
    # features the device's gradients equal the oracle's to 1e-7, fed the device's bf16 features the ORACLE reproduces the
    # device's norm to the digit — but the statistic: 65 % of |grad| of this 12-window minibatch is ONE net (throttle_lstm_0) and
    # it moves 1.3 % (1 sigma) under unbiased feature noise of the bf16 encoder's rms (2e-3 of max); every kernel selection is
    # another draw (-2.85 / -3.07 / -2.76 / -0.91 / +1.67 % for s2+s1x / s2 / s1x / neither / tile kernels only, bias / rms of the
    # feature error -0.05).  So: (1) per model, device vs oracle ON THE SAME bf16 features at 2e-5 — any change in an update kernel
    # shows here; (2) the bf16-vs-fp32 deviation against the spread unbiased noise of the same rms gives the oracle (8 draws).
    per_dev = {name: float(torch.sqrt(sum((q.grad.double() ** 2).sum() for q in mod.parameters()))) for name, mod in agent.model_dict.items()}
    p3 = ppo_ref.to_torch_params(st0, requires_grad=True)
    s_b = samples(fb, lambda x: x.contiguous())
    ppo_ref.update_policy(p3, s_b[0], s_b[1])
    worst_m = 0.0
    for m, d in p3.items():
        ref_m = float(torch.sqrt(sum((q.grad.double() ** 2).sum() for q in d.values())))
        worst_m = max(worst_m, abs(per_dev[m] - ref_m) / ref_m)
    dfe = (fb - fr)[..., :512]
    rms = float(dfe.pow(2).mean().sqrt())
    gen = torch.Generator().manual_seed(5)
    devs = []
    for _ in range(8):
        fn = fr.clone()
        fn[..., :512] += rms * torch.randn(fr[..., :512].shape, generator=gen)
        pn = ppo_ref.to_torch_params(st0, requires_grad=True)
        sn = samples(fn, lambda x: x.contiguous())
        ppo_ref.update_policy(pn, sn[0], sn[1])
        devs.append(float(torch.sqrt(sum((q.grad.double() ** 2).sum() for d in pn.values() for q in d.values()))) / gn_ref - 1.0)
    sigma = float(np.sqrt(np.mean(np.square(devs))))
    dev_bf16 = abs(gn - gn_ref) / gn_ref
    print("C3 contract: per-model |grad| device vs oracle on the same bf16 features: worst %.2e; bf16-vs-fp32 |grad| %.2f %% = %.1f sigma of "
          "unbiased feature noise (sigma %.2f %%, feature bias / rms %.3f)" % (worst_m, 100 * dev_bf16, dev_bf16 / sigma, 100 * sigma,
                                                                           float(dfe.mean()) / rms))
    assert worst_m < 2e-5
    assert abs(float(dfe.mean())) / rms < 0.2           # the encoder's error is noise, not an offset
    assert dev_bf16 < 4.0 * sigma and dev_bf16 < C3_GRAD_NORM_TOL


# ---------------------------------------------------------------------------------------------------------------
# C2 contract (fp32 end to end AT 288 x 288): observations -> act() (pre_process, DANet encoder, LSTM + heads, sampling)
# -> RolloutStorage.insert -> learner_section (bootstrap values, GAE, advantage normalisation, update_policy, clip, Adam)
# against the oracle chain encoder_ref.latent_feature -> ppo_ref act -> ppo_ref learner replay, as ONE chain
# (reference ppo_agent/agent.py:97-141, storage.py:45-58, train.py:55-110).  Bars = north_star's: action indices and
# advantage ordering bit-exact, fp32 losses within 1e-4 relative.  Both conv algorithms of the fp32 encoder: the
# default (Winograd F(4x4) / F(3x3) / fused layer-1 kernel — all ON exactly at this size) and CADRE_WINOGRAD=0.
C2_T = 16
_C2_ORACLE = {}


def _c2_oracle_chain():
    """The oracle side, once per session: features of the 16 windows, act outputs with torch's CPU generator seeded per
    step, storages as train.py:57-72 fills them."""
    if _C2_ORACLE:
        return _C2_ORACLE
    from oracle import encoder_ref, ppo_ref
    H = W = 288
    sd = synth.encoder_state(9, 9, 7)
    st0 = synth.ppo_state(11)
    params = ppo_ref.to_torch_params(st0)
    steps = synth.synth_rollout(C2_T, H, W, seed=2024)
    T = C2_T
    data = {hd: dict(obs=np.zeros((T + 1, 8, 530), np.float32), action=np.zeros((T + 1, 1), np.int64),
                     action_log_probs=np.zeros((T + 1, 1), np.float32), value_preds=np.zeros((T + 1, 1), np.float32),
                     rewards=np.zeros((T + 1, 1), np.float32), masks=np.zeros((T + 1, 1), np.float32),
                     command=np.zeros((T + 1, 1), np.int32), hn=np.zeros((T + 1, 530), np.float32),
                     cn=np.zeros((T + 1, 530), np.float32)) for hd in ("steer", "throttle")}
    margins = []
    for i, td in enumerate(steps):
        feat = encoder_ref.latent_feature(td["rgb"], td["route_fig"], td["measurements"], sd)
        torch.manual_seed(5000 + i)
        c = td["command"]
        for hd, K, j in (("steer", 33, 0), ("throttle", 3, 1)):
            with torch.no_grad():
                x, _ = ppo_ref.lstm_forward(feat, (torch.zeros(1, 530), torch.zeros(1, 530)), params["%s_lstm_%d" % (hd, c)])
                logits = ppo_ref.categorical_logits(x, params["%s_ppo_%d" % (hd, c)])
                val = ppo_ref.mlp3(x, params["%s_ppo_%d" % (hd, c)], "critic")
            q = torch.empty(1, K).exponential_(1)
            a = int(ppo_ref.sample_from_logits(logits, q))
            ratio = (torch.softmax(logits, -1) / q)[0]
            top2 = torch.topk(ratio, 2).values
            margins.append(float((top2[0] - top2[1]) / top2[0]))
            d = data[hd]
            d["obs"][i] = feat.numpy()
            d["action"][i, 0] = a
            d["action_log_probs"][i, 0] = float(logits[0, a])
            d["value_preds"][i, 0] = float(val)
            d["rewards"][i, 0] = float(td["reward"][j])
            d["masks"][i, 0] = 0.0 if bool(td["done"][j]) else 1.0          # train.py:61-62
            d["command"][i, 0] = c
    _C2_ORACLE.update(steps=steps, data=data, margins=margins, sd=sd, st0=st0)
    return _C2_ORACLE


@pytest.mark.parametrize("algo", ["default", "direct", "stem_bf16x3"])
def test_c2_fp32_contract_end_to_end(algo, monkeypatch):
    from ppo_agent.agent import CadreAgent
    from ppo_agent.models import Shared_grad_buffers
    from ppo_agent.storage import RolloutStorage
    from ppo_agent.train import learner_section
    from tests.helpers import oracle_learner_replay
    from tests.test_learner_gpu import per_model
    if algo == "direct":
        monkeypatch.setenv("CADRE_WINOGRAD", "0")
    else:
        monkeypatch.delenv("CADRE_WINOGRAD", raising=False)
    # (opt-in front: u8 pixels x three exact bf16 pieces of every fp32 weight on the bf16 matrix cores — the whole contract at the same bars)
    monkeypatch.setenv("CADRE_STEM_EXACT_BF16", "1" if algo == "stem_bf16x3" else "0")
    o = _c2_oracle_chain()
    T, H, W = C2_T, 288, 288
    cfg = dict(use_lstm=True, vae_device=0, device_num=0, vae_params="CoPM", measurement_dim=18,
               num_output=dict(steer=33, throttle=3), command_num=4, obs_hw=(H, W), weights_init="none",
               vae_state_dict=o["sd"], latent_cache=False)
    agent = CadreAgent(rank=0, model_cfg=cfg, frame=8, STEER_CONTROL={i: (i - 16) / 16.0 for i in range(33)},
                       THROTTLE_CONTROL={0: [0, 0], 1: [0, 1], 2: [0.6, 0]}, ent_coeff=0.01, value_coeff=0.1,
                       clip_coeff=1.0, clip=0.1)
    assert (agent.vae_model.winograd_convs() > 0) == (algo != "direct")
    assert agent.vae_model.stem_x3 == (algo == "stem_bf16x3")
    agent.arena.load_numpy_state(o["st0"])
    stor = [RolloutStorage(T, 2, 530, 8, 530, True, 0.99, 0.95) for _ in range(2)]
    for s in stor:
        s.to(agent.device)
    worst_f = worst_v = 0.0
    for i, td in enumerate(o["steps"]):                      # train.py:55-72
        torch.manual_seed(5000 + i)
        obs = dict(rgb=td["rgb"], route_fig=td["route_fig"].copy(), measurements=td["measurements"], command=td["command"])
        feat, a, lp, v, hid = agent.act(obs)
        worst_f = max(worst_f, rel(feat.cpu().numpy(), o["data"]["steer"]["obs"][i]))
        for j, hd in enumerate(("steer", "throttle")):
            d = o["data"][hd]
            assert int(a[j]) == int(d["action"][i, 0]), (i, hd, o["margins"][2 * i + j])          # bit-exact indices
            worst_v = max(worst_v, abs(lp[j].item() - d["action_log_probs"][i, 0]) / max(1.0, abs(d["action_log_probs"][i, 0])),
                          abs(v[j].item() - d["value_preds"][i, 0]) / max(1.0, abs(d["value_preds"][i, 0])))
            stor[j].insert(feat, a[j], lp[j], v[j], torch.tensor(float(td["reward"][j])),
                           torch.tensor([[0.0] if td["done"][j] else [1.0]]), hid, td["command"])
    assert worst_f < ENC_TOL and worst_v < 1e-4, (worst_f, worst_v)
    g = dict(T=T, mbn=2, epochs=2, ppo_seed=11, data_seed=0, torch_seed=31, names=np.array(sorted(o["st0"])))
    want = oracle_learner_replay(g, data=o["data"])
    shared = Shared_grad_buffers(agent.model_dict, agent.device)
    torch.manual_seed(31)
    vl, pl, el = learner_section(agent, stor[0], stor[1], False, dict(use_adv_norm=True, ppo_epoch=2, max_grad_norm=250.0), shared)
    e_loss = rel(np.array([vl, pl, el]).T, np.array(want["losses"]))
    for hd, s in zip(("steer", "throttle"), stor):
        mine = s.advantages.cpu().numpy()[:, 0]
        assert np.array_equal(np.argsort(mine, kind="stable"), np.argsort(want["adv_" + hd][:, 0], kind="stable")), hd
    names = [str(n) for n in g["names"]]
    ps = per_model(agent.arena, agent.arena.params, names, lambda ts: float(sum(t.sum() for t in ts)))
    e_par = rel(ps, want["param_sums"][-1])
    print("C2 contract (%s): feature err %.2e, value/log-prob err %.2e, losses rel err %.2e, parameter sums %.2e, min action margin %.2e"
          % (algo, worst_f, worst_v, e_loss, e_par, min(o["margins"])))
    assert e_loss < 1e-4                                   # north_star: fp32 losses within 1e-4 relative
    assert e_par < 1e-5
