"""GPU: the drop-in path (ppo_agent.agent / storage / chief mirrors on HIP kernels) against
golden vectors produced by the imported reference (tests/golden/{update,act,chief}.npz).
north_star bar: bit-exact action indices and advantage ordering, fp32 losses within 1e-4 rel."""
import numpy as np
import pytest
import torch

from cadre_amd import synth
from tests.helpers import fill_storages

pytestmark = pytest.mark.gpu
LOSS_TOL = 1e-4


def rel(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def make_agent(H, W, enc_seed=7, ppo_seed=11, command_num=4):
    from ppo_agent.agent import CadreAgent
    fh, fw = synth.feat_hw(H, W)
    cfg = dict(use_lstm=True, vae_device=0, device_num=0, vae_params="CoPM", measurement_dim=18,
               num_output=dict(steer=33, throttle=3), command_num=command_num, obs_hw=(H, W), weights_init="none",
               vae_state_dict=synth.encoder_state(fh, fw, enc_seed))
    steer = {i: (i - 16) / 16.0 for i in range(33)}
    thr = {0: [0, 0], 1: [0, 1], 2: [0.6, 0]}
    agent = CadreAgent(rank=0, model_cfg=cfg, frame=8, STEER_CONTROL=steer, THROTTLE_CONTROL=thr, ent_coeff=0.01,
                       value_coeff=0.1, clip_coeff=1.0, clip=0.1)
    agent.arena.load_numpy_state(synth.ppo_state(ppo_seed, command_num=command_num))
    return agent


def per_model(arena, buf, names, fn):
    out = []
    for n in names:
        v = arena.views(buf, n)
        out.append(fn([t.double() for t in v.values()]))
    return out


def test_learner_section_replay_vs_reference(golden):
    """F-update: train.py:76-110 replay (get_value -> GAE -> adv-norm -> 2 epochs x 2 minibatches of
    update_policy + per-model clip + Adam) on identical RolloutStorage inputs."""
    from ppo_agent.chief import chief_step
    from ppo_agent.models import Shared_grad_buffers
    from ppo_agent.storage import RolloutStorage
    g = golden("update")
    T, mbn, epochs = int(g["T"]), int(g["mbn"]), int(g["epochs"])
    names = [str(n) for n in g["names"]]
    agent = make_agent(84, 84, ppo_seed=int(g["ppo_seed"]))
    data = fill_storages(T, int(g["data_seed"]))
    stor = {}
    for hd in ("steer", "throttle"):
        s = RolloutStorage(T, mbn, 530, 8, 530, True, 0.99, 0.95)
        for k, v in data[hd].items():
            getattr(s, k).copy_(torch.from_numpy(v))
        s.to("cuda:0")
        stor[hd] = s
    nv_s, nv_t = agent.get_value(False, stor["steer"].get_last(), stor["throttle"].get_last())
    assert rel([nv_s.item(), nv_t.item()], g["next_value"]) < LOSS_TOL
    adv = {}
    for hd, nv in (("steer", nv_s), ("throttle", nv_t)):
        adv[hd] = stor[hd].compute_returns(nv.detach())
        mine = adv[hd].cpu().numpy()[:, 0]
        want = g["adv_" + hd][:, 0]
        assert rel(mine, want) < LOSS_TOL
    shared = Shared_grad_buffers(agent.model_dict, agent.device)
    opt = torch.optim.Adam([p for m in agent.model_dict.values() for p in m.parameters()], lr=3e-4)
    torch.manual_seed(int(g["torch_seed"]))
    step = 0
    for _ in range(epochs):
        g_s = stor["steer"].feed_forward_generator(adv["steer"])
        g_t = stor["throttle"].feed_forward_generator(adv["throttle"])
        for s_s, t_s in zip(g_s, g_t):
            l3 = agent.update_policy(s_s, t_s)
            assert rel(l3, g["losses"][step]) < LOSS_TOL, (step, l3, g["losses"][step])
            gn = per_model(agent.arena, agent.arena.grads, names, lambda ts: float(torch.sqrt(sum((t ** 2).sum() for t in ts))))
            assert rel(gn, g["grad_norms"][step]) < 5e-4, (step, gn, g["grad_norms"][step])
            gs = per_model(agent.arena, agent.arena.grads, names, lambda ts: float(sum(t.sum() for t in ts)))
            assert np.abs(np.array(gs) - g["grad_sums"][step]).max() < 5e-4 * np.abs(g["grad_norms"][step]).max()
            # .grad attributes are the arena views the reference's Shared_grad_buffers would read
            p = agent.model_dict["steer_lstm_0"].rnn.weight_hh
            assert p.grad is not None and p.grad.shape == p.shape and float(p.grad.abs().sum()) > 0
            shared.add_gradient(agent.model_dict)
            chief_step(shared, opt, 250.0)
            ps = per_model(agent.arena, agent.arena.params, names, lambda ts: float(sum(t.sum() for t in ts)))
            assert rel(ps, g["param_sums"][step]) < 1e-5, (step,)
            step += 1
    assert step == len(g["losses"])


@pytest.mark.parametrize("B,C", [(24, 4), (64, 4), (24, 3), (64, 2), (64, 6), (96, 16)])
def test_update_policy_matches_oracle_autograd_per_param(B, C):
    """Per-parameter gradient check of the explicit backward against oracle autograd; B=64 is the
    full C2 minibatch (T=128 / mini_batch_num=2).  C = command_num: the reference loops agent.py:170-182 over any count
    (agent_config.py ships 4); the row-sorted form (B >= 64) and the masked form (B = 24) both with other counts."""
    from oracle import ppo_ref
    agent = make_agent(84, 84, command_num=C)
    st0 = synth.ppo_state(11, command_num=C)
    params = ppo_ref.to_torch_params(st0, requires_grad=True)
    r = np.random.RandomState(5)
    S = 8
    samples, dsamples = [], []
    for hd, K in (("steer", 33), ("throttle", 3)):
        tup = (torch.from_numpy((r.standard_normal((S * B, 530)) * 0.5).astype(np.float32)),
               torch.from_numpy(r.randint(0, K, (B, 1)).astype(np.int64)),
               torch.from_numpy((0.3 * r.standard_normal((B, 1))).astype(np.float32)),
               torch.from_numpy(r.standard_normal((B, 1)).astype(np.float32)),
               torch.ones(B, 1),
               torch.from_numpy((-np.log(K) + 0.2 * r.standard_normal((B, 1))).astype(np.float32)),
               torch.from_numpy(r.standard_normal((B, 1)).astype(np.float32)),
               [torch.from_numpy((0.1 * r.standard_normal((B, 530))).astype(np.float32)),
                torch.from_numpy((0.1 * r.standard_normal((B, 530))).astype(np.float32))],
               torch.from_numpy(r.randint(0, C, (B, 1)).astype(np.int32)))
        samples.append(tup)
        dsamples.append(tuple(x.cuda() if not isinstance(x, list) else [y.cuda() for y in x] for x in tup))
    want = ppo_ref.update_policy(params, samples[0], samples[1], command_num=C)
    got = agent.update_policy(dsamples[0], dsamples[1])
    assert rel(got, want) < LOSS_TOL
    worst = 0.0
    for mn, d in params.items():
        gv = agent.arena.views(agent.arena.grads, mn)
        scale = max(float(p.grad.abs().max()) for p in d.values())
        for k, p in d.items():
            err = float((gv[k].cpu() - p.grad).abs().max()) / max(scale, 1e-12)
            worst = max(worst, err)
            assert err < 2e-4, (mn, k, err)
    print("worst per-parameter gradient error (rel. to model max |g|): %.2e" % worst)


def test_act_matches_reference(golden):
    """F-act: native 144x256 observations through encoder + LSTM + heads + sampling."""
    g = golden("act")
    agent = make_agent(144, 256)
    steps = synth.synth_rollout(6, 144, 256, seed=int(g["rollout_seed"]))
    torch.manual_seed(int(g["torch_seed"]))
    for i, td in enumerate(steps):
        obs = dict(rgb=td["rgb"], route_fig=td["route_fig"].copy(), measurements=td["measurements"],
                   command=td["command"])
        feat, a, lp, v, hid = agent.act(obs)
        assert set(np.unique(obs["route_fig"])) <= {0, 1}                    # caller's dict mutated (agent.py:51-54)
        assert float(hid[0].abs().sum()) == 0.0 and tuple(feat.shape) == (8, 530)
        assert rel(feat.cpu().numpy(), g["feats"][i]) < 2e-4
        assert [int(a[0]), int(a[1])] == list(g["actions"][i]), (i, g["margins"][2 * i:2 * i + 2])   # bit-exact
        assert a[0].dim() == 0 and a[0].dtype == torch.int64 and tuple(lp[0].shape) == (1, 1)
        assert rel([lp[0].item(), lp[1].item()], g["log_probs"][i]) < 1e-4      # (one bar for act-time heads: the C2 contract's)
        assert rel([v[0].item(), v[1].item()], g["values"][i]) < 1e-4
        ctl = agent.convert_action(a)
        assert len(ctl) == 3


def test_act_matches_reference_at_288(golden):
    """F-act at the size bench.py times (288 x 288: Winograd F(4x4) / F(3x3) / the fused layer-1 kernel are the fp32 defaults
    exactly here): agent.py:97-141 of the imported reference (tests/golden/act_288.npz, make_golden.py `act_288`) —
    features within 2e-4, action indices BIT-EXACT (smallest top-2 margin of p/q in the fixture: 1.1e-2, recorded in
    `margins`), log-probs and values within 1e-4 (the bar of the C2 contract test)."""
    g = golden("act_288")
    agent = make_agent(288, 288)
    steps = synth.synth_rollout(len(g["actions"]), 288, 288, seed=int(g["rollout_seed"]))
    torch.manual_seed(int(g["torch_seed"]))
    worst = 0.0
    for i, td in enumerate(steps):
        obs = dict(rgb=td["rgb"], route_fig=td["route_fig"].copy(), measurements=td["measurements"], command=td["command"])
        feat, a, lp, v, hid = agent.act(obs)
        e = rel(feat.cpu().numpy(), g["feats"][i])
        worst = max(worst, e)
        assert e < 2e-4, (i, e)
        assert [int(a[0]), int(a[1])] == list(g["actions"][i]), (i, g["margins"][2 * i:2 * i + 2])   # bit-exact
        assert rel([lp[0].item(), lp[1].item()], g["log_probs"][i]) < 1e-4
        assert rel([v[0].item(), v[1].item()], g["values"][i]) < 1e-4
    print("act at 288x288: worst feature error %.2e, min margin of the fixture %.2e" % (worst, float(g["margins"].min())))


def test_act_graph_equals_eager_launch_chain():
    """act() captured into one hipGraph per (window mode, command) (reference agent.py:114-141; ~150 launches per env
    step otherwise): identical features, actions, log-probs, values, route mutation and global-RNG consumption as the
    eager launch chain over 14 sliding-window steps (every command, first pass / warm-up / capture / replays), also when
    the eager path (get_latent_feature) runs in between and when the encoder workspace is re-shaped by a bigger batch."""
    import time
    H, W = 144, 256
    eager, graphed = make_agent(H, W), make_agent(H, W)
    eager.act_graph, graphed.act_graph = False, True
    steps = synth.synth_rollout(14, H, W, seed=31)
    for i in range(len(steps)):
        steps[i]["command"] = i % 4 if i < 12 else 1
    outs = []
    for ag in (eager, graphed):
        torch.manual_seed(77)
        res, t_steps = [], []
        for i, td in enumerate(steps):
            obs = dict(rgb=td["rgb"], route_fig=td["route_fig"].copy(), measurements=td["measurements"], command=td["command"])
            if ag is graphed and i == 9:      # the eager feature path in between: the latent cache stays consistent
                f2 = ag.get_latent_feature(dict(obs, route_fig=obs["route_fig"].copy()))
                assert tuple(f2.shape) == (8, 530)
            if ag is graphed and i == 11:     # a bigger encoder batch re-shapes workspace tensors the graphs were captured over
                big = torch.zeros(24, H, W, 3, dtype=torch.uint8, device="cuda"); bigr = torch.zeros(24, W, H, dtype=torch.uint8, device="cuda")
                ag.vae_model.max_frames = 24
                ag.vae_model.latent(big, bigr)
                ag.vae_model.latent(big[:16], bigr[:16])
            torch.cuda.synchronize(); t0 = time.perf_counter()
            feat, a, lp, v, hid = ag.act(obs)
            ai = [int(a[0]), int(a[1])]
            t_steps.append(time.perf_counter() - t0)
            res.append((feat.cpu(), ai, lp[0].cpu(), lp[1].cpu(), v[0].cpu(), v[1].cpu(), obs["route_fig"].copy()))
            assert a[0].dim() == 0 and a[0].dtype == torch.int64 and tuple(lp[0].shape) == (1, 1) and tuple(v[0].shape) == (1, 1)
            assert float(hid[0].abs().sum()) == 0.0
        outs.append((res, torch.rand(1).item(), t_steps))
    (r0, rng0, t0s), (r1, rng1, t1s) = outs
    assert rng0 == rng1                                               # same global-generator consumption
    for i, (x, y) in enumerate(zip(r0, r1)):
        assert torch.equal(x[0], y[0]), i
        assert x[1] == y[1], i
        for k in (2, 3, 4, 5):
            assert torch.equal(x[k], y[k]), (i, k)
        assert np.array_equal(x[6], y[6]), i
    assert len(graphed._ag["graphs"]) >= 1 and graphed._ag["gen"] > 0      # re-captured after the workspace moved
    print("act() per env step (144x256, incl. H2D / D2H): eager %.2f ms, hipGraph replay %.2f ms" % (
        1e3 * np.median(t0s[4:9]), 1e3 * np.median(t1s[6:9] + t1s[12:])))


def test_snapshot_roundtrip(tmp_path):
    agent = make_agent(84, 84)
    p = str(tmp_path / "ppo_model_0.pt")
    agent.save_snapshot(p)
    saved = torch.load(p, weights_only=False)
    assert sorted(saved) == sorted(["%s_%d" % (k, c) for c in range(4) for k in ("throttle_ppo", "steer_ppo", "steer_lstm")])
    before = agent.arena.params.clone()
    agent.arena.params.mul_(0.5)
    agent.load_snapshot(p, None)
    names = [n for n in agent.model_dict if not n.startswith("throttle_lstm")]
    for n in names:
        for k, v in agent.arena.views(agent.arena.params, n).items():
            assert torch.equal(v, agent.arena.views(before, n)[k])
    with pytest.raises(ImportError):
        agent.load_snapshot(str(tmp_path / "missing.pt"), None)


@pytest.mark.parametrize("C", [2, 6])
def test_act_get_value_and_snapshot_with_other_command_counts(C, tmp_path):
    """command_num other than the shipped 4 (models.py:201-217 builds that many nets per head, agent.py:113-141 picks by
    obs['command']): act() against the oracle chain on the same observations and exponential draws — indices bit-exact, log-prob and
    value at the C2 bar —, get_value of the last window, and the snapshot's key set (agent.py:458-484: no throttle_lstm keys)."""
    from oracle import encoder_ref, ppo_ref
    agent = make_agent(84, 84, command_num=C)
    sd = synth.encoder_state(3, 3, 7)
    st0 = synth.ppo_state(11, command_num=C)
    params = ppo_ref.to_torch_params(st0)
    steps = synth.synth_rollout(2 * C, 84, 84, seed=19)
    for i, td in enumerate(steps):
        c = i % C                                                  # every command net once or twice
        want = encoder_ref.latent_feature(td["rgb"], td["route_fig"], td["measurements"], sd)
        torch.manual_seed(600 + i)
        feat, a, lp, v, hid = agent.act(dict(rgb=td["rgb"], route_fig=td["route_fig"].copy(), measurements=td["measurements"], command=c))
        assert rel(feat.cpu().numpy(), want.numpy()) < 2e-4
        torch.manual_seed(600 + i)
        for hd, K, j in (("steer", 33, 0), ("throttle", 3, 1)):
            with torch.no_grad():
                x, _ = ppo_ref.lstm_forward(want, (torch.zeros(1, 530), torch.zeros(1, 530)), params["%s_lstm_%d" % (hd, c)])
                logits = ppo_ref.categorical_logits(x, params["%s_ppo_%d" % (hd, c)])
                val = ppo_ref.mlp3(x, params["%s_ppo_%d" % (hd, c)], "critic")
            q = torch.empty(1, K).exponential_(1)
            a_ref = int(ppo_ref.sample_from_logits(logits, q))
            assert int(a[j]) == a_ref, (i, hd)
            assert abs(lp[j].item() - logits[0, a_ref].item()) < 1e-4 * max(1.0, abs(logits[0, a_ref].item()))
            assert abs(v[j].item() - val.item()) < 1e-4 * max(1.0, abs(val.item()))
        nv = agent.get_value(False, (feat, c), (feat, c))
        nd = agent.get_value(False, (feat, torch.tensor(c, device=feat.device)), (feat, torch.tensor(c, device=feat.device)))
        assert all(torch.equal(x.cpu(), y.cpu()) for x, y in zip(nv, nd))
        for j, hd in enumerate(("steer", "throttle")):
            with torch.no_grad():
                x, _ = ppo_ref.lstm_forward(want, (torch.zeros(1, 530), torch.zeros(1, 530)), params["%s_lstm_%d" % (hd, c)])
                val = ppo_ref.mlp3(x, params["%s_ppo_%d" % (hd, c)], "critic")
            assert abs(float(nv[j]) - val.item()) < 1e-4 * max(1.0, abs(val.item()))
    p = str(tmp_path / "ppo_model_0.pt")
    agent.save_snapshot(p)
    saved = torch.load(p, weights_only=False)
    assert sorted(saved) == sorted(["%s_%d" % (k, c) for c in range(C) for k in ("throttle_ppo", "steer_ppo", "steer_lstm")])
    before = agent.arena.params.clone()
    agent.arena.params.mul_(0.5)
    agent.load_snapshot(p, None)
    for n in agent.model_dict:
        if not n.startswith("throttle_lstm"):
            for k, t in agent.arena.views(agent.arena.params, n).items():
                assert torch.equal(t, agent.arena.views(before, n)[k])


def test_fused_gather_path_equals_tuple_path():
    """update_policy_from_storages (gather kernel + hipGraph replay, 2 workers concatenated) gives the
    same losses and gradients as feed_forward_generator tuples -> update_policy(workers=2)."""
    from ppo_agent.storage import RolloutStorage
    agent = make_agent(84, 84)
    T = 16
    stor = []
    for w in range(2):
        data = fill_storages(T, 300 + w)
        pair = []
        for hd in ("steer", "throttle"):
            s = RolloutStorage(T, 2, 530, 8, 530, True, 0.99, 0.95)
            for k, v in data[hd].items():
                getattr(s, k).copy_(torch.from_numpy(v))
            s.to("cuda:0")
            s.compute_returns(torch.tensor([0.1 * (w + 1)]))
            pair.append(s)
        stor.append(pair)
    idx = [torch.randperm(T)[:8] for _ in range(4)]
    for rep in range(3):                                   # eager, eager+capture, graph replay
        tup_s = [stor[w][0].gather(idx[2 * w], stor[w][0].advantages) for w in range(2)]
        tup_t = [stor[w][1].gather(idx[2 * w + 1], stor[w][1].advantages) for w in range(2)]

        def cat(ts):
            B = ts[0][1].shape[0]
            out = [torch.stack([t[0].reshape(8, B, -1) for t in ts], 1).reshape(8 * 2 * B, -1)]
            out += [torch.cat([t[k] for t in ts], 0) for k in (1, 2, 3, 4, 5, 6)]
            out.append([torch.cat([t[7][0] for t in ts], 0), torch.cat([t[7][1] for t in ts], 0)])
            out.append(torch.cat([t[8] for t in ts], 0))
            return tuple(out)
        l_a = agent.update_policy(cat(tup_s), cat(tup_t), workers=2)
        g_a = agent.arena.grads.clone()
        l_b = agent.update_policy_from_storages(
            [(stor[w][0], idx[2 * w], stor[w][0].advantages, stor[w][1], idx[2 * w + 1], stor[w][1].advantages)
             for w in range(2)])
        assert l_a == l_b and torch.equal(g_a, agent.arena.grads), rep


def test_latent_cache_is_bit_identical():
    """Sliding-window latent cache on vs off: identical features, actions, log-probs over a rollout
    whose windows really slide (and one reset in the middle that must miss the cache)."""
    a_on, a_off = make_agent(84, 84), make_agent(84, 84)
    a_off.latent_cache = False
    steps = synth.synth_rollout(5, 84, 84, seed=9) + synth.synth_rollout(3, 84, 84, seed=10)
    for ag in (a_on, a_off):
        torch.manual_seed(3)
        ag._out = []
        for td in steps:
            obs = dict(rgb=td["rgb"], route_fig=td["route_fig"].copy(), measurements=td["measurements"],
                       command=td["command"])
            feat, a, lp, v, _ = ag.act(obs)
            ag._out.append((feat.clone(), int(a[0]), int(a[1]), lp[0].item(), lp[1].item(), obs["route_fig"].copy()))
    for x, y in zip(a_on._out, a_off._out):
        assert torch.equal(x[0], y[0]) and x[1:5] == y[1:5] and np.array_equal(x[5], y[5])


def test_ragged_minibatches_and_done_vs_oracle():
    """T=50, mini_batch_num=3 -> minibatches of 16,16,16,2 (BatchSampler drop_last=False,
    storage.py:94-97) and the done=True bootstrap (agent.py:145-147): HIP learner section vs the
    oracle replay on identical storages."""
    from ppo_agent.chief import chief_step
    from ppo_agent.models import Shared_grad_buffers
    from ppo_agent.storage import RolloutStorage
    from ppo_agent.train import learner_section
    from tests.helpers import oracle_learner_replay
    T, mbn = 50, 3
    g = dict(T=T, mbn=mbn, epochs=1, ppo_seed=11, data_seed=91, torch_seed=5,
             names=np.array(sorted(synth.ppo_state(11))))
    want = oracle_learner_replay(g)
    agent = make_agent(84, 84)
    data = fill_storages(T, 91)
    stor = []
    for hd in ("steer", "throttle"):
        s = RolloutStorage(T, mbn, 530, 8, 530, True, 0.99, 0.95)
        for k, v in data[hd].items():
            getattr(s, k).copy_(torch.from_numpy(v))
        s.to("cuda:0")
        stor.append(s)
    shared = Shared_grad_buffers(agent.model_dict, agent.device)
    torch.manual_seed(5)
    cfg = dict(use_adv_norm=True, ppo_epoch=1, max_grad_norm=250.0)
    vl, pl, el = learner_section(agent, stor[0], stor[1], False, cfg, shared)
    got = np.array([vl, pl, el]).T
    assert got.shape == (4, 3)
    assert rel(got, np.array(want["losses"])) < LOSS_TOL
    names = [str(n) for n in g["names"]]
    ps = per_model(agent.arena, agent.arena.params, names, lambda ts: float(sum(t.sum() for t in ts)))
    assert rel(ps, want["param_sums"][-1]) < 1e-5
    # done=True: zero bootstrap values on the host, like the reference
    nv = agent.get_value(True, stor[0].get_last(), stor[1].get_last())
    assert all(float(v) == 0.0 and not v.is_cuda for v in nv)


def test_module_level_api_matches_oracle():
    """LSTM.forward / Model.get_value / Model.evaluate_actions / Model.act called directly on the
    modules of model_dict (reference ppo_agent/models.py:139-208), against the oracle."""
    from oracle import ppo_ref
    agent = make_agent(84, 84)
    st0 = synth.ppo_state(11)
    P = ppo_ref.to_torch_params(st0)
    r = np.random.RandomState(2)
    N, T = 5, 8
    x = torch.from_numpy((r.standard_normal((T * N, 530)) * 0.5).astype(np.float32))
    h0 = torch.from_numpy((r.standard_normal((N, 530)) * 0.1).astype(np.float32))
    c0 = torch.from_numpy((r.standard_normal((N, 530)) * 0.1).astype(np.float32))
    lstm = agent.model_dict["throttle_lstm_2"]
    h, (h2, c2) = lstm(x.cuda(), (h0.cuda(), c0.cuda()))
    wh, (_, wc) = ppo_ref.lstm_forward(x, (h0, c0), P["throttle_lstm_2"])
    assert rel(h.detach().cpu().numpy(), wh.numpy()) < 1e-5 and rel(c2.detach().cpu().numpy(), wc.numpy()) < 1e-5
    h1, _ = lstm(x[:N].cuda(), (h0.cuda(), c0.cuda()))                     # single-step branch
    w1, _ = ppo_ref.lstm_forward(x[:N], (h0, c0), P["throttle_lstm_2"])
    assert rel(h1.detach().cpu().numpy(), w1.numpy()) < 1e-5
    model = agent.model_dict["steer_ppo_1"]
    feat = wh
    acts = torch.from_numpy(r.randint(0, 33, (N, 1)))
    v, lp, ent = model.evaluate_actions(feat.cuda(), acts.cuda())
    wv, wlp, went = ppo_ref.evaluate_actions(feat, acts, P["steer_ppo_1"])
    assert rel(v.detach().cpu().numpy(), wv.numpy()) < 1e-5 and rel(lp.detach().cpu().numpy(), wlp.numpy()) < 1e-5
    assert rel(ent.detach().cpu().numpy(), went.numpy()) < 1e-5
    assert rel(model.get_value(feat.cuda()).detach().cpu().numpy(), wv.numpy()) < 1e-5
    with torch.no_grad():                                                  # the no-graph path: same values
        v0, lp0, ent0 = model.evaluate_actions(feat.cuda(), acts.cuda())
        assert not v0.requires_grad and rel(v0.cpu().numpy(), wv.numpy()) < 1e-5 and rel(lp0.cpu().numpy(), wlp.numpy()) < 1e-5
        assert rel(ent0.cpu().numpy(), went.numpy()) < 1e-5
        hn, _ = lstm(x.cuda(), (h0.cuda(), c0.cuda()))
        assert not hn.requires_grad and rel(hn.cpu().numpy(), wh.numpy()) < 1e-5
    torch.manual_seed(7)
    value, action, _f = model.act(feat.cuda())
    torch.manual_seed(7)
    q = torch.empty(N, 33).exponential_(1)
    want_a = ppo_ref.sample_from_logits(ppo_ref.categorical_logits(feat, P["steer_ppo_1"]), q)
    assert torch.equal(action.cpu(), want_a)
    assert rel(model.get_log_probs(action).cpu().numpy(),
               ppo_ref.categorical_logits(feat, P["steer_ppo_1"]).gather(1, want_a.view(-1, 1)).numpy()) < 1e-5


def test_module_autograd_matches_torch():
    """A caller that builds its OWN loss on the stand-alone modules (reference ppo_agent/models.py:139-152 LSTM.forward,
    199-208 Model.evaluate_actions, 195-197 get_value — all differentiable there): gradients of every parameter and of
    the inputs equal torch autograd of the oracle restatement, and land in the parameters' `.grad` (arena views)."""
    from oracle import ppo_ref
    agent = make_agent(84, 84)
    st0 = synth.ppo_state(11)
    P = ppo_ref.to_torch_params(st0, requires_grad=True)
    r = np.random.RandomState(5)
    f32 = lambda *shape, s=1.0: torch.from_numpy((r.standard_normal(shape) * s).astype(np.float32))
    for N, T in ((5, 8), (33, 3), (4, 1)):
        agent.arena.grads.zero_()
        for d in P.values():
            for p in d.values():
                p.grad = None
        x, h0, c0 = f32(T * N, 530, s=0.5), f32(N, 530, s=0.1), f32(N, 530, s=0.1)
        wh_, wc_ = f32(N, 530), f32(N, 530)
        # ---- LSTM: loss = <h_T, wh> + <c_T, wc>
        xd, hd, cd = (t.cuda().requires_grad_() for t in (x, h0, c0))
        lstm = agent.model_dict["throttle_lstm_2"]
        h, (_h, c) = lstm(xd, (hd, cd))
        ((h * wh_.cuda()).sum() + (c * wc_.cuda()).sum()).backward()
        xr, hr, cr = (t.clone().requires_grad_() for t in (x, h0, c0))
        rh, (_rh, rc) = ppo_ref.lstm_forward(xr, (hr, cr), P["throttle_lstm_2"])
        ((rh * wh_).sum() + (rc * wc_).sum()).backward()
        errs = {"x": rel(xd.grad.cpu().numpy(), xr.grad.numpy()), "h0": rel(hd.grad.cpu().numpy(), hr.grad.numpy()),
                "c0": rel(cd.grad.cpu().numpy(), cr.grad.numpy())}
        for k, p in lstm.named_parameters():
            errs[k] = rel(p.grad.cpu().numpy(), P["throttle_lstm_2"][k].grad.numpy())
        # ---- towers: loss = <value, wv> + <log_prob, wl> + 0.3 sum(entropy) + sum(get_value^2)
        model = agent.model_dict["steer_ppo_1"]
        feat, acts = f32(N, 530, s=0.5), torch.from_numpy(r.randint(0, 33, (N, 1)))
        wv, wl = f32(N, 1), f32(N, 1)
        fd = feat.cuda().requires_grad_()
        v, lp, ent = model.evaluate_actions(fd, acts.cuda())
        ((v * wv.cuda()).sum() + (lp * wl.cuda()).sum() + 0.3 * ent.sum() + (model.get_value(fd) ** 2).sum()).backward()
        fr = feat.clone().requires_grad_()
        rv, rlp, rent = ppo_ref.evaluate_actions(fr, acts, P["steer_ppo_1"])
        ((rv * wv).sum() + (rlp * wl).sum() + 0.3 * rent.sum() + (ppo_ref.mlp3(fr, P["steer_ppo_1"], "critic") ** 2).sum()).backward()
        errs["feat"] = rel(fd.grad.cpu().numpy(), fr.grad.numpy())
        for k, p in model.named_parameters():
            errs["ppo." + k] = rel(p.grad.cpu().numpy(), P["steer_ppo_1"][k].grad.numpy())
            assert p.grad.data_ptr() >= agent.arena.grads.data_ptr() and \
                p.grad.data_ptr() < agent.arena.grads.data_ptr() + 4 * agent.arena.grads.numel()      # still the arena's view
        print("module autograd N=%d T=%d: worst %s" % (N, T, max(errs.items(), key=lambda kv: kv[1])))
        assert max(errs.values()) < 2e-5, errs
        # no other net's gradients were touched
        other = agent.arena.views(agent.arena.grads, "steer_lstm_0")
        assert all(float(t.abs().max()) == 0.0 for t in other.values())


def test_reference_topology_shared_nets_and_worker_agent():
    """reference main.py:38-70 shape in one process: a shared model_dict (create_model(load_vae=False))
    owned by the chief + a worker agent with its own nets.  add_gradient accumulates into the shared
    arena, chief_step updates the shared nets, update_model pulls them — same parameters as the
    single-arena in-process path."""
    from ppo_agent.chief import chief_step
    from ppo_agent.models import Shared_grad_buffers, arena_of, create_model
    cfg = dict(use_lstm=True, vae_device=0, device_num=0, vae_params="CoPM", measurement_dim=18,
               num_output=dict(steer=33, throttle=3), command_num=4, weights_init="none")
    _none, shared = create_model(cfg, load_vae=False)
    arena_of(shared).load_numpy_state(synth.ppo_state(11))
    worker, solo = make_agent(84, 84), make_agent(84, 84)
    worker.update_model(shared)
    assert torch.equal(worker.arena.params, arena_of(shared).params)
    bufs = Shared_grad_buffers(shared, worker.device)
    solo_bufs = Shared_grad_buffers(solo.model_dict, solo.device)
    assert set(bufs.grads) == {"%s_%s_grad" % (m, n) for m, mod in shared.items() for n, _ in mod.named_parameters()}
    r = np.random.RandomState(3)
    B = 8
    samp = []
    for K in (33, 3):
        samp.append((torch.from_numpy((r.standard_normal((8 * B, 530)) * 0.5).astype(np.float32)).cuda(),
                     torch.from_numpy(r.randint(0, K, (B, 1))).cuda(),
                     torch.from_numpy((0.3 * r.standard_normal((B, 1))).astype(np.float32)).cuda(),
                     torch.from_numpy(r.standard_normal((B, 1)).astype(np.float32)).cuda(), torch.ones(B, 1).cuda(),
                     torch.from_numpy((-np.log(K) + 0.2 * r.standard_normal((B, 1))).astype(np.float32)).cuda(),
                     torch.from_numpy(r.standard_normal((B, 1)).astype(np.float32)).cuda(),
                     [torch.zeros(B, 530).cuda(), torch.zeros(B, 530).cuda()],
                     torch.from_numpy(r.randint(0, 4, (B, 1)).astype(np.int32)).cuda()))
    opt = torch.optim.Adam([p for m in shared.values() for p in m.parameters()], lr=3e-4)
    for _ in range(3):
        l_w = worker.update_policy(samp[0], samp[1])
        bufs.add_gradient(worker.model_dict)
        chief_step(bufs, opt, 250.0)
        worker.update_model(shared)
        l_s = solo.update_policy(samp[0], samp[1])
        solo_bufs.add_gradient(solo.model_dict)
        chief_step(solo_bufs, None, 250.0)
        assert l_w == l_s
        assert torch.equal(worker.arena.params, solo.arena.params)
    assert float(arena_of(shared).grads.abs().max()) == 0.0          # chief resets the shared buffers (chief.py:22)


@pytest.mark.parametrize("Bw,nW", [(128, 1), (64, 4), (64, 1), (96, 1)])
def test_row_sorted_update_equals_unsorted(Bw, nW):
    """Minibatches >= 64 rows per head: rows sorted by command + GEMM tile skipping (stale rows of other
    command nets are masked to exact zeros in the backward) must reproduce the unsorted path: same
    losses, same gradients up to fp32 summation order — over several updates so that skipped tiles
    really hold stale data from earlier minibatches."""
    from ppo_agent.storage import RolloutStorage
    a_s, a_u, a_p = make_agent(84, 84), make_agent(84, 84), make_agent(84, 84)
    a_u.learner.use_sorted = False
    from cadre_amd import hip as _hip
    a_p.learner.persistent_lstm = _hip.has_ab_kernels()     # A/B build: forward LSTM as one persistent launch (cadre_lstm_seq_fwd), bit-identical
    assert a_s.learner.sorted_rows(Bw * nW) and not a_u.learner.sorted_rows(Bw * nW)
    T = 2 * Bw
    stor = []
    for w in range(nW):
        data = fill_storages(T, 700 + w)
        pair = []
        for hd in ("steer", "throttle"):
            s = RolloutStorage(T, 2, 530, 8, 530, True, 0.99, 0.95)
            for k, v in data[hd].items():
                getattr(s, k).copy_(torch.from_numpy(v))
            if w == 0 and hd == "steer":
                s.command[: T // 2] = 2                      # a heavily unbalanced command mix for one head
            s.to("cuda:0")
            s.compute_returns(torch.tensor([0.05 * (w + 1)]))
            pair.append(s)
        stor.append(pair)
    g = torch.Generator().manual_seed(1)
    for it in range(4):
        idx = [torch.randperm(T, generator=g)[:Bw] for _ in range(2 * nW)]
        batches = [(stor[w][0], idx[2 * w], stor[w][0].advantages, stor[w][1], idx[2 * w + 1], stor[w][1].advantages)
                   for w in range(nW)]
        l_s = a_s.update_policy_from_storages(batches)
        l_u = a_u.update_policy_from_storages(batches)
        assert rel(l_s, l_u) < 1e-5, (it, l_s, l_u)
        gs, gu = a_s.arena.grads, a_u.arena.grads
        assert float((gs - gu).abs().max() / gu.abs().max()) < 2e-5, it
        assert torch.isfinite(gs).all()
        l_p = a_p.update_policy_from_storages(batches)
        assert torch.equal(torch.as_tensor(l_p).cpu(), torch.as_tensor(l_s).cpu()) and torch.equal(a_p.arena.grads, gs), it
        assert int(a_p.learner.workspace(Bw * nW)["sync"][-1]) == 0      # no bounded spin ran out


@pytest.mark.parametrize("Bw,nW", [(64, 1), (64, 4), (48, 2)])
def test_one_launch_gather_sort_equals_three_launches(Bw, nW, monkeypatch):
    """cadre_gather_sorted_multi (round 6: gather + stable counting sort by command + placement in one launch) against the
    three-launch form it replaces (cadre_gather_minibatch_multi -> cadre_sort_rows_by_command -> cadre_permute_minibatch,
    CADRE_GATHER_SORTED=0): every workspace tensor of the minibatch, the run table, the positions, the losses and the whole
    gradient arena bit for bit, over several minibatches (agent.py:166-237: the per-command nets each read one run of rows)."""
    from ppo_agent.storage import RolloutStorage
    a_1, a_3 = make_agent(84, 84), make_agent(84, 84)
    assert a_1.learner.sorted_rows(Bw * nW)
    T = 2 * Bw
    stor = []
    for w in range(nW):
        data = fill_storages(T, 1700 + w)
        pair = []
        for hd in ("steer", "throttle"):
            s = RolloutStorage(T, 2, 530, 8, 530, True, 0.99, 0.95)
            for k, v in data[hd].items():
                getattr(s, k).copy_(torch.from_numpy(v))
            if w == 0 and hd == "throttle":
                s.command[: T // 2] = 3                      # an unbalanced command mix (one command may own no row at all)
            s.to("cuda:0")
            s.compute_returns(torch.tensor([0.05 * (w + 1)]))
            pair.append(s)
        stor.append(pair)
    g = torch.Generator().manual_seed(3)
    keys = ("X", "h0", "c0", "actions", "commands", "old_values", "returns", "old_logp", "adv", "seg", "pos")
    for it in range(3):
        idx = [torch.randperm(T, generator=g)[:Bw] for _ in range(2 * nW)]
        batches = [(stor[w][0], idx[2 * w], stor[w][0].advantages, stor[w][1], idx[2 * w + 1], stor[w][1].advantages)
                   for w in range(nW)]
        monkeypatch.setenv("CADRE_GATHER_SORTED", "1")
        l1 = a_1.update_policy_from_storages(batches)
        monkeypatch.setenv("CADRE_GATHER_SORTED", "0")
        l3 = a_3.update_policy_from_storages(batches)
        w1, w3 = a_1.learner.workspace(Bw * nW), a_3.learner.workspace(Bw * nW)
        for k in keys:
            assert torch.equal(w1[k], w3[k]), (it, k)
        assert torch.equal(torch.as_tensor(l1).cpu(), torch.as_tensor(l3).cpu()), it
        assert torch.equal(a_1.arena.grads, a_3.arena.grads), it
    seg = a_1.learner.workspace(Bw * nW)["seg"].cpu().view(2, -1, 2)
    assert int(seg[0, :, 1].sum()) == Bw * nW and int(seg[1, :, 1].sum()) == Bw * nW


def test_optimiser_step_that_writes_the_weight_copies_is_bit_identical():
    """CADRE_ADAM_PACK=1 (opt-in): clip_adam() writes the fragment-order W_hh copies itself and the next update carries
    no packing launch.  Several update -> optimiser-step rounds, plus a parameter write from outside in between (which
    must bring the separate packing launch back), against the default path: same losses, gradients, parameters."""
    from ppo_agent.storage import RolloutStorage
    a_f, a_d = make_agent(84, 84), make_agent(84, 84)
    a_f.learner.fused_pack = True
    assert not a_d.learner.fused_pack
    T, Bw = 128, 64
    data = fill_storages(T, 913)
    pair = []
    for hd in ("steer", "throttle"):
        s = RolloutStorage(T, 2, 530, 8, 530, True, 0.99, 0.95)
        for k, v in data[hd].items():
            getattr(s, k).copy_(torch.from_numpy(v))
        s.to("cuda:0")
        s.compute_returns(torch.tensor([0.1]))
        pair.append(s)
    g = torch.Generator().manual_seed(5)
    for it in range(6):
        idx = [torch.randperm(T, generator=g)[:Bw] for _ in range(2)]
        batches = [(pair[0], idx[0], pair[0].advantages, pair[1], idx[1], pair[1].advantages)]
        l_f, l_d = a_f.update_policy_from_storages(batches), a_d.update_policy_from_storages(batches)
        assert torch.equal(torch.as_tensor(l_f).cpu(), torch.as_tensor(l_d).cpu()), it
        assert torch.equal(a_f.arena.grads, a_d.arena.grads), it
        if it >= 1:
            assert a_f.learner._adam_fresh is not None       # the step before this update left the copies current
        a_f.learner.clip_adam(3e-3, 250.0)
        a_d.learner.clip_adam(3e-3, 250.0)
        assert torch.equal(a_f.arena.params, a_d.arena.params), it
        if it == 3:                                          # parameters replaced behind the learner's back (load / broadcast)
            for a in (a_f, a_d):
                a.arena.params.mul_(0.5)


@pytest.mark.parametrize("C", [4, 3, 6])
def test_full_size_learner_section_vs_oracle(C):
    """BASELINE C2 sizes (T=128, mini_batch_num=2 -> minibatch 64, fused-gather + hipGraph path):
    get_value, GAE, advantage normalisation and one epoch of updates + clip + Adam vs the oracle.  C = command_num: the count the
    reference ships (4) and two others (the reference builds command_num nets per head, models.py:201-217, and loops
    agent.py:170-182 over them)."""
    from ppo_agent.models import Shared_grad_buffers
    from ppo_agent.storage import RolloutStorage
    from ppo_agent.train import learner_section
    from tests.helpers import oracle_learner_replay
    T, mbn = 128, 2
    g = dict(T=T, mbn=mbn, epochs=1, ppo_seed=11, data_seed=123, torch_seed=8, command_num=C,
             names=np.array(sorted(synth.ppo_state(11, command_num=C))))
    data = fill_storages(T, 123)
    if C != 4:
        for i, hd in enumerate(("steer", "throttle")):
            data[hd]["command"] = np.random.RandomState(40 + i).randint(0, C, (T + 1, 1)).astype(np.int32)
    want = oracle_learner_replay(g, data=data)
    agent = make_agent(84, 84, command_num=C)
    assert len(agent.model_dict) == 4 * C
    stor = []
    for hd in ("steer", "throttle"):
        s = RolloutStorage(T, mbn, 530, 8, 530, True, 0.99, 0.95)
        for k, v in data[hd].items():
            getattr(s, k).copy_(torch.from_numpy(v))
        s.to("cuda:0")
        stor.append(s)
    shared = Shared_grad_buffers(agent.model_dict, agent.device)
    torch.manual_seed(8)
    vl, pl, el = learner_section(agent, stor[0], stor[1], False, dict(use_adv_norm=True, ppo_epoch=1, max_grad_norm=250.0), shared)
    assert rel(np.array([vl, pl, el]).T, np.array(want["losses"])) < LOSS_TOL
    for hd, s in zip(("steer", "throttle"), stor):
        mine = s.advantages.cpu().numpy()[:, 0]
        assert np.array_equal(np.argsort(mine, kind="stable"), np.argsort(want["adv_" + hd][:, 0], kind="stable"))
    names = [str(n) for n in g["names"]]
    ps = per_model(agent.arena, agent.arena.params, names, lambda ts: float(sum(t.sum() for t in ts)))
    assert rel(ps, want["param_sums"][-1]) < 1e-5


def test_ensemble_act_equals_per_agent_loop():
    """eval.py:52-60: the ensemble with one shared encoder pass returns what the per-agent act() loop
    returns (actions bit-exact, same RNG consumption), and refuses agents with different encoders."""
    from ppo_agent.agent import CadreAgent
    steps = synth.synth_rollout(3, 84, 84, seed=21)
    group_a = [make_agent(84, 84, ppo_seed=11 + i) for i in range(3)]
    group_b = [make_agent(84, 84, ppo_seed=11 + i) for i in range(3)]

    def obs_of(td):
        return dict(rgb=td["rgb"], route_fig=td["route_fig"].copy(), measurements=td["measurements"], command=td["command"])
    torch.manual_seed(5)
    loop = []
    for td in steps:
        o = obs_of(td)
        loop.append([ag.act(o) for ag in group_a])
    torch.manual_seed(5)
    for i, td in enumerate(steps):
        o = obs_of(td)
        ens = CadreAgent.ensemble_act(group_b, o)
        assert set(np.unique(o["route_fig"])) <= {0, 1}
        for (f0, a0, lp0, v0, _), (f1, a1, lp1, v1, _) in zip(loop[i], ens):
            assert torch.equal(f0, f1)
            assert [int(a0[0]), int(a0[1])] == [int(a1[0]), int(a1[1])]
            assert torch.equal(lp0[0], lp1[0]) and torch.equal(lp0[1], lp1[1])
            assert torch.equal(v0[0], v1[0]) and torch.equal(v0[1], v1[1])
        assert group_b[0].avg_action([e[1] for e in ens]) == group_a[0].avg_action([e[1] for e in loop[i]])
    other = make_agent(84, 84, enc_seed=8)
    with pytest.raises(ValueError):
        CadreAgent.ensemble_act([group_b[0], other], obs_of(steps[0]))


def test_get_values_equals_per_worker_get_value():
    """get_value (agent.py:143-164) for the W workers of a GPU in one pass over all command nets equals W separate
    calls: every command, ragged feature rows, both heads."""
    agent = make_agent(84, 84, ppo_seed=13)
    g = torch.Generator().manual_seed(3)
    W, S, D = 5, 8, 530
    batches = []
    for i in range(W):
        fs = (torch.randn(S, D, generator=g) * 0.5).cuda()
        ft = (torch.randn(S, D, generator=g) * 0.5).cuda()
        batches.append(((fs, i % 4), (ft, (i + 2) % 4)))
    want = [agent.get_value(False, sb, tb) for sb, tb in batches]
    got = agent.get_values(batches)
    # commands as 0-dim device tensors (RolloutStorage.get_last(as_tensor=True)): the net is picked on the device, same bits
    dev_batches = [((fs, torch.tensor(cs, dtype=torch.int32, device="cuda")), (ft, torch.tensor(ct, dtype=torch.int32, device="cuda")))
                   for (fs, cs), (ft, ct) in batches]
    got_dev = agent.get_values(dev_batches)
    one_dev = agent.get_value(False, dev_batches[2][0], dev_batches[2][1])
    for i, ((ws, wt), (gs, gt)) in enumerate(zip(want, got)):
        assert gs.shape == ws.shape == (1, 1) and gt.shape == wt.shape
        assert abs(float(ws) - float(gs)) <= 1e-6 * max(1.0, abs(float(ws)))
        assert abs(float(wt) - float(gt)) <= 1e-6 * max(1.0, abs(float(wt)))
        assert got_dev[i][0].shape == (1, 1) and torch.equal(got_dev[i][0], gs) and torch.equal(got_dev[i][1], gt)
    assert abs(float(one_dev[0]) - float(want[2][0])) <= 1e-6 * max(1.0, abs(float(want[2][0])))
    assert abs(float(one_dev[1]) - float(want[2][1])) <= 1e-6 * max(1.0, abs(float(want[2][1])))
