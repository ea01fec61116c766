"""GPU: the data-parallel learner (BASELINE config C4's math) proven on the one GPU a box has.

Two ranks — fresh processes on cuda:0, `gloo` process group over device tensors — each play `bench.learner_round`
(joint encode -> get_values -> GAE -> 8 x [batched update_policy + gradient exchange + per-model clip + Adam]) on a
C1-sized config with W = 2 workers and distinct worker seeds (tests/dp_ranks_driver.py).  Asserted:

  * both ranks end the round with BIT-IDENTICAL parameter arenas (replicated optimiser, no drift),
  * those parameters equal the oracle's "SUM over all four workers of ppo_ref.update_policy gradients -> per-model
    clip -> Adam" per step (reference ppo_agent/models.py:231-239, chief.py:13-21, train.py:76-110), starting from the
    raw synthetic observations (oracle encoder, get_value, GAE, sampler streams),
  * the three forms of the exchange — one all-reduce, MLP bucket beside the LSTM backward, reduce-scatter + sharded
    clip/Adam + all-gather — leave the same parameters, after the eager round and after a round of hipGraph replays.
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from cadre_amd import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LOSS_TOL = 1e-4           # north_star: fp32 losses within 1e-4 relative
PARAM_SUM_TOL = 1e-5      # per-model parameter sums after 8 clip + Adam steps (same bar as tests/test_learner_gpu.py)


def rel(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def run_ranks(out_dir, mode, world=2):
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""), HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, "-m", "tests.dp_ranks_driver", str(out_dir), mode, str(world)], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=1800)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("DP_RESULT ")]
    assert p.returncode == 0 and lines, "ranks failed rc=%s\nstdout:\n%s\nstderr:\n%s" % (p.returncode, p.stdout[-3000:], p.stderr[-8000:])
    res = json.loads(lines[-1][len("DP_RESULT "):])
    assert res["exitcodes"] == [0] * world
    return [np.load(os.path.join(str(out_dir), "rank%d.npz" % r)) for r in range(world)]


def oracle_round(world, feats_check=None):
    """The reference's learner section for world x 2 workers feeding ONE chief: every worker's update_policy
    gradients summed, per-model clip, one Adam step — 8 times (oracle/, CPU)."""
    import bench
    from oracle import encoder_ref, ppo_ref
    from tests.dp_ranks_driver import CFG
    T, H, W, nW = CFG["T"], CFG["H"], CFG["W"], CFG["workers"]
    enc_sd = synth.encoder_state(*synth.feat_hw(H, W), 7)
    params = ppo_ref.to_torch_params(synth.ppo_state(11), requires_grad=True)
    adam = {m: {k: (torch.zeros_like(p), torch.zeros_like(p)) for k, p in d.items()} for m, d in params.items()}
    stor, adv = {}, {}
    for r in range(world):
        for w in range(nW):
            wk = bench.Worker(dict(CFG), 1234 + 1000 * r + w, "cpu")
            win = wk.win.view(T, 8)
            feats = torch.stack([encoder_ref.latent_feature(wk.rgb[win[t]].numpy(), wk.route[win[t]].numpy(),
                                                           wk.meas[t * 8:(t + 1) * 8].numpy(), enc_sd) for t in range(T)])
            if feats_check is not None:
                e = rel(feats_check[r][w][:T], feats.numpy())
                assert e < 2e-4, ("encoder features inside the joint chunk", r, w, e)
            for j, hd in enumerate(("steer", "throttle")):
                s = wk.stor[j]
                d = dict(obs=torch.cat([feats, feats[-1:]]), action=s.action, action_log_probs=s.action_log_probs,
                         value_preds=s.value_preds.clone(), rewards=s.rewards, masks=s.masks, command=s.command,
                         hn=torch.zeros(T + 1, 530), cn=torch.zeros(T + 1, 530))
                cmd = int(d["command"][-1].item())
                with torch.no_grad():
                    x, _ = ppo_ref.lstm_forward(d["obs"][-1], (torch.zeros(1, 530), torch.zeros(1, 530)),
                                                params["%s_lstm_%d" % (hd, cmd)])
                    nv = ppo_ref.mlp3(x, params["%s_ppo_%d" % (hd, cmd)], "critic")
                ret, V = ppo_ref.gae_returns(d["rewards"][:, 0].numpy(), d["value_preds"][:, 0].numpy(),
                                             d["masks"][:, 0].numpy(), nv.item(), 0.99, 0.95)
                d["returns"] = torch.from_numpy(ret).view(-1, 1)
                d["value_preds"] = torch.from_numpy(V).view(-1, 1)
                stor[(r, w, j)] = d
                adv[(r, w, j)] = ppo_ref.advantages(ret, V).view(-1, 1)
    # sampler streams: rank r seeds torch's global generator with 100 + r; per epoch every worker draws its steer
    # permutation, then its throttle permutation (bench.learner_round)
    idx = {}
    for r in range(world):
        torch.manual_seed(100 + r)
        for ep in range(bench.PPO_EPOCH):
            for w in range(nW):
                idx[(r, ep, w, 0)] = ppo_ref.sampler_indices(T, bench.MINI_BATCH_NUM)
                idx[(r, ep, w, 1)] = ppo_ref.sampler_indices(T, bench.MINI_BATCH_NUM)
    losses = {r: [] for r in range(world)}
    step = 0
    for ep in range(bench.PPO_EPOCH):
        for b in range(bench.MINI_BATCH_NUM):
            gsum = {m: {k: torch.zeros_like(p) for k, p in d.items()} for m, d in params.items()}
            for r in range(world):
                l3 = np.zeros(3)
                for w in range(nW):
                    l = ppo_ref.update_policy(params, ppo_ref.gather_minibatch(stor[(r, w, 0)], idx[(r, ep, w, 0)][b], adv[(r, w, 0)]),
                                              ppo_ref.gather_minibatch(stor[(r, w, 1)], idx[(r, ep, w, 1)][b], adv[(r, w, 1)]))
                    l3 += np.array(l)
                    for m, d in params.items():
                        for k, p in d.items():
                            gsum[m][k] += p.grad
                losses[r].append(l3)                       # a rank reports the sum of its workers' per-worker means
            step += 1
            ppo_ref.chief_step(params, gsum, adam, step)
    return params, losses, adv


def model_sums(arena_flat, names):
    from cadre_amd.arena import PPOArena
    a = PPOArena("cpu")
    buf = torch.from_numpy(np.ascontiguousarray(arena_flat))
    return [float(sum(t.double().sum() for t in a.views(buf, n).values())) for n in names]


def test_two_ranks_one_gpu_match_the_oracle_and_each_other(tmp_path):
    ranks = run_ranks(tmp_path / "allreduce", "allreduce")
    assert all(int(r["n_exchange"]) == 16 for r in ranks)               # one exchange per optimiser step, 2 rounds x 8
    assert np.array_equal(ranks[0]["params1"], ranks[1]["params1"]), "ranks diverged inside one round"
    assert np.array_equal(ranks[0]["params2"], ranks[1]["params2"]), "ranks diverged on the hipGraph replays"
    params, losses, adv = oracle_round(2, feats_check=[r["feats"] for r in ranks])
    names = sorted(params)
    want = [float(sum(p.data.double().sum() for p in params[n].values())) for n in names]
    got = model_sums(ranks[0]["params1"], names)
    e_p = rel(got, want)
    e_l = max(rel(ranks[r]["losses"], np.array(losses[r])) for r in range(2))
    for r in range(2):                                                   # bit-exact advantage ordering (north_star)
        for w in range(2):
            for j in range(2):
                mine = ranks[r]["adv"][w][j][:, 0]
                assert np.array_equal(np.argsort(mine, kind="stable"), np.argsort(adv[(r, w, j)].numpy()[:, 0], kind="stable"))
    print("2 ranks x 2 workers vs oracle: losses rel %.2e, per-model parameter sums rel %.2e" % (e_l, e_p))
    assert e_l < LOSS_TOL and e_p < PARAM_SUM_TOL
    moved = float(np.abs(ranks[0]["params1"] - model_init()).max())
    assert moved > 1e-4                                                  # the optimiser really stepped
    # the other two forms of the exchange leave the same parameters
    for mode in ("buckets", "sharded"):
        other = run_ranks(tmp_path / mode, mode)
        assert str(other[0]["mode"]) == ("sharded" if mode == "sharded" else "allreduce")
        for key in ("params1", "params2"):
            assert np.array_equal(other[0][key], other[1][key]), (mode, key, "ranks diverged")
            assert np.array_equal(other[0][key], ranks[0][key]), (mode, key, "differs from the all-reduce form")
        assert np.array_equal(other[0]["losses"], ranks[0]["losses"])


def model_init():
    from cadre_amd.arena import PPOArena
    a = PPOArena("cpu")
    a.load_numpy_state(synth.ppo_state(11))
    return a.params.numpy()


def test_exchange_forms_at_world_one(tmp_path):
    """The same three forms with the collectives really issued at world size 1 (CADRE_BENCH_FORCE_DIST): identical
    parameters, and identical to a run without torch.distributed."""
    base = run_ranks(tmp_path / "allreduce", "allreduce", world=1)[0]
    assert int(base["n_exchange"]) == 16
    for mode in ("buckets", "sharded"):
        o = run_ranks(tmp_path / mode, mode, world=1)[0]
        assert np.array_equal(o["params1"], base["params1"]) and np.array_equal(o["params2"], base["params2"]), mode
