"""CPU: the oracle restatement vs the committed golden vectors produced by the imported
reference (tests/golden/make_golden.py).  This is what pins oracle/ (SURVEY.md §8c)."""
import numpy as np
import pytest
import torch

from cadre_amd import synth
from oracle import encoder_ref, ppo_ref

SEED_ENC, SEED_PPO = 7, 11


def rel(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


def test_prep_bytes(golden):
    g = golden("prep")
    out, route_after = encoder_ref.pre_process(g["rgb"], g["route"])
    assert out.dtype == np.float32 and np.array_equal(out, g["out"])
    assert np.array_equal(route_after, g["route_after"])          # uint8 truncation quirk, agent.py:51-54
    assert set(np.unique(route_after)) <= {0, 1}


@pytest.mark.parametrize("T", [32, 128, 200])
def test_gae_bit_exact(golden, T):
    g = golden("gae")
    ret, V = ppo_ref.gae_returns(g["T%d_rewards" % T], g["T%d_values" % T], g["T%d_masks" % T],
                                 float(g["T%d_next" % T]), 0.99, 0.95)
    assert np.array_equal(ret.view(np.uint32), g["T%d_returns" % T].view(np.uint32))
    adv = ppo_ref.advantages(ret, V).numpy()
    assert np.array_equal(adv.view(np.uint32), g["T%d_adv" % T].view(np.uint32))
    assert np.array_equal(np.argsort(adv, kind="stable"), g["T%d_argsort" % T])


@pytest.mark.parametrize("T,mbn", [(32, 2), (128, 2), (200, 2), (50, 3)])
def test_sampler_stream(golden, T, mbn):
    g = golden("sampler")
    want = np.split(g["T%d_m%d" % (T, mbn)], np.cumsum(g["T%d_m%d_lens" % (T, mbn)])[:-1])
    torch.manual_seed(1000 + T)
    got = []
    for _ in range(4):
        i1 = ppo_ref.sampler_indices(T, mbn)      # steer generator draws first (train.py:94-96)
        i2 = ppo_ref.sampler_indices(T, mbn)
        for a, b in zip(i1, i2):
            got += [np.array(a), np.array(b)]
    assert len(got) == len(want) and all(np.array_equal(a, b) for a, b in zip(got, want))


@pytest.mark.parametrize("tag", ["84", "native"])
def test_encoder_latent(golden, tag):
    g = golden("enc_" + tag)
    H, W, n = int(g["H"]), int(g["W"]), int(g["n"])
    fh, fw = synth.feat_hw(H, W)
    sd = synth.encoder_state(fh, fw, int(g["seed"]))
    r = np.random.RandomState(int(g["frame_seed"]))
    rgb = r.randint(0, 256, (n, H, W, 3)).astype(np.uint8)
    route = ((r.rand(n, W, H) < 0.15) * 255).astype(np.uint8)
    x, _ = encoder_ref.pre_process(rgb, route)
    lat, taps = encoder_ref.latent(x, sd, return_taps=True)
    assert rel(taps["layer4"].numpy(), g["layer4"]) < 1e-5
    assert rel(taps["da"].numpy(), g["da"]) < 1e-5
    assert rel(lat.numpy(), g["latent"]) < 1e-5


def test_chief_clip_adam(golden):
    g = golden("chief")
    names = [str(n) for n in g["names"]]
    st0 = synth.ppo_state(int(g["ppo_seed"]))
    params = ppo_ref.to_torch_params(st0)
    adam = {m: {k: (torch.zeros_like(p), torch.zeros_like(p)) for k, p in d.items()} for m, d in params.items()}
    for step in (1, 2, 3):
        grads = {}
        for i, n in enumerate(names):
            scale = 2.0 if i % 2 == 0 else 0.01
            grads[n] = {k: torch.from_numpy(synth.make_tensor("g%d.%s.%s" % (step, n, k), tuple(p.shape), "bias",
                                                              int(g["grad_seed"])) * scale * 10)
                        for k, p in params[n].items()}
        ppo_ref.chief_step(params, grads, adam, step)
        ps = [float(sum(p.double().sum() for p in params[n].values())) for n in names]
        pa = [float(sum(p.double().abs().sum() for p in params[n].values())) for n in names]
        assert rel(ps, g["param_sums"][step - 1]) < 1e-7
        assert rel(pa, g["param_abs"][step - 1]) < 1e-7


def test_update_losses(golden):
    """First update_policy call of the F-update replay (full replay runs on the GPU test)."""
    from tests.helpers import oracle_learner_replay
    g = golden("update")
    out = oracle_learner_replay(g, max_steps=2)
    assert rel(out["losses"], g["losses"][:2]) < 1e-5
    assert rel(out["grad_norms"], g["grad_norms"][:2]) < 1e-4
    assert rel(out["param_sums"], g["param_sums"][:2]) < 1e-6
    assert np.array_equal(out["adv_steer"].view(np.uint32), g["adv_steer"].view(np.uint32))
