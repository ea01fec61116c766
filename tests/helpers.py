"""Shared test helpers: seeded storage contents and the oracle replay of the learner section
(reference ppo_agent/train.py:76-110) used by both the CPU oracle-pinning test and the GPU
parity tests."""
import numpy as np
import torch

from cadre_amd import synth
from oracle import ppo_ref


def fill_storages(T, seed, with_hidden=True):
    """Must stay identical to tests/golden/make_golden.py:fill_storages (regenerable inputs)."""
    r = np.random.RandomState(seed)
    d = {}
    for hd, K in (("steer", 33), ("throttle", 3)):
        d[hd] = dict(
            obs=(r.standard_normal((T + 1, 8, 530)) * 0.5).astype(np.float32),
            action=r.randint(0, K, (T + 1, 1)).astype(np.int64),
            action_log_probs=(-np.log(K) + 0.1 * r.standard_normal((T + 1, 1))).astype(np.float32),
            value_preds=(0.3 * r.standard_normal((T + 1, 1))).astype(np.float32),
            rewards=r.rand(T + 1, 1).astype(np.float32),
            masks=(r.rand(T + 1, 1) >= 0.05).astype(np.float32),
            command=r.randint(0, 4, (T + 1, 1)).astype(np.int32),
            hn=((r.standard_normal((T + 1, 530)) * 0.1) if with_hidden else np.zeros((T + 1, 530))).astype(np.float32),
            cn=((r.standard_normal((T + 1, 530)) * 0.1) if with_hidden else np.zeros((T + 1, 530))).astype(np.float32),
        )
    return d


def oracle_learner_replay(g, max_steps=None, data=None):
    """train.py:76-110 on the oracle.  `data`: storage contents {head: {field: array}} to replay instead of the seeded
    ones of `fill_storages` (the end-to-end contract tests fill them from the oracle's own act path)."""
    T, mbn, epochs = int(g["T"]), int(g["mbn"]), int(g["epochs"])
    C = int(g["command_num"]) if "command_num" in g else 4            # (agent_config.py ships 4; agent.py:170-182 loops over any count)
    st0 = synth.ppo_state(int(g["ppo_seed"]), command_num=C)
    names = [str(n) for n in g["names"]]
    if data is None:
        data = fill_storages(T, int(g["data_seed"]))
    params = ppo_ref.to_torch_params(st0, requires_grad=True)
    adam = {m: {k: (torch.zeros_like(p), torch.zeros_like(p)) for k, p in d.items()} for m, d in params.items()}
    stor = {hd: {k: torch.from_numpy(v).clone() for k, v in data[hd].items()} for hd in data}
    adv = {}
    for hd in ("steer", "throttle"):
        cmd = int(stor[hd]["command"][-1].item())
        with torch.no_grad():
            x, _ = ppo_ref.lstm_forward(stor[hd]["obs"][-1], (torch.zeros(1, 530), torch.zeros(1, 530)),
                                        params["%s_lstm_%d" % (hd, cmd)])
            nv = ppo_ref.mlp3(x, params["%s_ppo_%d" % (hd, cmd)], "critic")
        ret, V = ppo_ref.gae_returns(stor[hd]["rewards"][:, 0].numpy(), stor[hd]["value_preds"][:, 0].numpy(),
                                     stor[hd]["masks"][:, 0].numpy(), nv.item(), 0.99, 0.95)
        stor[hd]["returns"] = torch.from_numpy(ret).view(-1, 1)
        stor[hd]["value_preds"] = torch.from_numpy(V).view(-1, 1)
        adv[hd] = ppo_ref.advantages(ret, V).view(-1, 1)
    torch.manual_seed(int(g["torch_seed"]))
    out = dict(losses=[], grad_norms=[], param_sums=[], adv_steer=adv["steer"].numpy(),
               adv_throttle=adv["throttle"].numpy())
    step = 0
    for _ in range(epochs):
        i_s = ppo_ref.sampler_indices(T, mbn)
        i_t = ppo_ref.sampler_indices(T, mbn)
        for a, b in zip(i_s, i_t):
            if max_steps is not None and step >= max_steps:
                break
            step += 1
            l3 = ppo_ref.update_policy(params, ppo_ref.gather_minibatch(stor["steer"], a, adv["steer"]),
                                       ppo_ref.gather_minibatch(stor["throttle"], b, adv["throttle"]), command_num=C)
            out["losses"].append(l3)
            out["grad_norms"].append([float(torch.sqrt(sum((p.grad.double() ** 2).sum() for p in params[n].values())))
                                      for n in names])
            grads = {m: {k: p.grad for k, p in d.items()} for m, d in params.items()}
            ppo_ref.chief_step(params, grads, adam, step)
            out["param_sums"].append([float(sum(p.data.double().sum() for p in params[n].values())) for n in names])
    return out


# ----------------------------------------------------------------------------- reference train() plumbing
class AD(dict):
    """attr-dict stand-in for addict's ConfigDict (reference ppo_agent/meta/config.py): cfg.key and cfg['key']."""
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    __setattr__ = dict.__setitem__


class SyntheticEnv(object):
    """Stand-in for reference env_wrapper.EnvWrapper (no CARLA in this image): deterministic synthetic
    observations with the interface train.py:50-75 uses (reset / step / work_dir).  Seeds torch's global CPU
    generator, so agent initialisation and action sampling are identical across launch topologies."""

    def __init__(self, env_cfg):
        self.cfg = env_cfg
        self.work_dir = env_cfg["work_dir"]
        H, W = env_cfg["obs_hw"]
        self.steps = synth.synth_rollout(env_cfg["total_steps"] + 2, H, W, seed=4242 + int(env_cfg["rank"]))
        self.i = 0
        torch.manual_seed(999 + int(env_cfg["rank"]))

    def _obs(self):
        td = self.steps[self.i % len(self.steps)]
        return dict(rgb=td["rgb"], route_fig=td["route_fig"].copy(), measurements=td["measurements"],
                    command=td["command"])

    def reset(self):
        return self._obs()

    def step(self, action):
        td = self.steps[self.i % len(self.steps)]
        self.i += 1
        done = bool(td["done"].any())
        return self._obs(), [float(td["reward"][0]), float(td["reward"][1])], done, {"action_done": [bool(td["done"][0]), bool(td["done"][1])]}


def topology_cfgs(work_dir, H=84, W=84, T=8, episodes=2):
    """The four config dicts reference main.py:28-33 reads from config_files/agent_config.py, sized for a test."""
    fh, fw = synth.feat_hw(H, W)
    model_cfg = AD(use_lstm=True, vae_device=0, device_num=0, vae_params="CoPM", measurement_dim=18,
                   num_output=AD(steer=33, throttle=3), command_num=4, obs_hw=(H, W), weights_init="none",
                   vae_state_dict=synth.encoder_state(fh, fw, 7), latent_cache=True)
    agent_cfg = AD(rank=0, model_cfg=model_cfg, frame=8, STEER_CONTROL={i: (i - 16) / 16.0 for i in range(33)},
                   THROTTLE_CONTROL={0: [0, 0], 1: [0, 1], 2: [0.6, 0]}, ent_coeff=0.01, value_coeff=0.1,
                   clip_coeff=1.0, clip=0.1)
    env_cfg = AD(num_processes=1, port=[2000], routes=["r"], scenarios=["s"], town=["Town01"], work_dir=work_dir,
                 obs_hw=(H, W), total_steps=T * episodes, rank=0)
    rollout_cfg = AD(num_steps=T, mini_batch_num=2, feature_dims=530, seq_length=8, use_gae=True, gamma=0.99, tau=0.95)
    train_cfg = AD(max_episode=episodes, ppo_epoch=1, use_adv_norm=True, max_grad_norm=250.0, lr=3e-4,
                   log_interval=1, save_interval=1)
    return train_cfg, agent_cfg, env_cfg, rollout_cfg
