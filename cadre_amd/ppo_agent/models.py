"""Mirror of reference ppo_agent/models.py: `create_model`, `get_vae_output`, `Model`, `LSTM`,
`Shared_grad_buffers` — same names, arguments, state_dict keys and pickle class paths.
The modules are parameter containers whose tensors are views into the flat HBM arena
(cadre_amd/arena.py); their math runs in cadre_amd/learner.py on HIP kernels."""
import os

import torch
import torch.nn as nn

from ..arena import PPOArena
from ..encoder import DANetEncoderHIP
from .. import hip
from .distributions import Categorical_1d
from .utils import Counter, init


def _cfg(cfg, key, default=None):
    """cfg.key or cfg['key'] (addict ConfigDict, plain dict or attr-dict), `default` when absent."""
    try:
        return cfg[key]
    except (KeyError, TypeError, IndexError):
        try:
            return getattr(cfg, key, default)
        except (KeyError, AttributeError):
            return default


def _device(num):
    if num == -1:
        raise hip.CadreHipError(
            "device_num/vae_device == -1 (CPU) is not supported by the MI355X-native learner: every op on this "
            "path is a HIP kernel and there is deliberately no CPU fallback (use the reference for CPU runs)")
    return torch.device("cuda:" + str(num))


class EncoderParams(object):
    """Stand-in for `danet_config()` (carla_perception/Config/auto_danet.py:104-171): only the
    fields the PPO path reads."""
    in_route = True

    def __init__(self, obs_hw=(144, 256)):
        self.obs_hw = tuple(obs_hw)
        root = os.environ.get("CHALLENGE_DIR", "")
        self.networks = {"autoencoder": {
            "net_name": "autoencoder", "model_name": "danet", "input_channel": 4, "z_dims": 256,
            "att_type": "transformer", "da_feature_channel": 512, "inter_att_dims": 512, "pred_bc": True,
            "pretrained_path": os.path.join(root, "carla_perception", "Experiments34",
                                            "danet912_nocrash_IL_n10_k1234_r40", "net_epoch90")}}


def get_vae_output(model_cfg):
    """models.py:33-42 -> (obs_dim, vae_params)."""
    vae_params = EncoderParams(_cfg(model_cfg, "obs_hw", (144, 256)))
    name = _cfg(model_cfg, "vae_params")
    z = vae_params.networks["autoencoder"]["z_dims"]
    md = _cfg(model_cfg, "measurement_dim")
    obs_dim = (2 * z if name in ("CoPM", "CoPM w/o att") else z) + md
    return obs_dim, vae_params


def _module_net(module):
    """(arena, learner, net index) of a module bound by create_model."""
    arena = getattr(module, "_cadre_arena", None)
    if arena is None:
        raise hip.CadreHipError("module is not bound to a parameter arena (build it with create_model)")
    learner = getattr(arena, "_learner", None)
    if learner is None:
        from ..learner import PPOLearnerHIP
        arena._learner = learner = PPOLearnerHIP(arena)
    head, _kind, c = module._cadre_name.split("_")
    return arena, learner, arena.net_index(head, int(c))


class LSTM(nn.Module):
    """models.py:130-152 container: `rnn` = nn.LSTMCell(input, hid) with orthogonal weights, zero biases."""

    def __init__(self, input_size, hid_size=128, num_layers=1):
        super().__init__()
        self.rnn = nn.LSTMCell(input_size, hid_size)
        nn.init.orthogonal_(self.rnn.weight_ih.data)
        nn.init.orthogonal_(self.rnn.weight_hh.data)
        self.rnn.bias_ih.data.fill_(0)
        self.rnn.bias_hh.data.fill_(0)

    def forward(self, x, hidden_state):
        """models.py:139-152 (inference; training gradients come from CadreAgent.update_policy):
        one cell step if x.size(0) == h.size(0), else [T*N, D] time-major unrolled over T."""
        arena, learner, g = _module_net(self)
        h, c = learner.lstm_module_forward(g, x, hidden_state[0], hidden_state[1])
        return h, (h, c)


class Model(nn.Module):
    """models.py:162-212 container: `critic` 530-128-128-1, `control` = Categorical_1d."""

    def __init__(self, num_input, num_output, trainable=True, hidsize=128):
        super().__init__()
        init_ = lambda m: init(m, nn.init.orthogonal_, lambda x: nn.init.constant_(x, 0))
        self.control = Categorical_1d(num_input, num_output)
        self.critic = nn.Sequential(
            init_(nn.Linear(num_input, hidsize)), nn.ReLU(),
            init_(nn.Linear(hidsize, hidsize)), nn.ReLU(),
            init_(nn.Linear(hidsize, 1)))
        self.train() if trainable else self.eval()
        self.trainable = trainable

    def to_device(self, device):
        self.critic.to(device)
        self.control.to_device(device)

    def get_log_probs(self, action):
        return self.control.log_probs(action)

    def get_value(self, obs_feature):
        """models.py:195-197 (inference)."""
        _a, learner, g = _module_net(self)
        return learner.mlp_module_forward(g, obs_feature)[1].clone()

    def act(self, obs_feature):
        """models.py:184-189: (value, sampled action, feature).  Sampling = argmax(p/q) with q drawn
        from the global torch CPU generator exactly like Categorical(probs).sample()."""
        arena, learner, g = _module_net(self)
        logits, value = learner.mlp_module_forward(g, obs_feature)
        R, K = obs_feature.shape[0], self.control.num_outputs
        q = torch.empty(R, K).exponential_(1).to(obs_feature.device)
        action = torch.empty(R, dtype=torch.int64, device=obs_feature.device)
        logp = torch.empty(R, 1, device=obs_feature.device)
        hip.check(hip.lib().cadre_sample(hip.ptr(logits), logits.stride(0), hip.ptr(q), K, R, K, hip.ptr(action),
                                         hip.ptr(logp), hip.stream()), "cadre_sample")
        self.control._last_action, self.control._last_logp = action, logp
        return value.clone(), action, obs_feature.clone().detach()

    def evaluate_actions(self, obs_feature, action):
        """models.py:199-208 forward (no autograd graph) -> (value [N,1], log_prob [N,1], entropy [N,1])."""
        arena, learner, g = _module_net(self)
        logits, value = learner.mlp_module_forward(g, obs_feature)
        R = obs_feature.shape[0]
        logp = torch.empty(R, 1, device=obs_feature.device)
        ent = torch.empty(R, 1, device=obs_feature.device)
        act = action.reshape(-1).to(torch.int64).contiguous()
        hip.check(hip.lib().cadre_categorical_eval(hip.ptr(logits), logits.stride(0), hip.ptr(act), R,
                                                   self.control.num_outputs, hip.ptr(logp), hip.ptr(ent),
                                                   hip.stream()), "cadre_categorical_eval")
        return value.clone(), logp, ent


# pickle class paths of the reference (agent.py:245-271 pickles whole modules)
LSTM.__module__ = "ppo_agent.models"
Model.__module__ = "ppo_agent.models"
Categorical_1d.__module__ = "ppo_agent.distributions"


def load_encoder_state(model_cfg, vae_params):
    sd = _cfg(model_cfg, "vae_state_dict")
    if sd is not None:
        return sd
    path = vae_params.networks["autoencoder"]["pretrained_path"]
    ck = torch.load(path, map_location="cpu", weights_only=False)     # raises like the reference if missing
    return ck["autoencoder"]                                          # experiments_builder.py:446-462


class _no_orthogonal_init(object):
    """`weights_init='none'`: skip the 32 QR factorisations of the reference's orthogonal init (≈ 10 s on
    a CPU) when the caller loads weights right after construction (snapshots, tests, bench)."""

    def __enter__(self):
        self._orig = nn.init.orthogonal_
        nn.init.orthogonal_ = lambda t, gain=1: t
        return self

    def __exit__(self, *exc):
        nn.init.orthogonal_ = self._orig


def create_model(model_cfg, load_vae=False):
    """models.py:44-126 -> (vae_model | None, model_dict).  All 16 trainable nets live in one arena."""
    if _cfg(model_cfg, "weights_init", "orthogonal") == "none":
        cfg2 = dict(model_cfg)
        cfg2["weights_init"] = "orthogonal"
        with _no_orthogonal_init():
            return create_model(cfg2, load_vae)
    obs_dim, vae_params = get_vae_output(model_cfg)
    vae_model = None
    if load_vae:
        vae_device = _device(_cfg(model_cfg, "vae_device"))
        H, W = vae_params.obs_hw
        vae_model = DANetEncoderHIP(load_encoder_state(model_cfg, vae_params), H, W, vae_device,
                                    max_frames=_cfg(model_cfg, "encoder_max_frames", 64),
                                    dtype=_cfg(model_cfg, "encoder_dtype", "f32"))
    device = _device(_cfg(model_cfg, "device_num"))
    if not _cfg(model_cfg, "use_lstm", True):
        raise hip.CadreHipError("use_lstm=False is not on the accelerated path (reference default is True)")
    command_num = _cfg(model_cfg, "command_num")
    if command_num != 4:
        raise hip.CadreHipError("command_num=%r: the HIP loss / row-sort kernels are built for the reference's 4 "
                                "navigation commands (agent_config.py: command_num=4)" % (command_num,))
    n_out = _cfg(model_cfg, "num_output")
    arena = PPOArena(device, obs_dim, {"steer": n_out["steer"], "throttle": n_out["throttle"]}, command_num)
    model_dict = {}
    for c in range(command_num):
        for head in ("steer", "throttle"):
            m = Model(obs_dim, n_out[head])
            m.to_device(device)
            model_dict["%s_ppo_%d" % (head, c)] = arena.bind("%s_ppo_%d" % (head, c), m)
    for c in range(command_num):
        for head in ("steer", "throttle"):
            l = LSTM(obs_dim, hid_size=obs_dim).to(device)
            model_dict["%s_lstm_%d" % (head, c)] = arena.bind("%s_lstm_%d" % (head, c), l)
    return vae_model, model_dict


def arena_of(model_dict):
    for m in model_dict.values():
        a = getattr(m, "_cadre_arena", None)
        if a is not None:
            return a
    raise hip.CadreHipError("model_dict was not built by cadre_amd create_model (no parameter arena attached)")


class Shared_grad_buffers(object):
    """models.py:219-258.  `.grads` keeps the reference's key scheme ('<model>_<param>_grad') as
    views of the flat gradient arena of `model_list`.  `add_gradient` is the gradient hand-off:
    SUM (never mean — chief.py:18, models.py:237) over every rank with one RCCL all-reduce of
    the arena when torch.distributed is initialised; nets living in a different arena (separate
    worker agents in one process) are accumulated first."""

    def __init__(self, model_list, device):
        self.arena = arena_of(model_list)
        self.device = device
        self.counter = Counter()
        self.lock = torch.multiprocessing.Lock()          # reference models.py:223,232
        self.grads = {}
        for model_name, model in model_list.items():
            gv = self.arena.views(self.arena.grads, model_name)
            for name, p in model.named_parameters():
                if p.requires_grad:
                    self.grads[model_name + "_" + name + "_grad"] = gv[name]

    def add_gradient(self, model_list):
        """models.py:231-239: accumulate one worker's gradients (SUM).  Nets bound to this arena already
        wrote their gradients into it (update_policy writes, it does not accumulate), so only a foreign
        arena is added.  The cross-rank reduction is a separate step (`all_reduce`, run once per
        optimiser step by `chief_step`) — several worker agents of one process feeding this buffer must
        not be all-reduced once each."""
        src = arena_of(model_list)
        with self.lock:
            if src is not self.arena:                     # a worker agent with its own nets (reference topology)
                self.arena.grads.add_(src.grads)
                if self.arena.grads.is_cuda:              # the worker overwrites src.grads in its next update
                    torch.cuda.current_stream().synchronize()
            self._pending = True
            self.counter.increment()

    def all_reduce(self):
        """One RCCL all-reduce(SUM) of the flat gradient arena over all ranks (chief.py:18 sums, never
        averages); a no-op outside torch.distributed, at world_size 1, or when nothing was handed in
        since the last reduction."""
        import torch.distributed as dist
        if not getattr(self, "_pending", False):
            return
        self._pending = False
        if dist.is_available() and dist.is_initialized() and (
                dist.get_world_size() > 1 or os.environ.get("CADRE_BENCH_FORCE_DIST") == "1"):
            dist.all_reduce(self.arena.grads, op=dist.ReduceOp.SUM)
            self.n_allreduce = getattr(self, "n_allreduce", 0) + 1

    def average_gradient(self):
        self.arena.grads.div_(max(1, self.counter.get()))

    def reset(self):
        self.counter.reset()
        self._pending = False
        self.arena.grads.zero_()
