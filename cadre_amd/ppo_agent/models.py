"""Mirror of reference ppo_agent/models.py: `create_model`, `get_vae_output`, `Model`, `LSTM`,
`Shared_grad_buffers` — same names, arguments, state_dict keys and pickle class paths.
The modules are parameter containers whose tensors are views into the flat HBM arena
(cadre_amd/arena.py); their math runs in cadre_amd/learner.py on HIP kernels."""
import os

import torch
import torch.nn as nn

from ..arena import PPOArena
from ..encoder import DANetEncoderHIP
from .. import autograd, hip
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
        """models.py:139-152: one cell step if x.size(0) == h.size(0), else [T*N, D] time-major unrolled over T.
        Differentiable (cadre_amd/autograd.py: backward through time on the fused step kernels) when gradients are
        enabled and the parameters or inputs require them; `update_policy` does not come through here."""
        arena, learner, g = _module_net(self)
        r = self.rnn
        if autograd.wants_grad(x, hidden_state[0], hidden_state[1], r.weight_ih, r.weight_hh, r.bias_ih, r.bias_hh):
            h, c = autograd.LstmSequence.apply(learner, g, x, hidden_state[0], hidden_state[1], r.weight_ih, r.weight_hh,
                                               r.bias_ih, r.bias_hh)
        else:
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

    def _towers(self, obs_feature):
        """(raw logits [N, n_out], value [N, 1]) with an autograd graph (cadre_amd/autograd.py), or None when no
        gradient is wanted."""
        params = autograd.tower_params(self)
        if not autograd.wants_grad(obs_feature, *params):
            return None
        _a, learner, g = _module_net(self)
        return autograd.TowerPair.apply(learner, g, self.control.num_outputs, obs_feature, *params)

    def get_value(self, obs_feature):
        """models.py:195-197."""
        tw = self._towers(obs_feature)
        if tw is not None:
            return tw[1]
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
        """models.py:199-208 -> (value [N,1], log_prob [N,1], entropy [N,1]); differentiable when gradients are wanted
        (towers on the HIP kernels, the categorical tail on [N, n_out] in torch)."""
        tw = self._towers(obs_feature)
        if tw is not None:
            raw, value = tw
            logits = raw - raw.logsumexp(dim=-1, keepdim=True)           # distributions.py:66-83 Categorical(logits=...)
            logp = logits.gather(1, action.reshape(-1, 1).to(torch.int64))
            ent = -(logits.exp() * logits).sum(-1, keepdim=True)
            return value, logp, ent
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
    if not (isinstance(command_num, int) and 1 <= command_num <= 16):
        raise hip.CadreHipError("command_num=%r: 1 .. 16 navigation commands (the row sort keeps one counter per command in a "
                                "wave; agent_config.py ships command_num=4)" % (command_num,))
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
    views of the flat gradient arena of `model_list`.  `add_gradient` is the gradient hand-off of ONE
    worker: it accumulates (SUM, never mean — chief.py:18, models.py:237) a foreign arena and counts the
    hand-in; the cross-rank exchange is a separate step run ONCE per optimiser step by `chief_step`:

      * `all_reduce()`            one RCCL all-reduce(SUM) of the 80 MB gradient arena (default), or
      * `reduce_scatter()` + sharded clip/Adam + `all_gather_params()`   (`CADRE_GRAD_EXCHANGE=sharded`):
        the same bytes on the wire, 1/N of the optimiser's HBM traffic per rank.

    Whether an exchange is due is derived from SHARED state (`counter` against `_reduced_at`, both
    mp.Value): in the reference's topology (main.py:57-70) workers and chief are separate processes, each
    with its own pickled copy of this object — a plain attribute set by a worker never reaches the chief."""

    def __init__(self, model_list, device):
        self.arena = arena_of(model_list)
        self.device = device
        self.counter = Counter()
        self._reduced_at = Counter()                       # value of `counter` covered by the last exchange
        self._n_exchange = Counter()                       # exchanges run (any process holding this object)
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
        arena is added.  No collective here: several worker agents of one process feeding this buffer must
        not be reduced over the ranks once each."""
        src = arena_of(model_list)
        with self.lock:
            if src is not self.arena:                     # a worker agent with its own nets (reference topology)
                self.arena.grads.add_(src.grads)
                self._accumulated = True
                if self.arena.grads.is_cuda:              # the worker overwrites src.grads in its next update
                    torch.cuda.current_stream().synchronize()
            self.counter.increment()

    # ------------------------------------------------------------------ cross-rank exchange
    @property
    def n_allreduce(self):
        return self._n_exchange.get()

    @staticmethod
    def dist_world():
        """World size of the gradient exchange: 0 when no exchange runs (no process group, or one rank
        without CADRE_BENCH_FORCE_DIST=1, which exercises the collectives at world size 1)."""
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()):
            return 0
        w = dist.get_world_size()
        return w if (w > 1 or os.environ.get("CADRE_BENCH_FORCE_DIST") == "1") else 0

    def exchange_mode(self):
        """'allreduce' (default) or 'sharded' (CADRE_GRAD_EXCHANGE=sharded; needs an arena divisible into
        16-byte-aligned equal shards, else falls back)."""
        w = self.dist_world()
        if w and os.environ.get("CADRE_GRAD_EXCHANGE", "allreduce") == "sharded" and self.arena.total % (4 * w) == 0:
            return "sharded"
        return "allreduce"

    def pending(self):
        return self.counter.get() != self._reduced_at.get()

    def _mark(self):
        self._reduced_at.val.value = self.counter.get()

    def shard(self):
        import torch.distributed as dist
        w, r = dist.get_world_size(), dist.get_rank()
        n = self.arena.total // w
        return r * n, (r + 1) * n

    @staticmethod
    def native_scatter_gather():
        """Whether the process group's backend has reduce-scatter / all-gather on device tensors: RCCL ('nccl') does,
        gloo does not.  Decided ONCE from the backend's name, identically on every rank — never from an exception: a rank
        that fell back to all_reduce after a genuine RCCL error while its peers wait in reduce_scatter would hang the
        job instead of failing it.  Collective errors propagate."""
        import torch.distributed as dist
        return str(dist.get_backend()).lower() == "nccl"

    def all_reduce(self):
        """One all-reduce(SUM) of the flat gradient arena over all ranks (chief.py:18 sums, never averages);
        a no-op outside torch.distributed, at world_size 1, or when nothing was handed in since the last
        exchange.  Buckets already reduced on the side stream (`reduce_bucket_async`) are only waited for."""
        import torch.distributed as dist
        if not self.pending():
            return
        self._mark()
        if not self.dist_world():
            return
        buckets = getattr(self, "_buckets", None) or []
        self._buckets = []
        if buckets:                                        # buckets that went out beside the backward: reduce what they left
            covered = sorted((lo, hi) for lo, hi, _w in buckets)
            pos, g = 0, self.arena.grads
            for lo, hi in covered + [(self.arena.total, self.arena.total)]:
                if lo > pos:
                    dist.all_reduce(g[pos:lo], op=dist.ReduceOp.SUM)
                pos = max(pos, hi)
            for _lo, _hi, work in buckets:
                work.wait()
        else:
            dist.all_reduce(self.arena.grads, op=dist.ReduceOp.SUM)
        self._n_exchange.increment()

    def reduce_bucket_async(self, lo, hi=None):
        """Start the all-reduce of grads[lo:hi] — a bucket whose gradients are final while the backward still runs (the
        MLP towers before the backward through time, the steer nets' LSTM gradients before the throttle nets' lstm_dw) —
        as an async collective on the backend's stream; `all_reduce` later reduces what the buckets left and waits for
        them.  No-op outside the all-reduce exchange."""
        import torch.distributed as dist
        if not self.dist_world() or self.exchange_mode() != "allreduce":
            return
        hi = self.arena.total if hi is None else hi
        if getattr(self, "_buckets", None) is None:
            self._buckets = []
        self._buckets.append((lo, hi, dist.all_reduce(self.arena.grads[lo:hi], op=dist.ReduceOp.SUM, async_op=True)))

    def overlap_hook(self, arena=None):
        """The callable `CadreAgent.update_policy_from_storages(mlp_grads_ready=...)` takes, or None when no exchange runs,
        bucketing is off, or the updating agent's arena is NOT the one behind this buffer.
        `arena`: the parameter arena of the agent whose update will call the hook.  In the reference's topology a worker
        owns its nets and `add_gradient` ADDS its gradients into the shared arena afterwards: a bucket of the shared arena
        reduced while the worker's backward still runs would go out before that add (and `add_` would then write into a
        buffer with a collective in flight) — such an agent gets the single blocking all-reduce of `chief_step`.
        OPT-IN (CADRE_GRAD_BUCKETS=1 / `bench.py --grad-buckets`): three hipGraph parts per step and async collectives
        beside the backward have only ever run over gloo and at RCCL world size 1 — no multi-GPU RCCL run of this form
        exists yet, so the default exchange is the one blocking all-reduce per optimiser step (the sharded exchange
        reduce-scatters the whole arena at once and never buckets)."""
        if not self.dist_world() or self.exchange_mode() != "allreduce" or os.environ.get("CADRE_GRAD_BUCKETS", "0") in ("", "0"):
            return None
        if arena is not None and arena is not self.arena:
            return None
        return self.reduce_bucket_async

    def reduce_scatter(self):
        """Sharded exchange, step 1: this rank's shard of the gradient arena receives the SUM over ranks
        (in place: the shard is a view of the arena).  Returns the shard bounds, or None when nothing is due."""
        import torch.distributed as dist
        if not self.pending():
            return None
        self._mark()
        lo, hi = self.shard()
        g = self.arena.grads
        if self.native_scatter_gather():
            dist.reduce_scatter_tensor(g[lo:hi], g, op=dist.ReduceOp.SUM)
        else:                       # gloo: the same sums by all-reduce (every rank takes this branch)
            dist.all_reduce(g, op=dist.ReduceOp.SUM)
        self._n_exchange.increment()
        return lo, hi

    def all_reduce_norms(self, norms2):
        """Sharded exchange, step 2: the per-model partial square norms (f64) of the shards -> global norms."""
        import torch.distributed as dist
        dist.all_reduce(norms2, op=dist.ReduceOp.SUM)

    def all_gather_params(self):
        """Sharded exchange, step 3: every rank's updated parameter shard to every rank."""
        import torch.distributed as dist
        lo, hi = self.shard()
        p = self.arena.params
        if self.native_scatter_gather():
            dist.all_gather_into_tensor(p, p[lo:hi])
        else:                       # gloo: x + 0 + ... + 0 is x
            p[:lo].zero_(); p[hi:].zero_()
            dist.all_reduce(p, op=dist.ReduceOp.SUM)

    def average_gradient(self):
        self.arena.grads.div_(max(1, self.counter.get()))

    def reset(self, zero=True):
        """models.py:255-258.  `zero=False` (chief_step, when only nets of this arena handed in): the next
        update_policy WRITES every gradient element of the arena (it does not accumulate), so the 80 MB fill would
        be overwritten unread; a foreign arena accumulated with `add_` always gets the fill."""
        for _lo, _hi, work in (getattr(self, "_buckets", None) or []):
            work.wait()             # a bucket started by reduce_bucket_async that no all_reduce() collected (nothing was
        self._buckets = []          # handed in afterwards): its collective must not outlive the buffers' reset
        self.counter.reset()
        self._reduced_at.reset()
        if zero or getattr(self, "_accumulated", False):
            self.arena.grads.zero_()
            self._accumulated = False
