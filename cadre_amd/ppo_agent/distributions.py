"""`Categorical_1d` policy head (reference ppo_agent/distributions.py:25-109): parameter
container with the reference's state_dict keys (`linear.{0,2,4}.{weight,bias}`) and the
reference's stand-alone API — `forward / sample / mode / softmax_sample / log_probs / entropy`.
The math (3-layer MLP, log-softmax, log-prob gather, entropy, argmax(p/q) sampling) runs on the
HIP kernels cadre_gemm_f32 / cadre_categorical_dist / cadre_categorical_eval / cadre_sample; the
fused training path (CadreAgent.update_policy -> cadre_ppo_loss) never goes through this class.
Inference only: gradients come from CadreAgent.update_policy."""
import torch
import torch.nn as nn

from .. import hip
from .utils import init


class CategoricalHIP(object):
    """What `Categorical(logits=x)` exposes to the reference's callers (distributions.py:81-105):
    `.logits` (normalised), `.probs`, `.sample()`, `.log_prob(a)`, `.entropy()` — device tensors
    produced by the HIP kernels from the raw head outputs `raw` [R, ld] (first K columns valid)."""

    def __init__(self, raw, K):
        self._raw, self.K = raw, K
        R = raw.shape[0]
        self.logits = torch.empty(R, K, device=raw.device)
        self.probs = torch.empty(R, K, device=raw.device)
        self._mode = torch.empty(R, dtype=torch.int64, device=raw.device)
        hip.check(hip.lib().cadre_categorical_dist(hip.ptr(raw), raw.stride(0), R, K, hip.ptr(self.logits),
                                                   hip.ptr(self.probs), hip.ptr(self._mode), hip.stream()),
                  "cadre_categorical_dist")

    def sample(self):
        """Categorical.sample() on torch CPU == argmax(p / q), q = empty_like(p).exponential_(1) from the
        global CPU generator — the draw is made on the host exactly like the reference."""
        R = self._raw.shape[0]
        q = torch.empty(R, self.K).exponential_(1).to(self._raw.device)
        action = torch.empty(R, dtype=torch.int64, device=self._raw.device)
        logp = torch.empty(R, 1, device=self._raw.device)
        hip.check(hip.lib().cadre_sample(hip.ptr(self._raw), self._raw.stride(0), hip.ptr(q), self.K, R, self.K,
                                         hip.ptr(action), hip.ptr(logp), hip.stream()), "cadre_sample")
        return action

    def _eval(self, action):
        R = self._raw.shape[0]
        act = action.reshape(-1).to(device=self._raw.device, dtype=torch.int64).contiguous()
        if act.numel() != R:
            raise ValueError("expected one action per row (%d), got %d" % (R, act.numel()))
        logp = torch.empty(R, device=self._raw.device)
        ent = torch.empty(R, device=self._raw.device)
        hip.check(hip.lib().cadre_categorical_eval(hip.ptr(self._raw), self._raw.stride(0), hip.ptr(act), R, self.K,
                                                   hip.ptr(logp), hip.ptr(ent), hip.stream()), "cadre_categorical_eval")
        return logp, ent

    def log_prob(self, action):
        return self._eval(action)[0]

    def entropy(self):
        return self._eval(torch.zeros(self._raw.shape[0], dtype=torch.int64, device=self._raw.device))[1]


class Categorical_1d(nn.Module):
    def __init__(self, num_inputs, num_outputs, name="none"):
        super().__init__()
        init_ = lambda m: init(m, nn.init.orthogonal_, lambda x: nn.init.constant_(x, 0), gain=0.01)
        self.linear = nn.Sequential(
            init_(nn.Linear(num_inputs, 128)), nn.ReLU(),
            init_(nn.Linear(128, 128)), nn.ReLU(),
            init_(nn.Linear(128, num_outputs)))
        self.num_outputs = num_outputs
        self.name = name
        self.dis_cat = None
        self.logits = 0
        self._last_action = None
        self._last_logp = None
        self.train()

    def to_device(self, device):
        self.linear.to(device)

    # ------------------------------------------------------------------ stand-alone head (distributions.py:66-105)
    def forward(self, x):
        """distributions.py:66-83: run the actor tower on x [R, D] and keep the resulting distribution."""
        arena = getattr(self, "_cadre_arena", None)
        if arena is None:
            raise hip.CadreHipError("Categorical_1d is not bound to a parameter arena (build it with create_model)")
        from .models import _module_net
        _a, learner, g = _module_net(self)
        raw, _value = learner.mlp_module_forward(g, x)
        self.dis_cat = CategoricalHIP(raw[:, :self.num_outputs].clone(), self.num_outputs)
        self.logits = self.dis_cat.logits
        self.probs = self.dis_cat.probs
        self._last_action = None
        return self.dis_cat

    def _dist(self):
        if self.dis_cat is None:
            raise RuntimeError("Categorical_1d: call forward(x) first")
        return self.dis_cat

    def sample(self):
        return self._dist().sample()

    def mode(self):
        """distributions.py:87-88: torch.argmax(self.probs) — the FLAT argmax (no dim), kept as is."""
        d = self._dist()
        if d.probs.shape[0] == 1:
            return d._mode[0]
        rowmax = d.probs.gather(1, d._mode.view(-1, 1)).view(-1)
        r = int(torch.argmax(rowmax))
        return d._mode[r] + r * self.num_outputs

    def softmax_sample(self):
        """distributions.py:96-99: Categorical(probs=softmax(logits)).sample() — the same argmax(p/q) rule."""
        return self._dist().sample()

    def log_probs(self, action):
        """distributions.py:101-102.  The action returned by the last `Model.act` is served from the value the
        sampling kernel already produced."""
        if self._last_action is not None and action is self._last_action:
            return self._last_logp
        return self._dist().log_prob(action.squeeze(-1) if action.dim() > 1 else action).unsqueeze(-1)

    def entropy(self):
        return self._dist().entropy()
