"""`Categorical_1d` policy head (reference ppo_agent/distributions.py:25-109): parameter
container with the reference's state_dict keys (`linear.{0,2,4}.{weight,bias}`).  The math
(3-layer MLP, log-softmax, log-prob gather, entropy, argmax(p/q) sampling) runs in
cadre_amd.learner through the HIP kernels cadre_gemm_f32 / cadre_ppo_loss / cadre_sample;
this class only caches what the last `Model.act` produced."""
import torch.nn as nn

from .utils import init


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
        self._last_action = None
        self._last_logp = None
        self.train()

    def to_device(self, device):
        self.linear.to(device)

    def log_probs(self, action):
        """log-prob of the action sampled by the last `act` (distributions.py:101-102)."""
        if self._last_action is None or action is not self._last_action:
            raise RuntimeError("Categorical_1d.log_probs: only the action returned by the last act() is cached; "
                               "minibatch evaluation runs inside CadreAgent.update_policy on the HIP path")
        return self._last_logp
