"""Cross-process primitives with the reference's names (ppo_agent/utils.py:31-125).

In the MI355X build one process drives one GPU and gradient exchange is an RCCL all-reduce,
so the worker<->chief barrier is no longer needed for correctness; the classes are kept (same
methods, same mp.Value semantics) because reference main.py constructs and passes them."""
import os

import torch.multiprocessing as mp


class Counter(object):
    def __init__(self, val=True):
        self.val = mp.Value("i", 0)
        self.lock = mp.Lock()

    def get(self):
        return self.val.value

    def increment(self):
        with self.lock:
            self.val.value += 1

    def reset(self):
        self.val.value = 0


class TrafficLight(object):
    def __init__(self, val=True):
        self.val = mp.Value("b", False)
        self.lock = mp.Lock()

    def get(self):
        return self.val.value

    def reset(self):
        self.val.value = False

    def switch(self):
        with self.lock:
            self.val.value = not self.val.value


def check_exist(local_path):
    os.makedirs(local_path, exist_ok=True)


def init(module, weight_init, bias_init, gain=1):
    weight_init(module.weight.data, gain=gain)
    if module.bias is not None:
        bias_init(module.bias.data)
    return module


def init_normc_(weight, gain=1):
    """reference ppo_agent/utils.py:27-29 (imported by the reference models.py)."""
    import torch
    weight.normal_(0, 1)
    weight *= gain / torch.sqrt(weight.pow(2).sum(1, keepdim=True))


class AddBias(__import__("torch").nn.Module):
    """reference ppo_agent/utils.py:7-19 (used only by the dead DiagGaussian heads; kept importable)."""

    def __init__(self, bias):
        super().__init__()
        import torch.nn as nn
        self._bias = nn.Parameter(bias.unsqueeze(1))

    def forward(self, x):
        b = self._bias.t().view(1, -1) if x.dim() == 2 else self._bias.t().view(1, -1, 1, 1)
        return x + b
