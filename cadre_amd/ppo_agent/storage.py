"""Mirror of reference ppo_agent/storage.py: per-worker, per-head rollout buffer living in HBM.

Same constructor, attributes, cursor semantics (including the modulo-(T+1) drift: `after_update`
is never called by the reference train loop, storage.py:57-66) and generator contract.  The
math is HIP: GAE + advantage normalisation = cadre_gae (strict fp32 order, bit-exact with
storage.py:69-76), the time-major minibatch gather = cadre_gather_obs.  Feature rows are
stored with a 544-float pitch (530 + zero pad) so they feed the GEMMs without re-packing;
`.obs` / `.hn` / `.cn` expose the reference's [.., 530] shapes as views."""
import torch

from .. import hip

PAD = 32


def _rup(x, m):
    return (x + m - 1) // m * m


class RolloutStorage(object):
    def __init__(self, num_steps, mini_batch_num, feature_dims, seq_length, hidden_size, use_gae, gamma, tau):
        T = num_steps
        self.mini_batch_num = mini_batch_num
        self.num_steps = T
        self.z_dims = feature_dims
        self.seq_length = seq_length
        self.hid_size = hidden_size
        self.use_gae = use_gae
        self.gamma = gamma
        self.tau = tau
        self.step = 0
        self._ldo = _rup(feature_dims, PAD)
        self._ldh = _rup(hidden_size, PAD)
        self._alloc(torch.device("cpu"))

    def _alloc(self, device):
        T = self.num_steps
        z = lambda *s, dtype=torch.float32: torch.zeros(*s, dtype=dtype, device=device)
        self.device = device
        self._obs = z(T + 1, self.seq_length, self._ldo)
        self._hn = z(T + 1, self._ldh)
        self._cn = z(T + 1, self._ldh)
        self.obs = self._obs[:, :, :self.z_dims]
        self.hn = self._hn[:, :self.hid_size]
        self.cn = self._cn[:, :self.hid_size]
        self.command = z(T + 1, 1, dtype=torch.int)
        self.rewards = z(T + 1, 1)
        self.value_preds = z(T + 1, 1)
        self.returns = z(T + 1, 1)
        self.action_log_probs = z(T + 1, 1)
        self.action = z(T + 1, 1, dtype=torch.long)
        self.masks = z(T + 1, 1)
        self.advantages = z(T, 1)            # filled by compute_returns (train.py:82-88)
        self._next = z(1)

    def to(self, device):
        device = torch.device(device)
        old = {k: getattr(self, k) for k in ("_obs", "_hn", "_cn", "command", "rewards", "value_preds", "returns",
                                             "action_log_probs", "action", "masks", "advantages")}
        self._alloc(device)
        for k, v in old.items():
            getattr(self, k).copy_(v)

    def insert(self, obs, action, action_log_probs, value_preds, rewards, masks, hidden_state, command):
        """storage.py:45-58."""
        s = self.step
        self.action[s].copy_(torch.as_tensor(action).reshape(-1)[:1])
        self.action_log_probs[s].copy_(torch.as_tensor(action_log_probs).reshape(-1)[:1])
        self.value_preds[s].copy_(torch.as_tensor(value_preds).reshape(-1)[:1])
        self.rewards[s].copy_(torch.as_tensor(rewards, dtype=torch.float32).reshape(-1)[:1])
        self.obs[s].copy_(obs.reshape(self.seq_length, self.z_dims))
        if hidden_state is not None and s < self.num_steps:
            hn, cn = hidden_state
            self.hn[s + 1].copy_(hn.reshape(-1))
            self.cn[s + 1].copy_(cn.reshape(-1))
        self.masks[s].copy_(torch.as_tensor(masks).reshape(-1)[:1])
        self.command[s] = command
        self.step = (s + 1) % (self.num_steps + 1)

    def after_update(self, hidden_state):
        self.step = 0
        if hidden_state is not None:
            hn, cn = hidden_state
            self.hn[0].copy_(hn.reshape(-1))
            self.cn[0].copy_(cn.reshape(-1))

    def compute_returns(self, next_value, normalise=True):
        """storage.py:68-76 (GAE branch) + the caller-side advantage lines train.py:82-88.
        `self.advantages` holds (ret[:-1]-V[:-1]) normalised with the unbiased std when `normalise`."""
        if not self.use_gae:
            raise NotImplementedError("use_gae=False branch (storage.py:77-86) is dead in the reference config")
        if not self.returns.is_cuda:
            raise hip.CadreHipError("RolloutStorage.compute_returns runs on the HIP device: call .to('cuda:N') first")
        import numpy as np
        self._next.copy_(torch.as_tensor(next_value, dtype=torch.float32).reshape(-1)[:1])
        g32 = float(np.float32(self.gamma))
        gt32 = float(np.float32(self.gamma * self.tau))          # double product, then one rounding (storage.py:75)
        hip.check(hip.lib().cadre_gae(hip.ptr(self.rewards), hip.ptr(self.value_preds), hip.ptr(self.masks),
                                      hip.ptr(self._next), hip.ptr(self.returns), hip.ptr(self.advantages), 1,
                                      self.num_steps, g32, gt32, 1 if normalise else 0, hip.stream()), "cadre_gae")
        return self.advantages

    def get_last(self, as_tensor=False):
        """storage.py:88-91: (obs[-1], command[-1].item()).  `.item()` on a device tensor is a host sync — in the learner
        section it waits for the whole encoder pass and leaves the GPU idle while the host then enqueues the bootstrap
        values, GAE and the first minibatch (traced: 1.2-2 ms per round).  as_tensor=True returns the command as the
        0-dim device tensor instead; `CadreAgent.get_value(s)` then pick the command net on the device."""
        if as_tensor:
            return self.obs[-1], self.command[-1]
        return self.obs[-1], int(self.command[-1].item())

    def sample_indices(self):
        """BatchSampler(SubsetRandomSampler(range(T)), T // mini_batch_num, drop_last=False)
        (storage.py:94-97): ONE torch.randperm(T) from the global CPU generator."""
        T = self.num_steps
        bs = T // self.mini_batch_num
        perm = torch.randperm(T)
        return [perm[i:i + bs] for i in range(0, T, bs)]

    def gather(self, indices, advantages):
        idx = indices.to(self.device, non_blocking=True)
        B, S = idx.numel(), self.seq_length
        x = torch.empty(S * B, self._ldo, device=self.device)
        hip.check(hip.lib().cadre_gather_obs(hip.ptr(self._obs), self._ldo, S, hip.ptr(idx), B, hip.ptr(x),
                                             self._ldo, self.z_dims, hip.stream()), "cadre_gather_obs")
        hidden = [self._hn.index_select(0, idx)[:, :self.hid_size], self._cn.index_select(0, idx)[:, :self.hid_size]]
        return (x[:, :self.z_dims], self.action[idx], self.value_preds[idx], self.returns[idx], self.masks[idx],
                self.action_log_probs[idx], advantages[idx], hidden, self.command[idx])

    def feed_forward_generator(self, advantages):
        """storage.py:93-120: yields (obs [S*B,D] time-major, action, V_old, ret, mask, logp_old, adv,
        [hn, cn], command)."""
        for indices in self.sample_indices():
            yield self.gather(indices, advantages)
