"""Flat, padded HBM arena for the 16 trainable nets (8 LSTM + 8 actor-critic `Model`s).

MI355X-first layout of what the reference keeps as 112 separate nn.Parameters
(ppo_agent/models.py:100-125): ONE fp32 buffer for parameters, ONE for gradients, two for
Adam state.  Consequences:
  * every per-net matmul becomes a strided batch of one GEMM launch (uniform net stride);
  * input width 530 is padded to 544 (= 17 k-tiles of 32) with zero columns, so GEMM rows are
    16-B aligned and there is no K tail; n_out (33 / 3) is padded to 64 zero rows;
  * the reference's gradient hand-off (`Shared_grad_buffers.add_gradient`, models.py:231-239)
    is one all-reduce over the gradient arena, and chief.py:13-21 is one fused clip+Adam pass;
  * nn.Module parameters handed to the caller (`agent.model_dict`, state_dict/pickle
    compatibility) are strided *views* into the arena, `.grad` likewise.

Net index g = head * command_num + command, head 0 = steer, 1 = throttle.
"""
import torch

HEADS = ("steer", "throttle")


def _rup(x, m):
    return (x + m - 1) // m * m


class PPOArena:
    # Tensors that travel to other processes as HIP-IPC handles (reference main.py:57-70: the shared nets, the gradient
    # buffers and this object are pickled to the spawned chief and workers) come from a PRIVATE, never-split memory pool: each
    # is its own hipMalloc.  The caching allocator would carve them out of segments it shares with unrelated tensors, and
    # `storage._share_cuda_()` exports the whole SEGMENT: two launch rounds of one parent process then export the same
    # segment twice, the second time after the first round's importers have opened and closed it — the situation in which
    # `hipIpcGetMemHandle: invalid argument` was seen on this pool (tests/test_topology_gpu.py; the export alone succeeds in
    # every allocator layout: profiles/r05_hip_ipc_export_probe.txt).  CADRE_ARENA_POOL=0 restores the shared allocator.
    _pools = {}

    def _alloc_shared(self, fn):
        import os
        if self.device.type != "cuda" or os.environ.get("CADRE_ARENA_POOL", "1") == "0" or not hasattr(torch.cuda, "MemPool"):
            return fn()
        idx = self.device.index if self.device.index is not None else torch.cuda.current_device()
        pool = PPOArena._pools.get(idx)
        if pool is None:
            try:
                pool = torch.cuda.MemPool(no_split=True)
            except TypeError:                                  # (older torch: no no_split argument)
                pool = torch.cuda.MemPool()
            PPOArena._pools[idx] = pool                        # (the pool outlives every tensor allocated from it)
        with torch.cuda.use_mem_pool(pool, device=idx):
            return fn()

    def __init__(self, device, obs_dim=530, n_out=None, command_num=4, hid=128):
        n_out = n_out or {"steer": 33, "throttle": 3}
        self.device = torch.device(device)
        self.D = obs_dim
        self.DP = _rup(obs_dim, 32)
        self.H4 = 4 * obs_dim
        self.H4P = 4 * self.DP                # pitch of the gate rows in the update workspace: 4 x 34 k-blocks of 16
        self.C = command_num
        self.Z = 2 * command_num
        self.hid = hid
        self.n_out = (int(n_out["steer"]), int(n_out["throttle"]))
        self.NP = 64
        assert max(self.n_out) <= self.NP and self.H4 % 4 == 0
        # LSTM block
        self.o_wih = 0
        self.o_whh = self.H4 * self.DP
        self.o_bih = 2 * self.H4 * self.DP
        self.o_bhh = self.o_bih + self.H4
        self.size_L = self.o_bhh + self.H4
        # PPO tower (actor = tower 0 'control.linear', critic = tower 1)
        self.t_w1 = 0
        self.t_b1 = hid * self.DP
        self.t_w2 = self.t_b1 + hid
        self.t_b2 = self.t_w2 + hid * hid
        self.t_w3 = self.t_b2 + hid
        self.t_b3 = self.t_w3 + self.NP * hid
        self.size_T = self.t_b3 + self.NP
        self.size_P = 2 * self.size_T
        self.P0 = self.Z * self.size_L
        self.total = self.P0 + self.Z * self.size_P
        assert self.size_L % 4 == 0 and self.size_T % 4 == 0
        self.params = self._alloc_shared(lambda: torch.zeros(self.total, device=self.device))
        self.grads = self._alloc_shared(lambda: torch.zeros(self.total, device=self.device))
        self.exp_avg = None
        self.exp_avg_sq = None
        self._bound = []
        self.step = 0
        # clip segments in reference model order is irrelevant for the math; one segment per model
        offs = [g * self.size_L for g in range(self.Z)] + [self.P0 + g * self.size_P for g in range(self.Z)] + [self.total]
        # the three small device tensors of the optimiser step share ONE pooled block (with a never-split pool each tensor is its
        # own hipMalloc: ADVICE r5): [segment offsets (int64) | per-model square norms (float64) | Adam step count (int32)]
        n_off = len(offs)
        n_nrm = 2 * self.Z + 2
        self._small = self._alloc_shared(lambda: torch.zeros(n_off + n_nrm + 1, dtype=torch.int64, device=self.device))
        self.seg_off = self._small[:n_off]
        self.seg_off.copy_(torch.tensor(offs, dtype=torch.int64))
        self.norms2 = self._small[n_off:n_off + n_nrm].view(torch.float64)
        self.step_dev = self._small[n_off + n_nrm:].view(torch.int32)[:1]                 # Adam step count (graph replay)

    # ------------------------------------------------------------------ naming
    def net_index(self, head, command):
        return HEADS.index(head) * self.C + command

    def model_names(self):
        return ["%s_lstm_%d" % (h, c) for h in HEADS for c in range(self.C)] + \
               ["%s_ppo_%d" % (h, c) for h in HEADS for c in range(self.C)]

    def segment_of(self, model_name):
        head, kind, c = model_name.split("_")
        g = self.net_index(head, int(c))
        return g if kind == "lstm" else self.Z + g

    # ------------------------------------------------------------------ views
    def lstm_views(self, buf, g, base=None):
        """parameter-shaped views of net g's LSTM block in `buf` (arena layout; `base`: the block starts there instead —
        a buffer that holds just this block)"""
        b = g * self.size_L if base is None else base
        D, DP, H4 = self.D, self.DP, self.H4
        return {
            "rnn.weight_ih": buf[b + self.o_wih: b + self.o_wih + H4 * DP].view(H4, DP)[:, :D],
            "rnn.weight_hh": buf[b + self.o_whh: b + self.o_whh + H4 * DP].view(H4, DP)[:, :D],
            "rnn.bias_ih": buf[b + self.o_bih: b + self.o_bih + H4],
            "rnn.bias_hh": buf[b + self.o_bhh: b + self.o_bhh + H4],
        }

    def ppo_views(self, buf, g, base=None):
        out = {}
        hid, DP, D, NP = self.hid, self.DP, self.D, self.NP
        head = g // self.C
        for tower, name in ((0, "control.linear"), (1, "critic")):
            b = (self.P0 + g * self.size_P if base is None else base) + tower * self.size_T
            n3 = self.n_out[head] if tower == 0 else 1
            out[name + ".0.weight"] = buf[b + self.t_w1: b + self.t_w1 + hid * DP].view(hid, DP)[:, :D]
            out[name + ".0.bias"] = buf[b + self.t_b1: b + self.t_b1 + hid]
            out[name + ".2.weight"] = buf[b + self.t_w2: b + self.t_w2 + hid * hid].view(hid, hid)
            out[name + ".2.bias"] = buf[b + self.t_b2: b + self.t_b2 + hid]
            out[name + ".4.weight"] = buf[b + self.t_w3: b + self.t_w3 + NP * hid].view(NP, hid)[:n3]
            out[name + ".4.bias"] = buf[b + self.t_b3: b + self.t_b3 + n3]
        return out

    def views(self, buf, model_name):
        head, kind, c = model_name.split("_")
        g = self.net_index(head, int(c))
        return self.lstm_views(buf, g) if kind == "lstm" else self.ppo_views(buf, g)

    # ------------------------------------------------------------------ module binding
    def bind(self, model_name, module):
        """Move `module`'s parameters into the arena (values preserved) and point p.data / p.grad at
        the arena views.  Idempotent."""
        pv, gv = self.views(self.params, model_name), self.views(self.grads, model_name)
        with torch.no_grad():
            for name, p in module.named_parameters():
                v = pv[name]
                if p.data.data_ptr() != v.data_ptr():
                    v.copy_(p.data.to(self.device))
                    p.data = v
                p.grad = gv[name]
                self._bound.append((p, gv[name]))
        module._cadre_arena = self
        module._cadre_name = model_name
        control = getattr(module, "control", None)        # Categorical_1d stand-alone API (distributions.py:66-105)
        if control is not None:
            control._cadre_arena = self
            control._cadre_name = model_name
        return module

    def __getstate__(self):
        """Pickling (reference main.py:57-70 hands the shared nets to spawned processes): the tensors travel
        as HIP-IPC handles; the learner (hipGraphs, workspaces) is per process and is rebuilt on first use."""
        d = dict(self.__dict__)
        d.pop("_learner", None)
        return d

    def attach_grads(self, model_dict=None):
        """(Re-)attach arena gradient views — `zero_grad(set_to_none=True)` by a caller detaches them."""
        for p, g in self._bound:
            if p.grad is not g:
                p.grad = g

    def load_numpy_state(self, state):
        """state: {model_name: {param_name: ndarray}} (cadre_amd.synth.ppo_state layout)."""
        with torch.no_grad():
            for mn, d in state.items():
                v = self.views(self.params, mn)
                for k, arr in d.items():
                    v[k].copy_(torch.as_tensor(arr).to(self.device))

    def ensure_adam(self):
        if getattr(self, "_shard", None) is not None:
            raise RuntimeError("this arena's Adam state is sharded over the data-parallel ranks (elements [%d, %d)); "
                               "the replicated optimiser step cannot follow sharded ones" % self._shard)
        if self.exp_avg is None:
            # (the moments are pickled with the arena once they exist — main.py:57-70 hands the optimiser's owner to the chief —,
            #  so they come from the private pool like everything else that travels: ADVICE r5)
            self.exp_avg = self._alloc_shared(lambda: torch.zeros_like(self.params))
            self.exp_avg_sq = self._alloc_shared(lambda: torch.zeros_like(self.params))
