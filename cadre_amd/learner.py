"""PPO inner loop on MI355X: the math of `CadreAgent.update_policy` / `act` / `get_value`
(reference ppo_agent/agent.py:114-237) and of `chief` (ppo_agent/chief.py:13-21) as batched
HIP launches over the parameter arena (cadre_amd/arena.py).

The reference runs 8 (head x command) LSTM+MLP nets one after another, each through 8
LSTMCell calls, and lets autograd replay ~1000 tiny kernels.  Here the 8 nets are ONE strided
batch: the input projections of all 8 time steps and 4 command nets of a head are one GEMM,
each recurrent step is ONE launch for all 8 nets (product + cell math, csrc/ppo_update.hip), the
MLP towers three launches, the backward pass is written out explicitly (same formulas autograd
would apply) and writes straight into the flat gradient arena.  No autograd graph, no
per-parameter tensors, nothing on the host between launches — the whole update is 24 launches
on one stream, replayed as a hipGraph.
"""
import os

import torch

from . import hip


class PPOLearnerHIP:
    SORT_MIN_B = 64

    def sorted_rows(self, B):
        """Row-sorted update (rows of a minibatch grouped by command; every kernel of the step then works on exactly
        the run of rows a command net owns).  At minibatch 64 a command net owns ~16 of the 64 rows, so the
        unsorted form spends 4x the fp32-MFMA time the update needs — and the recurrent steps are bound by
        exactly that (the matrix pipe on their critical path), not by the weight stream."""
        return self.use_sorted and B >= self.SORT_MIN_B and B % 32 == 0

    def __init__(self, arena, clip=0.1, value_coeff=0.1, clip_coeff=1.0, ent_coeff=0.01, seq_length=8):
        self.a = arena
        self.clip, self.vc, self.cc, self.ec = float(clip), float(value_coeff), float(clip_coeff), float(ent_coeff)
        self.S = seq_length
        self._ws = {}
        self._graphs = {}
        self._wp = None            # recurrent weights in MFMA fragment order (forward, backward), re-packed per optimiser step
        self._wp_key = None
        self.pack_outside_capture = False      # act() graphs: the copies are refreshed eagerly before each replay
        self.launches = {}
        self.use_graphs = os.environ.get("CADRE_HIP_GRAPHS", "1") != "0"
        self.use_sorted = os.environ.get("CADRE_SORTED_UPDATE", "1") != "0"
        # forward LSTM of the update as one persistent launch (cadre_lstm_seq_fwd) instead of one launch per time step:
        # opt-in, measured slower in place (C2 208 vs 192 us, C3 376 vs 295 us for the 8 steps)
        # (A/B build only: the default library does not export it)
        self.persistent_lstm = os.environ.get("CADRE_LSTM_PERSISTENT", "0") != "0" and hip.has_ab_kernels()
        # MLP towers of the update as three fused launches (cadre_mlp_fwd / _bwd / _dw) instead of 17 GEMM / column-sum /
        # mask launches; CADRE_FUSED_MLP=0 keeps the GEMM chain (A/B)
        self.fused_mlp = os.environ.get("CADRE_FUSED_MLP", "1") != "0"
        # CADRE_ADAM_PACK=1: the optimiser step writes the fragment-order copies of W_hh itself (cadre_clip_adam_pack_graph)
        # and the update that follows an in-process clip_adam() carries no packing launch.  Opt-in: bit-identical, one kernel
        # fewer (23 vs 24 per step) and 110 MB less traffic, but NOT faster — same box, C3: 0.987 / 0.992 vs 0.988 / 0.980 ms
        # per step: the 4 x 4-block thread mapping scatters its 16-byte stores into both copies
        self.fused_pack = os.environ.get("CADRE_ADAM_PACK", "0") != "0"
        self._adam_fresh = None    # _pkey() right after a fused optimiser step of THIS learner: the copies are current
        self._skip_pack = False
        self._mlp_offs = None
        hip.lib()

    # ------------------------------------------------------------------ workspace
    def workspace(self, B, Z=None, S=None):
        a = self.a
        Z = a.Z if Z is None else Z
        S = self.S if S is None else S
        key = (B, Z, S)
        w = self._ws.get(key)
        if w is None:
            dev = a.device
            z = lambda *shape, dtype=torch.float32: torch.zeros(*shape, dtype=dtype, device=dev)
            w = dict(
                X=z(2, S, B, a.DP), h0=z(2, B, a.DP), c0=z(2, B, a.DP),
                G=z(Z, S, B, a.H4P), dG=z(Z, S, B, a.H4P),         # gate rows [i f g o] x D, zero padded to 4 x 34 k-blocks
                Hs=z(Z, S + 1, B, a.DP), Cs=z(Z, S + 1, B, a.DP), TC=z(Z, S + 1, B, a.DP),
                A1=z(2 * Z, B, a.hid), A2=z(2 * Z, B, a.hid), O3=z(2 * Z, B, a.NP),
                dO3=z(2 * Z, B, a.NP), dA2=z(2 * Z, B, a.hid), dA1=z(2 * Z, B, a.hid),
                dH=z(Z, B, a.DP), dC=z(Z, B, a.DP), dGp=z(2, Z, (B + 15) // 16, 16 * a.H4P),
                actions=z(2, B, dtype=torch.int64), commands=z(2, B, dtype=torch.int32),
                old_values=z(2, B), returns=z(2, B), old_logp=z(2, B), adv=z(2, B),
                losses=z(3), loss_scratch=z(4 + 6 * ((B + 15) // 16)), sync=z(Z * S + 1, dtype=torch.int32),
                pos=z(2, B, dtype=torch.int32), seg=z(2 * a.C, 2, dtype=torch.int32),
            )
            if self.sorted_rows(B) and Z == a.Z:      # unsorted staging for gather -> sort -> permute
                w.update(Xu=z(2, S, B, a.DP), h0u=z(2, B, a.DP), c0u=z(2, B, a.DP),
                         actions_u=z(2, B, dtype=torch.int64), commands_u=z(2, B, dtype=torch.int32),
                         old_values_u=z(2, B), returns_u=z(2, B), old_logp_u=z(2, B), adv_u=z(2, B))
            self._ws[key] = w
        return w

    # ------------------------------------------------------------------ packed weights
    def packed_weights(self, g0, gs, Z):
        """Fragment-order copies of the recurrent weights of all 8 nets (cadre_pack_lstm_weights), refreshed when the
        parameters changed: the optimiser step count and `params._version` (in-place loads) key the copy.  Returns
        (forward copy of net g0, net stride gs nets)."""
        a = self.a
        key = self._pkey()
        self._alloc_wp()
        capturing = torch.cuda.is_current_stream_capturing()
        if capturing and self.pack_outside_capture:
            return self._wp[0, g0:], gs * self._wp.stride(1)     # (the caller refreshed the copies before the replay)
        if self._skip_pack:                                      # (update() checked: the last optimiser step left them current)
            return self._wp[0, g0:], gs * self._wp.stride(1)
        if key != self._wp_key or capturing:
            hip.check(hip.lib().cadre_pack_lstm_weights(hip.ptr(a.params[a.o_whh:]), a.size_L, a.DP, a.D, a.Z, hip.ptr(self._wp[0]),
                                                        hip.ptr(self._wp[1]), self._wp.stride(1), hip.stream()),
                      "cadre_pack_lstm_weights")
            self._wp_key = None if torch.cuda.is_current_stream_capturing() else key
        return self._wp[0, g0:], gs * self._wp.stride(1)

    def _pkey(self):
        a = self.a
        return (a.step, a.params._version, a.params.data_ptr())

    def _alloc_wp(self):
        a = self.a
        if self._wp is None:
            n = ((a.D + 15) // 16) * 4 * (a.DP // 16) * 256
            self._wp = torch.zeros(2, a.Z, n, device=a.device)

    # ------------------------------------------------------------------ forward
    def _forward(self, w, B, nets, x_div, S=None, mlp=True, seg=None, fused_mlp=False):
        """LSTM (S steps) + (optionally) both MLP towers for `Z` nets.  nets = (g0, g_stride, Z): arena net
        indices g0 + i*g_stride.  Net i reads inputs X[i // x_div], h0/c0[i // x_div]."""
        a = self.a
        S = self.S if S is None else S
        g0, gs, Z = nets
        P, st = a.params, hip.stream()
        L = hip.lib()
        DP, H4, H4P, hid, NP = a.DP, a.H4, a.H4P, a.hid, a.NP
        pL = P[g0 * a.size_L:]
        sL = gs * a.size_L
        X, G, Hs, Cs, TC = w["X"], w["G"], w["Hs"], w["Cs"], w["TC"]
        # h_{-1}, c_{-1} (hidden_state_batch, agent.py:166-175) into slot 0 of every net; dC <- 0 for the backward
        hip.check(L.cadre_lstm_init(hip.ptr(w["h0"]), hip.ptr(w["c0"]), hip.ptr(Hs), hip.ptr(Cs), hip.ptr(w["dC"]),
                                    B * DP, (S + 1) * B * DP, x_div, Z, st), "cadre_lstm_init")
        # all input projections x_t W_ih^T + b_ih: one GEMM [S*B, DP] x [DP, H4] per net
        sg1 = None if seg is None else (3, seg, B, 1)         # exactly each net's run of rows, step after step (compact M)
        sgp = None if seg is None else hip.ptr(seg)           # fused steps: the same rows
        hip.gemm(X, pL[a.o_wih:], G, S * B, H4, DP, DP, DP, H4P, shift=pL[a.o_bih:], batch=Z,
                 a_z=(x_div, 0, S * B * DP), b_z=(1, 0, sL), c_z=(1, 0, S * B * H4P), s_z=(1, 0, sL), seg=sg1)
        wpf, wps = self.packed_weights(g0, gs, Z)
        if self.persistent_lstm and Z == a.Z and gs == 1 and "sync" in w:
            # all S steps in one persistent launch: weights resident in registers, h_t exchanged through L2
            hip.check(L.cadre_lstm_seq_fwd(hip.ptr(wpf), wps, hip.ptr(pL[a.o_bhh:]), sL, hip.ptr(G), H4P, S * B * H4P, hip.ptr(Hs),
                                           hip.ptr(Cs), hip.ptr(TC), DP, (S + 1) * B * DP, B, a.D, S, Z, sgp, hip.ptr(w["sync"]), st),
                      "cadre_lstm_seq_fwd")
        for t in range(0 if not (self.persistent_lstm and Z == a.Z and gs == 1 and "sync" in w) else S, S):   # models.py:148-151
            # gates = x-projection + h_{t-1} W_hh^T + b_hh, cell math, h_t / c_t / tanh(c_t): one launch for all nets
            hip.check(L.cadre_lstm_step_fwd(hip.ptr(wpf), wps, hip.ptr(pL[a.o_bhh:]), sL, hip.ptr(G[:, t]), H4P,
                                            S * B * H4P, hip.ptr(Hs[:, t]), hip.ptr(Cs[:, t]), hip.ptr(Hs[:, t + 1]),
                                            hip.ptr(Cs[:, t + 1]), hip.ptr(TC[:, t + 1]), DP, (S + 1) * B * DP, B, a.D, Z,
                                            sgp, t & 1, st), "cadre_lstm_step_fwd")
        if mlp:
            self._mlp(w, B, nets, Hs[:, S], (S + 1) * B * DP, seg=seg, fused=fused_mlp)

    def mlp_offs(self):
        """(W1, b1, W2, b2, W3, b3) float offsets inside a tower, as the int32[6] the cadre_mlp_* entry points take."""
        if self._mlp_offs is None:
            import ctypes
            a = self.a
            self._mlp_offs = (ctypes.c_int32 * 6)(a.t_w1, a.t_b1, a.t_w2, a.t_b2, a.t_w3, a.t_b3)
        return self._mlp_offs

    def _mlp(self, w, B, nets, inp, inp_zstride, seg=None, fused=False):
        """actor (tower 0) + critic (tower 1) of each net on `inp` ([Z][B][DP] rows, net stride
        inp_zstride): z = 2*i + tower  (models.py:171-177, distributions.py:34-40)."""
        a = self.a
        g0, gs, Z = nets
        P = a.params
        DP, hid, NP = a.DP, a.hid, a.NP
        pP = P[a.P0 + g0 * a.size_P:]
        sT = a.size_T if gs == 1 else None
        A1, A2, O3 = w["A1"], w["A2"], w["O3"]
        if sT and fused and self.fused_mlp:
            hip.check(hip.lib().cadre_mlp_fwd(hip.ptr(pP), a.size_T, self.mlp_offs(), hip.ptr(inp), DP, inp_zstride, hip.ptr(A1),
                                              hip.ptr(A2), hip.ptr(O3), B, 2 * Z, None if seg is None else hip.ptr(seg), hip.stream()),
                      "cadre_mlp_fwd")
            return
        for tower in ((None,) if sT else (0, 1)):
            if sT:      # contiguous nets: 2Z towers with uniform stride
                pw, zb, nb, div, zs, cs = pP, 0, 2 * Z, 2, a.size_T, 1
            else:       # strided nets (act/get_value): one launch per tower
                pw, zb, nb, div, zs, cs = pP[tower * a.size_T:], tower, Z, 1, gs * a.size_P, 2
            sg = None if (seg is None or not sT) else (1, seg, B, 2)      # z = 2*net + tower
            hip.gemm(inp, pw[a.t_w1:], A1[zb:], B, hid, DP, DP, DP, hid, shift=pw[a.t_b1:], act=1, batch=nb,
                     a_z=(div, 0, inp_zstride), b_z=(1, 0, zs), c_z=(1, 0, cs * B * hid), s_z=(1, 0, zs), seg=sg)
            hip.gemm(A1[zb:], pw[a.t_w2:], A2[zb:], B, hid, hid, hid, hid, hid, shift=pw[a.t_b2:], act=1, batch=nb,
                     a_z=(1, 0, cs * B * hid), b_z=(1, 0, zs), c_z=(1, 0, cs * B * hid), s_z=(1, 0, zs), seg=sg)
            hip.gemm(A2[zb:], pw[a.t_w3:], O3[zb:], B, NP, hid, hid, hid, NP, shift=pw[a.t_b3:], batch=nb,
                     a_z=(1, 0, cs * B * hid), b_z=(1, 0, zs), c_z=(1, 0, cs * B * NP), s_z=(1, 0, zs), seg=sg)

    # ------------------------------------------------------------------ update_policy
    def update(self, B, inv_b, sorted_rows=False, mlp_grads_ready=None):
        """Forward + loss + backward for the packed minibatch in workspace(B).  Gradients of all 16
        nets are written (not accumulated) into arena.grads.  Returns the device tensor
        losses[3] = (value_loss*vc, action_loss*cc, ent_loss*ec) (agent.py:226-237).
        The launch sequence has fixed shapes and pointers, so after one eager run it is captured
        into a hipGraph per (B, inv_b) and replayed (launch-bound otherwise: ~2 ms of host time).
        With `mlp_grads_ready` (a callable taking an arena range) the sequence is cut where gradient buckets become
        FINAL, and the callable runs at each cut with that bucket's element range — the data-parallel exchange starts the
        bucket's all-reduce there, beside the kernels that follow (Shared_grad_buffers.reduce_bucket_async):
          after the MLP-tower backward            arena[P0:]            (6 MB)   — beside the backward through time
          after the steer nets' weight gradients  arena[:4 size_L]      (37 MB)  — beside the throttle nets' lstm_dw
        the throttle nets' bucket arena[4 size_L:P0] is final when the step ends (the chief's all_reduce takes it)."""
        a = self.a
        # The packed W_hh copies are current iff this learner's own fused optimiser step produced the parameters that are
        # in the arena now (a chief in another process, a broadcast or a checkpoint load change them behind our back: then
        # the update's graph carries the packing launch, as in round 3).
        self._skip_pack = bool(self.fused_pack and self._adam_fresh is not None and self._adam_fresh == self._pkey())
        try:
            if mlp_grads_ready is None:
                self._run("all", B, inv_b, sorted_rows)
                return self.workspace(B)["losses"]
            half = (a.Z // 2) * a.size_L
            for part, rng in (("front", (a.P0, a.total)), ("mid", (0, half)), ("back", None)):
                self._run(part, B, inv_b, sorted_rows)
                if rng is not None:
                    # the hook's arity is read from its signature — never from a TypeError, which would also swallow one
                    # raised INSIDE the hook (a failed collective on this rank only: its peers would wait for ever)
                    if self._hook_takes_range(mlp_grads_ready):
                        mlp_grads_ready(*rng)
                    elif part == "front":                    # (a round-3 style hook without arguments: the MLP bucket only)
                        mlp_grads_ready()
            return self.workspace(B)["losses"]
        finally:
            self._skip_pack = False                          # (act / get_value outside an update always check the copies)

    @staticmethod
    def _hook_takes_range(hook):
        import inspect
        try:
            ps = [q for q in inspect.signature(hook).parameters.values()]
        except (TypeError, ValueError):
            return True
        if any(q.kind == q.VAR_POSITIONAL for q in ps):
            return True
        return sum(q.kind in (q.POSITIONAL_ONLY, q.POSITIONAL_OR_KEYWORD) for q in ps) >= 1

    def _run(self, part, B, inv_b, sorted_rows):
        if not self.use_graphs:
            return self._update_body(B, inv_b, sorted_rows, part)
        key = (part, B, inv_b, sorted_rows, self._skip_pack)
        g = self._graphs.get(key)
        if g is None:
            n0 = hip.N_CALLS
            self._update_body(B, inv_b, sorted_rows, part)          # eager warm-up (func attributes, lazy init)
            self.launches[(part, B)] = hip.N_CALLS - n0             # kernel launches of this part of the step (latest variant)
            if self._graphs.get(("warm",) + key):
                torch.cuda.synchronize()
                self._graphs[key] = self._capture(lambda: self._update_body(B, inv_b, sorted_rows, part))
            else:
                self._graphs[("warm",) + key] = True
            return
        g.replay()

    @staticmethod
    def _capture(fn):
        """Capture fn() into a hipGraph.  The cyclic garbage collector is held off for the duration:
        a finaliser that runs inside the capture window (an older agent's graphs or tensors being
        destroyed) issues HIP calls that are illegal while the stream is capturing and aborts the
        process from a destructor."""
        import gc
        g = torch.cuda.CUDAGraph()
        was = gc.isenabled()
        gc.collect()
        gc.disable()
        try:
            with torch.cuda.graph(g, capture_error_mode="thread_local"):
                fn()
        finally:
            if was:
                gc.enable()
        return g

    def _update_body(self, B, inv_b, sorted_rows=False, part="all"):
        """part: "all", or the three cuts of the same launch sequence: "front" (forward, loss, MLP-tower backward, dh_S),
        "mid" (backward through time + the weight gradients of the steer nets, arena nets 0 .. Z/2 - 1), "back" (the weight
        gradients of the throttle nets)."""
        a, S = self.a, self.S
        w = self.workspace(B)
        Z, C = a.Z, a.C
        L, st = hip.lib(), hip.stream()
        DP, H4, hid, NP = a.DP, a.H4, a.hid, a.NP
        seg = w["seg"] if sorted_rows else None         # rows sorted by command: skip tiles a net does not own
        cmd = hip.ptr(w["commands"]) if sorted_rows else None
        sgM1 = None if seg is None else (1, seg, B, 1)  # M tiles, z = net
        sgM2 = None if seg is None else (1, seg, B, 2)  # M tiles, z = 2*net + tower
        sgK1 = None if seg is None else (2, seg, B, 1)  # k tiles (rows), z = net
        sgK2 = None if seg is None else (2, seg, B, 2)
        front, back = part in ("all", "front"), part in ("all", "mid", "back")
        O3, dO3 = w["O3"], w["dO3"]
        if front:
            self._forward(w, B, (0, 1, Z), C, seg=seg, fused_mlp=True)
            hip.check(L.cadre_ppo_loss(hip.ptr(O3), NP, 2 * B * NP, hip.ptr(O3[1]), NP, 2 * B * NP,
                                       hip.ptr(w["actions"]), hip.ptr(w["commands"]), hip.ptr(w["old_values"]),
                                       hip.ptr(w["returns"]), hip.ptr(w["old_logp"]), hip.ptr(w["adv"]), B, C,
                                       a.n_out[0], a.n_out[1], self.clip, self.vc, self.cc, self.ec, inv_b,
                                       hip.ptr(w["losses"]), hip.ptr(dO3), hip.ptr(dO3[1]), hip.ptr(w["loss_scratch"]),
                                       hip.ptr(w["sync"][Z * S:]), st), "cadre_ppo_loss")
        # ---------------- backward: MLP towers (16 = 2Z batched)
        Gr = a.grads
        pP, gP = a.params[a.P0:], Gr[a.P0:]
        sT, nb = a.size_T, 2 * Z
        A1, A2, dA1, dA2 = w["A1"], w["A2"], w["dA1"], w["dA2"]
        Hs = w["Hs"]
        zT = (1, 0, sT)

        def layer_bwd(dY, n_y, Xin, ldx_, n_x, x_z, o_w, o_b, dX):
            # dW = dY^T X ; db = colsum(dY) ; dX = dY W
            hip.gemm(dY, Xin, gP[o_w:], n_y, n_x, B, n_y, ldx_, n_x, a_mode=1, b_mode=1, batch=nb,
                     a_z=(1, 0, B * n_y), b_z=x_z, c_z=zT, seg=sgK2)
            hip.check(L.cadre_colsum(hip.ptr(dY), n_y, B * n_y, hip.ptr(gP[o_b:]), sT, B, n_y, nb, 0, st), "cadre_colsum")
            if dX is not None:
                hip.gemm(dY, pP[o_w:], dX, B, n_x, n_y, n_y, n_x, n_x, b_mode=1, batch=nb,
                         a_z=(1, 0, B * n_y), b_z=zT, c_z=(1, 0, B * n_x), seg=sgM2)

        dH, dC = w["dH"], w["dC"]
        if front and self.fused_mlp:
            sgq = None if seg is None else hip.ptr(seg)
            hip.check(L.cadre_mlp_bwd(hip.ptr(pP), sT, self.mlp_offs(), hip.ptr(dO3), hip.ptr(A1), hip.ptr(A2), hip.ptr(dA1), hip.ptr(dA2),
                                      hip.ptr(dH), DP, B * DP, B, nb, sgq, st), "cadre_mlp_bwd")
            hip.check(L.cadre_mlp_dw(hip.ptr(dO3), hip.ptr(dA2), hip.ptr(dA1), hip.ptr(A2), hip.ptr(A1), hip.ptr(Hs[:, S]), DP,
                                     (S + 1) * B * DP, hip.ptr(gP), sT, self.mlp_offs(), B, nb, sgq, st), "cadre_mlp_dw")
        elif front:
            layer_bwd(dO3, NP, A2, hid, hid, (1, 0, B * hid), a.t_w3, a.t_b3, dA2)
            hip.check(L.cadre_relu_bwd(hip.ptr(A2), hip.ptr(dA2), nb * B * hid, cmd, B, hid, C, st), "cadre_relu_bwd")
            layer_bwd(dA2, hid, A1, hid, hid, (1, 0, B * hid), a.t_w2, a.t_b2, dA1)
            hip.check(L.cadre_relu_bwd(hip.ptr(A1), hip.ptr(dA1), nb * B * hid, cmd, B, hid, C, st), "cadre_relu_bwd")
            layer_bwd(dA1, hid, Hs[:, S], DP, DP, (2, 0, (S + 1) * B * DP), a.t_w1, a.t_b1, None)
            # dh_S = dZ1_actor W1_actor + dZ1_critic W1_critic   (two launches, second accumulates)
            for tower in (0, 1):
                hip.gemm(dA1[tower:], pP[tower * sT + a.t_w1:], dH, B, DP, hid, hid, DP, DP, b_mode=1, batch=Z,
                         a_z=(1, 0, 2 * B * hid), b_z=(1, 0, a.size_P), c_z=(1, 0, B * DP),
                         resid=dH if tower else None, ldr=DP, r_z=(1, 0, B * DP), seg=sgM1)
        if not back:
            return w["losses"]
        # ---------------- backward through time (autograd of models.py:148-151)
        G, dG, Cs, TC, X = w["G"], w["dG"], w["Cs"], w["TC"], w["X"]
        pL, gL, sL = a.params, Gr, a.size_L
        H4P = a.H4P
        sgp = None if seg is None else hip.ptr(seg)
        dGp = w["dGp"]                                      # dG of a step in fragment order: ping-pong pair
        gps = dGp.stride(1)
        if part != "back":
            for t in range(S, 0, -1):
                # t == S: dh_{S-1} = dH (MLP towers), no product; else dh_{t-1} = dG_t W_hh.  Then the cell backward of step
                # t-1 in the same launch: dG_{t-1} (row-major for the weight gradients, fragment order for the next step), dc_{t-2}
                src = None if t == S else hip.ptr(dGp[t & 1])
                hip.check(L.cadre_lstm_step_bwd(hip.ptr(self._wp[1]), self._wp[1].stride(0), src, hip.ptr(dGp[(t - 1) & 1]), gps,
                                                hip.ptr(dG[:, t - 1]), hip.ptr(G[:, t - 1]), H4P, S * B * H4P,
                                                hip.ptr(dH) if t == S else None, hip.ptr(dC), B * DP, hip.ptr(TC[:, t]),
                                                hip.ptr(Cs[:, t - 1]), DP, (S + 1) * B * DP, B, a.D, Z, cmd, C, sgp, t & 1, st),
                          "cadre_lstm_step_bwd")
        # dW_hh = sum_t dG_t^T h_{t-1} ; dW_ih = sum_t dG_t^T x_t ; db_ih = db_hh = colsum(dG): one launch for all nets, or —
        # when the exchange overlaps — one per head: nets [z0, z0 + nz) of the arena (X: one input block per head, x_div = C)
        z0, nz = {"all": (0, Z), "mid": (0, Z // 2), "back": (Z // 2, Z - Z // 2)}[part]
        gz = gL[z0 * sL:]
        hip.check(L.cadre_lstm_dw(hip.ptr(dG[z0:]), H4P, S * B * H4P, hip.ptr(Hs[z0:]), hip.ptr(X[z0 // C:]), DP, (S + 1) * B * DP,
                                  S * B * DP, C, hip.ptr(gz[a.o_whh:]), hip.ptr(gz[a.o_wih:]), hip.ptr(gz[a.o_bih:]),
                                  hip.ptr(gz[a.o_bhh:]), DP, sL, B, S, H4, DP, nz, None if seg is None else hip.ptr(seg[z0:]), st),
                  "cadre_lstm_dw")
        return w["losses"]

    # ------------------------------------------------------------------ optimiser (chief.py:13-21)
    def clip_adam(self, lr=3e-4, max_grad_norm=250.0, betas=(0.9, 0.999), eps=1e-8):
        """Three launches (prep, per-model square norms, Adam) with the step count in device memory,
        captured into a hipGraph per hyper-parameter set."""
        a = self.a
        a.ensure_adam()
        a.step += 1
        fused = self.fused_pack
        key = ("adam", float(lr), float(max_grad_norm), float(betas[0]), float(betas[1]), float(eps), fused)
        if fused:
            self._alloc_wp()

        def body():
            if fused:
                hip.check(hip.lib().cadre_clip_adam_pack_graph(
                    hip.ptr(a.params), hip.ptr(a.grads), hip.ptr(a.exp_avg), hip.ptr(a.exp_avg_sq), hip.ptr(a.seg_off),
                    2 * a.Z, hip.ptr(a.norms2), key[2], key[1], key[3], key[4], key[5], hip.ptr(a.step_dev),
                    a.Z, a.size_L, a.o_whh, a.H4, a.DP, a.D, hip.ptr(self._wp[0]), hip.ptr(self._wp[1]), self._wp.stride(1),
                    hip.stream()), "cadre_clip_adam_pack_graph")
            else:
                hip.check(hip.lib().cadre_clip_adam_graph(
                    hip.ptr(a.params), hip.ptr(a.grads), hip.ptr(a.exp_avg), hip.ptr(a.exp_avg_sq), hip.ptr(a.seg_off),
                    2 * a.Z, hip.ptr(a.norms2), key[2], key[1], key[3], key[4], key[5], hip.ptr(a.step_dev),
                    hip.stream()), "cadre_clip_adam_graph")

        def done():
            self._adam_fresh = self._pkey() if fused else None      # (the copies now match the stepped parameters)
            self._wp_key = self._adam_fresh if fused else self._wp_key
        if not self.use_graphs:
            body()
            return done()
        g = self._graphs.get(key)
        if g is None:
            if self._graphs.get(("warm",) + key):
                torch.cuda.synchronize()
                g = self._capture(body)
                self._graphs[key] = g
                g.replay()
                return done()
            self._graphs[("warm",) + key] = True
            body()
            return done()
        g.replay()
        done()

    def clip_adam_sharded(self, lo, hi, all_reduce_norms, lr=3e-4, max_grad_norm=250.0, betas=(0.9, 0.999), eps=1e-8):
        """The same step on arena elements [lo, hi) only (data-parallel ranks after a reduce-scatter of the
        gradient arena, chief.py:13-21 semantics): partial per-model square norms of the shard ->
        `all_reduce_norms(norms2[:n_models])` (SUM of 16 doubles over the ranks) -> clip + Adam on the shard.
        The Adam moments exist for the shard only (1/N of the state and of the pass's HBM traffic)."""
        a = self.a
        if getattr(a, "_shard", None) != (lo, hi):
            if a.step:
                raise hip.CadreHipError("the optimiser shard changed after %d steps (Adam state is per shard)" % a.step)
            a._shard = (lo, hi)
            a.exp_avg = torch.zeros(hi - lo, device=a.device)
            a.exp_avg_sq = torch.zeros(hi - lo, device=a.device)
        a.step += 1
        L, st, nm = hip.lib(), hip.stream(), 2 * a.Z
        hip.check(L.cadre_clip_adam_norms(hip.ptr(a.grads), hip.ptr(a.seg_off), nm, hip.ptr(a.norms2), float(lr),
                                          float(betas[0]), float(betas[1]), hip.ptr(a.step_dev), lo, hi, st),
                  "cadre_clip_adam_norms")
        all_reduce_norms(a.norms2[:nm])
        hip.check(L.cadre_clip_adam_apply(hip.ptr(a.params), hip.ptr(a.grads), hip.ptr(a.exp_avg), hip.ptr(a.exp_avg_sq),
                                          hip.ptr(a.seg_off), nm, hip.ptr(a.norms2), float(max_grad_norm),
                                          float(betas[0]), float(betas[1]), float(eps), lo, hi, st),
                  "cadre_clip_adam_apply")

    # ------------------------------------------------------------------ inference (act / get_value)
    def infer(self, feats, commands, h0=None, c0=None):
        """LSTM + both towers of net (steer, commands[0]) and (throttle, commands[1]) at batch 1.
        feats: [2][S][DP-padded] views or one shared [S][D] feature block.  Returns (O3 [4][1][NP]:
        rows (steer actor, steer critic, throttle actor, throttle critic))."""
        a, S = self.a, feats.shape[-2]
        w = self.workspace(1, 2, S)
        X = w["X"]
        if feats.dim() == 2:
            X[:, :, 0, :a.D].copy_(feats.unsqueeze(0).expand(2, S, a.D))
        else:
            X[:, :, 0, :a.D].copy_(feats)
        if h0 is None:
            w["h0"].zero_(); w["c0"].zero_()
        else:
            w["h0"][:, 0, :a.D].copy_(h0); w["c0"][:, 0, :a.D].copy_(c0)
        g_s, g_t = commands[0], a.C + commands[1]
        self._forward(w, 1, (g_s, g_t - g_s, 2), 1, S=S)
        return w["O3"], w["Hs"][:, S], w["Cs"][:, S]

    def infer_rows(self, feats):
        """LSTM + both towers of ALL command nets of both heads on W independent rows in one pass (zero initial
        state, agent.py:38-40): feats [2][W][S][D] (head, row).  Returns O3 [4*C][W][NP] (tower z = 2*net + t,
        net = head*C + c) — the caller picks each row's command.  One launch chain instead of W (get_value for every
        worker of a GPU, train.py:76-80)."""
        a, W, S = self.a, feats.shape[1], feats.shape[2]
        w = self.workspace(W, a.Z, S)
        w["X"].view(2, S, W, a.DP)[:, :, :, :a.D].copy_(feats.permute(0, 2, 1, 3))
        w["h0"].zero_(); w["c0"].zero_()
        self._forward(w, W, (0, 1, a.Z), a.C, S=S)
        return w["O3"]

    # ------------------------------------------------------------------ stand-alone module calls
    def lstm_module_forward(self, g, x, h0, c0):
        """`LSTM.forward` (models.py:139-152) of arena net g: x [T*N, D] time-major (or [N, D]), hidden
        [N, D] -> (h_T, c_T) [N, D].  Inference only."""
        a = self.a
        N = h0.shape[0]
        S = x.shape[0] // N
        w = self.workspace(N, 1, S)
        w["X"][0].view(S * N, a.DP)[:, :a.D].copy_(x)
        w["h0"][0][:, :a.D].copy_(h0)
        w["c0"][0][:, :a.D].copy_(c0)
        self._forward(w, N, (g, 1, 1), 1, S=S, mlp=False)
        return w["Hs"][0, S, :, :a.D].clone(), w["Cs"][0, S, :, :a.D].clone()

    def mlp_module_forward(self, g, feat):
        """critic + actor of arena net g on feat [B, D] -> (raw logits [B, NP], values [B, 1]) views."""
        a = self.a
        B = feat.shape[0]
        w = self.workspace(B, 1, 1)
        inp = w["Hs"][0, 1]
        inp[:, :a.D].copy_(feat)
        self._mlp(w, B, (g, 1, 1), w["Hs"][:, 1], 2 * B * a.DP)
        return w["O3"][0], w["O3"][1, :, :1]
