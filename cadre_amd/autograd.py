"""Autograd through the stand-alone modules of the reference API: `LSTM.forward` (ppo_agent/models.py:139-152) and the
actor / critic towers behind `Model.evaluate_actions / get_value` (models.py:171-177, 195-208; distributions.py:34-40).

The training path of the reference (`update_policy`, agent.py:166-237) does not come through here — it is the fused
learner step.  These two `torch.autograd.Function`s serve a caller that builds its OWN loss on the modules: forward and
backward run on the same HIP kernels as the fused step (cadre_lstm_step_fwd / _bwd / cadre_lstm_dw, cadre_mlp_fwd /
_bwd / _dw), for one net at a time; the module parameters are views into the arena, so the gradients autograd routes
to `p.grad` land in the gradient arena like those of `update_policy`."""
import torch

from . import hip


class LstmSequence(torch.autograd.Function):
    """(h_T, c_T) = LSTMCell unrolled over x [S*N, D] (time-major) from (h0, c0) [N, D] for arena net g."""

    @staticmethod
    def forward(ctx, learner, g, x, h0, c0, w_ih, w_hh, b_ih, b_hh):
        a = learner.a
        N = h0.shape[0]
        S = x.shape[0] // N
        w = learner.workspace(N, 1, S)
        w["X"][0].view(S * N, a.DP)[:, :a.D].copy_(x.detach())
        w["h0"][0][:, :a.D].copy_(h0.detach())
        w["c0"][0][:, :a.D].copy_(c0.detach())
        learner._forward(w, N, (g, 1, 1), 1, S=S, mlp=False)
        ctx.learner, ctx.g, ctx.N, ctx.S = learner, g, N, S
        # what the backward reads, taken out of the shared workspace (later calls overwrite it)
        ctx.saved = tuple(w[k][:1].clone() for k in ("G", "Hs", "Cs", "TC")) + (w["X"][:1].clone(),)
        return w["Hs"][0, S, :, :a.D].clone(), w["Cs"][0, S, :, :a.D].clone()

    @staticmethod
    def backward(ctx, dh, dc):
        learner, g, N, S = ctx.learner, ctx.g, ctx.N, ctx.S
        a = learner.a
        L, st = hip.lib(), hip.stream()
        G, Hs, Cs, TC, X = ctx.saved
        dev = G.device
        DP, H4, H4P, D = a.DP, a.H4, a.H4P, a.D
        z = lambda *shape: torch.zeros(*shape, device=dev)
        dH, dC = z(1, N, DP), z(1, N, DP)
        if dh is not None:
            dH[0, :, :D].copy_(dh)
        if dc is not None:
            dC[0, :, :D].copy_(dc)
        dG = z(1, S, N, H4P)
        dGp = z(2, 1, (N + 15) // 16, 16 * H4P)
        learner.packed_weights(0, 1, a.Z)                    # (refreshed when the parameters changed)
        wp_b = learner._wp[1, g:]
        for t in range(S, 0, -1):                            # as learner._update_body: cell t-1 per launch
            src = None if t == S else hip.ptr(dGp[t & 1])
            hip.check(L.cadre_lstm_step_bwd(hip.ptr(wp_b), learner._wp.stride(1), src, hip.ptr(dGp[(t - 1) & 1]), dGp.stride(1),
                                            hip.ptr(dG[:, t - 1]), hip.ptr(G[:, t - 1]), H4P, S * N * H4P,
                                            hip.ptr(dH) if t == S else None, hip.ptr(dC), N * DP, hip.ptr(TC[:, t]),
                                            hip.ptr(Cs[:, t - 1]), DP, (S + 1) * N * DP, N, D, 1, None, 1, None, t & 1, st),
                      "cadre_lstm_step_bwd")
        grads = z(a.size_L)
        hip.check(L.cadre_lstm_dw(hip.ptr(dG), H4P, S * N * H4P, hip.ptr(Hs), hip.ptr(X), DP, (S + 1) * N * DP, S * N * DP, 1,
                                  hip.ptr(grads[a.o_whh:]), hip.ptr(grads[a.o_wih:]), hip.ptr(grads[a.o_bih:]), hip.ptr(grads[a.o_bhh:]),
                                  DP, a.size_L, N, S, H4, DP, 1, None, st), "cadre_lstm_dw")
        gv = a.lstm_views(grads, g, base=0)
        pL = a.params[g * a.size_L:]
        # dx_t = dG_t W_ih ; dh_{-1} = dG_0 W_hh  (k-major B: W [4D][DP] row-major)
        need = ctx.needs_input_grad
        dx = dh0 = None
        if need[2]:
            dxp = z(S * N, DP)
            hip.gemm(dG[0].view(S * N, H4P), pL[a.o_wih:], dxp, S * N, DP, H4, H4P, DP, DP, b_mode=1)
            dx = dxp[:, :D]
        if need[3]:
            dhp = z(N, DP)
            hip.gemm(dG[0, 0], pL[a.o_whh:], dhp, N, DP, H4, H4P, DP, DP, b_mode=1)
            dh0 = dhp[:, :D]
        dc0 = dC[0, :, :D] if need[4] else None
        return (None, None, dx, dh0, dc0, gv["rnn.weight_ih"], gv["rnn.weight_hh"], gv["rnn.bias_ih"], gv["rnn.bias_hh"])


class TowerPair(torch.autograd.Function):
    """(raw logits [B, n_out], value [B, 1]) of arena net g's actor and critic towers on feat [B, D].
    `params`: actor (control.linear.0/2/4 weight, bias) then critic (0/2/4 weight, bias) — 12 tensors."""

    @staticmethod
    def forward(ctx, learner, g, n_out, feat, *params):
        a = learner.a
        B = feat.shape[0]
        dev = feat.device
        L, st = hip.lib(), hip.stream()
        H = torch.zeros(1, B, a.DP, device=dev)
        H[0, :, :a.D].copy_(feat.detach())
        A1 = torch.empty(2, B, a.hid, device=dev)
        A2 = torch.empty(2, B, a.hid, device=dev)
        O3 = torch.empty(2, B, a.NP, device=dev)
        pP = a.params[a.P0 + g * a.size_P:]
        hip.check(L.cadre_mlp_fwd(hip.ptr(pP), a.size_T, learner.mlp_offs(), hip.ptr(H), a.DP, B * a.DP, hip.ptr(A1), hip.ptr(A2),
                                  hip.ptr(O3), B, 2, None, st), "cadre_mlp_fwd")
        ctx.learner, ctx.g, ctx.n_out, ctx.B = learner, g, n_out, B
        ctx.saved = (H, A1, A2)
        return O3[0, :, :n_out].clone(), O3[1, :, :1].clone()

    @staticmethod
    def backward(ctx, dlogits, dvalue):
        learner, g, n_out, B = ctx.learner, ctx.g, ctx.n_out, ctx.B
        a = learner.a
        L, st = hip.lib(), hip.stream()
        H, A1, A2 = ctx.saved
        dev = H.device
        dO3 = torch.zeros(2, B, a.NP, device=dev)
        if dlogits is not None:
            dO3[0, :, :n_out].copy_(dlogits)
        if dvalue is not None:
            dO3[1, :, :1].copy_(dvalue)
        dA1, dA2 = torch.empty_like(A1), torch.empty_like(A2)
        dH = torch.empty(1, B, a.DP, device=dev)
        pP = a.params[a.P0 + g * a.size_P:]
        hip.check(L.cadre_mlp_bwd(hip.ptr(pP), a.size_T, learner.mlp_offs(), hip.ptr(dO3), hip.ptr(A1), hip.ptr(A2), hip.ptr(dA1),
                                  hip.ptr(dA2), hip.ptr(dH), a.DP, B * a.DP, B, 2, None, st), "cadre_mlp_bwd")
        grads = torch.zeros(a.size_P, device=dev)
        hip.check(L.cadre_mlp_dw(hip.ptr(dO3), hip.ptr(dA2), hip.ptr(dA1), hip.ptr(A2), hip.ptr(A1), hip.ptr(H), a.DP, B * a.DP,
                                 hip.ptr(grads), a.size_T, learner.mlp_offs(), B, 2, None, st), "cadre_mlp_dw")
        gv = a.ppo_views(grads, g, base=0)
        order = [t + k for t in ("control.linear", "critic") for k in (".0.weight", ".0.bias", ".2.weight", ".2.bias", ".4.weight", ".4.bias")]
        dfeat = dH[0, :, :a.D] if ctx.needs_input_grad[3] else None
        return (None, None, None, dfeat) + tuple(gv[k] for k in order)


def tower_params(model):
    """the 12 parameter tensors of a `Model` in TowerPair's order"""
    lin, cr = model.control.linear, model.critic
    return [p for seq in (lin, cr) for i in (0, 2, 4) for p in (seq[i].weight, seq[i].bias)]


def wants_grad(*tensors):
    return torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in tensors)
