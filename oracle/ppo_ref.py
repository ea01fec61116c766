"""Oracle: PPO learner math on CPU (TEST INFRASTRUCTURE ONLY).

numpy float32 with explicit operation order for the strict parts (GAE scan, sampling rule),
plain torch-CPU fp32 + autograd for the floating-point network math.  Each function cites
the reference lines it restates; tests/golden/make_golden.py pins them against the imported
reference.

Parameters are passed as {model_name: {param_name: tensor}} using the reference's
model_dict / state_dict names (ppo_agent/models.py:100-125).
"""
import numpy as np
import torch
import torch.nn.functional as F


# ----------------------------------------------------------------------------- networks
def lstm_cell(x, h, c, p):
    """nn.LSTMCell forward (gate order i,f,g,o), the cell wrapped by models.py:130-137."""
    g = F.linear(x, p["rnn.weight_ih"], p["rnn.bias_ih"]) + F.linear(h, p["rnn.weight_hh"], p["rnn.bias_hh"])
    i, f, gg, o = g.chunk(4, dim=1)
    c2 = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(gg)
    h2 = torch.sigmoid(o) * torch.tanh(c2)
    return h2, c2


def lstm_forward(x, hidden, p):
    """models.py:139-152: one step if x.size(0)==h.size(0), else time-major [T*N,D] unrolled
    over T, returning the last h."""
    h, c = hidden
    if x.size(0) == h.size(0):
        h, c = lstm_cell(x, h, c, p)
    else:
        N = h.size(0)
        T = x.size(0) // N
        xs = x.view(T, N, x.size(1))
        for t in range(T):
            h, c = lstm_cell(xs[t], h, c, p)
    return h, (h, c)


def mlp3(x, p, tower):
    """critic (models.py:171-177) / actor (distributions.py:34-40): Linear-ReLU-Linear-ReLU-Linear."""
    y = F.relu(F.linear(x, p[tower + ".0.weight"], p[tower + ".0.bias"]))
    y = F.relu(F.linear(y, p[tower + ".2.weight"], p[tower + ".2.bias"]))
    return F.linear(y, p[tower + ".4.weight"], p[tower + ".4.bias"])


def categorical_logits(x, p):
    """distributions.py:66-83: Categorical(logits=x).logits == x - logsumexp(x)."""
    raw = mlp3(x, p, "control.linear")
    return raw - raw.logsumexp(dim=-1, keepdim=True)


def evaluate_actions(x, action, p):
    """models.py:199-208 -> (value [N,1], log_prob [N,1], entropy [N,1])."""
    value = mlp3(x, p, "critic")
    logits = categorical_logits(x, p)
    lp = logits.gather(1, action.view(-1, 1).long())
    probs = torch.softmax(logits, dim=-1)          # Categorical.probs = logits_to_probs(logits)
    min_real = torch.finfo(logits.dtype).min
    ent = -(torch.clamp(logits, min=min_real) * probs).sum(-1, keepdim=True)
    return value, lp, ent


def sample_from_logits(logits, q):
    """models.py:184-189 + distributions.py:96-99: Categorical(probs=softmax(logits)).sample().
    On torch CPU this is argmax(p / q) with q = empty_like(p).exponential_(1) drawn from the
    global CPU generator (re-verified by make_golden.py).  `q` is supplied."""
    probs = torch.softmax(logits, dim=-1)
    probs = probs / probs.sum(-1, keepdim=True)    # Categorical(probs=...) renormalises
    return torch.argmax(probs / q, dim=-1)


# ----------------------------------------------------------------------------- storage math
def gae_returns(rewards, value_preds, masks, next_value, gamma, tau):
    """storage.py:69-76, strict fp32 left-to-right, no fused multiply-add.
    rewards/value_preds/masks: float32 [T+1]; returns (returns [T+1] with [T]=0 untouched
    semantics handled by caller, value_preds with [T]=next_value)."""
    f = np.float32
    T = rewards.shape[0] - 1
    V = value_preds.astype(np.float32).copy()
    V[T] = f(next_value)
    ret = np.zeros(T + 1, np.float32)
    g32 = f(gamma)                 # python float * f32 tensor -> scalar rounded to f32
    gt32 = f(gamma * tau)          # double product first (storage.py:75), then rounded
    gae = f(0.0)
    for t in range(T - 1, -1, -1):
        t1 = f(g32 * V[t + 1])
        t2 = f(t1 * masks[t])
        t3 = f(rewards[t] + t2)
        delta = f(t3 - V[t])
        u1 = f(gt32 * masks[t])
        u2 = f(u1 * gae)
        gae = f(delta + u2)
        ret[t] = f(gae + V[t])
    return ret, V


def advantages(returns, value_preds, normalise=True):
    """train.py:82-88: adv = ret[:-1]-V[:-1]; (adv-mean)/(std_unbiased+1e-8)."""
    adv = torch.as_tensor(returns[:-1]) - torch.as_tensor(value_preds[:-1])
    if normalise:
        adv = (adv - adv.mean()) / (adv.std() + 1e-8)
    return adv


def sampler_indices(T, mini_batch_num):
    """storage.py:93-98: BatchSampler(SubsetRandomSampler(range(T)), T//mbn, drop_last=False).
    One torch.randperm(T) from the global CPU generator."""
    perm = torch.randperm(T).tolist()
    bs = T // mini_batch_num
    return [perm[i:i + bs] for i in range(0, T, bs)]


def gather_minibatch(st, idx, adv):
    """storage.py:99-120 over a dict of storage tensors -> the 9-tuple."""
    idx = torch.as_tensor(idx, dtype=torch.long)
    obs = st["obs"][idx].permute(1, 0, 2)
    obs = obs.reshape(-1, obs.size(-1))
    return (obs, st["action"][idx], st["value_preds"][idx], st["returns"][idx], st["masks"][idx],
            st["action_log_probs"][idx], adv[idx], [st["hn"][idx], st["cn"][idx]], st["command"][idx])


# ----------------------------------------------------------------------------- update
def head_losses(params, head, samples, clip, command_num=4):
    """agent.py:166-196 for one head: command-masked sum over all command nets, then the three
    per-head loss terms (action, value, entropy)."""
    obs, act, old_v, ret, _m, old_lp, adv, hidden, cmd = samples
    cur_v = cur_lp = ent = 0
    for c in range(command_num):
        x, _ = lstm_forward(obs.clone(), hidden, params["%s_lstm_%d" % (head, c)])
        v, lp, e = evaluate_actions(x, act, params["%s_ppo_%d" % (head, c)])
        m = (cmd == c)
        cur_v = cur_v + v * m
        cur_lp = cur_lp + lp * m
        ent = ent + e * m
    ratio = torch.exp(cur_lp - old_lp)
    s1 = ratio * adv
    s2 = torch.clamp(ratio, 1.0 - clip, 1.0 + clip) * adv
    a_loss = -torch.min(s1, s2).mean()
    vpc = old_v + (cur_v - old_v).clamp(-clip, clip)
    v_loss = 0.5 * torch.max((cur_v - ret).pow(2), (vpc - ret).pow(2)).mean()
    return a_loss, v_loss, ent.mean()


def update_policy(params, steer_samples, throttle_samples, ent_coeff=0.01, value_coeff=0.1,
                  clip_coeff=1.0, clip=0.1, command_num=4):
    """agent.py:166-237.  `params` leaves must have requires_grad=True; grads are zeroed then
    populated by backward.  Returns the three floats of agent.py:237."""
    a1, v1, e1 = head_losses(params, "steer", steer_samples, clip, command_num)
    a2, v2, e2 = head_losses(params, "throttle", throttle_samples, clip, command_num)
    value_loss = (v1 + v2) * value_coeff
    action_loss = (a1 + a2) * clip_coeff
    ent_loss = (e1 + e2) * ent_coeff
    total = value_loss + action_loss - ent_loss
    for m in params.values():
        for p in m.values():
            p.grad = None
    total.backward()
    for m in params.values():            # zero_grad() semantics for params the loss did not touch
        for p in m.values():
            if p.grad is None:
                p.grad = torch.zeros_like(p)
    return value_loss.item(), action_loss.item(), ent_loss.item()


def chief_step(params, grads, adam_state, step, lr=3e-4, max_grad_norm=250.0,
               betas=(0.9, 0.999), eps=1e-8):
    """chief.py:13-21 + main.py:55: per-model clip_grad_norm_(max_grad_norm) on the summed
    gradients, then one Adam step (torch defaults) over every parameter.  In place on
    params / adam_state ({model:{param:(m,v)}}); `step` is the 1-based step count."""
    b1, b2 = betas
    with torch.no_grad():
        for mn, mp_ in params.items():
            g = grads[mn]
            total = torch.linalg.vector_norm(
                torch.stack([torch.linalg.vector_norm(g[k], 2.0) for k in mp_]), 2.0)
            coef = torch.clamp(max_grad_norm / (total + 1e-6), max=1.0)
            for k, p in mp_.items():
                gk = g[k] * coef
                m, v = adam_state[mn][k]
                m.mul_(b1).add_(gk, alpha=1 - b1)
                v.mul_(b2).addcmul_(gk, gk, value=1 - b2)
                bc1 = 1 - b1 ** step
                bc2 = 1 - b2 ** step
                denom = (v.sqrt() / (bc2 ** 0.5)).add_(eps)
                p.addcdiv_(m, denom, value=-(lr / bc1))


def to_torch_params(state, requires_grad=False):
    out = {}
    for m, d in state.items():
        out[m] = {k: torch.from_numpy(np.ascontiguousarray(v)).clone().requires_grad_(requires_grad)
                  for k, v in d.items()}
    return out
