"""Mirror of reference ppo_agent/chief.py:8-27: once every worker has handed in its gradients,
clip each model's gradient norm to `max_grad_norm`, take one Adam step over all 16 nets,
clear the buffers and flip the traffic light — as ONE fused HIP pass over the arena
(cadre_clip_adam) instead of 16 clip calls + a 112-tensor optimizer.step()."""
import time

import torch

from ..learner import PPOLearnerHIP


def _hyper(optimizer):
    g = optimizer.param_groups[0]
    return g["lr"], tuple(g.get("betas", (0.9, 0.999))), g.get("eps", 1e-8)


def chief_step(shared_grad_buffers, optimizer, max_grad_norm, lr=None, zero_grads=True):
    """One optimiser step (chief.py:13-23) on the arena behind `shared_grad_buffers`: the pending
    cross-rank SUM of the gradients (ONE exchange per optimiser step, however many worker agents of this
    process handed gradients in), then per-model clip + Adam, then clear the buffers.  Two forms of the
    exchange (Shared_grad_buffers.exchange_mode): all-reduce + replicated optimiser, or reduce-scatter +
    optimiser on this rank's shard + all-gather of the parameters — identical parameters either way.
    Hyper-parameters come from `optimizer` (the reference's optim.Adam, main.py:52) or, without one,
    from `lr` (train_cfg.lr) with Adam's defaults.
    `zero_grads=True` is the reference's `shared_grad_buffers.reset()` (models.py:255-258): the gradient arena is
    cleared.  A caller whose NEXT writer of the arena is the fused `update_policy` — which writes every element, it
    does not accumulate — may pass False and save the 80 MB fill (`learner_section` does; the stand-alone modules'
    autograd path and foreign arenas ACCUMULATE into `p.grad` / the arena and need the fill)."""
    arena = shared_grad_buffers.arena
    if optimizer is not None:
        lr, betas, eps = _hyper(optimizer)
    else:
        lr, betas, eps = (3e-4 if lr is None else float(lr)), (0.9, 0.999), 1e-8
    step = getattr(arena, "_learner", None)
    if step is None:
        arena._learner = step = PPOLearnerHIP(arena)
    if shared_grad_buffers.exchange_mode() == "sharded":
        # (nothing handed in since the last exchange: no collective, but still the SHARDED optimiser — the arena's Adam
        #  moments exist for this rank's shard only; every rank's gradients are then its own, as in the replicated form)
        rng = shared_grad_buffers.reduce_scatter() or shared_grad_buffers.shard()
        step.clip_adam_sharded(rng[0], rng[1], shared_grad_buffers.all_reduce_norms, lr=lr,
                               max_grad_norm=max_grad_norm, betas=betas, eps=eps)
        shared_grad_buffers.all_gather_params()
    else:
        shared_grad_buffers.all_reduce()
        step.clip_adam(lr=lr, max_grad_norm=max_grad_norm, betas=betas, eps=eps)
    shared_grad_buffers.reset(zero=zero_grads)


def chief(update_threshold, traffic_light, counter, shared_model_list, shared_grad_buffers, optimizer,
          son_process_counter, max_grad_norm, total_thread):
    while True:
        if counter.get() >= update_threshold:
            chief_step(shared_grad_buffers, optimizer, max_grad_norm, zero_grads=True)
            # separate-process mode (main.py:57-60): the workers' weight pull and next add_gradient run on
            # their own streams in other processes — the clip+Adam graph and the gradient clear must have
            # finished on the device before the light flips
            torch.cuda.current_stream().synchronize()
            counter.reset()
            traffic_light.switch()
        elif son_process_counter.get() >= total_thread:
            print("chief finished.")
            break
        else:
            time.sleep(0.0005)      # the reference polls with sleep(1): <= 1 optimiser step / s
