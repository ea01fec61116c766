"""Mirror of reference ppo_agent/train.py: `train(rank, ...)` worker loop with the learner
section (train.py:76-110) factored into `learner_section` so it can be driven by replayed /
synthetic rollouts (tests, bench.py) as well as by the live CARLA `EnvWrapper`."""
import os

import numpy as np
import torch

from .agent import CadreAgent
from .chief import chief_step
from .models import arena_of, get_vae_output
from .storage import RolloutStorage
from .utils import check_exist


def _get(cfg, key, default=None):
    try:
        return cfg[key]
    except (KeyError, TypeError, IndexError):
        return getattr(cfg, key, default)


def _default_logger():
    """The reference logs per-episode losses through utils.logger.logger (train.py:11,104-110); use it when
    the Cadre checkout is importable, else stay silent."""
    try:
        from utils.logger import logger
        return logger
    except Exception:
        return None


def learner_section(agent, steer_rollout, throttle_rollout, done, train_cfg, shared_grad_buffers,
                    optimizer=None, traffic_light=None, counter=None, shared_model_list=None, in_process_chief=True,
                    fused_gather=True, losses_on_device=False, step_events=None):
    """train.py:76-110.  Returns (value_loss_list, policy_loss_list, ent_loss_list).
    With `in_process_chief` (one process per GPU) the optimiser step runs right after the gradient
    all-reduce instead of waiting on a separate chief process.  `fused_gather` uses the storage ->
    workspace gather kernel and keeps the per-minibatch losses on the device until the end of the
    section (same numbers as the generator/tuple path, one host sync instead of eight).
    The stored command of the bootstrap observation stays on the device (`get_last(as_tensor=True)`): the reference's
    `.item()` (storage.py:88-91) would wait for every kernel enqueued so far — the whole encoder pass — before the host
    may enqueue the rest of the section (0.5-1.1 ms of idle GPU per round).  `losses_on_device` (needs fused_gather):
    return the [steps, 3] loss tensor instead of the three lists, so the caller chooses when to wait; `step_events`:
    a list that receives one timing event before every minibatch step and one after the last."""
    use_adv_norm = train_cfg["use_adv_norm"]
    nv_s, nv_t = agent.get_value(done, steer_rollout.get_last(as_tensor=True), throttle_rollout.get_last(as_tensor=True))
    steer_adv = steer_rollout.compute_returns(nv_s.detach(), normalise=use_adv_norm)
    throttle_adv = throttle_rollout.compute_returns(nv_t.detach(), normalise=use_adv_norm)
    dev_losses = []
    vl, pl, el = [], [], []
    # several ranks, CADRE_GRAD_BUCKETS=1: gradient buckets leave as soon as they are final, beside the rest of the backward
    # (Shared_grad_buffers.overlap_hook) — only with the in-process chief, which collects them in chief_step, and only
    # when the agent's nets live in the arena behind `shared_grad_buffers` (a foreign worker arena is ADDED afterwards)
    hook = shared_grad_buffers.overlap_hook(arena_of(agent.model_dict)) if in_process_chief else None
    for _ in range(train_cfg["ppo_epoch"]):
        if fused_gather:
            idx_s, idx_t = steer_rollout.sample_indices(), throttle_rollout.sample_indices()   # steer draws first
            steps = [("idx", a, b) for a, b in zip(idx_s, idx_t)]
        else:
            steps = [("tup", a, b) for a, b in zip(steer_rollout.feed_forward_generator(steer_adv),
                                                     throttle_rollout.feed_forward_generator(throttle_adv))]
        for kind, a, b in steps:
            if step_events is not None:
                step_events.append(torch.cuda.Event(enable_timing=True)); step_events[-1].record()
            if kind == "idx":
                dev_losses.append(agent.update_policy_from_storages(
                    [(steer_rollout, a, steer_adv, throttle_rollout, b, throttle_adv)], sync=False, mlp_grads_ready=hook))
            else:
                v, p, e = agent.update_policy(a, b)
                vl.append(v); pl.append(p); el.append(e)
            if in_process_chief:
                shared_grad_buffers.add_gradient(agent.model_dict)
                # (the next writer of the gradient arena is the next fused update, which overwrites every element)
                chief_step(shared_grad_buffers, optimizer, train_cfg["max_grad_norm"], lr=_get(train_cfg, "lr"),
                           zero_grads=False)
            else:
                signal_init = traffic_light.get()
                shared_grad_buffers.add_gradient(agent.model_dict)
                counter.increment()
                while traffic_light.get() == signal_init:
                    pass
            if shared_model_list is not None:
                agent.update_model(shared_model_list)
    if step_events is not None:
        step_events.append(torch.cuda.Event(enable_timing=True)); step_events[-1].record()
    if losses_on_device:
        if not dev_losses:
            raise ValueError("losses_on_device needs fused_gather=True")
        return torch.stack(dev_losses)
    if dev_losses:
        for v, p, e in torch.stack(dev_losses).tolist():
            vl.append(v); pl.append(p); el.append(e)
    return vl, pl, el


def train(rank, train_cfg, agent_cfg, env_cfg, rollout_cfg, traffic_light=None, counter=None,
          shared_model_list=None, shared_grad_buffers=None, son_process_counter=None, env_cls=None, logger=None,
          recorder=None):
    if env_cls is None:
        from env_wrapper import EnvWrapper as env_cls        # needs the CARLA stack (reference env_wrapper.py)
    if logger is None:
        logger = _default_logger()
    env_cfg.rank = rank
    for k in ("port", "routes", "scenarios", "town"):
        env_cfg[k] = env_cfg[k][rank]
    env_cfg.seq_length = rollout_cfg.seq_length
    env = env_cls(env_cfg)
    model_dir = os.path.join(env.work_dir, "models")
    check_exist(model_dir)
    num_steps = rollout_cfg.num_steps
    hidden_size, _ = get_vae_output(agent_cfg.model_cfg)
    agent_cfg.rank = rank
    agent = CadreAgent(**agent_cfg)
    device = torch.device("cuda:" + str(agent_cfg.model_cfg.device_num))
    rollout_cfg.hidden_size = hidden_size
    steer_rollout = RolloutStorage(**rollout_cfg); steer_rollout.to(device)
    throttle_rollout = RolloutStorage(**rollout_cfg); throttle_rollout.to(device)
    if shared_grad_buffers is None:              # single-process use: the agent's own arena is the shared one
        from .models import Shared_grad_buffers
        shared_grad_buffers = Shared_grad_buffers(agent.model_dict, device)
    obs = env.reset()
    done = False
    for episode in range(train_cfg.max_episode):
        for _ in range(num_steps):
            command = obs["command"]
            raw = dict(obs, rgb=obs["rgb"].copy(), route_fig=obs["route_fig"].copy()) if recorder is not None else None
            feat, action, alp, values, hidden = agent.act(obs)
            obs, reward, done, info = env.step(agent.convert_action(action))
            ad = info["action_done"]
            if recorder is not None:            # cadre_amd.replay.RolloutRecorder
                recorder.step(raw, action, alp, values, reward, ad)
            steer_rollout.insert(feat, action[0], alp[0], values[0], reward[0],
                                 torch.tensor([[0.0] if ad[0] else [1.0]]), hidden, command)
            throttle_rollout.insert(feat, action[1], alp[1], values[1], reward[1],
                                    torch.tensor([[0.0] if ad[1] else [1.0]]), hidden, command)
            if done:
                obs = env.reset()
        if recorder is not None:
            recorder.end_episode()
        vl, pl, el = learner_section(agent, steer_rollout, throttle_rollout, done, train_cfg, shared_grad_buffers,
                                     traffic_light=traffic_light, counter=counter,
                                     shared_model_list=shared_model_list,
                                     in_process_chief=traffic_light is None)   # no chief process -> step in-process
        if episode % train_cfg.log_interval == 0 and rank == 0 and logger is not None:
            logger.log("Episode: {}, value loss: {:.4f}, policy loss: {:.4f}, entropy loss: {:.4f}".format(
                episode, np.mean(vl), np.mean(pl), np.mean(el)))
        if episode % train_cfg.save_interval == 0 and rank == 0:
            agent.save_snapshot(os.path.join(model_dir, "ppo_model_{}.pt".format(episode)))
    if son_process_counter is not None:
        son_process_counter.increment()
    print("process {} finished.".format(rank))
