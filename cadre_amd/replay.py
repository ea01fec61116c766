"""Rollout record / replay on disk (BASELINE config C5).

The reference keeps rollouts only in memory (nothing under result/ but models and csv), so there
is no format to be compatible with; the nearest precedent is carla_project's per-episode layout
(carla_project/src/dataset.py:132-147).  This is a compact learner-side format: one `.npz` per
(worker, episode) holding the DISTINCT camera frames once plus, per transition, the indices of
its 8-frame window — the env's sliding window (env_wrapper.py:899-904) makes 7 of 8 frames repeat,
so a 128-step episode at 288x288 is 135 frames (34 MB) instead of 1024 (254 MB).

    rec = RolloutRecorder(dir, worker=0)
    rec.step(obs, action, action_log_probs, values, reward, action_done)   # after env.step, train.py:57-72
    rec.end_episode()
    ep = load_episode(path)            # dict of numpy arrays, see FIELDS
"""
import json
import os

import numpy as np

VERSION = 1
FIELDS = ("rgb", "route", "measurements", "window", "command", "reward", "done", "action", "action_log_prob", "value")


class RolloutRecorder(object):
    def __init__(self, directory, worker=0):
        self.dir = directory
        self.worker = worker
        self.episode = 0
        os.makedirs(directory, exist_ok=True)
        self._reset()

    def _reset(self):
        self.frames_rgb, self.frames_route, self.frames_meas = [], [], []
        self.window, self.rows = [], []
        self._prev = None

    def step(self, obs, action, action_log_probs, values, reward, action_done):
        """obs: the tick_data dict given to agent.act (BEFORE act mutated route_fig, or after — both
        forms are recognised as the same frame on replay since normalisation is idempotent)."""
        rgb, route, meas = obs["rgb"], obs["route_fig"], obs["measurements"]
        S = rgb.shape[0]
        if self._prev is not None and np.array_equal(rgb[:-1], self._prev[0][1:]) and \
                np.array_equal(route[:-1], self._prev[1][1:]):
            first = S - 1
            win = self.window[-1][1:] + [len(self.frames_rgb)]
        else:
            first = 0
            win = list(range(len(self.frames_rgb), len(self.frames_rgb) + S))
        for s in range(first, S):
            self.frames_rgb.append(np.array(rgb[s], np.uint8))
            self.frames_route.append(np.array(route[s], np.uint8))
            self.frames_meas.append(np.array(meas[s], np.float64))
        self.window.append(win)
        self._prev = (np.array(rgb), np.array(route))
        f = lambda x: float(x.item() if hasattr(x, "item") else x)
        self.rows.append((int(obs["command"]), [f(reward[0]), f(reward[1])],
                          [bool(action_done[0]), bool(action_done[1])],
                          [int(action[0]), int(action[1])],
                          [f(action_log_probs[0]), f(action_log_probs[1])], [f(values[0]), f(values[1])]))

    def end_episode(self):
        if not self.rows:
            return None
        path = os.path.join(self.dir, "w%02d_ep%06d.npz" % (self.worker, self.episode))
        cmd, rew, done, act, alp, val = zip(*self.rows)
        np.savez_compressed(
            path, version=VERSION, rgb=np.stack(self.frames_rgb), route=np.stack(self.frames_route),
            measurements=np.stack(self.frames_meas), window=np.array(self.window, np.int32),
            command=np.array(cmd, np.int32), reward=np.array(rew, np.float32), done=np.array(done, np.uint8),
            action=np.array(act, np.int64), action_log_prob=np.array(alp, np.float32), value=np.array(val, np.float32))
        with open(os.path.join(self.dir, "meta.json"), "w") as fh:
            json.dump({"version": VERSION, "fields": FIELDS, "H": int(self.frames_rgb[0].shape[0]),
                       "W": int(self.frames_rgb[0].shape[1]), "seq_length": len(self.window[0])}, fh)
        self.episode += 1
        self._reset()
        return path


def load_episode(path):
    z = np.load(path, allow_pickle=False)
    if int(z["version"]) != VERSION:
        raise ValueError("unsupported rollout record version %s" % z["version"])
    return {k: z[k] for k in FIELDS}


def list_episodes(directory, worker=None):
    names = sorted(n for n in os.listdir(directory) if n.endswith(".npz"))
    if worker is not None:
        names = [n for n in names if n.startswith("w%02d_" % worker)]
    return [os.path.join(directory, n) for n in names]


def windows(ep, t):
    """The observation dict of transition t, as the env would have produced it."""
    w = ep["window"][t]
    return dict(rgb=ep["rgb"][w], route_fig=ep["route"][w].copy(), measurements=ep["measurements"][w],
                command=int(ep["command"][t]))
