#!/usr/bin/env python3
"""Write synthetic rollouts in the on-disk record format (cadre_amd/replay.py) so that
`bench.py --replay DIR` can be exercised without CARLA (BASELINE config C5 uses real recordings)."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cadre_amd import replay, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--out", required=True)
ap.add_argument("--episodes", type=int, default=1)
ap.add_argument("--steps", type=int, default=128)
ap.add_argument("--size", type=int, nargs=2, default=(288, 288))
a = ap.parse_args()
for e in range(a.episodes):
    rec = replay.RolloutRecorder(a.out, worker=e)
    for i, td in enumerate(synth.synth_rollout(a.steps, a.size[0], a.size[1], seed=500 + e)):
        rec.step(dict(rgb=td["rgb"], route_fig=td["route_fig"], measurements=td["measurements"], command=td["command"]),
                 (i % 33, i % 3), (-3.4, -1.1), (0.0, 0.0), td["reward"], td["done"])
    print(rec.end_episode())
