"""Import shim for the upstream reference (/root/reference) — FIXTURE GENERATION ONLY.

Used only by tests/golden/make_golden.py in the build container; /root/reference does not
exist on the GPU box and nothing under tests/ run by pytest imports this module.
Recipe follows SURVEY.md §8(c): no bytecode writes into the reference tree, stub the
never-called `torchsnooper`, pre-register an empty `carla_perception` package so its
__init__ (cv2/skimage/tensorboardX) is skipped, provide CHALLENGE_DIR with a seeded
encoder checkpoint.
"""
import os
import sys
import types

sys.dont_write_bytecode = True
REF = "/root/reference"


class AD(dict):
    """attr-dict standing in for addict.ConfigDict (ppo_agent/meta/config.py:25-38)."""
    __getattr__ = dict.__getitem__
    __setattr__ = dict.__setitem__


def install(challenge_dir):
    os.environ["CHALLENGE_DIR"] = challenge_dir
    for p in (REF, os.path.join(REF, "carla_perception")):
        if p not in sys.path:
            sys.path.insert(0, p)
    sys.modules.setdefault("torchsnooper", types.ModuleType("torchsnooper"))
    if "carla_perception" not in sys.modules:
        pkg = types.ModuleType("carla_perception")
        pkg.__path__ = [os.path.join(REF, "carla_perception")]
        sys.modules["carla_perception"] = pkg
    # utils.logger is imported by ppo_agent.agent; it needs tabulate/dateutil (installed)


def encoder_ckpt_path(challenge_dir):
    return os.path.join(challenge_dir, "carla_perception", "Experiments34",
                        "danet912_nocrash_IL_n10_k1234_r40", "net_epoch90")
