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
    # The reference's ppo_agent/ has no __init__.py (namespace package), so this repo's regular
    # `ppo_agent` package would shadow it whatever the sys.path order: pin the name to the reference
    # directory explicitly, and refuse to run if the mirror was imported first.
    for name in list(sys.modules):
        if name == "ppo_agent" or name.startswith("ppo_agent."):
            mod = sys.modules[name]
            if not getattr(mod, "__file__", "").startswith(REF) and getattr(mod, "__path__", [""])[0] != os.path.join(REF, "ppo_agent"):
                raise RuntimeError("this repo's ppo_agent mirror is already imported; run the generator in a fresh process")
    if "ppo_agent" not in sys.modules:
        ref_pkg = types.ModuleType("ppo_agent")
        ref_pkg.__path__ = [os.path.join(REF, "ppo_agent")]
        sys.modules["ppo_agent"] = ref_pkg


def assert_reference(module):
    """Fail loudly if `module` did not come from /root/reference."""
    f = getattr(module, "__file__", "") or ""
    if not f.startswith(REF):
        raise RuntimeError("expected the reference implementation, got %s" % f)
    # utils.logger is imported by ppo_agent.agent; it needs tabulate/dateutil (installed)


def encoder_ckpt_path(challenge_dir):
    return os.path.join(challenge_dir, "carla_perception", "Experiments34",
                        "danet912_nocrash_IL_n10_k1234_r40", "net_epoch90")
