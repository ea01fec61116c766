"""Drop-in import path for the reference ppo_agent package (filled by cadre_amd.ppo_agent).

The reference's `ppo_agent/` is a namespace package (no __init__.py) that also holds
`ppo_agent/meta/{config,module_utils,path_utils}.py`, which main.py:4, eval.py:4 and
simple_test.py:2 import (`from ppo_agent.meta.config import Config`).  A regular package would
shadow that directory completely, so this one appends every other `ppo_agent` directory found
on sys.path to its `__path__`: `ppo_agent.agent / storage / chief / models / distributions /
utils / train` resolve to the MI355X modules here (searched first), `ppo_agent.meta.*` — which is
config plumbing, outside the hot path — keeps resolving to the Cadre checkout.
"""
import pkgutil

__path__ = pkgutil.extend_path(__path__, __name__)
