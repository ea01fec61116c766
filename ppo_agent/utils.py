from cadre_amd.ppo_agent.utils import *  # noqa: F401,F403
from cadre_amd.ppo_agent import utils as _m
globals().update({k: v for k, v in vars(_m).items() if not k.startswith('__')})
