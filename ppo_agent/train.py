from cadre_amd.ppo_agent.train import *  # noqa: F401,F403
from cadre_amd.ppo_agent import train as _m
globals().update({k: v for k, v in vars(_m).items() if not k.startswith('__')})
