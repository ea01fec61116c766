from cadre_amd.ppo_agent.models import *  # noqa: F401,F403
from cadre_amd.ppo_agent import models as _m
globals().update({k: v for k, v in vars(_m).items() if not k.startswith('__')})
