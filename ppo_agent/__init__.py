"""Drop-in import path for the reference ppo_agent package (filled by cadre_amd.ppo_agent)."""
