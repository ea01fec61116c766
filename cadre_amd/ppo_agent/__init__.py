"""MI355X-native mirror of the reference `ppo_agent` package (agent / storage / chief / models /
distributions / utils / train).  Importable as `ppo_agent.*` through the top-level shim
package so the CARLA rollout loop (reference main.py, eval.py) runs unchanged."""
