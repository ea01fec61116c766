"""CPU oracle for the Cadre PPO-learner hot path — TEST INFRASTRUCTURE ONLY.

This package is a CPU restatement (numpy for the byte/integer/strict-order parts, plain
torch-CPU fp32 for the floating-point parts) of the reference algorithms on the hot path
(SURVEY.md §8a).  It is the checker, never the product:

  * only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import it;
  * nothing under `cadre_amd/` or `ppo_agent/` imports it — the product path is the HIP
    library behind `include/cadre_hip.h` and it fails loudly when that library is missing.

Parity pinning: the reference ships no tests or golden vectors, so the oracle
is pinned against outputs of the reference itself, imported unmodified in the build
container by `tests/golden/make_golden.py` (committed), with the resulting vectors committed
under `tests/golden/*.npz`.  `tests/test_oracle_golden.py` re-checks the oracle against those
vectors on every CPU run.
"""
