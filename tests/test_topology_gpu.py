"""GPU: the two launch topologies that cannot be tested inside the pytest process.

  * reference main.py:57-70 — one `chief` process + `train` workers started with the `spawn` method (also with the
    chief inside an RCCL process group: the gradient exchange is due by SHARED state, whoever handed in), the
    shared nets / `Shared_grad_buffers` / optimizer handed over by pickling (HIP-IPC handles of views into the
    parameter arena).  Two episodes (two worker<->chief barrier rounds each); the worker's final snapshot must
    equal the one the in-process hand-off produces, bit for bit.
  * RCCL (`nccl` backend) initialised on the real device at world_size 1: `add_gradient` + `chief_step`
    all-reduce the gradient arena once per optimiser step and leave the same parameters as without
    torch.distributed.

Each runs in its own fresh process (tests/spawn_topology_driver.py, tests/rccl_world1_driver.py)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(module, args, tag, timeout=900):
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""), HSA_ENABLE_IPC_MODE_LEGACY="0")
    # (this pytest process may hold gigabytes of cached device memory from the tests before: hand it back before the driver and
    #  its children export and import blocks over HIP IPC)
    try:
        import torch
        if torch.cuda.is_initialized():
            torch.cuda.synchronize()
            torch.cuda.empty_cache()
    except Exception:
        pass
    for attempt in (0, 1):
        p = subprocess.run([sys.executable, "-m", module] + args, cwd=ROOT, env=env, capture_output=True, text=True,
                           timeout=timeout)
        # ONE retry, and only for the pool's HIP-IPC export flake (`storage._share_cuda_()` -> hipIpcGetMemHandle:
        # invalid argument in the driver's second mode — seen on some boxes when the test runs inside the full suite, never in
        # the test file alone); any other failure, and a second IPC failure, fails the test
        out = p.stdout + p.stderr
        if p.returncode == 0 or attempt == 1 or not ("hipIpc" in out or "_share_cuda_" in out):
            break
        print("HIP-IPC export flake, retrying once:\n" + p.stderr[-1500:])
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith(tag + " ")]
    assert p.returncode == 0 and lines, "driver failed rc=%s\nstdout:\n%s\nstderr:\n%s" % (
        p.returncode, p.stdout[-3000:], p.stderr[-6000:])
    return json.loads(lines[-1][len(tag) + 1:])


def test_spawned_chief_and_worker_match_in_process_handoff(tmp_path):
    res = _run("tests.spawn_topology_driver", [str(tmp_path)], "TOPOLOGY_RESULT")
    print(res)
    assert res["exitcodes"] == [0, 0]
    assert res["spawned_snapshot"] and res["in_process_snapshot"]
    assert res["tensors_compared"] >= 12 * 4 and res["max_abs_update"] > 0.0      # the optimiser really stepped
    assert res["max_abs_param_diff"] == 0.0
    assert res["spawned_shared_param_sum"] == res["in_process_shared_param_sum"]
    # the same topology with the chief inside a process group (RCCL, world size 1 forced): the hand-ins happen in the
    # WORKER process, the exchange must still run in the chief — once per optimiser step (2 episodes x 2 minibatches)
    assert res["exitcodes_dist"] == [0, 0]
    assert res["spawned_exchanges"] == 0 and res["spawned_dist_exchanges"] == 4
    assert res["max_abs_param_diff_dist"] == 0.0


def test_rccl_allreduce_on_device_world1():
    res = _run("tests.rccl_world1_driver", [], "RCCL_RESULT")
    print(res)
    assert res["backend"] == "nccl" and res["world"] == 1 and res["warmup_sum"] == 4.0
    assert res["allreduce_calls"] == 3 and res["allreduce_calls_nodist"] == 0     # one per optimiser step
    assert res["params_equal"] and res["losses_equal"] and res["broadcast_equal"]
