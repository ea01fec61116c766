"""CPU: host-side logic of the drop-in mirror — storage cursor / sampler stream, arena layout,
C-ABI symbol table, loud failure without a HIP device.  No kernel is launched here."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cabi_exports_every_declared_symbol():
    from cadre_amd import hip
    hdr = open(os.path.join(ROOT, "include", "cadre_hip.h")).read()
    declared = set(re.findall(r"\b(cadre_[a-z0-9_]+)\s*\(", hdr))
    declared.discard("cadre_gemm_t")
    assert declared == set(hip.SYMBOLS), declared ^ set(hip.SYMBOLS)
    L = ctypes.CDLL(hip.LIB_PATH)
    for name in declared:
        assert hasattr(L, name), name
    assert hip.lib().cadre_abi_version() == hip.ABI_VERSION == 15
    # the default library exports only entry points the product dispatches: the superseded kernels live in the A/B build
    if os.path.basename(hip.LIB_PATH) == "libcadre_hip.so":
        assert not hip.has_ab_kernels()
        for name in hip.AB_SYMBOLS:              # (cadre_lstm_seq_fwd moved there in round 4: measured slower, opt-in only)
            assert not hasattr(L, name), name
    assert ctypes.sizeof(hip.GemmDesc) == 264      # static_assert-ed in gemm_f32.hip


def test_no_cpu_fallback():
    from cadre_amd import hip
    from ppo_agent.models import create_model
    from ppo_agent.storage import RolloutStorage
    cfg = dict(use_lstm=True, vae_device=-1, device_num=-1, vae_params="CoPM", measurement_dim=18,
               num_output=dict(steer=33, throttle=3), command_num=4)
    with pytest.raises(hip.CadreHipError):
        create_model(cfg, load_vae=False)
    st = RolloutStorage(8, 2, 530, 8, 530, True, 0.99, 0.95)
    with pytest.raises(hip.CadreHipError):
        st.compute_returns(torch.zeros(1))
    with pytest.raises(hip.CadreHipError):
        hip.ptr(torch.zeros(4))


def test_product_never_imports_oracle():
    for base in ("cadre_amd", "ppo_agent"):
        for dp, _, fs in os.walk(os.path.join(ROOT, base)):
            for f in fs:
                if f.endswith(".py"):                        # (no exception: the smoke checker lives in tests/ since round 6)
                    src = open(os.path.join(dp, f)).read()
                    assert "oracle" not in src, os.path.join(dp, f)


def test_storage_insert_cursor_drift(golden):
    """F-insert: modulo-(T+1) cursor, hn/cn written at step+1 (storage.py:45-58)."""
    from ppo_agent.storage import RolloutStorage
    g = golden("insert")
    T = int(g["T"])
    s = RolloutStorage(T, 1, 6, 2, 6, True, 0.99, 0.95)
    r = np.random.RandomState(3)
    for i in range(T + 3):
        obs = r.standard_normal((2, 6)).astype(np.float32)
        hn = r.standard_normal((1, 6)).astype(np.float32)
        cn = r.standard_normal((1, 6)).astype(np.float32)
        s.insert(torch.from_numpy(obs), torch.tensor(i % 3), torch.tensor([[-0.1 * i]]), torch.tensor([[0.5 * i]]),
                 torch.tensor(0.25 * i), torch.tensor([[float(i % 2)]]),
                 (torch.from_numpy(hn), torch.from_numpy(cn)), i % 4)
    assert s.step == int(g["step"])
    for mine, key in ((s.obs, "obs"), (s.action, "action"), (s.action_log_probs, "alp"), (s.value_preds, "values"),
                      (s.rewards, "rewards"), (s.masks, "masks"), (s.command, "command"), (s.hn, "hn"), (s.cn, "cn")):
        assert np.array_equal(mine.numpy(), g[key]), key
    assert s._obs[:, :, 6:].abs().max() == 0


@pytest.mark.parametrize("T,mbn", [(32, 2), (128, 2), (200, 2), (50, 3)])
def test_storage_sampler_stream_matches_reference(golden, T, mbn):
    from ppo_agent.storage import RolloutStorage
    g = golden("sampler")
    want = np.split(g["T%d_m%d" % (T, mbn)], np.cumsum(g["T%d_m%d_lens" % (T, mbn)])[:-1])
    a = RolloutStorage(T, mbn, 4, 8, 4, True, 0.99, 0.95)
    b = RolloutStorage(T, mbn, 4, 8, 4, True, 0.99, 0.95)
    torch.manual_seed(1000 + T)
    got = []
    for _ in range(4):
        ia, ib = a.sample_indices(), b.sample_indices()      # steer draws first (train.py:94-96)
        for x, y in zip(ia, ib):
            got += [x.numpy(), y.numpy()]
    assert len(got) == len(want) and all(np.array_equal(x, y) for x, y in zip(got, want))


def test_arena_layout():
    from cadre_amd.arena import PPOArena
    a = PPOArena.__new__(PPOArena)
    # layout arithmetic only (no device allocation)
    import types
    a2 = types.SimpleNamespace()
    D, DP, H4 = 530, 544, 2120
    size_L = 2 * H4 * DP + 2 * H4
    size_T = 128 * DP + 128 + 128 * 128 + 128 + 64 * 128 + 64
    assert size_L == 2310800 and size_T == 94528
    assert 8 * size_L + 16 * size_T == 19998848
    # real parameter count of the reference nets
    real = 8 * (2 * H4 * D + 2 * H4) + 4 * ((128 * D + 128 + 128 * 128 + 128) * 2 + 33 * 128 + 33 + 128 + 1) \
        + 4 * ((128 * D + 128 + 128 * 128 + 128) * 2 + 3 * 128 + 3 + 128 + 1)
    assert real == 19382808


def test_snapshot_key_quirk(golden):
    g = golden("insert")
    keys = sorted(str(k) for k in g["snapshot_keys"])
    assert keys == sorted(["%s_%d" % (k, c) for c in range(4) for k in ("throttle_ppo", "steer_ppo", "steer_lstm")])


def test_rollout_record_replay_roundtrip(tmp_path):
    """§8f-2: recorder de-duplicates the sliding window, replay reproduces every observation."""
    from cadre_amd import replay, synth
    steps = synth.synth_rollout(9, 20, 28, seed=4) + synth.synth_rollout(4, 20, 28, seed=5)   # one env reset inside
    rec = replay.RolloutRecorder(str(tmp_path), worker=3)
    for i, td in enumerate(steps):
        obs = dict(rgb=td["rgb"], route_fig=td["route_fig"], measurements=td["measurements"], command=td["command"])
        rec.step(obs, (i % 33, i % 3), (-0.5 * i, -0.1), (0.25 * i, 1.0), td["reward"], td["done"])
    path = rec.end_episode()
    ep = replay.load_episode(path)
    assert ep["rgb"].shape[0] == (8 + 8) + (8 + 3)                      # 2 fresh windows + slides, not 13*8
    assert replay.list_episodes(str(tmp_path), worker=3) == [path]
    for i, td in enumerate(steps):
        o = replay.windows(ep, i)
        assert np.array_equal(o["rgb"], td["rgb"]) and np.array_equal(o["route_fig"], td["route_fig"])
        assert np.array_equal(o["measurements"], td["measurements"]) and o["command"] == td["command"]
    assert ep["action"][5].tolist() == [5, 2] and ep["done"].dtype == np.uint8


def test_cabi_rejects_bad_arguments_without_launching():
    """Error behaviour of the C ABI: negative status + readable message, checked before any HIP call
    (so this runs without a GPU)."""
    import ctypes as C
    from cadre_amd import hip
    L = hip.lib()
    d = hip.GemmDesc()
    assert L.cadre_gemm_f32(C.byref(d), None) == -1 and b"null operand" in L.cadre_last_error()
    d.A, d.B, d.C, d.M, d.N, d.K = 16, 16, 16, 4, 4, 6
    assert L.cadre_gemm_f32(C.byref(d), None) == -1 and b"K%4" in L.cadre_last_error()
    d.K, d.lda, d.ldb, d.ldc, d.a_mode, d.Cin, d.KH, d.KW = 64, 64, 64, 4, 2, 48, 1, 1
    assert L.cadre_gemm_f32(C.byref(d), None) == -1 and b"Cin%32" in L.cadre_last_error()
    assert L.cadre_gemm_bf16(C.byref(d), None) == -1
    assert L.cadre_gae(None, None, None, None, None, None, 1, 8, 0.99, 0.94, 1, None) == -1
    assert b"cadre_gae" in L.cadre_last_error()
    assert L.cadre_pam(16, 16, 0.5, 16, 1, 2000, None) == -1 and b"Np<=1024" in L.cadre_last_error()
    assert L.cadre_sample(16, 64, 16, 64, 1, 65, 16, 16, None) == -1
    assert L.cadre_clip_adam(16, 16, 16, 16, 16, 0, 16, 250.0, 3e-4, 0.9, 0.999, 1e-8, 1, None) == -1
    with pytest.raises(hip.CadreHipError, match="cadre_gemm_f32"):
        hip.check(-1, "cadre_gemm_f32")


def test_missing_library_fails_loudly():
    """No silent fallback: with the shared library absent every entry into the product path raises."""
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from cadre_amd import hip\n"
            "try:\n"
            "    hip.lib()\n"
            "except hip.CadreHipError as e:\n"
            "    print('RAISED', 'no CPU fallback' in str(e))\n" % ROOT)
    out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, CADRE_HIP_LIB="/nonexistent/libcadre_hip.so"),
                         capture_output=True, text=True, timeout=120)
    assert "RAISED True" in out.stdout, out.stdout + out.stderr


def test_gemm_tile_choice_is_host_logic():
    """cadre_gemm_pick_tile runs on the host (no launch): the shapes of the hot path get the tiles DESIGN.md
    documents, and an explicit tile id is passed through."""
    import ctypes as C
    from cadre_amd import hip
    L = hip.lib()

    def pick(M, N, K, a_mode=0, batch=1, split_k=1, seg=0, period=0, tile=0):
        d = hip.GemmDesc()
        d.M, d.N, d.K, d.a_mode, d.batch, d.split_k, d.tile = M, N, K, a_mode, batch, split_k, tile
        d.seg_mode, d.seg_period = seg, period
        d.ldc = N
        return L.cadre_gemm_pick_tile(C.byref(d))
    F = 1024
    # Cin=4 stem as a plain conv launch (geometries the fused front does not cover): 64x64; streamed 64x64 in the A/B build
    assert pick(F * 144 * 144, 64, 224, a_mode=3) == (12 if hip.has_ab_kernels() else 3)
    assert pick(F * 72 * 72, 64, 576, a_mode=2) == 3               # stage 1 (N = 64)
    assert pick(F * 36 * 36, 128, 1152, a_mode=2) == 8             # stage 2 (N = 128): 128x128 on 8 waves
    assert pick(F * 18 * 18, 256, 2304, a_mode=2) == 3
    assert pick(F * 81, 512, 4608, a_mode=2) == 3
    assert pick(F * 81, 128, 4608, a_mode=2) == 3                  # head convs: too few tiles for the big one
    assert pick(4096, 4096, 4096) == 8                             # big dense
    assert pick(64, 2120, 544, batch=8) == 3                       # unsorted recurrent step
    assert pick(64, 2120, 544, batch=8, seg=1, period=64) == 9     # row-sorted: 32-row tiles
    assert pick(512, 2120, 544, batch=8, seg=1, period=64) == 9
    assert pick(300, 200, 544, tile=2) == 2


def test_window_conv_policy_is_host_logic():
    """cadre_conv3x3_ring_supported / _ntile run on the host: every stride-1 3x3 conv of the bf16 encoder goes to the
    8-wave ping-pong window kernel, fp32 only the 64-channel stage, and geometry the window cannot
    hold is refused."""
    from cadre_amd import hip
    L = hip.lib()
    F = 1024
    trunk = [(72, 64, 64), (36, 128, 128), (18, 256, 256), (9, 512, 512), (9, 512, 128), (9, 128, 128)]
    for hw, cin, n in trunk:
        assert L.cadre_conv3x3_ring_supported(F, hw, hw, cin, n, 1 | 2 | 8 | 4) == 1, (hw, cin, n)
        # ntile + 1000 * waves-along-positions + 100000 * ping-pong + 1000000 * G: G = 9 names the weight-stationary
        # 64 -> 64 stage (conv3x3_c64s_kernel, one 128-byte chunk of input channels); the 128-channel tiles run the
        # one-k-tile-per-slot ping-pong kernel (the G-k-tiles-per-slot form lives in the A/B build: measured slower)
        code = L.cadre_conv3x3_ring_ntile(F, hw, hw, cin, n, 1)
        assert code == (64 + 9000000 if (n == 64 and cin == 64) else 128) + 4000 + 100000, (hw, cin, n, code)
    assert L.cadre_conv3x3_ring_ntile(F, 36, 36, 64, 128, 1) == 128 + 4000 + 100000
    assert L.cadre_conv3x3_ring_ntile(F, 72, 72, 128, 64, 1) == 64 + 4000 + 100000      # two chunks: the ping-pong kernel
    assert L.cadre_conv3x3_ring_ntile(F, 80, 80, 64, 64, 1) == 64 + 4000 + 100000       # W = 80: three windows do not fit LDS
    assert L.cadre_conv3x3_ring_supported(F, 72, 72, 64, 64, 0) == 1
    assert L.cadre_conv3x3_ring_ntile(F, 72, 72, 64, 64, 0) == 64 + 4000 + 100000      # fp32 64-channel stage: ping-pong, one k-tile per slot
    assert L.cadre_conv3x3_ring_ntile(F, 36, 36, 128, 128, 0) == 128 + 4000             # (a forced fp32 128-channel tile: lockstep)
    for hw, cin, n in trunk[1:4]:
        assert L.cadre_conv3x3_ring_supported(F, hw, hw, cin, n, 0) == 0  # fp32 N >= 128: the tile kernels
    assert L.cadre_conv3x3_ring_supported(1, 144, 144, 64, 64, 1) == 0    # W > 95: two windows do not fit LDS
    assert L.cadre_conv3x3_ring_supported(F, 36, 36, 48, 64, 1) == 0      # channel chunk not 128 bytes
    # every tensor at its own element size against the 32-bit offset bound: the joint 2048-frame chunk's layer-1 maps
    # (10,616,832 x 64) fit as bf16 (1.27 GiB) and not as fp32 (2.53 GiB) — output and residual separately
    assert L.cadre_conv3x3_ring_supported(2048, 72, 72, 64, 64, 1 | 2) == 1
    assert L.cadre_conv3x3_ring_supported(2048, 72, 72, 64, 64, 1) == 0
    assert L.cadre_conv3x3_ring_supported(2048, 72, 72, 64, 64, 1 | 2 | 8 | 4) == 1
    assert L.cadre_conv3x3_ring_supported(2048, 72, 72, 64, 64, 1 | 2 | 8) == 0
    assert L.cadre_conv3x3_ring_supported(2048, 9, 9, 512, 512, 1) == 1    # conv5a / conv5c: fp32 out of bf16 operands


def test_drop_in_package_keeps_reference_meta_importable(tmp_path):
    """The reference's ppo_agent/ is a namespace package that also holds ppo_agent/meta/ (main.py:4, eval.py:4,
    simple_test.py:2: `from ppo_agent.meta.config import Config`).  With both roots on sys.path — in either
    order — the hot-path modules must resolve to this repo and `ppo_agent.meta.*` to the Cadre checkout."""
    import subprocess
    import sys
    fake = tmp_path / "cadre_checkout"
    (fake / "ppo_agent" / "meta").mkdir(parents=True)
    (fake / "ppo_agent" / "meta" / "config.py").write_text("class Config(object):\n    where = 'reference'\n")
    (fake / "ppo_agent" / "agent.py").write_text("raise ImportError('the reference module must be shadowed')\n")
    code = ("import os, ppo_agent\n"
            "from ppo_agent.meta.config import Config\n"                       # main.py:4
            "from ppo_agent.models import create_model, Shared_grad_buffers, get_vae_output\n"   # main.py:6,11 train.py:4
            "from ppo_agent.chief import chief\n"                               # main.py:8
            "from ppo_agent.utils import TrafficLight, Counter, check_exist\n"  # main.py:9 train.py:8
            "from ppo_agent.train import train\n"                               # main.py:13
            "from ppo_agent.agent import CadreAgent\n"                          # eval.py:5 train.py:1
            "from ppo_agent.storage import RolloutStorage\n"                    # train.py:2
            "from ppo_agent.distributions import Categorical_1d\n"
            "import ppo_agent.agent as a, ppo_agent.meta.config as c\n"
            "assert Config.where == 'reference'\n"
            "assert os.path.realpath(a.__file__).startswith(os.path.realpath(%r)), a.__file__\n"
            "assert os.path.realpath(c.__file__).startswith(os.path.realpath(%r)), c.__file__\n"
            "print('ok')\n" % (ROOT, str(fake)))
    for order in ((ROOT, str(fake)), (str(fake), ROOT)):
        env = dict(os.environ, PYTHONPATH=os.pathsep.join(order))
        p = subprocess.run([sys.executable, "-B", "-c", code], cwd=str(tmp_path), env=env, capture_output=True, text=True)
        assert p.returncode == 0 and p.stdout.strip() == "ok", (order, p.stdout, p.stderr[-2000:])


def test_arena_pickle_drops_process_local_state():
    """reference main.py:57-70 pickles the shared nets into spawned processes: the arena must not drag the
    learner (hipGraphs) along and the bound parameters / gradient views must come back attachable."""
    import pickle

    import torch
    from cadre_amd.arena import PPOArena
    from ppo_agent.models import LSTM, _no_orthogonal_init
    arena = PPOArena("cpu", 530, {"steer": 33, "throttle": 3}, 4)
    with _no_orthogonal_init():
        m = arena.bind("steer_lstm_1", LSTM(530, hid_size=530))
    arena._learner = object()
    assert "_learner" not in arena.__getstate__()
    m2 = pickle.loads(pickle.dumps(m))
    a2 = m2._cadre_arena
    assert not hasattr(a2, "_learner") and a2 is not arena
    assert tuple(m2.rnn.weight_hh.shape) == (2120, 530) and m2._cadre_name == "steer_lstm_1"
    assert torch.equal(m2.rnn.weight_hh.detach(), m.rnn.weight_hh.detach())
    # (storage aliasing between the unpickled views and the unpickled arena is a property of
    #  torch.multiprocessing's HIP-IPC reductions: tests/test_topology_gpu.py checks it on the device)
    a2.attach_grads()
    assert m2.rnn.weight_hh.grad is not None and tuple(m2.rnn.weight_hh.grad.shape) == (2120, 530)


def test_bench_kernel_names_are_keys_of_the_hbm_traffic_file():
    """profiles/ reproduces the line.  `bench.py --section headline|c3` is the command tools/prof_bench.sh
    profiles, ONE section per command; profiles/hbm_traffic.json holds, per section, the rocprofv3 FETCH_SIZE / WRITE_SIZE averages
    and dispatch counts of every kernel of that command.  For both sections: every kernel name the profiled line prints
    (roofline.kernel and per_kernel) is a key of the section's table — no `traffic: null` through a naming drift between
    bench.kname() and the profiler — and the launch populations MATCH: launches per round from the HIP events of the timed region
    == dispatches / rounds of the profiled command.  The committed full line (profiles/r06_bench_c2_c3.json) carries a traffic
    figure for the dominant kernel of the headline and of the c3 section, at least the algorithmic bytes."""
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    traffic = json.load(open(os.path.join(root, "profiles", "hbm_traffic.json")))
    for sec in ("headline", "c3"):
        line = open(os.path.join(root, "profiles", "r06_bench_under_rocprof_%s.json" % sec)).read().strip().splitlines()[-1]
        b = json.loads(line)
        assert b.get("section") == sec and "c3" not in b and "cpu_baseline" not in b and "c2_latent_cache" not in b
        t = traffic[sec]
        assert t["rounds"] == b["steps"] + b["warmup"] + 3
        r = b["roofline"]
        names = {r["kernel"]} | set(r["per_kernel"])
        assert len(names) >= 5
        missing = sorted(n for n in names if n not in t["kernels"])
        assert not missing, (sec, missing)
        for n in names:
            e = t["kernels"][n]
            assert e["hbm_bytes_per_launch"] > 0 and e["launches"] > 0
            if n in r["per_kernel"]:
                assert e["launches"] == r["per_kernel"][n]["launches_per_step"] * t["rounds"], (sec, n, e["launches"])
    full = json.loads(open(os.path.join(root, "profiles", "r06_bench_c2_c3.json")).read().strip().splitlines()[-1])
    for sect in (full, full["c3"]):
        r = sect["roofline"]
        assert r["traffic"] is not None and r["traffic_launches_per_round"] == r["launches_per_round"]
        assert r["traffic"] >= 0.95 * r["bytes_per_launch"], (r["kernel"], r["traffic"], r["bytes_per_launch"])
