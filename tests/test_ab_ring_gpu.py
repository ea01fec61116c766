"""The window-conv variants that round 4 built, measured and left in the A/B library (csrc/conv3x3_ring.hip under
CADRE_AB_KERNELS): the G-k-tiles-per-slot ping-pong kernel (CADRE_RING_G=2) and the one-wave-per-SIMD
streamed-weights kernel (CADRE_RING_1W=1).  Both stay parity-green against torch on every shape class of
test_conv3x3_ring — each in its own process: the switches are read once, when the library loads.
Reference layers: carla_perception/Networks/danet_blocks/resnet.py:26-55, danet.py:21-41."""
import os
import subprocess
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
AB = os.path.join(ROOT, "cadre_amd", "csrc", "libcadre_hip_ab.so")
pytestmark = pytest.mark.gpu


def _ab_current():
    from cadre_amd import build as b
    return os.path.exists(AB) and not b.needs_build(ab=True)


@pytest.mark.skipif(not _ab_current(), reason="A/B library missing or older than its sources (CADRE_BUILD_AB=1 python -m cadre_amd.build)")
@pytest.mark.parametrize("switch", ["CADRE_RING_G=2", "CADRE_RING_1W=1"])
def test_ab_window_conv_variants_match_torch(switch):
    k, v = switch.split("=")
    env = dict(os.environ, CADRE_HIP_LIB=AB, CADRE_RING_C64S="0", **{k: v})
    p = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_kernels_gpu.py"), "-q", "-x", "-m", "gpu",
                        "-k", "test_conv3x3_ring and bf16"], env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-2000:]
    assert " passed" in p.stdout


@pytest.mark.skipif(not _ab_current(), reason="A/B library missing or older than its sources (CADRE_BUILD_AB=1 python -m cadre_amd.build)")
@pytest.mark.parametrize("ks", ["4", "8"])
def test_ab_lstm_dw_with_operands_shared_through_lds(ks):
    """lstm_dw_lds_kernel<KS> (measured slower, A/B build, CADRE_DW_LDS=4|8) stays parity-green on the
    weight-gradient test (float64 reference, NaN in foreign rows, bit-identical on repeat)."""
    env = dict(os.environ, CADRE_HIP_LIB=AB, CADRE_DW_LDS=ks)
    p = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_kernels_gpu.py"), "-q", "-x", "-m", "gpu",
                        "-k", "test_lstm_dw_fused"], env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-2000:]
    assert " passed" in p.stdout


@pytest.mark.skipif(not _ab_current(), reason="A/B library missing or older than its sources (CADRE_BUILD_AB=1 python -m cadre_amd.build)")
def test_ab_w128_wave_tile_window_conv_matches_torch():
    """conv3x3_w128_kernel (round 5, VERDICT r4 item 2: 128 x 128 wave tile, one wave per SIMD, accumulators in AGPRs, weights streamed
    to registers — a tie with the ping-pong kernel, A/B build) stays parity-green against torch-CPU on both tile shapes and is
    bit-repeatable under load."""
    env = dict(os.environ, CADRE_HIP_LIB=AB)
    p = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_kernels_gpu.py"), "-q", "-x", "-m", "gpu",
                        "-k", "conv3x3_w128"], env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-2000:]
    assert " passed" in p.stdout and "skipped" not in p.stdout.splitlines()[-1]
