"""Build libcadre_hip.so (HIP kernels + C ABI) in-tree with hipcc for gfx950."""
import os
import subprocess

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
SOURCES = ["gemm_f32.hip", "gemm_f32_skinny.hip", "conv_stream_f32.hip", "gemm_bf16.hip", "conv_stream_bf16.hip", "stem_pool.hip", "conv3x3_c64_bf16.hip", "conv3x3_ring.hip", "peaks.hip", "cadre_kernels.hip"]
LIB = os.path.join(CSRC, "libcadre_hip.so")


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + [
        os.path.join(os.path.dirname(CSRC), "..", "include", "cadre_hip.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    if not force and not needs_build():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-o", LIB] + \
          [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force=True)
