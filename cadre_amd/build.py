"""Build libcadre_hip.so (HIP kernels + C ABI) in-tree with hipcc for gfx950.

Default build: the kernels the product dispatches.  `CADRE_BUILD_AB=1 python -m cadre_amd.build` additionally compiles
csrc/ab/ (superseded kernels kept as measured comparison points, include/cadre_hip_ab.h) with -DCADRE_AB_KERNELS into
libcadre_hip_ab.so; point CADRE_HIP_LIB at it to run the A/B tools and tests.

Sources compile to objects in parallel (one hipcc per file, csrc/build/), only the stale ones, then link."""
import os
import subprocess
from concurrent.futures import ThreadPoolExecutor

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
INCLUDE = os.path.join(os.path.dirname(CSRC), "..", "include")
SOURCES = ["gemm_f32.hip", "gemm_bf16.hip", "gemm_bf16_w128.hip", "stem_pool.hip", "conv3x3_ring.hip", "conv3x3_s2.hip", "conv3x3_s1x.hip", "ppo_update.hip", "peaks.hip", "winograd.hip", "winograd_c64.hip", "winograd_fused.hip",
           "cadre_kernels.hip"]
AB_SOURCES = ["ab/gemm_f32_skinny.hip", "ab/gemm_stream_f32.hip", "ab/conv_stream_f32.hip", "ab/conv_stream_bf16.hip", "ab/conv3x3_c64_bf16.hip", "ab/conv3x3_w128.hip"]
# per-file flags: the fused Winograd kernel is written in issue order; the machine scheduler's reordering costs it 40 spills.  Its
# step loop (8 steps x 128 inline-asm MFMAs + the epilogue's pinned accumulator reads) is past the default size limit of
# "#pragma unroll": not unrolled, the register sets indexed by step parity would live in scratch
EXTRA_FLAGS = {"winograd_c64.hip": ["-mllvm", "-enable-misched=0", "-mllvm", "-pragma-unroll-threshold=262144"],
               # (written in issue order too: the scheduler sinks the fragment reads of the next slot down to their MFMAs)
               "winograd_fused.hip": ["-mllvm", "-enable-misched=0", "-mllvm", "-pragma-unroll-threshold=262144"],
               "gemm_bf16_w128.hip": ["-mllvm", "-enable-misched=0", "-mllvm", "-pragma-unroll-threshold=262144"],
               "ab/conv3x3_w128.hip": ["-mllvm", "-enable-misched=0", "-mllvm", "-pragma-unroll-threshold=262144"]}
LIB = os.path.join(CSRC, "libcadre_hip.so")
LIB_AB = os.path.join(CSRC, "libcadre_hip_ab.so")


def _ab():
    return os.environ.get("CADRE_BUILD_AB", "0") not in ("", "0")


def _plan(ab):
    srcs = SOURCES + (AB_SOURCES if ab else [])
    odir = os.path.join(CSRC, "build", "ab" if ab else "default")
    hdrs = [os.path.join(INCLUDE, "cadre_hip.h"), os.path.join(CSRC, "winograd_mats.h")] + ([os.path.join(INCLUDE, "cadre_hip_ab.h")] if ab else [])
    return srcs, odir, hdrs, (LIB_AB if ab else LIB)


def needs_build(ab=None):
    ab = _ab() if ab is None else ab
    srcs, _odir, hdrs, lib = _plan(ab)
    if not os.path.exists(lib):
        return True
    t = os.path.getmtime(lib)
    return any(os.path.getmtime(d) > t for d in [os.path.join(CSRC, s) for s in srcs] + hdrs)


def build(force=False, verbose=True, ab=None):
    ab = _ab() if ab is None else ab
    srcs, odir, hdrs, lib = _plan(ab)
    if not force and not needs_build(ab):
        return lib
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(odir, exist_ok=True)
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC"] + (["-DCADRE_AB_KERNELS"] if ab else [])
    hdr_t = max(os.path.getmtime(h) for h in hdrs)
    jobs, objs = [], []
    for s in srcs:
        src = os.path.join(CSRC, s)
        obj = os.path.join(odir, os.path.basename(s)[:-4] + ".o")
        objs.append(obj)
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), hdr_t):
            jobs.append([hipcc] + flags + EXTRA_FLAGS.get(s, []) + ["-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    workers = max(1, min(len(jobs), int(os.environ.get("CADRE_BUILD_JOBS", str(min(8, os.cpu_count() or 1))))))
    if jobs:
        with ThreadPoolExecutor(workers) as ex:
            list(ex.map(run, jobs))
    run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs)
    return lib


if __name__ == "__main__":
    build(force=True)
