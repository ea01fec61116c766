"""ctypes binding of libcadre_hip.so (include/cadre_hip.h) + thin tensor-pointer helpers.

PyTorch-ROCm is plumbing here: it owns device memory and streams; every call below hands
raw device pointers and the current HIP stream to the C ABI.  There is NO CPU fallback:
a missing library raises at first use (`lib()`), and every kernel wrapper requires device
tensors.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CADRE_HIP_LIB") or os.path.join(_HERE, "csrc", "libcadre_hip.so")
_lib = None

i32, i64, f32, f64, vp = C.c_int32, C.c_int64, C.c_float, C.c_double, C.c_void_p


class GemmDesc(C.Structure):
    _fields_ = [("A", vp), ("B", vp), ("C", vp), ("scale", vp), ("shift", vp), ("resid", vp),
                ("lda", i64), ("ldb", i64), ("ldc", i64), ("ldr", i64),
                ("M", i32), ("N", i32), ("K", i32), ("a_mode", i32), ("b_mode", i32),
                ("act", i32), ("slope", f32), ("batch", i32),
                ("a_div", i32), ("a_mod", i32), ("b_div", i32), ("b_mod", i32), ("c_div", i32), ("c_mod", i32),
                ("s_div", i32), ("s_mod", i32), ("r_div", i32), ("r_mod", i32),
                ("a_str", i64), ("b_str", i64), ("c_str", i64), ("s_str", i64), ("r_str", i64),
                ("H", i32), ("W", i32), ("Cin", i32), ("Ho", i32), ("Wo", i32), ("KH", i32), ("KW", i32),
                ("stride", i32), ("pad", i32), ("split_k", i32), ("tile", i32), ("flags", i32), ("seg_mode", i32),
                ("row_seg", vp), ("seg_period", i32), ("seg_div", i32)]


# name -> argtypes (restype is int unless noted); must list every symbol of include/cadre_hip.h
SYMBOLS = {
    "cadre_abi_version": [],
    "cadre_last_error": [],
    "cadre_gemm_f32": [C.POINTER(GemmDesc), vp],
    "cadre_gemm_pick_tile": [C.POINTER(GemmDesc)],
    "cadre_gemm_bf16": [C.POINTER(GemmDesc), vp],
    "cadre_gemm_bf16_pick_tile": [C.POINTER(GemmDesc)],
    "cadre_conv3x3_ring": [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp],
    "cadre_conv3x3_ring_supported": [i32, i32, i32, i32, i32, i32],
    "cadre_conv3x3_ring_ntile": [i32, i32, i32, i32, i32, i32],
    "cadre_conv3x3_s2": [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp],
    "cadre_conv3x3_s2_supported": [i32, i32, i32, i32, i32],
    "cadre_conv3x3_s1x": [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp],
    "cadre_conv3x3_s1x_supported": [i32, i32, i32, i32, i32, i32],
    "cadre_conv3x3_s1x_stages": [i32, i32],
    "cadre_gemm_bf16_w128": [vp, vp, vp, i32, i32, i32, i32, i32, i32, vp],
    "cadre_gemm_bf16_w128_supported": [i32, i32, i32, i32, i32, i32],
    "cadre_maxpool3x3s2_bf16": [vp, vp, i32, i32, i32, i32, vp],
    "cadre_pam_bf16out": [vp, vp, f32, vp, i32, i32, vp],
    "cadre_cam_bf16out": [vp, f32, vp, i32, i32, vp],
    "cadre_splitk_reduce": [vp, i32, i64, i64, vp, i64, i32, i32, vp, vp, i32, f32, vp, i32, vp],
    "cadre_preprocess": [vp, vp, vp, vp, vp, vp, i32, i32, i32, vp],
    "cadre_preprocess_bf16pad": [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp],
    "cadre_maxpool3x3s2": [vp, vp, i32, i32, i32, i32, vp],
    "cadre_pack_obs": [vp, vp, vp, vp, vp, i32, i32, i32, vp, i32, vp],
    "cadre_stem_pool": [vp, vp, vp, vp, vp, i32, i32, i32, i32, i64, i64, i32, i64, vp],
    "cadre_div255_selfcheck": [vp, vp, vp],
    "cadre_stem_pool_supported": [i32, i32],
    "cadre_winograd_c64": [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp],
    "cadre_winograd_in": [vp, vp, i32, i32, i32, i32, i32, vp],
    "cadre_winograd_out": [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp],
    "cadre_winograd_fused_supported": [i32, i32, i32, i32, i32, i32],
    "cadre_winograd_fused_capable": [i32, i32, i32, i32, i32, i32],
    "cadre_winograd_fused_ntb": [i32, i32],
    "cadre_winograd_frag_elems": [i32, i32, i32, i32, i32],
    "cadre_winograd_in_frag": [vp, vp, i32, i32, i32, i32, i32, vp],
    "cadre_winograd_gemm_out": [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp],
    "cadre_pam": [vp, vp, f32, vp, i32, i32, vp],
    "cadre_cam": [vp, f32, vp, i32, i32, vp],
    "cadre_intertask_att": [vp, vp, i64, i32, f32, vp],
    "cadre_append_measurements": [vp, vp, i64, i32, vp],
    "cadre_gae": [vp, vp, vp, vp, vp, vp, i32, i32, f32, f32, i32, vp],
    "cadre_gather_obs": [vp, i64, i32, vp, i32, vp, i64, i32, vp],
    "cadre_gather_minibatch": [vp, i64, i32, vp, vp, i64, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32,
                               vp, i64, vp, vp, i64, vp, vp, vp, vp, vp, vp, vp],
    "cadre_gather_minibatch_multi": [vp, i32, i64, i32, i64, vp, i32, i32, i32, i32, vp, i64, i64, vp, vp, i64, i64,
                                     vp, vp, vp, vp, vp, vp, vp],
    "cadre_gather_sorted_multi": [vp, i32, i64, i32, i64, vp, i32, i32, i32, i32, i32, vp, i64, i64, vp, vp, i64, i64,
                                  vp, vp, vp, vp, vp, vp, vp, vp, vp],
    "cadre_pack_lstm_weights": [vp, i64, i32, i32, i32, vp, vp, i64, vp],
    "cadre_lstm_step_fwd": [vp, i64, vp, i64, vp, i32, i64, vp, vp, vp, vp, vp, i32, i64, i32, i32, i32, vp, i32, vp],
    "cadre_lstm_step_bwd": [vp, i64, vp, vp, i64, vp, vp, i32, i64, vp, vp, i64, vp, vp, i32, i64, i32, i32, i32, vp, i32, vp, i32, vp],
    "cadre_mlp_fwd": [vp, i64, vp, vp, i32, i64, vp, vp, vp, i32, i32, vp, vp],
    "cadre_mlp_bwd": [vp, i64, vp, vp, vp, vp, vp, vp, vp, i32, i64, i32, i32, vp, vp],
    "cadre_mlp_dw": [vp, vp, vp, vp, vp, vp, i32, i64, vp, i64, vp, i32, i32, vp, vp],
    "cadre_lstm_dw": [vp, i32, i64, vp, vp, i32, i64, i64, i32, vp, vp, vp, vp, i32, i64, i32, i32, i32, i32, i32, vp, vp],
    "cadre_colsum": [vp, i64, i64, vp, i64, i32, i32, i32, i32, vp],
    "cadre_relu_bwd": [vp, vp, i64, vp, i32, i32, i32, vp],
    "cadre_lstm_init": [vp, vp, vp, vp, vp, i64, i64, i32, i32, vp],
    "cadre_mfma_peak": [i32, i32, i32, vp, vp],
    "cadre_hbm_stream": [i32, vp, vp, i64, vp, vp],
    "cadre_mfma_shape": [i32, i32, i32, vp, vp],
    "cadre_sort_rows_by_command": [vp, i32, i32, vp, vp, vp],
    "cadre_permute_minibatch": [vp, i32, i32, vp, vp, i64, vp, vp, vp, vp, i64, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i64, i64, vp],
    "cadre_ppo_loss": [vp, i64, i64, vp, i64, i64, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, f32, f32, f32, f32, f32, vp, vp, vp, vp, vp, vp],
    "cadre_sample": [vp, i64, vp, i64, i32, i32, vp, vp, vp],
    "cadre_categorical_eval": [vp, i64, vp, i32, i32, vp, vp, vp],
    "cadre_categorical_dist": [vp, i64, i32, i32, vp, vp, vp, vp],
    "cadre_clip_adam": [vp, vp, vp, vp, vp, i32, vp, f64, f64, f64, f64, f64, i32, vp],
    "cadre_clip_adam_graph": [vp, vp, vp, vp, vp, i32, vp, f64, f64, f64, f64, f64, vp, vp],
    "cadre_clip_adam_pack_graph": [vp, vp, vp, vp, vp, i32, vp, f64, f64, f64, f64, f64, vp, i32, i64, i64, i32, i32, i32, vp, vp, i64, vp],
    "cadre_clip_adam_norms": [vp, vp, i32, vp, f64, f64, f64, vp, i64, i64, vp],
    "cadre_clip_adam_apply": [vp, vp, vp, vp, vp, i32, vp, f64, f64, f64, f64, i64, i64, vp],
}
# entry points of the A/B build only (include/cadre_hip_ab.h; CADRE_BUILD_AB=1 python -m cadre_amd.build, then
# CADRE_HIP_LIB=.../libcadre_hip_ab.so): bound when the loaded library has them
AB_SYMBOLS = {
    "cadre_lstm_pointwise_fwd": [vp, i64, i64, vp, i64, i32, vp, vp, vp, i64, i64, i32, i32, i32, vp, vp],
    "cadre_lstm_pointwise_bwd": [vp, vp, i64, i64, vp, vp, i64, vp, vp, i64, i32, i64, i64, i32, i32, i32, vp, i32, vp, vp],
    "cadre_colsum2": [vp, i64, i64, vp, vp, i64, i32, i32, i32, vp, i32, vp],
    "cadre_conv3x3_c64_bf16": [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp],
    "cadre_lstm_seq_fwd": [vp, i64, vp, i64, vp, i32, i64, vp, vp, vp, i32, i64, i32, i32, i32, i32, vp, vp, vp],
    "cadre_conv3x3_w128": [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp],
    "cadre_conv3x3_w128_supported": [i32, i32, i32, i32, i32],
}


ABI_VERSION = 15


class CadreHipError(RuntimeError):
    pass


def has_ab_kernels():
    """True when the loaded library is the A/B build (superseded kernels of csrc/ab/ present)."""
    return hasattr(lib(), "cadre_conv3x3_c64_bf16")


def lib():
    """Load the HIP library; fail loudly if it is missing (no fallback path exists)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise CadreHipError(
                "libcadre_hip.so not built (%s). Run `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `python -m cadre_amd.build`. There is no CPU fallback for the Cadre MI355X learner." % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        for name, args in SYMBOLS.items():
            fn = getattr(L, name)
            fn.argtypes = args
            fn.restype = C.c_char_p if name == "cadre_last_error" else (C.c_int64 if name == "cadre_winograd_frag_elems" else C.c_int)
        if L.cadre_abi_version() != ABI_VERSION:
            raise CadreHipError("libcadre_hip.so ABI version mismatch (library %d, binding %d): rebuild with "
                                "`python -m cadre_amd.build`" % (L.cadre_abi_version(), ABI_VERSION))
        for name, args in AB_SYMBOLS.items():
            fn = getattr(L, name, None)
            if fn is not None:
                fn.argtypes = args
                fn.restype = C.c_int
        _lib = L
    return _lib


N_CALLS = 0          # C-ABI calls checked so far (one kernel launch each, cadre_clip_adam_graph three): launch census


def check(rc, what):
    global N_CALLS
    N_CALLS += 1
    if rc != 0:
        msg = lib().cadre_last_error().decode() if rc < 0 else "hipError_t %d" % rc
        raise CadreHipError("%s failed: %s" % (what, msg))


def ptr(t):
    if t is None:
        return None
    if not t.is_cuda:
        raise CadreHipError("cadre_amd kernels need device (HIP) tensors; got a CPU tensor — there is no CPU path")
    return t.data_ptr()


def stream():
    return torch.cuda.current_stream().cuda_stream


# Optional launch profiler (bench.py): when PROFILE is a list, every gemm launch is bracketed by
# HIP events recorded on the launch stream and appended as (key, flops, start, end, shape, bytes):
#   key   = (tile, a_mode, b_mode) for cadre_gemm_f32, ("bf16", tile, a_mode) for cadre_gemm_bf16
#   flops = algorithmic FLOPs of the launch, bytes = algorithmic HBM bytes (each operand once)
#   shape = (M, N, K, batch, split_k, seg_mode)
PROFILE = None


def gemm(A, B, Cout, M, N, K, lda, ldb, ldc, a_mode=0, b_mode=0, scale=None, shift=None, resid=None, ldr=0,
         act=0, slope=0.01, batch=1, a_z=(1, 0, 0), b_z=(1, 0, 0), c_z=(1, 0, 0), s_z=(1, 0, 0), r_z=(1, 0, 0),
         conv=None, split_k=1, tile=0, bf16=False, flags=0, seg=None):
    """C = act((A . B^T) * scale + shift + resid).  *_z = (div, mod, stride) batch addressing.
    bf16=True: A/B are bfloat16 tensors (cadre_gemm_bf16); flags bit1/bit2: C / resid are bf16."""
    d = GemmDesc()
    d.A, d.B, d.C = ptr(A), ptr(B), ptr(Cout)
    d.scale, d.shift, d.resid = ptr(scale), ptr(shift), ptr(resid)
    d.lda, d.ldb, d.ldc, d.ldr = lda, ldb, ldc, ldr
    d.M, d.N, d.K, d.a_mode, d.b_mode, d.act, d.slope, d.batch = M, N, K, a_mode, b_mode, act, slope, batch
    (d.a_div, d.a_mod, d.a_str), (d.b_div, d.b_mod, d.b_str) = a_z, b_z
    (d.c_div, d.c_mod, d.c_str), (d.s_div, d.s_mod, d.s_str), (d.r_div, d.r_mod, d.r_str) = c_z, s_z, r_z
    if conv is not None:
        d.H, d.W, d.Cin, d.Ho, d.Wo, d.KH, d.KW, d.stride, d.pad = conv
    d.split_k, d.tile = split_k, tile
    d.flags = flags
    if seg is not None:                  # (mode, row_seg tensor, period, div)
        d.seg_mode, d.seg_period, d.seg_div = seg[0], seg[2], seg[3]
        d.row_seg = ptr(seg[1])
    fn = lib().cadre_gemm_bf16 if bf16 else lib().cadre_gemm_f32
    name = "cadre_gemm_bf16" if bf16 else "cadre_gemm_f32"
    if PROFILE is None or torch.cuda.is_current_stream_capturing():
        check(fn(C.byref(d), stream()), name)
        return
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    check(fn(C.byref(d), stream()), name)
    e1.record()
    nb = max(1, batch)
    k_alg = conv[5] * conv[6] * conv[2] if conv is not None else K      # algorithmic K (the stems pad 196 -> 224 / 256)
    esz = 2 if bf16 else 4
    a_bytes = (M // (conv[3] * conv[4])) * conv[0] * conv[1] * conv[2] * esz if conv is not None else M * K * esz * nb
    c_esz = 2 if (flags & 2) else 4
    r_esz = 2 if (flags & 4) else 4
    nbytes = a_bytes + N * k_alg * esz * nb + M * N * c_esz * nb * max(1, split_k) + (M * N * r_esz * nb if resid is not None else 0)
    if bf16:
        key = ("bf16", lib().cadre_gemm_bf16_pick_tile(C.byref(d)), a_mode)
    else:
        key = (lib().cadre_gemm_pick_tile(C.byref(d)), a_mode, b_mode)
    PROFILE.append((key, 2.0 * M * N * k_alg * nb, e0, e1, (M, N, K, nb, split_k, seg[0] if seg is not None else 0), nbytes))


def conv3x3_c64_bf16(x, w, scale, shift, resid, out, F, H, W, relu):
    """cadre_conv3x3_c64_bf16 (A/B build only) with the same profiling hook as gemm(): key ("bf16", 64, 2)."""
    if not has_ab_kernels():
        raise CadreHipError("cadre_conv3x3_c64_bf16 exists only in the A/B build (CADRE_BUILD_AB=1 python -m cadre_amd.build)")
    fn = lib().cadre_conv3x3_c64_bf16
    args = (ptr(x), ptr(w), ptr(scale), ptr(shift), ptr(resid), ptr(out), F, H, W, relu, stream())
    if PROFILE is None or torch.cuda.is_current_stream_capturing():
        check(fn(*args), "cadre_conv3x3_c64_bf16")
        return
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    check(fn(*args), "cadre_conv3x3_c64_bf16")
    e1.record()
    M = F * H * W
    nbytes = M * 64 * 2 * (3 if resid is not None else 2) + 64 * 576 * 2
    PROFILE.append((("bf16", 64, 2), 2.0 * M * 64 * 576, e0, e1, (M, 64, 576, 1, 1, 0), nbytes))


def winograd_c64(x, u, scale, shift, resid, out, F, H, W, act):
    """cadre_winograd_c64 (fused Winograd F(2x2,3x3) of the fp32 64 -> 64 stage) with the profiling hook of gemm():
    key ("wino_c64", residual); FLOPs = the EXECUTED ones (16 planes x tiles x 64 x 64)."""
    fn = lib().cadre_winograd_c64
    args = (ptr(x), ptr(u), ptr(scale), ptr(shift), ptr(resid), ptr(out), F, H, W, act, stream())
    if PROFILE is None or torch.cuda.is_current_stream_capturing():
        check(fn(*args), "cadre_winograd_c64")
        return
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    check(fn(*args), "cadre_winograd_c64")
    e1.record()
    T = F * ((H + 1) // 2) * ((W + 1) // 2)
    nbytes = F * H * W * 64 * 4 * (3 if resid is not None else 2) + 16 * 64 * 64 * 4
    PROFILE.append((("wino_c64", resid is not None), 2.0 * 16 * T * 64 * 64, e0, e1, (T, 64, 64 * 16, 1, 1, 0), nbytes))


def winograd_fused(x, V, u_frag, scale, shift, resid, out, F, H, W, Cin, N, act, m):
    """cadre_winograd_in_frag + cadre_winograd_gemm_out (Winograd F(m x m, 3x3): the plane products and the inverse transform in one
    kernel, csrc/winograd_fused.hip).  Profiling key ("wgo", m, ntb) on the product kernel: wino_gemm_out_kernel<m, ntb>; FLOPs = the EXECUTED
    ones ((m+2)^2 planes x tiles x Cin x N), bytes = V read once + the output written (+ the residual read) + U."""
    L = lib()
    check(L.cadre_winograd_in_frag(ptr(x), ptr(V), F, H, W, Cin, m, stream()), "cadre_winograd_in_frag")
    fn = L.cadre_winograd_gemm_out
    args = (ptr(V), ptr(u_frag), ptr(scale), ptr(shift), ptr(resid), ptr(out), F, H, W, Cin, N, act, m, stream())
    if PROFILE is None or torch.cuda.is_current_stream_capturing():
        check(fn(*args), "cadre_winograd_gemm_out")
        return
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    check(fn(*args), "cadre_winograd_gemm_out")
    e1.record()
    P, T = (m + 2) ** 2, F * -(-H // m) * -(-W // m)
    nbytes = (P * T * Cin + P * N * Cin + F * H * W * N * (2 if resid is not None else 1)) * 4
    PROFILE.append((("wgo", m, int(L.cadre_winograd_fused_ntb(T, N))), 2.0 * P * T * Cin * N, e0, e1, (T, N, Cin * P, 1, 1, 0), nbytes))


def conv3x3_s2(x, w_s2, scale, shift, out, F, H, W, Cin, N, act):
    """cadre_conv3x3_s2 (3x3 / s2 / p1 on bf16 NHWC: four parity-plane windows in LDS); profiling key ("s2", npw):
    conv3x3_s2_kernel<npw>, npw = window pieces per wave (9 for output rows of <= 31 pixels, else 10)."""
    fn = lib().cadre_conv3x3_s2
    args = (ptr(x), ptr(w_s2), ptr(scale), ptr(shift), ptr(out), F, H, W, Cin, N, act, stream())
    if PROFILE is None or torch.cuda.is_current_stream_capturing():
        check(fn(*args), "cadre_conv3x3_s2")
        return
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    check(fn(*args), "cadre_conv3x3_s2")
    e1.record()
    M = F * (H // 2) * (W // 2)
    nbytes = F * H * W * Cin * 2 + N * 9 * Cin * 2 + M * N * 2
    npw = 9 if (256 + W // 2 + 1 + 7) // 8 <= 36 else 10
    PROFILE.append((("s2", npw), 2.0 * M * N * 9 * Cin, e0, e1, (M, N, 9 * Cin, 1, 1, 0), nbytes))


def conv3x3_s1x(x, x2, w_s1x, shift, out, F, H, W, C1, Cd, N, act):
    """cadre_conv3x3_s1x (3x3 / s1 conv + the block's 1x1 / s2 shortcut as extra k-tiles, one accumulation); profiling key
    ("s1x", nps, nstg): conv3x3_s1x_kernel<nps, nstg>, nps = stride-1 window pieces per wave (9 / 10 / 11 by map width), nstg =
    weight stages (2; CADRE_S1X_STAGES=3 selects the symmetric-issue form)."""
    fn = lib().cadre_conv3x3_s1x
    args = (ptr(x), ptr(x2), ptr(w_s1x), ptr(shift), ptr(out), F, H, W, C1, Cd, N, act, stream())
    if PROFILE is None or torch.cuda.is_current_stream_capturing():
        check(fn(*args), "cadre_conv3x3_s1x")
        return
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    check(fn(*args), "cadre_conv3x3_s1x")
    e1.record()
    M = F * H * W
    nbytes = (M * C1 + M * Cd + N * (9 * C1 + Cd) + M * N) * 2          # (the shortcut reads one pixel in four of x2)
    pa = ((256 + 2 * W + 2 + 7) // 8 * 8) // 8
    nps = 9 if pa <= 36 else (10 if pa <= 40 else 11)
    nstg = int(lib().cadre_conv3x3_s1x_stages(W, N))       # (what the library launches: a request for 3 stages falls back to 2 where they do not fit)
    PROFILE.append((("s1x", nps, nstg), 2.0 * M * N * (9 * C1 + Cd), e0, e1, (M, N, 9 * C1 + Cd, 1, 1, 0), nbytes))


def gemm_bf16_w128(A, B_frag, slabs, M, N, K, lda, ldc, split_k):
    """cadre_gemm_bf16_w128 (dense bf16 split-K product, 128 x 128 wave tile, B streamed to registers in fragment order); profiling
    key ("gw128",): gemm_bf16_w128_kernel."""
    fn = lib().cadre_gemm_bf16_w128
    args = (ptr(A), ptr(B_frag), ptr(slabs), M, N, K, lda, ldc, split_k, stream())
    if PROFILE is None or torch.cuda.is_current_stream_capturing():
        check(fn(*args), "cadre_gemm_bf16_w128")
        return
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    check(fn(*args), "cadre_gemm_bf16_w128")
    e1.record()
    nbytes = (M * K + N * K) * 2 + M * N * 4 * split_k
    PROFILE.append((("gw128",), 2.0 * M * N * K, e0, e1, (M, N, K, 1, split_k, 0), nbytes))


def w128_shape(W, N):
    """(MW, NPW) template arguments cadre_conv3x3_w128 picks (conv3x3_w128.hip w128_pick) — the profiling key / kernel name."""
    mw = 2 if N % 256 == 0 else 4
    need = (128 * mw + 2 * W + 2 + 31) // 32
    if mw == 2:
        return mw, (9 if need <= 9 else (10 if need <= 10 else 11))
    return mw, (17 if need <= 17 else 19)


def conv3x3_w128(x, w_frag, shift, resid, out, F, H, W, Cin, N, act):
    """cadre_conv3x3_w128 (3x3 / s1 window conv, 128 x 128 wave tile, weights streamed to registers); profiling key
    ("w128", mw, npw, residual): conv3x3_w128_kernel<mw, npw, residual>.  A/B build only."""
    if not has_ab_kernels():
        raise CadreHipError("cadre_conv3x3_w128 exists only in the A/B build (CADRE_BUILD_AB=1 python -m cadre_amd.build)")
    fn = lib().cadre_conv3x3_w128
    args = (ptr(x), ptr(w_frag), None, ptr(shift), ptr(resid), ptr(out), F, H, W, Cin, N, act, stream())
    if PROFILE is None or torch.cuda.is_current_stream_capturing():
        check(fn(*args), "cadre_conv3x3_w128")
        return
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    check(fn(*args), "cadre_conv3x3_w128")
    e1.record()
    M = F * H * W
    nbytes = (M * Cin + N * 9 * Cin + M * N * (2 if resid is not None else 1)) * 2
    mw, npw = w128_shape(W, N)
    PROFILE.append((("w128", mw, npw, resid is not None), 2.0 * M * N * 9 * Cin, e0, e1, (M, N, 9 * Cin, 1, 1, 0), nbytes))


def conv3x3_ring(x, w_ring, scale, shift, resid, out, F, H, W, Cin, N, act):
    """cadre_conv3x3_ring (3x3 / s1 / p1, each pixel through LDS once per channel chunk); profiling key
    ("ring", bf16, ntile, res, out_bf16, WVM, pp, G): pp 0 = conv3x3_ring_kernel<bf16, ntile, res, out_bf16, WVM>,
    pp 1, G 0 = conv3x3_ring_pp_kernel<bf16, ntile, res, out_bf16, false> (the 8-wave ping-pong kernel), G > 0 =
    conv3x3_ring_pp2_kernel<ntile, res, out_bf16, G> (G k-tiles per ping-pong slot)."""
    bf = x.dtype == torch.bfloat16
    flags = (1 if bf else 0) | (2 if out.dtype == torch.bfloat16 else 0) | (4 if (resid is not None and resid.dtype == torch.bfloat16) else 0)
    fn = lib().cadre_conv3x3_ring
    args = (ptr(x), ptr(w_ring), ptr(scale), ptr(shift), ptr(resid), ptr(out), F, H, W, Cin, N, act, flags, stream())
    if PROFILE is None or torch.cuda.is_current_stream_capturing():
        check(fn(*args), "cadre_conv3x3_ring")
        return
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    check(fn(*args), "cadre_conv3x3_ring")
    e1.record()
    M, esz = F * H * W, (2 if bf else 4)
    nbytes = M * Cin * esz + N * 9 * Cin * esz + M * N * out.element_size() + (M * N * resid.element_size() if resid is not None else 0)
    code = lib().cadre_conv3x3_ring_ntile(F, H, W, Cin, N, 1 if bf else 0)      # ntile + 1000 * WVM + 100000 * ping-pong + 1000000 * G
    res = 0 if resid is None else (2 if resid.dtype == torch.bfloat16 else 1)
    key = ("ring", bf, code % 1000, res, out.dtype == torch.bfloat16, code // 1000 % 100, code // 100000 % 10, code // 1000000)
    PROFILE.append((key, 2.0 * M * N * 9 * Cin, e0, e1, (M, N, 9 * Cin, 1, 1, 0), nbytes))
