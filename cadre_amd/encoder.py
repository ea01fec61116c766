"""DANet act-time encoder forward on MI355X: `DANet.get_latent_feature(x, "concate")`
(reference carla_perception/Networks/danet.py:216-238) as a chain of HIP launches.

Data layout: activations NHWC fp32 in HBM (one [F*Ho*Wo, C] row-major matrix per layer, so
every conv is an implicit GEMM whose output is already the next layer's input), conv weights
[Cout][KH][KW][Cin] (k contiguous), eval BatchNorm folded to a per-channel (scale, shift)
epilogue, the six inter-task first-layer matrices re-ordered once at load time from the
reference's NCHW flatten (c*hw+pos) to the NHWC flatten (pos*512+c) and concatenated per
branch so the 128 M weights (288x288) stream through HBM once per frame batch.

Weights come in the reference's state_dict naming so a real `net_epoch<N>` checkpoint
(`{'autoencoder': state_dict}`, experiments_builder.py:442-462) drops in.  Any input size
whose layer-4 map has <= 1024 positions is supported (the reference hard-codes 5x8).
"""
import os

import numpy as np
import torch

from . import hip, synth

BN_EPS = 1e-5


def _fold_bn(sd, bn, conv_bias=None):
    w, b = sd[bn + ".weight"].double(), sd[bn + ".bias"].double()
    mean, var = sd[bn + ".running_mean"].double(), sd[bn + ".running_var"].double()
    scale = w / torch.sqrt(var + BN_EPS)
    shift = b - mean * scale
    if conv_bias is not None:
        shift = shift + conv_bias.double() * scale
    return scale.float(), shift.float()


def _khwc(w):
    """OIHW -> [O][KH*KW*I] with k contiguous; the Cin=4 stem -> [O][KH][32] rows (a_mode 3: one k-tile
    per kernel row = 8 NHWC4 pixels, zero weights past KW)."""
    o, i, kh, kw = w.shape
    k = w.permute(0, 2, 3, 1)
    if i == 4 and kw <= 8:
        rows = torch.zeros(o, kh, 32, dtype=w.dtype)
        rows[:, :, :kw * 4] = k.reshape(o, kh, kw * 4)
        return rows.reshape(o, kh * 32).contiguous()
    return k.reshape(o, -1).contiguous()


def _stem_taps(w, ntaps, row8=False):
    """OIHW Cin=4 stem weights -> tap-major [O][ntaps][4], the B operand of the fused front (cadre_stem_pool).
    fp32: tap = ky*KW + kx, zero taps appended (K = 4*50).  bf16 (row8): tap = ky*8 + kx with zero weights at kx = 7
    (K = 4*56): no pair of adjacent taps straddles a kernel row (stem_pool.hip, second form)."""
    o, i, kh, kw = w.shape
    if row8:
        assert kw <= 8 and ntaps == kh * 8
        out = torch.zeros(o, kh, 8, i, dtype=torch.float32)
        out[:, :, :kw] = w.permute(0, 2, 3, 1)
        return out.reshape(o, ntaps * i).contiguous()
    out = torch.zeros(o, ntaps, i, dtype=torch.float32)
    out[:, :kh * kw] = w.permute(0, 2, 3, 1).reshape(o, kh * kw, i)
    return out.reshape(o, ntaps * i).contiguous()


def _stem_taps_x3(w):
    """OIHW Cin=4 fp32 stem weights -> [3 pieces][O][216] bf16 for cadre_stem_pool mode 2 (stem_pool.hip X3): k = tap * 4 + channel,
    tap = ky*7 + kx, zeros past tap 48 (52 taps = 13 k-steps of 16; row pitch 216).  p1 = bf16(w), p2 = bf16(w - p1),
    p3 = w - p1 - p2: the three pieces sum to the fp32 weight EXACTLY (8 + 8 + 8 significand bits; asserted here)."""
    o, i, kh, kw = w.shape
    flat = torch.zeros(o, 216, dtype=torch.float32)
    flat[:, :kh * kw * i] = w.float().permute(0, 2, 3, 1).reshape(o, kh * kw * i)
    p1 = flat.to(torch.bfloat16)
    r1 = flat - p1.float()
    p2 = r1.to(torch.bfloat16)
    r2 = r1 - p2.float()
    p3 = r2.to(torch.bfloat16)
    if not torch.equal(p1.float() + p2.float() + p3.float(), flat) or not torch.equal(p3.float(), r2):
        raise hip.CadreHipError("stem weights do not split into three bf16 pieces exactly (subnormal weights?)")
    return torch.stack([p1, p2, p3]).contiguous()


def _stem_rows_bf16(w):
    """OIHW Cin=4 stem weights -> [O][ceil(KH/2)][64] bf16 rows for cadre_gemm_bf16 a_mode 4: k-tile kt
    holds kernel rows 2kt, 2kt+1; within it chunk cc (8 values) = pixels 2(cc&3), 2(cc&3)+1 x 4 channels
    of row 2kt + (cc>>2); zero where kh >= KH or kw >= KW."""
    o, i, kh, kw = w.shape
    nkt = (kh + 1) // 2
    out = torch.zeros(o, nkt, 8, 2, 4, dtype=torch.float32)
    for kt in range(nkt):
        for cc in range(8):
            r = 2 * kt + (cc >> 2)
            for e2 in range(2):
                c = 2 * (cc & 3) + e2
                if r < kh and c < kw:
                    out[:, kt, cc, e2, :] = w[:, :, r, c]
    return out.reshape(o, nkt * 64).contiguous()


def _ring_w(w, cpc):
    """OIHW 3x3 weights -> [O][Cin/cpc][9][cpc] (channel chunk of 128 bytes, tap, channel): cadre_conv3x3_ring's B."""
    o, i, kh, kw = w.shape
    return w.permute(0, 2, 3, 1).reshape(o, kh * kw, i // cpc, cpc).permute(0, 2, 1, 3).contiguous()


S2_TAPS = ((0, 0), (0, 2), (2, 0), (2, 2), (1, 0), (1, 2), (0, 1), (2, 1), (1, 1))


def _s2_w(w):
    """OIHW 3x3 weights -> [O][Cin/64][9][64], the nine taps of a 64-channel chunk in PLANE order (the parity plane of the input
    pixel a tap reads under stride 2: (odd, odd) x 4, (even, odd) x 2, (odd, even) x 2, (even, even) x 1): cadre_conv3x3_s2's B."""
    o, i, kh, kw = w.shape
    assert kh == 3 and kw == 3 and i % 64 == 0
    taps = torch.stack([w[:, :, a, b] for a, b in S2_TAPS], dim=1)            # [O][9][I]
    return taps.reshape(o, 9, i // 64, 64).permute(0, 2, 1, 3).contiguous()


def _w128_w(w):
    """3x3 / s1 weights [O][I][3][3] (fp32, BN scale folded) in the FRAGMENT order of cadre_conv3x3_w128 (include/cadre_hip.h):
    [O/128][I/64][9 taps][4 k-steps][4 blocks][lane half lh][32 lanes l31][8] with channel 128 g + 32 cb + l31, input channel
    64 c + 16 s + 8 lh + e."""
    O, I = w.shape[0], w.shape[1]
    t = w.reshape(O // 128, 4, 32, I // 64, 4, 2, 8, 9)                       # g cb l31 c s lh e tap
    return t.permute(0, 3, 7, 4, 1, 5, 2, 6).contiguous().reshape(-1)          # g c tap s cb lh l31 e


def _w128_dense_b(w):
    """Dense weights [N][K] (fp32) in the FRAGMENT order of cadre_gemm_bf16_w128 (include/cadre_hip.h): [N/128][K/16][4 blocks][lane
    half lh][32 lanes l31][8] with row 128 g + 32 cb + l31, column 16 q + 8 lh + e."""
    N, K = w.shape
    t = w.reshape(N // 128, 4, 32, K // 16, 2, 8)                              # g cb l31 q lh e
    return t.permute(0, 3, 1, 4, 2, 5).contiguous().reshape(-1)                # g q cb lh l31 e


def _s1x_w(w2, wd):
    """conv2 weights OIHW [O][C1][3][3] + shortcut weights [O][Cd][1][1] (both with their folded-BN scale already multiplied in)
    -> [O][9 C1/64 + Cd/64][64] in the k-tile order of cadre_conv3x3_s1x: per 64-channel chunk c of C1 the nine taps kh*3 + kw,
    then (c < Cd/64) chunk c of the shortcut."""
    o, c1 = w2.shape[0], w2.shape[1]
    cd = wd.shape[1]
    assert c1 % 64 == 0 and cd % 64 == 0 and cd <= c1 and (cd == 0 or wd.shape[0] == o)
    rows = []
    for c in range(c1 // 64):
        rows.append(w2[:, 64 * c:64 * c + 64].permute(0, 2, 3, 1).reshape(o, 9, 64))
        if c < cd // 64:
            rows.append(wd[:, 64 * c:64 * c + 64, 0, 0].reshape(o, 1, 64))
    return torch.cat(rows, dim=1).contiguous()


def _winograd_min_c():
    """Stride-1 3x3 convs of the fp32 model with at least this many input channels (and >= 128 output channels) run as
    Winograd F(3x3, 3x3) / F(2x2, 3x3): 128 = layer2, layer3, layer4, head — at 64 channels (layer1) the transform-domain
    traffic costs more than the fewer MACs save.  CADRE_WINOGRAD=0: direct convolution everywhere
    (returns 0).  Read when an encoder is built."""
    if os.environ.get("CADRE_WINOGRAD", "1") in ("", "0"):
        return 0
    return int(os.environ.get("CADRE_WINOGRAD_MIN_C", "128"))


_WINO_G = {2: [[1.0, 0.0, 0.0], [0.5, 0.5, 0.5], [0.5, -0.5, 0.5], [0.0, 0.0, 1.0]],
           3: [[8.0 / 9, 0.0, 0.0], [-32.0 / 45, -8.0 / 15, -2.0 / 5], [-32.0 / 99, 8.0 / 33, -2.0 / 11], [8.0 / 55, 16.0 / 55, 32.0 / 55],
               [0.0, 0.0, 1.0]],
           # F(4x4): points 0, 3/4, -3/4, 3/2, -3/2, infinity; rows scaled against the powers of two taken out of B^T (winograd.hip)
           4: [[64.0 / 81, 0.0, 0.0], [128.0 / 243, 128.0 / 324, 128.0 / 432], [128.0 / 243, -128.0 / 324, 128.0 / 432],
               [32.0 / 243, 32.0 / 162, 32.0 / 108], [32.0 / 243, -32.0 / 162, 32.0 / 108], [0.0, 0.0, 1.0]],
           # F(6x6): points 0, +-1/2, +-1, +-2, infinity (tools/dbg/wino_points.py prints G for the B^T / A^T of csrc/winograd_mats.h)
           6: [[-2.0, 0.0, 0.0], [64.0 / 45, 32.0 / 45, 16.0 / 45], [64.0 / 45, -32.0 / 45, 16.0 / 45], [-4.0 / 9, -4.0 / 9, -4.0 / 9],
               [-4.0 / 9, 4.0 / 9, -4.0 / 9], [1.0 / 90, 1.0 / 45, 2.0 / 45], [1.0 / 90, -1.0 / 45, 2.0 / 45], [0.0, 0.0, 2.0]]}


def _winograd_u(w, m=2):
    """OIHW 3x3 weights -> U[(m+2)^2][O][I] = (G g G^T)[xi], xi = (m+2) i + j (float64 product, one rounding to fp32): the B
    operands of the batched GEMM between cadre_winograd_in and cadre_winograd_out (csrc/winograd.hip; Cook-Toom points
    0, 1, -1, infinity for m = 2, 0, 3/4, -3/4, 2, infinity for m = 3, 0, +-3/4, +-3/2, infinity for m = 4 — G matches the kernels' B^T / A^T)."""
    G = torch.tensor(_WINO_G[m], dtype=torch.float64)
    u = torch.einsum("ik,ockl,jl->ijoc", G, torch.as_tensor(w).double(), G)        # [m+2][m+2][O][I]
    return u.reshape((m + 2) ** 2, w.shape[0], w.shape[1]).float().contiguous()


def _winograd_u_frag(w, m):
    """OIHW 3x3 weights -> U = (G g G^T)[xi][O][I] in the fragment order of cadre_winograd_gemm_out (csrc/winograd_fused.hip):
    [O/32][I/16][(m+2)^2][2 nb][4 kk][16 couts][4 e] with output channel 32 nt + 16 nb + co and input channel 16 c + 4 kk + e — the
    2 KB block (nt, c, xi) is what the two channel-block waves read as one ds_read_b128 per lane (lane = 16 kk + co)."""
    u = _winograd_u(w, m)                                            # [P][O][I]
    P, O, I = u.shape
    assert O % 32 == 0 and I % 16 == 0
    t = u.reshape(P, O // 32, 2, 16, I // 16, 4, 4)                  # p nt nb co c kk e
    return t.permute(1, 4, 0, 2, 5, 3, 6).contiguous().reshape(-1)   # nt c p nb kk co e


def _winograd_u_c64(w, cin_pairs=True):
    """OIHW 64 x 64 x 3 x 3 -> the fused kernel's layout [8 chunks of 8 cin][16 planes][64 positions][8 cin]
    (csrc/winograd_c64.hip); position 16 b + n of the cout axis holds output channel 4 n + b: column n of the kernel's MFMA
    block b — a lane then owns four consecutive channels of a pixel (16-byte stores); chunk c = 2d + e, index k = 2q + s of the
    cin axis hold input channel 16 d + 4 q + 2 e + s (16-byte patch requests; cin_pairs=False: natural order, older builds)."""
    u = _winograd_u(w, 2)                                          # [16][O = 64][I = 64]
    pos = torch.arange(64)
    u = u[:, 4 * (pos % 16) + pos // 16, :]
    if cin_pairs:
        # chunk c = 2d + e, index k = 2q + s in the chunk  <->  input channel 16 d + 4 q + 2 e + s: a lane (q) requests the four
        # channels 16 d + 4 q .. + 3 of a pixel at once (16 bytes) and feeds steps 2d and 2d + 1 from them
        c, k = pos // 8, pos % 8
        u = u[:, :, 16 * (c // 2) + 4 * (k // 2) + 2 * (c % 2) + (k % 2)]
    return u.reshape(16, 64, 8, 8).permute(2, 0, 1, 3).contiguous()


def _winograd_m(H, W):
    """Output tile edge of the Winograd form for an H x W map: the one with fewer transform-domain multiplies,
    (m+2)^2 * ceil(H/m) * ceil(W/m) (F(3x3) tiles the 9x9 maps of the 288x288 model exactly; F(4x4) is taken only where 4x4 tiles
    cover the map exactly and it is the cheapest of m <= 4: the 36x36 maps, where the fused kernel of csrc/winograd_fused.hip runs;
    F(6x6) — round 6: 64 planes, 5.06x fewer multiplies, rounding error 4-5x the F(4x4) set's — where 6x6 tiles cover the map
    exactly, it is cheaper still, and the map's larger edge lies in [CADRE_WINOGRAD_M6_MIN, CADRE_WINOGRAD_M6_MAX] (default 18 .. 18:
    the 18x18 maps of layer3 at 288x288 — the small maps of the 84x84 test model keep F(3x3), whose error the 16-step data-parallel
    parameter test was tuned on; the 36x36 maps keep the fused F(4x4) kernel).
    CADRE_WINOGRAD_M forces 2, 3, 4 or 6; CADRE_WINOGRAD_M4=0 / CADRE_WINOGRAD_M6=0: never."""
    e = os.environ.get("CADRE_WINOGRAD_M", "")
    if e in ("2", "3", "4", "6"):
        return int(e)
    cost = {m: (m + 2) ** 2 * -(-H // m) * -(-W // m) for m in (2, 3)}
    best = 3 if cost[3] <= cost[2] else 2
    bc = cost[best]
    # F(4x4) only where 4 x 4 tiles cover the map EXACTLY (36 x 36: layer2 of the 288 x 288 model) and beat the others: its
    # rounding error is 1.6x the F(3x3) set's (CADRE_WINOGRAD_M4=0: never)
    if (os.environ.get("CADRE_WINOGRAD_M4", "1") != "0" and H % 4 == 0 and W % 4 == 0
            and 36 * (H // 4) * (W // 4) < bc):
        best, bc = 4, 36 * (H // 4) * (W // 4)
    if (os.environ.get("CADRE_WINOGRAD_M6", "1") != "0" and H % 6 == 0 and W % 6 == 0 and 64 * (H // 6) * (W // 6) < bc
            and int(os.environ.get("CADRE_WINOGRAD_M6_MIN", "18")) <= max(H, W) <= int(os.environ.get("CADRE_WINOGRAD_M6_MAX", "18"))):
        best = 6
    return best


class _Conv:
    __slots__ = ("w", "w_ring", "ring_folded", "scale", "shift", "cin", "cout", "k", "stride", "pad", "act", "w_wino", "_w_oihw", "w_wino_c64", "w_s2")

    def wino_u(self, m, dev):
        if m not in self.w_wino:
            self.w_wino[m] = _winograd_u(self._w_oihw, m).to(dev)
        return self.w_wino[m]

    def wino_u_frag(self, m, dev):
        if ("frag", m) not in self.w_wino:
            self.w_wino[("frag", m)] = _winograd_u_frag(self._w_oihw, m).to(dev)
        return self.w_wino[("frag", m)]

    def __init__(self, w, scale, shift, k, stride, pad, act, dev, wdtype=torch.float32):
        self.w = _khwc(w).to(dev).to(wdtype)
        cpc = 64 if wdtype == torch.bfloat16 else 32
        self.w_ring = None
        self.ring_folded = False
        if k == 3 and stride == 1 and pad == 1 and w.shape[1] % cpc == 0:
            wr = w
            if wdtype == torch.bfloat16 and scale is not None:
                # bf16 model: the folded-BN scale of an output channel goes INTO its window-kernel weight row (fp32 product,
                # ONE rounding to bf16), the kernels then get scale = NULL: their epilogue has no multiply and the
                # weight-stationary / one-wave kernels take the shift as the accumulator's initial value.  (fp32 model:
                # weights untouched, scale applied in fp32 — rounding within 1 ulp of conv -> BN)
                wr = w * scale.reshape(-1, 1, 1, 1).to(w.dtype)
                self.ring_folded = True
            self.w_ring = _ring_w(wr, cpc).to(dev).to(wdtype)
        self.w_s2 = None
        if k == 3 and stride == 2 and pad == 1 and wdtype == torch.bfloat16 and w.shape[1] % 64 == 0 and w.shape[0] % 32 == 0:
            # bf16 model, stride-2 3x3 convs (layer2.0 / 3.0 / 4.0 conv1): plane-window kernel (csrc/conv3x3_s2.hip); the folded-BN
            # scale goes into the weight rows like on the stride-1 window kernels (fp32 product, one rounding to bf16)
            ws = w if scale is None else w * scale.reshape(-1, 1, 1, 1).to(w.dtype)
            self.w_s2 = _s2_w(torch.as_tensor(ws).float()).to(dev).to(wdtype)
        self.scale = None if scale is None else scale.contiguous().to(dev)
        self.shift = None if shift is None else shift.contiguous().to(dev)
        self.cout, self.cin = w.shape[0], w.shape[1]
        self.k, self.stride, self.pad, self.act = k, stride, pad, act
        self.w_wino = None
        self.w_wino_c64 = None
        if (_winograd_min_c() and os.environ.get("CADRE_WINOGRAD_C64", "1") != "0" and wdtype == torch.float32
                and k == 3 and stride == 1 and pad == 1 and tuple(w.shape[:2]) == (64, 64)):
            # fused F(2x2) kernel of the 64 -> 64 stage (csrc/winograd_c64.hip): 1.66 / 1.75 vs 2.97 ms per 1024 frames of 72 x 72
            self.w_wino_c64 = _winograd_u_c64(torch.as_tensor(w).float()).to(dev)
        wmin = _winograd_min_c()
        if (wmin and wdtype == torch.float32 and k == 3 and stride == 1 and pad == 1 and w.shape[1] >= wmin
                and w.shape[0] >= 128 and w.shape[1] % 4 == 0 and w.shape[0] % 4 == 0):
            self.w_wino = {}                       # {m: U} filled on first use (the tile edge depends on the map size)
            self._w_oihw = torch.as_tensor(w).float().cpu()


class DANetEncoderHIP:
    """Frozen encoder; `latent(rgb_u8, route_u8)` -> [F,512] features on device."""

    def __init__(self, state_dict, H, W, device="cuda:0", max_frames=64, dtype="f32"):
        """dtype "f32": everything fp32 (BASELINE C2).  dtype "bf16" (BASELINE C3): activations and
        conv / inter-task weights in bf16 from the max-pool on, fp32 MFMA accumulation and fp32
        epilogues, fp32 stem (Cin=4), fp32 attention math (PAM/CAM/inter-task softmax)."""
        if dtype not in ("f32", "bf16"):
            raise ValueError("dtype must be 'f32' or 'bf16'")
        self.bf16 = dtype == "bf16"
        self.dtype = dtype
        wd = torch.bfloat16 if self.bf16 else torch.float32
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise hip.CadreHipError("DANetEncoderHIP needs a HIP device (device_num/vae_device >= 0); no CPU path")
        hip.lib()
        self.H, self.W = H, W
        self.max_frames = max_frames
        sd = {k: (v if isinstance(v, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(v))).float().cpu()
              for k, v in state_dict.items() if not k.endswith("num_batches_tracked")}
        dev = self.device
        fh, fw = synth.feat_hw(H, W)
        self.fh, self.fw, self.Np = fh, fw, fh * fw
        # identity of the loaded checkpoint (CadreAgent.ensemble_act shares one pass between equal encoders)
        self.fingerprint = (dtype, H, W) + tuple(float(sd[k].sum(dtype=torch.float64)) for k in sorted(sd))
        if self.Np > 1024:
            raise hip.CadreHipError("layer-4 map %dx%d > 1024 positions not supported by the PAM/CAM kernels" % (fh, fw))
        need = 512 * self.Np
        got = sd["inter_task_att.visual_query_layer.1.weight"].shape[1]
        if got != need:
            raise hip.CadreHipError("encoder weights are for a %d-wide inter-task input, this %dx%d input needs %d "
                                    "(reference hard-codes 5x8: intertask_att.py:17-18)" % (got, H, W, need))
        # ---- trunk (resnet.py:111-115, 152-166)
        sc, sh = _fold_bn(sd, "backbone.bn1", sd["backbone.conv1.bias"])
        self.stem = _Conv(sd["backbone.conv1.weight"], sc, sh, 7, 2, 3, 1, dev)
        # A/B build only (csrc/ab/conv3x3_c64_bf16.hip): 0 off, 1 fallback when the window kernel declines, 2 preferred
        self.c64_kernel = int(os.environ.get("CADRE_C64_KERNEL", "1")) if hip.has_ab_kernels() else 0
        self.ring_conv = os.environ.get("CADRE_RING_CONV", "1") != "0"
        # fused front (pack -> LUT -> stem conv + BN + ReLU -> max-pool in one kernel, stem_pool.hip)
        self.fused_stem = bool(hip.lib().cadre_stem_pool_supported(H, W)) and os.environ.get("CADRE_FUSED_STEM", "1") != "0"
        self.stem_x3 = False
        if self.fused_stem:
            w1 = torch.as_tensor(sd["backbone.conv1.weight"]).float()
            if self.bf16:                            # BN scale folded into the bf16 weights (the kernel starts its sums at the shift)
                w1 = w1 * self.stem.scale.detach().cpu().float().view(-1, 1, 1, 1)
            wt = _stem_taps(w1, 56 if self.bf16 else 50, row8=self.bf16)
            self.stem_taps = wt.to(dev).to(torch.bfloat16 if self.bf16 else torch.float32)
            # opt-in (CADRE_STEM_EXACT_BF16=1, fp32 model): the front on the bf16 matrix cores with exact products — three bf16 pieces
            # of every fp32 weight, pixel bytes exact in bf16, fp32 sums and fp32 epilogue (stem_pool.hip X3).  Reported by bench.py as
            # its own section; the default and the headline stay on v_mfma_f32.
            self.stem_x3 = (not self.bf16) and os.environ.get("CADRE_STEM_EXACT_BF16", "0") == "1"
            if self.stem_x3:
                self.stem_taps_x3 = _stem_taps_x3(w1).to(dev)
        if self.bf16:
            # bf16 stem on a zero-padded NHWC4 image (3 px halo; row pitch padded so every 8-pixel tap
            # row is in-bounds and 16-B aligned): no halo masks, two kernel rows per 64-deep k-tile
            self.stem_w16 = _stem_rows_bf16(sd["backbone.conv1.weight"]).to(dev).to(torch.bfloat16)
            Ho, Wo = (H + 6 - 7) // 2 + 1, (W + 6 - 7) // 2 + 1
            self.Hp = max(H + 6, (Ho - 1) * 2 + 8)
            self.Wp = max(W + 6, (Wo - 1) * 2 + 8)
            self.Wp += self.Wp & 1
        self.blocks = []
        self.s1x = {}                                   # block index -> (fused conv2 + shortcut weights, summed shifts)
        self.use_s1x = os.environ.get("CADRE_S1X_CONV", "1") != "0"
        for li in range(1, 5):
            for bi in range(2):
                pre = "backbone.layer%d.%d" % (li, bi)
                stride = 2 if (li > 1 and bi == 0) else 1
                c1 = _Conv(sd[pre + ".conv1.weight"], *_fold_bn(sd, pre + ".bn1"), 3, stride, 1, 1, dev, wd)
                c2 = _Conv(sd[pre + ".conv2.weight"], *_fold_bn(sd, pre + ".bn2"), 3, 1, 1, 1, dev, wd)
                down = None
                if (pre + ".downsample.0.weight") in sd:
                    down = _Conv(sd[pre + ".downsample.0.weight"], *_fold_bn(sd, pre + ".downsample.1"), 1, stride, 0, 0, dev, wd)
                self.blocks.append((c1, c2, down))
                if down is not None and self.bf16 and c2.cin % 64 == 0 and down.cin % 64 == 0 and down.cin <= c2.cin:
                    # bf16 model: the block's shortcut (1x1 / s2 conv + BN) rides as extra k-tiles of conv2 (csrc/conv3x3_s1x.hip):
                    # both folded-BN scales go into the weight rows (fp32 product, one rounding to bf16), the shifts add up
                    w2 = sd[pre + ".conv2.weight"].float() * c2.scale.detach().cpu().float().view(-1, 1, 1, 1)
                    wdn = sd[pre + ".downsample.0.weight"].float() * down.scale.detach().cpu().float().view(-1, 1, 1, 1)
                    self.s1x[len(self.blocks) - 1] = (_s1x_w(w2, wdn).to(dev).to(wd), (c2.shift + down.shift).contiguous())
        # ---- DANet head (danet.py:21-41)
        hd = "da_head."
        self.conv5a = _Conv(sd[hd + "conv5a.0.weight"], *_fold_bn(sd, hd + "conv5a.1"), 3, 1, 1, 1, dev, wd)
        self.conv5c = _Conv(sd[hd + "conv5c.0.weight"], *_fold_bn(sd, hd + "conv5c.1"), 3, 1, 1, 1, dev, wd)
        self.conv51 = _Conv(sd[hd + "conv51.0.weight"], *_fold_bn(sd, hd + "conv51.1"), 3, 1, 1, 1, dev, wd)
        self.conv52 = _Conv(sd[hd + "conv52.0.weight"], *_fold_bn(sd, hd + "conv52.1"), 3, 1, 1, 1, dev, wd)
        self.pam_w = torch.cat([sd[hd + "sa.%s_conv.weight" % n].reshape(-1, 128) for n in ("query", "key", "value")]).contiguous().to(dev)
        self.pam_b = torch.cat([sd[hd + "sa.%s_conv.bias" % n] for n in ("query", "key", "value")]).contiguous().to(dev)
        self.pam_gamma = float(sd[hd + "sa.gamma"].item())
        self.cam_gamma = float(sd[hd + "sc.gamma"].item())
        self.conv8 = _Conv(sd[hd + "conv8.1.weight"], None, sd[hd + "conv8.1.bias"], 1, 1, 0, 0, dev, wd)
        self.visual_conv = _Conv(sd["visual_conv.weight"], None, sd["visual_conv.bias"], 1, 1, 0, 0, dev, wd)
        self.bc_conv = _Conv(sd["bc_conv.weight"], None, sd["bc_conv.bias"], 1, 1, 0, 0, dev, wd)
        # ---- inter-task attention MLPs (intertask_att.py:39-80); order q,k,v per branch
        Np = self.Np
        self.ita_w1, self.ita_b1, self.ita_w1f = [], [], []
        w2, b2 = [], []
        for br in ("visual", "bc"):
            ws, bs = [], []
            for role in ("query", "key", "value"):
                pre = "inter_task_att.%s_%s_layer" % (br, role)
                w = sd[pre + ".1.weight"].view(512, 512, Np).permute(0, 2, 1).reshape(512, Np * 512)
                ws.append(w)
                bs.append(sd[pre + ".1.bias"])
                w2.append(sd[pre + ".3.weight"])
                b2.append(sd[pre + ".3.bias"])
            self.ita_w1.append(torch.cat(ws).contiguous().to(dev).to(wd))  # [1536][Np*512]
            # (the same matrix in fragment order for cadre_gemm_bf16_w128 — bf16 model, frame batches > 64 — is built on the first
            #  such call: _ita_frag; 127 MB per branch at 288 x 288 that an act()-only agent never needs: ADVICE r5)
            self.ita_w1f.append(None)
            self.ita_b1.append(torch.cat(bs).contiguous().to(dev))
        self.ita_w2 = torch.stack(w2).contiguous().to(dev)                # [6][256][512]
        self.ita_b2 = torch.stack(b2).contiguous().to(dev)                # [6][256]
        self.temperature = 256 ** 0.5                                     # intertask_att.py:29
        self.lut255 = torch.from_numpy((np.arange(256) / 255.).astype(np.float32)).to(dev)   # agent.py:46
        if self.fused_stem and not self.bf16:
            # the fused front converts bytes arithmetically (x * fl(1/255) + one correction) instead of through the table:
            # bit-exactness with float32(i / 255.) depends on the compiler keeping that sequence — checked once per
            # encoder on the device, the table path takes over if a single byte differs
            bad = torch.zeros(1, dtype=torch.int32, device=dev)
            hip.check(hip.lib().cadre_div255_selfcheck(hip.ptr(self.lut255), hip.ptr(bad), hip.stream()), "cadre_div255_selfcheck")
            if int(bad.item()) != 0:
                self.fused_stem = False
        self._ws = {}
        self._ws_flat = {}
        self._pass_frames = 0
        self.ws_generation = 0
        self.n_weights = sum(t.numel() for t in self._all_weight_tensors())

    def _ita_frag(self, b):
        """Inter-task first-layer matrix of branch b in the fragment order of cadre_gemm_bf16_w128, built on first use (on the device,
        from the bf16 matrix: a permutation, the same bits)."""
        if self.ita_w1f[b] is None:
            self.ita_w1f[b] = _w128_dense_b(self.ita_w1[b]).contiguous()
        return self.ita_w1f[b]

    def _all_weight_tensors(self):
        convs = [self.stem, self.conv5a, self.conv5c, self.conv51, self.conv52, self.conv8, self.visual_conv, self.bc_conv]
        for c1, c2, d in self.blocks:
            convs += [c1, c2] + ([d] if d is not None else [])
        ts = [self.pam_w, self.pam_b, self.ita_w2, self.ita_b2] + self.ita_w1 + self.ita_b1
        for c in convs:
            ts += [t for t in (c.w, c.scale, c.shift) if t is not None]
        return ts

    # ------------------------------------------------------------------ workspace
    def _buf(self, key, shape, dtype=torch.float32, zero=False):
        """Workspace tensor `key` of the given shape.  Slots are kept per use: the two most recent SMALL shapes (<= 64 frames:
        act() alternates between 1-frame and 8-frame passes, and hipGraphs captured over these tensors hold their addresses)
        and the two most recent LARGE ones (a learner's full chunk and its remainder chunk) — an agent that acts and learns
        on one encoder no longer evicts its act() tensors every round.  Replacing a small slot bumps
        `ws_generation`: holders of captured graphs re-capture when it moved."""
        small = self._pass_frames <= 64                      # (set by preprocess / forward_nhwc from the pass's frame count)
        slot = self._ws.setdefault((key, small), [])
        for i, t in enumerate(slot):
            if t.shape == torch.Size(shape) and t.dtype == dtype:
                if i:
                    slot.insert(0, slot.pop(i))
                return t
        t = (torch.zeros if zero else torch.empty)(shape, dtype=dtype, device=self.device)
        slot.insert(0, t)
        if len(slot) > 2:
            slot.pop()
            if small:
                self.ws_generation += 1
        return t

    def _flat(self, key, n):
        """First n elements of a flat fp32 workspace that only grows (layers of different shapes share it, one after another)."""
        t = self._ws_flat.get(key)
        if t is None or t.numel() < n:
            t = torch.empty(n, dtype=torch.float32, device=self.device)
            self._ws_flat[key] = t
            self.ws_generation += 1
        return t[:n]

    # ------------------------------------------------------------------ layers
    def _conv(self, c, x, F, H, W, key, resid=None, act=None, out_f32=False, reuse_v=False):
        Ho = (H + 2 * c.pad - c.k) // c.stride + 1
        Wo = (W + 2 * c.pad - c.k) // c.stride + 1
        odt = torch.bfloat16 if (self.bf16 and not out_f32) else torch.float32
        out = self._buf(key, (F, Ho, Wo, c.cout), odt)
        K = c.k * 32 if c.cin == 4 else c.k * c.k * c.cin
        M = F * Ho * Wo
        act = c.act if act is None else act
        wbf = c.w.dtype == torch.bfloat16
        flags = (2 if odt == torch.bfloat16 else 0) | (4 if (resid is not None and resid.dtype == torch.bfloat16) else 0)
        use_c64 = (wbf and self.c64_kernel and c.k == 3 and c.stride == 1 and c.pad == 1 and c.cin == 64 and c.cout == 64
                   and odt == torch.bfloat16 and (act & 16) == 0 and (resid is None or resid.dtype == torch.bfloat16)
                   and M * 128 < 2 ** 31)
        ring_flags = ((1 if wbf else 0) | (2 if odt == torch.bfloat16 else 0)
                      | (0 if resid is None else (8 | (4 if resid.dtype == torch.bfloat16 else 0))))
        use_ring = (c.w_ring is not None and self.ring_conv and x.dtype == c.w_ring.dtype and (act & 15) <= 1
                    and bool(hip.lib().cadre_conv3x3_ring_supported(F, H, W, c.cin, c.cout, ring_flags)))
        if (c.w_wino_c64 is not None and x.dtype == torch.float32 and odt == torch.float32 and act in (0, 1)
                and F * H * W * 256 < 2 ** 31 and (resid is None or resid.dtype == torch.float32)):
            # fused Winograd F(2x2, 3x3) of the 64 -> 64 stage: transforms and the 16 plane products in one kernel
            hip.winograd_c64(x, c.w_wino_c64, c.scale, c.shift, resid, out, F, H, W, act)
        elif c.w_wino is not None and x.dtype == torch.float32 and odt == torch.float32 and (act & 15) <= 1:
            # Winograd F(2x2, 3x3): input transform -> one batched GEMM over the 16 transform planes -> inverse transform + BN + residual + ReLU
            m = _winograd_m(H, W)
            L = hip.lib()
            if L.cadre_winograd_fused_supported(F, H, W, c.cin, c.cout, m):
                # round 6: the plane products and the inverse transform in one kernel (csrc/winograd_fused.hip): V goes out in
                # MFMA-fragment order, the (m+2)^2 product planes never reach HBM
                V = self._flat("wino_v", int(L.cadre_winograd_frag_elems(F, H, W, c.cin, m)))
                hip.winograd_fused(x, V, c.wino_u_frag(m, self.device), c.scale, c.shift, resid, out, F, H, W, c.cin, c.cout, act, m)
                return out, Ho, Wo
            P, T = (m + 2) ** 2, F * -(-H // m) * -(-W // m)
            V = self._flat("wino_v", P * T * c.cin).view(P, T, c.cin)
            Mx = self._flat("wino_m", P * T * c.cout).view(P, T, c.cout)
            # (reuse_v: the caller's previous conv transformed this very input — the head's conv5a / conv5c both read layer4 — and
            #  nothing has written the transform-domain buffer since: one input transform for the two)
            if not reuse_v:
                hip.check(L.cadre_winograd_in(hip.ptr(x), hip.ptr(V), F, H, W, c.cin, m, hip.stream()), "cadre_winograd_in")
            hip.gemm(V, c.wino_u(m, self.device), Mx, T, c.cout, c.cin, c.cin, c.cin, c.cout, batch=P,
                     a_z=(1, P, T * c.cin), b_z=(1, P, c.cout * c.cin), c_z=(1, P, T * c.cout))
            hip.check(L.cadre_winograd_out(hip.ptr(Mx), hip.ptr(c.scale), hip.ptr(c.shift), hip.ptr(resid), hip.ptr(out),
                                           F, H, W, c.cout, act, m, hip.stream()), "cadre_winograd_out")
        elif use_c64 and (self.c64_kernel == 2 or not use_ring):
            # stage-1 convs of the bf16 encoder with the weights resident in LDS (bit-identical to cadre_gemm_bf16);
            # the window kernel below measures 3 % faster on them and takes precedence unless CADRE_C64_KERNEL=2
            hip.conv3x3_c64_bf16(x, c.w, c.scale, c.shift, resid, out, F, H, W, 1 if act == 1 else 0)
        elif use_ring:
            # stride-1 3x3 convs: each pixel through LDS once per channel chunk, weights streamed (conv3x3_ring.hip)
            hip.conv3x3_ring(x, c.w_ring, None if c.ring_folded else c.scale, c.shift, resid, out, F, H, W, c.cin, c.cout, act)
        elif (c.w_s2 is not None and x.dtype == torch.bfloat16 and odt == torch.bfloat16 and resid is None and act in (0, 1)
              and bool(hip.lib().cadre_conv3x3_s2_supported(F, H, W, c.cin, c.cout))):
            # stride-2 3x3 convs of the bf16 model: four parity-plane windows in LDS, each pixel once per channel chunk
            hip.conv3x3_s2(x, c.w_s2, None, c.shift, out, F, H, W, c.cin, c.cout, act)
        elif c.k == 1 and c.stride == 1:
            hip.gemm(x, c.w, out, M, c.cout, K, K, K, c.cout, scale=c.scale, shift=c.shift, resid=resid,
                     ldr=c.cout, act=act, bf16=wbf, flags=flags)
        else:
            hip.gemm(x, c.w, out, M, c.cout, K, 0, K, c.cout, a_mode=3 if c.cin == 4 else 2, scale=c.scale,
                     shift=c.shift, resid=resid, ldr=c.cout, act=act,
                     conv=(H, W, c.cin, Ho, Wo, c.k, c.k, c.stride, c.pad), bf16=wbf, flags=flags)
        return out, Ho, Wo

    def preprocess(self, rgb_d, route_d, route_norm_d=None, frame_idx=None):
        """agent.py:43-75 on device: u8 [F,H,W,3] + u8 [F,W,H] -> f32 NHWC [F,H,W,4] (or the packed u8 image of the
        fused front).  frame_idx (i64 device tensor): output frame f is source frame frame_idx[f] (sliding windows)."""
        if frame_idx is not None and not self.fused_stem:
            rgb_d, route_d, frame_idx = rgb_d.index_select(0, frame_idx), route_d.index_select(0, frame_idx), None
        F = rgb_d.shape[0] if frame_idx is None else frame_idx.numel()
        self._pass_frames = int(F)
        n_src = int(rgb_d.shape[0])
        fmax = self._buf("fmax", (max(F, n_src),), torch.int32)
        L = hip.lib()
        if self.fused_stem:  # packed u8 pixels (one dword each): the LUT conversion happens inside the stem kernel
            x = self._buf("packed", (F, self.H, self.W), torch.int32)
            hip.check(L.cadre_pack_obs(hip.ptr(rgb_d), hip.ptr(route_d), hip.ptr(x), hip.ptr(route_norm_d), hip.ptr(fmax),
                                       F, self.H, self.W, hip.ptr(frame_idx), n_src, hip.stream()), "cadre_pack_obs")
            return x
        if self.bf16:       # zero-bordered bf16 image for the bf16 stem; the border is written never
            x = self._buf("pre_pad", (F, self.Hp, self.Wp, 4), torch.bfloat16, zero=True)
            hip.check(L.cadre_preprocess_bf16pad(hip.ptr(rgb_d), hip.ptr(route_d), hip.ptr(self.lut255), hip.ptr(x),
                                                 hip.ptr(route_norm_d), hip.ptr(fmax), F, self.H, self.W, self.Hp,
                                                 self.Wp, 3, 3, hip.stream()), "cadre_preprocess_bf16pad")
            return x
        x = self._buf("pre", (F, self.H, self.W, 4))
        hip.check(L.cadre_preprocess(hip.ptr(rgb_d), hip.ptr(route_d), hip.ptr(self.lut255), hip.ptr(x),
                                     hip.ptr(route_norm_d), hip.ptr(fmax), F, self.H, self.W, hip.stream()),
                  "cadre_preprocess")
        return x

    def forward_nhwc(self, x, out=None, ldo=512, taps=None):
        """x f32 NHWC [F,H,W,4] (device) -> latent written into out[:, :512] (row pitch ldo)."""
        L = hip.lib()
        st = hip.stream()
        F = x.shape[0]
        self._pass_frames = int(F)
        H, W = self.H, self.W
        if x.dtype == torch.int32:              # packed observation from preprocess(): fused front
            if not self.fused_stem or tuple(x.shape[1:]) != (H, W):
                raise hip.CadreHipError("packed observation does not match this encoder (fused front off or wrong size)")
            Hs, Ws = (H + 6 - 7) // 2 + 1, (W + 6 - 7) // 2 + 1
            Hp, Wp = (Hs + 2 - 3) // 2 + 1, (Ws + 2 - 3) // 2 + 1
            p = self._buf("pool", (F, Hp, Wp, 64), torch.bfloat16 if self.bf16 else torch.float32)
            x3 = self.stem_x3
            hip.check(L.cadre_stem_pool(hip.ptr(x), hip.ptr(self.stem_taps_x3 if x3 else self.stem_taps),
                                        None if self.bf16 else hip.ptr(self.stem.scale), hip.ptr(self.stem.shift),
                                        hip.ptr(p), F, H, W, 1 if self.bf16 else (2 if x3 else 0),
                                        Hp * Wp * 64, Wp * 64, 64, 0, st), "cadre_stem_pool")
            return self._trunk(p, F, Hp, Wp, out, ldo, taps)
        if self.bf16:
            if x.dtype != torch.bfloat16 or tuple(x.shape[1:]) != (self.Hp, self.Wp, 4):
                raise hip.CadreHipError("bf16 encoder expects the padded bf16 image from preprocess()")
            H, W = (H + 6 - 7) // 2 + 1, (W + 6 - 7) // 2 + 1
            y = self._buf("stem", (F, H, W, 64), torch.bfloat16)
            hip.gemm(x, self.stem_w16, y, F * H * W, 64, self.stem_w16.shape[1], 0, self.stem_w16.shape[1], 64, a_mode=4,
                     scale=self.stem.scale, shift=self.stem.shift, act=1, conv=(self.Hp, self.Wp, 4, H, W, 7, 7, 2, 0),
                     bf16=True, flags=2)
        else:
            y, H, W = self._conv(self.stem, x, F, H, W, "stem")
        Hp, Wp = (H + 2 - 3) // 2 + 1, (W + 2 - 3) // 2 + 1
        p = self._buf("pool", (F, Hp, Wp, 64), y.dtype)
        if self.bf16:
            hip.check(L.cadre_maxpool3x3s2_bf16(hip.ptr(y), hip.ptr(p), F, H, W, 64, st), "cadre_maxpool3x3s2_bf16")
        else:
            hip.check(L.cadre_maxpool3x3s2(hip.ptr(y), hip.ptr(p), F, H, W, 64, st), "cadre_maxpool3x3s2")
        return self._trunk(p, F, Hp, Wp, out, ldo, taps)

    def _trunk(self, p, F, H, W, out, ldo, taps):
        """layer1..layer4, DANet head and inter-task attention on the pooled stem map p [F,H,W,64]."""
        L = hip.lib()
        st = hip.stream()
        cur = p
        if taps is not None:
            taps["pool"] = p
        for i, (c1, c2, down) in enumerate(self.blocks):                      # resnet.py:40-55
            t, H2, W2 = self._conv(c1, cur, F, H, W, "b%d_t" % i)
            if (i in self.s1x and self.use_s1x and H == 2 * H2 and W == 2 * W2 and t.dtype == torch.bfloat16
                    and bool(L.cadre_conv3x3_s1x_supported(F, H2, W2, c2.cin, down.cin, c2.cout))):
                # down-sampling block of the bf16 model: relu(bn2(conv2(t)) + bn_d(conv_d(cur))) as ONE accumulation — the
                # shortcut's 1 / 2 / 4 k-tiles ride behind conv2's, no shortcut tensor is written or read back
                w_f, sh_f = self.s1x[i]
                blk_out = self._buf("b%d_o" % i, (F, H2, W2, c2.cout), torch.bfloat16)
                hip.conv3x3_s1x(t, cur, w_f, sh_f, blk_out, F, H2, W2, c2.cin, down.cin, c2.cout, 1)
                cur, H, W = blk_out, H2, W2
                continue
            idt = cur
            if down is not None:
                idt, _, _ = self._conv(down, cur, F, H, W, "b%d_d" % i)
            cur, H, W = self._conv(c2, t, F, H2, W2, "b%d_o" % i, resid=idt)
        l4 = cur
        Np = H * W
        assert Np == self.Np
        # ---- da_head (danet.py:43-69)
        adt = torch.bfloat16 if self.bf16 else torch.float32
        pam_fn = L.cadre_pam_bf16out if self.bf16 else L.cadre_pam      # attention math stays fp32
        cam_fn = L.cadre_cam_bf16out if self.bf16 else L.cadre_cam
        f1, _, _ = self._conv(self.conv5a, l4, F, H, W, "f1", out_f32=True)
        # conv5c reads layer4 too: run it right behind conv5a, on conv5a's input transform when both took the three-launch Winograd path
        # with the same tile edge (the same V bits: the results do not change)
        share_v = (self.conv5a.w_wino is not None and self.conv5c.w_wino is not None and l4.dtype == torch.float32
                   and self.conv5a.cin == self.conv5c.cin and not L.cadre_winograd_fused_supported(F, H, W, self.conv5a.cin, self.conv5a.cout, _winograd_m(H, W))
                   and not L.cadre_winograd_fused_supported(F, H, W, self.conv5c.cin, self.conv5c.cout, _winograd_m(H, W))
                   and os.environ.get("CADRE_HEAD_SHARE_V", "1") != "0")
        f2, _, _ = self._conv(self.conv5c, l4, F, H, W, "f2", out_f32=True, reuse_v=share_v)
        qkv = self._buf("pam_qkv", (F * Np, 160))
        hip.gemm(f1, self.pam_w, qkv, F * Np, 160, 128, 128, 128, 160, shift=self.pam_b)
        sa = self._buf("sa", (F, H, W, 128), adt)
        hip.check(pam_fn(hip.ptr(f1), hip.ptr(qkv), self.pam_gamma, hip.ptr(sa), F, Np, st), "cadre_pam")
        sa_conv, _, _ = self._conv(self.conv51, sa, F, H, W, "sa_conv")
        sc = self._buf("sc", (F, H, W, 128), adt)
        hip.check(cam_fn(hip.ptr(f2), self.cam_gamma, hip.ptr(sc), F, Np, st), "cadre_cam")
        # feat_sum = sa_conv + sc_conv (danet.py:57) fused into conv52's epilogue as a residual added
        # AFTER its ReLU (act|16): relu(bn(conv52(sc))) + sa_conv
        feat_sum, _, _ = self._conv(self.conv52, sc, F, H, W, "feat_sum", resid=sa_conv, act=1 | 16)
        da, _, _ = self._conv(self.conv8, feat_sum, F, H, W, "da")
        vis, _, _ = self._conv(self.visual_conv, da, F, H, W, "vis")
        bc, _, _ = self._conv(self.bc_conv, da, F, H, W, "bc")
        if taps is not None:
            taps.update(layer4=l4, da=da, vis=vis, bc=bc)
        # ---- inter-task attention (intertask_att.py:121-176)
        Kin = Np * 512
        hid = self._buf("ita_hid", (F, 3072))
        # split-K chosen from K alone: every output element is then summed in the same order for any
        # frame batch, so per-frame results are bit-identical across batch sizes (latent cache, §8f-1)
        # (K per slice ~ 2592: at 288 x 288 (Kin = 41472) 16 slices — 48 tiles of 256 x 256 x 16 = 768 workgroups = three full rounds
        #  of the 256 CUs (bf16; fp32: 96 tiles of 128 x 128 x 16 = 1536 = three rounds at two per CU) and a fifth less fp32 slab
        #  traffic than the 20 slices of rounds 1-4, which left 960 / 1920 workgroups = 3.75 rounds)
        split = int(max(1, min(32, (Kin + 1296) // 2592)))
        for b, src in enumerate((vis, bc)):
            if split > 1:
                slabs = self._buf("ita_slab", (split, F, 1536))
                if self.bf16 and F > 64 and L.cadre_gemm_bf16_w128_supported(F, 1536, Kin, Kin, 1536, split):
                    # 256 x 256 tiles, weights streamed in fragment order: the same slices and k order, bit-identical partial sums
                    hip.gemm_bf16_w128(src, self._ita_frag(b), slabs, F, 1536, Kin, Kin, 1536, split)
                else:
                    hip.gemm(src, self.ita_w1[b], slabs, F, 1536, Kin, Kin, Kin, 1536, split_k=split,
                             tile=3 if F <= 64 else 0, bf16=self.bf16)
                hip.check(L.cadre_splitk_reduce(hip.ptr(slabs), split, F * 1536, 1536, hid.data_ptr() + 4 * 1536 * b,
                                                3072, F, 1536, None, hip.ptr(self.ita_b1[b]), 2, 0.01, None, 0, st),
                          "cadre_splitk_reduce")
            else:
                hip.gemm(src, self.ita_w1[b], hid[:, 1536 * b:], F, 1536, Kin, Kin, Kin, 3072, shift=self.ita_b1[b],
                         act=2, slope=0.01, bf16=self.bf16)
        qkv2 = self._buf("ita_qkv", (F, 6, 256))
        hip.gemm(hid, self.ita_w2, qkv2, F, 256, 512, 3072, 512, 1536, shift=self.ita_b2, batch=6,
                 a_z=(1, 0, 512), b_z=(1, 0, 256 * 512), c_z=(1, 0, 256), s_z=(1, 0, 256))
        if out is None:
            out = torch.zeros(F, ldo, device=self.device)
        hip.check(L.cadre_intertask_att(hip.ptr(qkv2), hip.ptr(out), out.stride(0), F, self.temperature, st),
                  "cadre_intertask_att")
        return out

    def latent(self, rgb_d, route_d, out=None, route_norm_d=None, taps=None):
        """Frames are processed in chunks of `max_frames` (workspace sized once)."""
        F = rgb_d.shape[0]
        if out is None:
            out = torch.zeros(F, 512, device=self.device)
        for s in range(0, F, self.max_frames):
            e = min(F, s + self.max_frames)
            rn = None if route_norm_d is None else route_norm_d[s:e]
            x = self.preprocess(rgb_d[s:e], route_d[s:e], rn)
            self.forward_nhwc(x, out[s:e], taps=taps)
        return out

    # ------------------------------------------------------------------ accounting
    def algorithmic_bytes(self, frames):
        """Layer-wise HBM model of SURVEY.md §8(d): every fused conv / attention op reads its input once
        and writes its output once per frame, residuals are re-read, weights are read once per batch."""
        e = 2 if self.bf16 else 4
        H, W = self.H, self.W
        act = 0

        def conv(c, H, W, resid=False):
            Ho = (H + 2 * c.pad - c.k) // c.stride + 1
            Wo = (W + 2 * c.pad - c.k) // c.stride + 1
            return H * W * c.cin + Ho * Wo * c.cout * (2 if resid else 1), Ho, Wo
        a, H, W = conv(self.stem, H, W); act += a
        Hp, Wp = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        act += H * W * 64 + Hp * Wp * 64
        H, W = Hp, Wp
        for c1, c2, d in self.blocks:
            a, H2, W2 = conv(c1, H, W); act += a
            if d is not None:
                act += conv(d, H, W)[0]
            a, H, W = conv(c2, H2, W2, resid=True); act += a
        Np = H * W
        for c in (self.conv5a, self.conv5c, self.conv51, self.conv52, self.conv8, self.visual_conv, self.bc_conv):
            act += conv(c, H, W)[0]
        act += Np * (128 + 160) + 2 * Np * (160 + 128 + 128) + Np * 128 * 2      # PAM qkv + PAM/CAM in/out
        act += 2 * Np * 512 + 3072 + 6 * 256 + 512                               # inter-task inputs/outputs
        return e * act * frames + e * self.n_weights

    def winograd_convs(self):
        """Number of conv layers that run as Winograd F(2x2, 3x3) (0 for the bf16 model and under CADRE_WINOGRAD=0)."""
        cs = [c for blk in self.blocks for c in blk[:2]] + [self.conv5a, self.conv5c, self.conv51, self.conv52]
        return sum(c.w_wino is not None or c.w_wino_c64 is not None for c in cs)

    def flops_per_frame(self, executed=False):
        """Direct-convolution (algorithmic) FLOPs of one frame; executed=True: what the launches multiply — differs only
        under CADRE_WINOGRAD=1 (16 planes x ceil(H/2) x ceil(W/2) tiles x Cin x Cout per Winograd conv)."""
        H, W = self.H, self.W
        total = 0
        def conv(c, H, W):
            Ho = (H + 2 * c.pad - c.k) // c.stride + 1
            Wo = (W + 2 * c.pad - c.k) // c.stride + 1
            if executed and c.w_wino_c64 is not None:
                return 2 * 16 * -(-H // 2) * -(-W // 2) * c.cout * c.cin, Ho, Wo
            if executed and c.w_wino is not None:
                m = _winograd_m(H, W)
                return 2 * (m + 2) ** 2 * -(-H // m) * -(-W // m) * c.cout * c.cin, Ho, Wo
            return 2 * Ho * Wo * c.cout * c.cin * c.k * c.k, Ho, Wo
        f, H, W = conv(self.stem, H, W); total += f
        H, W = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        for c1, c2, d in self.blocks:
            f, H2, W2 = conv(c1, H, W); total += f
            if d is not None:
                total += conv(d, H, W)[0]
            f, H, W = conv(c2, H2, W2); total += f
        for c in (self.conv5a, self.conv5c, self.conv51, self.conv52, self.conv8, self.visual_conv, self.bc_conv):
            total += conv(c, H, W)[0]
        Np = H * W
        total += 2 * Np * 160 * 128 + 2 * (Np * Np * 16 + Np * Np * 128) + 2 * 2 * (128 * 128 * Np)
        total += 2 * (2 * 1536 * Np * 512 + 6 * 256 * 512)
        return total
