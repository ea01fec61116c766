#!/usr/bin/env python3
"""Accuracy gate for bf16 Winograd F(2x2, 3x3) on the stride-1 convs of the bf16 trunk: fp32 transforms,
bf16 MFMA operands (V = B^T d B and U = G g G^T rounded to bf16), fp32 accumulation, fp32 inverse transform, bf16 output — emulated
on the CPU on the trunk's layer shapes and compared with the direct bf16 conv (bf16 operands, fp32 accumulation, bf16 output) against
a float64 conv of the same fp32 weights and bf16 activations.  The C3 feature bar (1.5e-2 of max, measured 8e-3 with direct convs)
leaves less than 2x headroom over 20 layers."""
import torch

G2 = torch.tensor([[1.0, 0.0, 0.0], [0.5, 0.5, 0.5], [0.5, -0.5, 0.5], [0.0, 0.0, 1.0]], dtype=torch.float64)
BT2 = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float32)
AT2 = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float32)


def bf(x):
    return x.to(torch.bfloat16).float()


def run(F, H, C, N, seed=0):
    g = torch.Generator().manual_seed(seed)
    x = bf(torch.relu(torch.randn(F, C, H, H, generator=g)))               # bf16 activations (what the previous layer stored)
    w = torch.randn(N, C, 3, 3, generator=g) / (9 * C) ** 0.5               # fp32 master weights (BN scale folded)
    ref = torch.nn.functional.conv2d(x.double(), w.double(), padding=1)
    s = float(ref.abs().max())
    direct = bf(torch.nn.functional.conv2d(x, bf(w), padding=1))            # bf16 operands, fp32 sums, bf16 result
    TH = -(-H // 2)
    xp = torch.zeros(F, C, 2 * TH + 2, 2 * TH + 2)
    xp[:, :, 1:H + 1, 1:H + 1] = x
    d = xp.unfold(2, 4, 2).unfold(3, 4, 2).contiguous()                     # [F][C][TH][TW][4][4]
    V = torch.einsum("ik,fcabkl,jl->fcabij", BT2, d, BT2)                   # fp32 transform of bf16 pixels: exact sums of 4
    U = torch.einsum("ik,ockl,jl->ocij", G2, w.double(), G2).float()
    out = {}
    for name, Vq, Uq in (("bf16 V, bf16 U", bf(V), bf(U)), ("fp32 V, bf16 U", V, bf(U)), ("bf16 V, fp32 U", bf(V), U)):
        M = torch.einsum("fcabij,ocij->foabij", Vq, Uq)
        Y = torch.einsum("ik,foabkl,jl->foabij", AT2, M, AT2)
        y = bf(Y.permute(0, 1, 2, 4, 3, 5).reshape(F, N, 2 * TH, 2 * TH)[:, :, :H, :H])
        e = y.double() - ref
        out[name] = (float(e.abs().max()) / s, float((e ** 2).mean().sqrt()) / s)
    e = direct.double() - ref
    out["direct"] = (float(e.abs().max()) / s, float((e ** 2).mean().sqrt()) / s)
    # the part of the error that is NOT the final bf16 rounding of the output (which both forms share)
    rnd = bf(ref.float()).double() - ref
    out["output rounding alone"] = (float(rnd.abs().max()) / s, float((rnd ** 2).mean().sqrt()) / s)
    return out


if __name__ == "__main__":
    for (F, H, C, N) in ((2, 36, 128, 128), (2, 18, 256, 256), (4, 9, 512, 512)):
        r = run(F, H, C, N)
        print("%dx%d %d->%d:" % (H, H, C, N))
        for k, (mx, rms) in r.items():
            print("    %-24s max %.2e  rms %.2e   (rms / direct rms = %.2f)" % (k, mx, rms, rms / r["direct"][1]))
