#!/usr/bin/env python3
"""Micro-benchmark of cadre_gemm_f32 on the encoder's conv shapes (288x288, F frames) and the
PPO-update GEMM shapes.  Prints TFLOP/s per shape (HIP-event timed, median of reps)."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cadre_amd import hip  # noqa: E402


def timeit(fn, reps=10, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e-3)
    ts.sort()
    return ts[len(ts) // 2]


def conv_case(F, H, W, Cin, Cout, k, s, p, resid, tile=0):
    Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    x = torch.randn(F, H, W, Cin, device="cuda")
    w = torch.randn(Cout, k * 32 if Cin == 4 else k * k * Cin, device="cuda") * 0.05
    sc = torch.rand(Cout, device="cuda") + 0.5
    sh = torch.randn(Cout, device="cuda")
    r = torch.randn(F, Ho, Wo, Cout, device="cuda") if resid else None
    out = torch.empty(F, Ho, Wo, Cout, device="cuda")
    K = k * 32 if Cin == 4 else k * k * Cin
    M = F * Ho * Wo

    def run():
        hip.gemm(x, w, out, M, Cout, K, 0, K, Cout, a_mode=3 if Cin == 4 else 2, scale=sc, shift=sh, resid=r,
                 ldr=Cout, act=1, conv=(H, W, Cin, Ho, Wo, k, k, s, p), tile=tile)
    t = timeit(run)
    return 2.0 * M * Cout * (k * k * Cin) / t / 1e12, t      # algorithmic FLOPs (stem: 196, not the padded 224)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=128)
    ap.add_argument("--tiles", default="0")
    args = ap.parse_args()
    F = args.frames
    hip.lib()
    cases = [("stem 7x7s2 4->64 @288", 288, 288, 4, 64, 7, 2, 3, False),
             ("layer1 3x3 64->64 @72", 72, 72, 64, 64, 3, 1, 1, False),
             ("layer1 3x3 64->64 @72 +res", 72, 72, 64, 64, 3, 1, 1, True),
             ("layer2.0 3x3s2 64->128", 72, 72, 64, 128, 3, 2, 1, False),
             ("layer2 3x3 128->128 @36 +res", 36, 36, 128, 128, 3, 1, 1, True),
             ("layer3 3x3 256->256 @18 +res", 18, 18, 256, 256, 3, 1, 1, True),
             ("layer4 3x3 512->512 @9 +res", 9, 9, 512, 512, 3, 1, 1, True),
             ("conv5a 3x3 512->128 @9", 9, 9, 512, 128, 3, 1, 1, False)]
    tiles = [int(x) for x in args.tiles.split(",")]
    for name, H, W, ci, co, k, s, p, res in cases:
        row = []
        for tl in tiles:
            if tl in (1, 4, 5, 8, 9) and co <= 64:
                row.append("   --  ")
                continue
            tf, t = conv_case(F, H, W, ci, co, k, s, p, res, tile=tl)
            row.append("%6.1f " % tf)
        print("%-34s F=%d  TFLOP/s by tile %s: %s" % (name, F, tiles, " ".join(row)), flush=True)
    for tl in tiles:
        M = N = K = 4096
        A = torch.randn(M, K, device="cuda"); B = torch.randn(N, K, device="cuda"); Cc = torch.empty(M, N, device="cuda")
        t = timeit(lambda: hip.gemm(A, B, Cc, M, N, K, K, K, N, tile=tl))
        print("dense 4096^3 NT tile %d: %7.2f TFLOP/s" % (tl, 2.0 * M * N * K / t / 1e12), flush=True)
    # dense GEMMs of the PPO update (batched over 8 nets)
    for name, M, N, K, Z in (("lstm x-proj [512,544]x[544,2120] x8", 512, 2120, 544, 8),
                             ("lstm step  [64,544]x[544,2120] x8", 64, 2120, 544, 8),
                             ("dW_hh^T    [2120,512]x[512,544] x8", 2120, 544, 512, 8)):
        A = torch.randn(Z, M, K, device="cuda"); B = torch.randn(Z, N, K, device="cuda"); Cc = torch.empty(Z, M, N, device="cuda")
        t = timeit(lambda: hip.gemm(A, B, Cc, M, N, K, K, K, N, batch=Z, a_z=(1, 0, M * K), b_z=(1, 0, N * K), c_z=(1, 0, M * N)))
        print("%-34s       %7.2f TFLOP/s  %8.1f us" % (name, 2.0 * M * N * K * Z / t / 1e12, t * 1e6), flush=True)
    # inter-task first layer, split-K 20
    Kin = 81 * 512
    A = torch.randn(F, Kin, device="cuda"); B = torch.randn(1536, Kin, device="cuda") * 0.01
    slabs = torch.empty(20, F, 1536, device="cuda")
    t = timeit(lambda: hip.gemm(A, B, slabs, F, 1536, Kin, Kin, Kin, 1536, split_k=20))
    print("%-34s F=%d  %7.2f TFLOP/s  %8.1f us  (weights %.2f TB/s)" % ("intertask L1 split-K20", F, 2.0 * F * 1536 * Kin / t / 1e12,
                                                                  t * 1e6, 1536 * Kin * 4 / t / 1e12), flush=True)


if __name__ == "__main__":
    main()
