#!/usr/bin/env python3
"""The two large GEMMs of an update step (input projections of all 8 time steps; LSTM weight gradients) on every
tile that takes them: the auto choice (32x128 for the row-sorted projection, 64x64 for the k-segmented gradient) is the
fastest at both minibatch sizes."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cadre_amd import hip
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from skinny_gemm_bench import timeit
hip.lib()
Z, N, K = 8, 2120, 544
for B, S in ((64, 8), (256, 8)):
    q = B // 4
    seg = torch.tensor([[0, q], [q, q], [2 * q, q], [3 * q, q]] * 2, dtype=torch.int32, device="cuda")
    M = S * B
    A = torch.randn(2, M, K, device="cuda"); W = torch.randn(Z, N, K, device="cuda") * 0.05
    b = torch.randn(Z, N, device="cuda")
    C = torch.zeros(Z, M, N, device="cuda")
    for tile in (9, 3, 2, 1, 8):
        try:
            t = timeit(lambda: hip.gemm(A, W, C, M, N, K, K, K, N, shift=b, batch=Z, a_z=(4, 0, M * K), b_z=(1, 0, N * K), c_z=(1, 0, M * N),
                                        s_z=(1, 0, N), tile=tile, seg=(1, seg, B, 1)), reps=10, warm=3, inner=10)
            print("x-proj B=%d (M=%d) tile %d: %.1f us" % (B, M, tile, t * 1e6), flush=True)
        except Exception as e:
            print("tile", tile, "failed", str(e)[:80])
    # weight gradient: dW[2120 x 544] = dG^T [2120 x M] X [M x 544], k-tiles skipped by segment (seg_mode 2)
    dG = torch.randn(Z, M, N, device="cuda"); X = torch.randn(2, M, K, device="cuda"); dW = torch.zeros(Z, N, K, device="cuda")
    for tile in (0, 3, 1, 2, 8):
        try:
            t = timeit(lambda: hip.gemm(dG, X, dW, N, K, M, N, K, K, a_mode=1, b_mode=1, batch=Z, a_z=(1, 0, M * N), b_z=(4, 0, M * K), c_z=(1, 0, N * K),
                                        tile=tile, seg=(2, seg, B, 1)), reps=10, warm=3, inner=10)
            print("dW    B=%d (K=%d) tile %d: %.1f us" % (B, M, tile, t * 1e6), flush=True)
        except Exception as e:
            print("tile", tile, "failed", str(e)[:80])
