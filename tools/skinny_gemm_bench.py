#!/usr/bin/env python3
"""Recurrent-step GEMM of the PPO update (8 nets x [B, 544] x [2120, 544]^T): time vs split_k and tile."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cadre_amd import hip
from tools.gemm_bench import timeit
hip.lib()
Z, K, N = 8, 544, 2120
W = torch.randn(Z, N, K, device="cuda") * 0.05
for B in (64, 256):
    A = torch.randn(Z, B, K, device="cuda")
    for tile in (9, 3):
        for sk in (1, 2, 4, 8):
            C = torch.zeros(sk, Z, B, N, device="cuda")
            t = timeit(lambda: hip.gemm(A, W, C, B, N, K, K, K, N, batch=Z, a_z=(1, 0, B * K), b_z=(1, 0, N * K), c_z=(1, 0, B * N),
                                        split_k=sk, tile=tile), reps=30, warm=5)
            print("B=%d tile %d split_k %d: %.1f us" % (B, tile, sk, t * 1e6), flush=True)
# backward dh: [B, 2120] x [2120, 544] (b_mode 1)
for B in (64, 256):
    dG = torch.randn(Z, B, N, device="cuda")
    for sk in (1, 4, 8, 16):
        C = torch.zeros(sk, Z, B, K, device="cuda")
        t = timeit(lambda: hip.gemm(dG, W, C, B, K, N, N, K, K, b_mode=1, batch=Z, a_z=(1, 0, B * N), b_z=(1, 0, N * K), c_z=(1, 0, B * K),
                                    split_k=sk), reps=30, warm=5)
        print("dh B=%d split_k %d: %.1f us" % (B, sk, t * 1e6), flush=True)
print("--- floor: same grid, K = 32 / 128 / 544")
for B in (64,):
    for Kx in (32, 128, 544):
        A = torch.randn(Z, B, Kx, device="cuda"); Wx = torch.randn(Z, N, Kx, device="cuda") * 0.05
        C = torch.zeros(1, Z, B, N, device="cuda")
        for tile in (9, 3):
            t = timeit(lambda: hip.gemm(A, Wx, C, B, N, Kx, Kx, Kx, N, batch=Z, a_z=(1, 0, B * Kx), b_z=(1, 0, N * Kx), c_z=(1, 0, B * N), tile=tile), reps=30, warm=5)
            print("B=%d K=%d tile %d: %.1f us" % (B, Kx, tile, t * 1e6), flush=True)
# same launch repeated back to back (weights hot in L2/MALL?)
A = torch.randn(Z, 64, K, device="cuda"); C = torch.zeros(1, Z, 64, N, device="cuda")
def rep4():
    for _ in range(4):
        hip.gemm(A, W, C, 64, N, K, K, K, N, batch=Z, a_z=(1, 0, 64 * K), b_z=(1, 0, N * K), c_z=(1, 0, 64 * N), tile=9)
print("4 back-to-back launches: %.1f us" % (timeit(rep4, reps=20, warm=3) * 1e6))
W1 = W[:1].contiguous()
t = timeit(lambda: hip.gemm(A, W1, C, 64, N, K, K, K, N, batch=Z, a_z=(1, 0, 64 * K), b_z=(1, 1, 0), c_z=(1, 0, 64 * N), tile=9), reps=30, warm=5)
print("all nets share ONE weight matrix (4.6 MB, L2-resident): %.1f us" % (t * 1e6))
