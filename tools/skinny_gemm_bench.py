#!/usr/bin/env python3
"""Recurrent-step GEMMs of the PPO update (8 nets x [B, 544] x [2120, 544]^T and the backward [B, 2120] x [2120, 544]):
time per launch vs tile (9: 32x128 LDS-tiled, 11: register-direct skinny kernel), K, and row segments."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cadre_amd import hip


def timeit(fn, reps=30, warm=5, inner=20):
    """median time of one launch inside a back-to-back run of `inner` (launch gaps overlap: close to the kernel duration)"""
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(inner):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e-3 / inner)
    ts.sort()
    return ts[len(ts) // 2]


def main():
    hip.lib()
    Z, N = 8, 2120
    for B in (64, 256):
        q = B // 4
        seg = torch.tensor([[0, q], [q, q], [2 * q, q], [3 * q, q]] * 2, dtype=torch.int32, device="cuda")
        for K in (32, 128, 544):
            A = torch.randn(Z, B, K, device="cuda"); W = torch.randn(Z, N, K, device="cuda") * 0.05
            C = torch.zeros(Z, B, N, device="cuda")
            for tile in (9, 11):
                for sg in (None, (1, seg, B, 1)):
                    t = timeit(lambda: hip.gemm(A, W, C, B, N, K, K, K, N, batch=Z, a_z=(1, 0, B * K), b_z=(1, 0, N * K), c_z=(1, 0, B * N),
                                                tile=tile, seg=sg), reps=30, warm=5)
                    print("fwd B=%d K=%d tile %d %s: %.1f us" % (B, K, tile, "seg " if sg else "full", t * 1e6), flush=True)
        # backward dh: [B, 2120] x [2120, 544] (b_mode 1)
        dG = torch.randn(Z, B, N, device="cuda"); W = torch.randn(Z, N, 544, device="cuda") * 0.05
        C = torch.zeros(Z, B, 544, device="cuda")
        for tile in (9, 11):
            for sg in (None, (1, seg, B, 1)):
                t = timeit(lambda: hip.gemm(dG, W, C, B, 544, N, N, 544, 544, b_mode=1, batch=Z, a_z=(1, 0, B * N), b_z=(1, 0, N * 544),
                                            c_z=(1, 0, B * 544), tile=tile, seg=sg), reps=30, warm=5)
                print("bwd B=%d tile %d %s: %.1f us" % (B, tile, "seg " if sg else "full", t * 1e6), flush=True)


if __name__ == "__main__":
    main()
