#!/usr/bin/env python3
"""cadre_gemm_f32 on the batched GEMMs of the Winograd convs (25 planes, 1024 frames of the 288x288 model), by tile."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cadre_amd import hip  # noqa: E402
from tools.gemm_bench import timeit  # noqa: E402

hip.lib()
tiles = [int(t) for t in (sys.argv[1] if len(sys.argv) > 1 else "0,1,2,3,4,5,6,8,10").split(",")]
P = 25
for name, T, N, K in (("layer2 128->128 @36", 147456, 128, 128), ("layer3 256->256 @18", 36864, 256, 256), ("layer4 512->512 @9", 9216, 512, 512),
                      ("head 512->128 @9", 9216, 128, 512), ("head 128->128 @9", 9216, 128, 128)):
    V = torch.randn(P, T, K, device="cuda"); U = torch.randn(P, N, K, device="cuda") * 0.05
    Mx = torch.empty(P, T, N, device="cuda")
    row = []
    for tl in tiles:
        try:
            t = timeit(lambda: hip.gemm(V, U, Mx, T, N, K, K, K, N, batch=P, a_z=(1, P, T * K), b_z=(1, P, N * K), c_z=(1, P, T * N), tile=tl))
            row.append("%d: %4.0f us %5.1f TF" % (tl, t * 1e6, 2.0 * P * T * N * K / t / 1e12))
        except Exception as e:
            row.append("%d: --" % tl)
    print("%-22s (%.1f GB in+out) %s" % (name, P * T * (K + N) * 4 / 1e9, " | ".join(row)), flush=True)
