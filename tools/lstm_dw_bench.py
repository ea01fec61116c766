#!/usr/bin/env python3
"""cadre_lstm_dw alone at the update's shapes (8 nets, 8 steps, minibatch 64 / 256 rows per head in four sorted runs).
CADRE_DW_LDS=0 / 4 / 8 selects the kernel (read once per process)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cadre_amd import hip  # noqa: E402
from tools.gemm_bench import timeit  # noqa: E402

L = hip.lib()
Z, S, C, D, DP, H4, H4P = 8, 8, 4, 530, 544, 2120, 2176
for B in (64, 256):
    cuts = [0, B // 4 + 3, B // 2 - 5, 3 * B // 4 + 2, B]
    run = [(cuts[i], cuts[i + 1] - cuts[i]) for i in range(4)]
    seg = torch.tensor(run + run, dtype=torch.int32, device="cuda")
    dG = torch.randn(Z, S, B, H4P, device="cuda"); Hs = torch.randn(Z, S + 1, B, DP, device="cuda"); X = torch.randn(2, S, B, DP, device="cuda")
    sL = 2 * H4 * DP + 2 * H4
    grads = torch.zeros(Z * sL, device="cuda")

    def f():
        hip.check(L.cadre_lstm_dw(dG.data_ptr(), H4P, S * B * H4P, Hs.data_ptr(), X.data_ptr(), DP, (S + 1) * B * DP, S * B * DP, C,
                                  grads.data_ptr() + 4 * H4 * DP, grads.data_ptr(), grads.data_ptr() + 4 * 2 * H4 * DP,
                                  grads.data_ptr() + 4 * (2 * H4 * DP + H4), DP, sL, B, S, H4, DP, Z, seg.data_ptr(), hip.stream()), "dw")
    t = timeit(f)
    fl = 2.0 * 2 * Z * (B / 4) * S * H4 * D
    print("CADRE_DW_LDS=%s minibatch %3d: %6.1f us  %.1f TFLOP/s" % (os.environ.get("CADRE_DW_LDS", "0"), B, t * 1e6, fl / t / 1e12), flush=True)
