#!/usr/bin/env python3
"""cadre_gemm_bf16_w128 against cadre_gemm_bf16 (tile kernel, split-K) on the inter-task first-layer shapes, same box, interleaved
rounds.   python tools/gw128_bench.py [frames]"""
import os
import sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cadre_amd import hip
from cadre_amd.encoder import _w128_dense_b


def main():
    F = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
    for name, K in (("288x288 (Np = 81)", 41472), ("144x256 (Np = 40)", 20480)):
        N = 1536
        split = int(max(1, min(32, (K + 1296) // 2592)))
        g = torch.Generator(device="cuda").manual_seed(K)
        A = torch.randn(F, K, device="cuda", generator=g).to(torch.bfloat16)
        B = (torch.randn(N, K, device="cuda", generator=g) * 0.05).to(torch.bfloat16)
        Bf = _w128_dense_b(B.float().cpu()).to(torch.bfloat16).cuda()
        o1 = torch.empty(split, F, N, device="cuda"); o2 = torch.empty_like(o1)

        def f_new():
            hip.gemm_bf16_w128(A, Bf, o1, F, N, K, K, N, split)

        def f_old():
            hip.gemm(A, B, o2, F, N, K, K, K, N, split_k=split, bf16=True)
        t = {"w128": [], "tile": []}
        for f in (f_new, f_old):
            f()
        torch.cuda.synchronize()
        for rnd in range(6):
            for key, f in (("w128", f_new), ("tile", f_old)):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(5):
                    f()
                e1.record()
                torch.cuda.synchronize()
                t[key].append(e0.elapsed_time(e1) / 5)
        fl = 2.0 * F * N * K
        tw, tt = np.median(t["w128"]), np.median(t["tile"])
        print("inter-task first layer %s: [%d][%d] x [1536][%d]^T, %d slices   w128 %.3f ms (%.1f TFLOP/s) | tile kernel %.3f ms (%.1f TFLOP/s)   bit-identical: %s"
              % (name, F, K, K, split, tw, fl / tw / 1e9, tt, fl / tt / 1e9, bool(torch.equal(o1, o2))), flush=True)


if __name__ == "__main__":
    main()
