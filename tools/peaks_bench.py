#!/usr/bin/env python3
"""Measured peaks of the box next to the datasheet figures: HBM stream copy / read, and the sustained
MFMA rate (cadre_mfma_peak: register-operand MFMA chains on every SIMD) in fp32 and bf16 with the clock it implies."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cadre_amd import hip  # noqa: E402


def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e-3)
    return best


def measure():
    out = {}
    n = 1 << 30                                              # 4 GiB of fp32 each
    a = torch.empty(n, device="cuda").normal_(); b = torch.empty_like(a)
    sink = torch.zeros(4, device="cuda")
    L = hip.lib()
    t = timed(lambda: b.copy_(a))
    out["hbm_copy_GBps_torch"] = round(2 * 4 * n / t / 1e9, 1)     # read + write (torch's copy kernel)
    # own streaming kernels (peaks.hip): 16 B per lane, 8 loads in flight per lane, nothing but the stream
    t = timed(lambda: hip.check(L.cadre_hbm_stream(1, hip.ptr(a), hip.ptr(b), 4 * n, hip.ptr(sink), hip.stream()), "cadre_hbm_stream"))
    out["hbm_copy_GBps"] = round(2 * 4 * n / t / 1e9, 1)
    t = timed(lambda: hip.check(L.cadre_hbm_stream(0, hip.ptr(a), None, 4 * n, hip.ptr(sink), hip.stream()), "cadre_hbm_stream"))
    out["hbm_read_GBps"] = round(4 * n / t / 1e9, 1)
    del a, b
    for name, bf, flop, per_clk in (("f32", 0, 4096.0, 64.0), ("bf16", 1, 32768.0, 1024.0)):
        for wps in (1, 2):                                   # waves per SIMD
            wgs, iters = 256 * wps, 20000 if bf else 10000
            t = timed(lambda: hip.check(L.cadre_mfma_peak(bf, wgs, iters, hip.ptr(sink), hip.stream()), "cadre_mfma_peak"), reps=3)
            tf = wgs * 4 * iters * 8 * flop / t / 1e12
            out["mfma_%s_%dwave_TFLOPs" % (name, wps)] = round(tf, 1)
            out["mfma_%s_%dwave_clock_GHz" % (name, wps)] = round(tf * 1e12 / (1024 * per_clk) / 1e9, 3)
    # bf16 MFMA shapes on random operands: FLOP/s at the clock the chip holds for each
    for shape in (32, 16):
        wgs, iters = 512, 20000
        t = timed(lambda: hip.check(L.cadre_mfma_shape(shape, wgs, iters, hip.ptr(sink), hip.stream()), "cadre_mfma_shape"), reps=3)
        out["mfma_bf16_random_%s_TFLOPs" % ("32x32x16" if shape == 32 else "16x16x32")] = round(wgs * 4 * iters * 262144.0 / t / 1e12, 1)
    # the fp32 pipe on random operands (constants above): what the chip holds when every multiplier input toggles
    for shape, nm, flop in ((2, "32x32x2", 8 * 4096.0), (4, "16x16x4", 32 * 2048.0)):
        for wps in (1, 2):
            wgs, iters = 256 * wps, 20000
            t = timed(lambda: hip.check(L.cadre_mfma_shape(shape, wgs, iters, hip.ptr(sink), hip.stream()), "cadre_mfma_shape"), reps=3)
            out["mfma_f32_random_%s_%dwave_TFLOPs" % (nm, wps)] = round(wgs * 4 * iters * flop / t / 1e12, 1)
    return out


if __name__ == "__main__":
    r = measure()
    for k, v in r.items():
        print("%-32s %s" % (k, v))
    print("datasheet: HBM 8000 GB/s, fp32 MFMA 157.3 TFLOP/s, bf16 MFMA 2500 TFLOP/s (2.4 GHz)")
