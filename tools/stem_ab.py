#!/usr/bin/env python3
"""Time cadre_pack_obs + cadre_stem_pool alone (1024 frames, 288x288), fp32 and bf16."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cadre_amd import synth
from cadre_amd.encoder import DANetEncoderHIP
from tools.gemm_bench import timeit
F, H, W = 1024, 288, 288
gen = torch.Generator(device="cuda").manual_seed(1)
rgb = torch.randint(0, 256, (F, H, W, 3), dtype=torch.uint8, device="cuda", generator=gen)
route = ((torch.rand(F, W, H, device="cuda", generator=gen) < 0.15) * 255).to(torch.uint8)
for dt in ("f32", "bf16"):
    enc = DANetEncoderHIP(synth.encoder_state(*synth.feat_hw(H, W), 7), H, W, "cuda:0", max_frames=F, dtype=dt)
    x = enc.preprocess(rgb, route)
    from cadre_amd import hip
    L = hip.lib()
    p = enc._buf("pool", (F, 72, 72, 64), torch.bfloat16 if dt == "bf16" else torch.float32)
    def run():
        hip.check(L.cadre_stem_pool(hip.ptr(x), hip.ptr(enc.stem_taps), (None if dt == 'bf16' else hip.ptr(enc.stem.scale)), hip.ptr(enc.stem.shift), hip.ptr(p),
                                    F, H, W, 1 if dt == "bf16" else 0, 72 * 72 * 64, 72 * 64, 64, 0, hip.stream()), "stem")
    t = timeit(run)
    fl = 2.0 * F * 144 * 144 * 64 * 196
    print("stem_pool %s: %.3f ms  %.1f TFLOP/s (algorithmic K=196)" % (dt, t * 1e3, fl / t / 1e12), flush=True)
