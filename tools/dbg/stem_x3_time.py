import torch, sys
sys.path.insert(0, '/root/repo')
from cadre_amd import hip
from cadre_amd.encoder import _stem_taps, _stem_taps_x3
L = hip.lib()
F, H = 1024, 288
Hp = 72
img = torch.randint(0, 2 ** 31 - 1, (F, H, H), dtype=torch.int32, device="cuda")
w = torch.randn(64, 4, 7, 7) * 0.05
w0 = _stem_taps(w, 50).cuda(); w3 = _stem_taps_x3(w).cuda()
sc, sh = torch.rand(64, device="cuda") + 0.5, torch.randn(64, device="cuda") * 0.1
out = torch.empty(F, Hp, Hp, 64, device="cuda")
res = {}
for rnd in range(4):
    for mode, wt in ((0, w0), (2, w3)):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            hip.check(L.cadre_stem_pool(hip.ptr(img), hip.ptr(wt), hip.ptr(sc), hip.ptr(sh), hip.ptr(out), F, H, H, mode, Hp * Hp * 64, Hp * 64, 64, 0, hip.stream()), "stem")
        e1.record(); torch.cuda.synchronize()
        res.setdefault(mode, []).append(e0.elapsed_time(e1) / 3)
        if rnd == 0: res["o%d" % mode] = out.clone()
print("mode 0 (v_mfma_f32): %.3f ms   mode 2 (exact bf16 pieces): %.3f ms   max |diff| / max %.2e" % (min(res[0]), min(res[2]), float((res["o0"] - res["o2"]).abs().max() / res["o0"].abs().max())))
