import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cadre_amd import hip
from cadre_amd.encoder import _ring_w
def run(F,H,W,Cin,N,use_resid,act=1):
    r = np.random.RandomState(1)
    td=torch.bfloat16
    x = torch.from_numpy(r.standard_normal((F, H, W, Cin)).astype(np.float32)).cuda().to(td)
    w = torch.from_numpy((r.standard_normal((N, Cin, 3, 3)) * (1.5 / np.sqrt(9 * Cin))).astype(np.float32)).to(td)
    sc = torch.from_numpy((0.5 + r.rand(N)).astype(np.float32)).cuda()
    sh = torch.from_numpy(r.standard_normal(N).astype(np.float32)).cuda()
    res = torch.from_numpy(r.standard_normal((F, H, W, N)).astype(np.float32)).cuda().to(td) if use_resid else None
    out = torch.full((F, H, W, N), 7.0, device="cuda", dtype=td)
    wr = _ring_w(w.float(), 64).to(td).cuda()
    hip.conv3x3_ring(x, wr, sc, sh, res, out, F, H, W, Cin, N, act)
    torch.cuda.synchronize()
    ref = torch.nn.functional.conv2d(x.float().permute(0, 3, 1, 2).cpu(), w.float(), padding=1).permute(0, 2, 3, 1) * sc.cpu() + sh.cpu()
    if use_resid: ref = ref + res.float().cpu()
    if act & 1: ref = torch.relu(ref)
    d = (out.float().cpu() - ref).abs().reshape(-1, N)
    M = d.shape[0]
    print("shape", (F,H,W,Cin,N), "resid", use_resid, "code", hip.lib().cadre_conv3x3_ring_ntile(F,H,W,Cin,N,1), "max err", float(d.max()), "ref max", float(ref.abs().max()))
    # per 32-position block x 8-channel block error map (first 256 positions)
    for p0 in range(0, min(M, 512), 32):
        print("pos %4d:" % p0, " ".join("%5.2f" % float(d[p0:p0+32, c0:c0+8].max()) for c0 in range(0, N, 8)))
    bad = (d > 0.1).float().mean()
    print("fraction bad", float(bad))
if __name__ == "__main__":
    run(1, 16, 16, 64, 64, False)
    run(1, 16, 16, 128, 128, False)
