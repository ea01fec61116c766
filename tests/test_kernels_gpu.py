"""GPU: every C-ABI kernel of libcadre_hip.so against plain torch-CPU fp32 / the oracle.
Tolerances: fp32 accumulate-order differences only (<= 2e-5 relative to the tensor's max)
unless the op is integer/strict-order (bit-exact asserted)."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _has_ab():
    """The superseded kernels of csrc/ab/ (tiles 11 / 12, cadre_conv3x3_c64_bf16) exist only in the A/B build
    (CADRE_BUILD_AB=1 python -m cadre_amd.build; CADRE_HIP_LIB=.../libcadre_hip_ab.so): their tests skip otherwise."""
    from cadre_amd import hip as h
    return h.has_ab_kernels()


needs_ab = pytest.mark.skipif(not _has_ab(), reason="A/B build only (CADRE_BUILD_AB=1)")


def ab(*args):
    return pytest.param(*args, marks=needs_ab)


def dev(x):
    return torch.as_tensor(x).cuda()


def rel(a, b):
    a = a.detach().double().cpu(); b = b.detach().double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


@pytest.fixture(scope="module")
def hip():
    from cadre_amd import hip as h
    h.lib()
    return h


# ----------------------------------------------------------------------------- GEMM
@pytest.mark.parametrize("M,N,K,tile", [(200, 136, 544, 1), (77, 50, 64, 3), (300, 64, 96, 2), (64, 2120, 544, 0),
                                        (5, 33, 128, 3), (300, 200, 544, 8), (70, 300, 96, 9), (300, 100, 160, 10)])
@pytest.mark.parametrize("a_mode,b_mode", [(0, 0), (0, 1), (1, 0), (1, 1)])
def test_gemm_modes(hip, M, N, K, tile, a_mode, b_mode):
    if a_mode == 1 and M % 4:
        M += 4 - M % 4
    if b_mode == 1 and N % 4:
        N += 4 - N % 4
    g = torch.Generator().manual_seed(M * 7 + N)
    A = torch.randn(M, K, generator=g)
    B = torch.randn(N, K, generator=g)
    scale = torch.rand(N, generator=g) + 0.5
    shift = torch.randn(N, generator=g)
    resid = torch.randn(M, N, generator=g)
    want = F.leaky_relu((A @ B.t()) * scale + shift + resid, 0.1)
    Ad = dev(A if a_mode == 0 else A.t().contiguous())
    Bd = dev(B if b_mode == 0 else B.t().contiguous())
    out = torch.zeros(M, N + 3, device="cuda")
    hip.gemm(Ad, Bd, out, M, N, K, K if a_mode == 0 else M, K if b_mode == 0 else N, N + 3, a_mode, b_mode,
             scale=dev(scale), shift=dev(shift), resid=dev(resid), ldr=N, act=2, slope=0.1, tile=tile)
    torch.cuda.synchronize()
    assert rel(out[:, :N], want) < 2e-5
    assert float(out[:, N:].abs().max()) == 0.0


def test_gemm_batched_and_splitk(hip):
    g = torch.Generator().manual_seed(3)
    Z, M, N, K = 8, 96, 160, 544
    A = torch.randn(2, M, K, generator=g)          # shared by groups of 4 (z // 4)
    B = torch.randn(Z, N, K, generator=g)
    bias = torch.randn(Z, N, generator=g)
    want = torch.stack([A[z // 4] @ B[z].t() + bias[z] for z in range(Z)])
    out = torch.empty(Z, M, N, device="cuda")
    hip.gemm(dev(A), dev(B), out, M, N, K, K, K, N, shift=dev(bias), batch=Z, a_z=(4, 0, M * K), b_z=(1, 0, N * K),
             c_z=(1, 0, M * N), s_z=(1, 0, N))
    torch.cuda.synchronize()
    assert rel(out, want) < 2e-5
    # split-K with slab reduce + bias + LeakyReLU
    K2 = 4608
    A2 = torch.randn(40, K2, generator=g); B2 = torch.randn(200, K2, generator=g) * 0.05
    b2 = torch.randn(200, generator=g)
    want2 = F.leaky_relu(A2 @ B2.t() + b2, 0.01)
    S = 6
    slabs = torch.empty(S, 40, 200, device="cuda")
    hip.gemm(dev(A2), dev(B2), slabs, 40, 200, K2, K2, K2, 200, split_k=S)
    out2 = torch.empty(40, 200, device="cuda")
    b2_d = dev(b2)
    hip.check(hip.lib().cadre_splitk_reduce(slabs.data_ptr(), S, 40 * 200, 200, out2.data_ptr(), 200, 40, 200, None,
                                            b2_d.data_ptr(), 2, 0.01, None, 0, hip.stream()), "reduce")
    torch.cuda.synchronize()
    assert rel(out2, want2) < 2e-5


@pytest.mark.parametrize("Z,M,N,K,use_resid,act", [(5, 1000, 128, 128, False, 0), (3, 517, 200, 96, True, 2), (1, 64, 64, 64, True, 1),
                                                  (25, 700, 256, 256, False, 0), (2, 130, 52, 68, True, 0)])
@needs_ab
def test_gemm_streamed_short_k_tile(hip, Z, M, N, K, use_resid, act):
    """Tile 13 (ab/gemm_stream_f32.hip, A/B build: several M-tiles per workgroup, prefetch across tile boundaries; the batched GEMMs of the
    Winograd convs): vs torch, and BIT-IDENTICAL to the one-tile 64x64 kernel (tile 3) — same k order, same epilogue —
    on batches, ragged M / N edges, a K that is not a multiple of the 32-wide k-tile, residual + activation."""
    g = torch.Generator().manual_seed(Z * 1000 + M)
    A = torch.randn(Z, M, K, generator=g)
    B = torch.randn(Z, N, K, generator=g) * 0.1
    sc, sh = torch.rand(Z, N, generator=g) + 0.5, torch.randn(Z, N, generator=g)
    res = torch.randn(Z, M, N, generator=g) if use_resid else None
    want = torch.einsum("zmk,znk->zmn", A, B) * sc[:, None, :] + sh[:, None, :]
    if res is not None:
        want = want + res
    if act == 1:
        want = torch.relu(want)
    if act == 2:
        want = F.leaky_relu(want, 0.1)
    Ad, Bd, scd, shd, rd = dev(A), dev(B), dev(sc), dev(sh), (None if res is None else dev(res))
    outs = {}
    for tile in (13, 3):
        out = torch.full((Z, M, N), 7.0, device="cuda")
        hip.gemm(Ad, Bd, out, M, N, K, K, K, N, scale=scd, shift=shd, resid=rd, ldr=N, act=act, slope=0.1, batch=Z,
                 a_z=(1, Z, M * K), b_z=(1, Z, N * K), c_z=(1, Z, M * N), s_z=(1, Z, N), r_z=(1, Z, M * N), tile=tile)
        outs[tile] = out
    assert rel(outs[13], want) < 2e-5
    assert torch.equal(outs[13], outs[3])


def test_gemm_batched_splitk(hip):
    """dh_{t-1} = dG_t . W_hh shape: tiny [B, 544] outputs, K = 2120, 8 nets, b_mode 1, split-K slabs."""
    g = torch.Generator().manual_seed(8)
    Z, M, N, K, S = 8, 24, 544, 2120, 8
    A = torch.randn(Z, M, K, generator=g) * 0.1
    W = torch.randn(Z, K, N, generator=g) * 0.1
    want = torch.bmm(A, W)
    slabs = torch.empty(S, Z, M, N, device="cuda")
    Ad, Wd = dev(A), dev(W)
    hip.gemm(Ad, Wd, slabs, M, N, K, K, N, N, b_mode=1, batch=Z, a_z=(1, 0, M * K), b_z=(1, 0, K * N),
             c_z=(1, 0, M * N), split_k=S)
    out = torch.empty(Z, M, N, device="cuda")
    hip.check(hip.lib().cadre_splitk_reduce(slabs.data_ptr(), S, Z * M * N, N, out.data_ptr(), N, Z * M, N, None, None,
                                            0, 0.0, None, 0, hip.stream()), "reduce")
    assert rel(out, want) < 2e-5


@pytest.mark.parametrize("Cin,Cout,H,W,k,s,p,tile", [(64, 64, 18, 22, 3, 1, 1, 0), (64, 128, 18, 22, 3, 2, 1, 0),
                                                      (64, 128, 17, 21, 1, 2, 0, 0), (128, 160, 9, 9, 1, 1, 0, 0),
                                                      (4, 64, 30, 36, 7, 2, 3, 0), (128, 256, 18, 18, 3, 1, 1, 8),
                                                      (4, 64, 30, 36, 7, 2, 3, 8), (64, 64, 18, 22, 3, 1, 1, 10),
                                                      (4, 64, 30, 36, 7, 2, 3, 10), (128, 256, 10, 13, 3, 2, 1, 9),
                                                      ab(64, 64, 18, 22, 3, 1, 1, 12), ab(4, 64, 30, 36, 7, 2, 3, 12), ab(64, 128, 17, 21, 1, 2, 0, 12)])
def test_conv_implicit_gemm(hip, Cin, Cout, H, W, k, s, p, tile):
    g = torch.Generator().manual_seed(Cin + Cout + k)
    Nimg = 3
    x = torch.randn(Nimg, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5
    scale = torch.rand(Cout, generator=g) + 0.5
    shift = torch.randn(Cout, generator=g)
    y = F.conv2d(x, w, None, s, p)
    Ho, Wo = y.shape[2], y.shape[3]
    res = torch.randn(Nimg, Cout, Ho, Wo, generator=g)
    want = F.relu(y * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1) + res)
    from cadre_amd.encoder import _khwc
    xd = dev(x.permute(0, 2, 3, 1).contiguous())
    wd = dev(_khwc(w))                                   # [O][KH*KW*I]; the Cin=4 stem -> [O][KH][32] rows
    rd = dev(res.permute(0, 2, 3, 1).contiguous())
    out = torch.empty(Nimg, Ho, Wo, Cout, device="cuda")
    K = wd.shape[1]
    hip.gemm(xd, wd, out, Nimg * Ho * Wo, Cout, K, 0, K, Cout, a_mode=3 if Cin == 4 else 2, scale=dev(scale),
             shift=dev(shift), resid=rd, ldr=Cout, act=1, conv=(H, W, Cin, Ho, Wo, k, k, s, p), tile=tile)
    torch.cuda.synchronize()
    assert rel(out.permute(0, 3, 1, 2), want) < 2e-5


# ----------------------------------------------------------------------------- encoder pieces
def test_preprocess_bit_exact(hip, golden):
    g = golden("prep")
    rgb, route = g["rgb"], g["route"]
    Fn, H, W = rgb.shape[:3]
    lut = torch.from_numpy((np.arange(256) / 255.).astype(np.float32)).cuda()
    out = torch.empty(Fn, H, W, 4, device="cuda")
    rn = torch.empty(Fn, W, H, dtype=torch.uint8, device="cuda")
    fm = torch.empty(Fn, dtype=torch.int32, device="cuda")
    rgb_d, route_d = dev(rgb), dev(route)          # keep device temporaries alive across the call
    hip.check(hip.lib().cadre_preprocess(rgb_d.data_ptr(), route_d.data_ptr(), lut.data_ptr(), out.data_ptr(),
                                         rn.data_ptr(), fm.data_ptr(), Fn, H, W, hip.stream()), "prep")
    torch.cuda.synchronize()
    assert np.array_equal(out.permute(0, 3, 1, 2).cpu().numpy(), g["out"])        # byte-for-byte
    assert np.array_equal(rn.cpu().numpy(), g["route_after"])


def test_maxpool(hip):
    x = torch.randn(2, 64, 15, 18)
    want = F.max_pool2d(x, 3, 2, 1)
    out = torch.empty(2, want.shape[2], want.shape[3], 64, device="cuda")
    xd = dev(x.permute(0, 2, 3, 1).contiguous())
    hip.check(hip.lib().cadre_maxpool3x3s2(xd.data_ptr(), out.data_ptr(), 2, 15,
                                           18, 64, hip.stream()), "maxpool")
    assert torch.equal(out.permute(0, 3, 1, 2).cpu(), want)


@pytest.mark.parametrize("h,w", [(3, 3), (5, 8), (9, 9), (10, 11), (11, 11), (8, 16), (12, 12), (11, 13), (16, 16), (20, 25)])
def test_pam_cam(hip, h, w):
    from oracle import encoder_ref
    g = torch.Generator().manual_seed(h * w)
    Fn, Np = 3, h * w
    x = torch.randn(Fn, 128, h, w, generator=g) * 0.4
    sd = {"da_head.sa.gamma": torch.tensor([0.5]), "da_head.sc.gamma": torch.tensor([0.7])}
    for nm, co in (("query", 16), ("key", 16), ("value", 128)):
        sd["da_head.sa.%s_conv.weight" % nm] = torch.randn(co, 128, 1, 1, generator=g) * 0.1
        sd["da_head.sa.%s_conv.bias" % nm] = torch.randn(co, generator=g) * 0.1
    want_p = encoder_ref.pam(x, sd)
    want_c = encoder_ref.cam(x, sd)
    xd = dev(x.permute(0, 2, 3, 1).contiguous())
    wqkv = torch.cat([sd["da_head.sa.%s_conv.weight" % n].view(-1, 128) for n in ("query", "key", "value")])
    bqkv = torch.cat([sd["da_head.sa.%s_conv.bias" % n] for n in ("query", "key", "value")])
    qkv = torch.empty(Fn * Np, 160, device="cuda")
    hip.gemm(xd, dev(wqkv), qkv, Fn * Np, 160, 128, 128, 128, 160, shift=dev(bqkv))
    y = torch.empty_like(xd)
    hip.check(hip.lib().cadre_pam(xd.data_ptr(), qkv.data_ptr(), 0.5, y.data_ptr(), Fn, Np, hip.stream()), "pam")
    assert rel(y.permute(0, 3, 1, 2), want_p) < 2e-5
    y2 = torch.empty_like(xd)
    hip.check(hip.lib().cadre_cam(xd.data_ptr(), 0.7, y2.data_ptr(), Fn, Np, hip.stream()), "cam")
    assert rel(y2.permute(0, 3, 1, 2), want_c) < 2e-5


def test_attention_kernel_forms_give_the_same_bits(hip, tmp_path):
    """The default attention kernels (pam_large_kernel: a workgroup per 32 query rows; cam_split_kernel: two per frame) against the
    one-workgroup-per-frame kernels they replaced (CADRE_PAM_LARGE=0 / CADRE_CAM_SPLIT=0, read once per process: ONE child process
    for all sizes): same fma chains, same bits."""
    import subprocess, sys, os
    g = torch.Generator().manual_seed(3)
    cases = {}
    for (h, w) in ((3, 3), (5, 8), (9, 9), (11, 11), (8, 16)):
        Fn, Np = 3, h * w
        x = dev((torch.randn(Fn, Np, 128, generator=g) * 0.4).contiguous())
        qkv = dev((torch.randn(Fn * Np, 160, generator=g) * 0.3).contiguous())
        yp, yc = torch.empty_like(x), torch.empty_like(x)
        hip.check(hip.lib().cadre_pam(x.data_ptr(), qkv.data_ptr(), 0.5, yp.data_ptr(), Fn, Np, hip.stream()), "pam")
        hip.check(hip.lib().cadre_cam(x.data_ptr(), 0.7, yc.data_ptr(), Fn, Np, hip.stream()), "cam")
        cases["%dx%d" % (h, w)] = dict(x=x.cpu(), qkv=qkv.cpu(), yp=yp.cpu(), yc=yc.cpu(), Fn=Fn, Np=Np)
    io = str(tmp_path / "io.pt")
    torch.save(cases, io)
    code = ("import torch, sys; sys.path.insert(0, %r); from cadre_amd import hip; L = hip.lib(); cases = torch.load(%r); bad = []\n"
            "for k, d in cases.items():\n"
            "    x = d['x'].cuda(); q = d['qkv'].cuda(); yp = torch.empty_like(x); yc = torch.empty_like(x)\n"
            "    hip.check(L.cadre_pam(x.data_ptr(), q.data_ptr(), 0.5, yp.data_ptr(), d['Fn'], d['Np'], hip.stream()), 'pam')\n"
            "    hip.check(L.cadre_cam(x.data_ptr(), 0.7, yc.data_ptr(), d['Fn'], d['Np'], hip.stream()), 'cam')\n"
            "    torch.cuda.synchronize()\n"
            "    if not torch.equal(yp.cpu(), d['yp']): bad.append('pam ' + k)\n"
            "    if not torch.equal(yc.cpu(), d['yc']): bad.append('cam ' + k)\n"
            "print(bad); sys.exit(3 if bad else 0)\n" % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), io))
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, CADRE_PAM_LARGE="0", CADRE_CAM_SPLIT="0"), capture_output=True, text=True)
    assert r.returncode == 0, (r.returncode, r.stdout[-300:], r.stderr[-500:])


def test_intertask_tail_and_measurements(hip):
    g = torch.Generator().manual_seed(5)
    Fn = 4
    qkv = torch.randn(Fn, 6, 256, generator=g) * 2.0
    vq, vk, vv, bq, bk, bv = (qkv[:, i] for i in range(6))

    def cross(q, k, v):
        e = torch.bmm((q.view(Fn, 1, 256).permute(0, 2, 1)) / 16.0, k.view(Fn, 1, 256))
        att = torch.softmax(e, -1)
        return torch.bmm(v.view(Fn, 1, 256), att.permute(0, 2, 1)).view(Fn, -1) + v
    want = torch.cat([cross(bq, vk, vv), cross(vq, bk, bv)], -1)
    meas = torch.rand(Fn, 3, dtype=torch.float64)
    out = torch.zeros(Fn, 544, device="cuda")
    qkv_d, meas_d = dev(qkv), dev(meas)
    hip.check(hip.lib().cadre_intertask_att(qkv_d.data_ptr(), out.data_ptr(), 544, Fn, 16.0, hip.stream()), "ita")
    hip.check(hip.lib().cadre_append_measurements(meas_d.data_ptr(), out.data_ptr(), 544, Fn, hip.stream()), "meas")
    assert rel(out[:, :512], want) < 2e-5
    assert torch.equal(out[:, 512:530].cpu(), meas.repeat(1, 6).float())
    assert float(out[:, 530:].abs().max()) == 0.0


# ----------------------------------------------------------------------------- storage math
@pytest.mark.parametrize("T", [32, 128, 200])
def test_gae_bit_exact_and_ordering(hip, golden, T):
    g = golden("gae")
    nseq = 3                                            # same sequence replicated + a perturbed one
    r = np.stack([g["T%d_rewards" % T]] * nseq); v = np.stack([g["T%d_values" % T]] * nseq)
    m = np.stack([g["T%d_masks" % T]] * nseq)
    r[2] = r[2][::-1]
    nv = torch.full((nseq,), float(g["T%d_next" % T])).cuda()
    rd, vd, md = dev(r.copy()), dev(v.copy()), dev(m.copy())
    ret = torch.zeros(nseq, T + 1, device="cuda"); adv = torch.empty(nseq, T, device="cuda")
    hip.check(hip.lib().cadre_gae(rd.data_ptr(), vd.data_ptr(), md.data_ptr(), nv.data_ptr(), ret.data_ptr(),
                                  adv.data_ptr(), nseq, T, float(np.float32(0.99)), float(np.float32(0.99 * 0.95)), 1, hip.stream()),
              "gae")
    ret, adv, vd = ret.cpu().numpy(), adv.cpu().numpy(), vd.cpu().numpy()
    for s in (0, 1):
        assert np.array_equal(ret[s].view(np.uint32), g["T%d_returns" % T].view(np.uint32))   # bit-exact
        assert vd[s][T] == g["T%d_next" % T]
        assert np.array_equal(np.argsort(adv[s], kind="stable"), g["T%d_argsort" % T])        # advantage ordering
        assert np.abs(adv[s] - g["T%d_adv" % T]).max() <= 4e-6 * np.abs(g["T%d_adv" % T]).max()
    from oracle import ppo_ref
    o_ret, o_V = ppo_ref.gae_returns(r[2], v[2], m[2], float(g["T%d_next" % T]), 0.99, 0.95)
    assert np.array_equal(ret[2].view(np.uint32), o_ret.view(np.uint32))
    # un-normalised
    adv2 = torch.empty(nseq, T, device="cuda")
    v2, ret2 = dev(v.copy()), torch.zeros(nseq, T + 1, device="cuda")
    hip.check(hip.lib().cadre_gae(rd.data_ptr(), v2.data_ptr(), md.data_ptr(), nv.data_ptr(),
                                  ret2.data_ptr(), adv2.data_ptr(), nseq, T,
                                  float(np.float32(0.99)), float(np.float32(0.99 * 0.95)), 0, hip.stream()), "gae")
    assert np.array_equal(adv2[0].cpu().numpy().view(np.uint32), g["T%d_adv_raw" % T].view(np.uint32))


def test_gather_obs(hip):
    obs = torch.randn(33, 8, 530)
    obs_d = torch.zeros(33, 8, 544, device="cuda"); obs_d[:, :, :530] = obs.cuda()
    idx = torch.randperm(32)[:16]
    x = torch.full((8, 16, 544), 7.0, device="cuda")
    idx_d = idx.cuda()
    hip.check(hip.lib().cadre_gather_obs(obs_d.data_ptr(), 544, 8, idx_d.data_ptr(), 16, x.data_ptr(), 544, 530,
                                         hip.stream()), "gather")
    want = obs[idx].permute(1, 0, 2)
    assert torch.equal(x[:, :, :530].cpu(), want) and float(x[:, :, 530:].abs().max()) == 0.0


# ----------------------------------------------------------------------------- LSTM pointwise
@needs_ab
def test_lstm_pointwise_fwd_bwd(hip):
    g = torch.Generator().manual_seed(9)
    Z, B, Hd, ldg, ldh = 8, 6, 530, 2120, 544
    G = torch.randn(Z, B, 4 * Hd, generator=g)
    c0 = torch.randn(2, B, Hd, generator=g)             # shared per head (z // 4)
    dh = torch.randn(Z, B, Hd, generator=g); dc_in = torch.randn(Z, B, Hd, generator=g)
    Gr = G.clone().requires_grad_(True); c0r = c0.clone().requires_grad_(True)
    i, f, gg, o = Gr.chunk(4, -1)
    cp = c0r[torch.arange(Z) // 4]
    c = torch.sigmoid(f) * cp + torch.sigmoid(i) * torch.tanh(gg)
    h = torch.sigmoid(o) * torch.tanh(c)
    Gd = dev(G.clone())
    c0d = torch.zeros(2, B, ldh, device="cuda"); c0d[:, :, :Hd] = c0.cuda()
    cd = torch.zeros(Z, B, ldh, device="cuda"); hd_ = torch.zeros_like(cd); tcd = torch.zeros_like(cd)
    L = hip.lib()
    hip.check(L.cadre_lstm_pointwise_fwd(Gd.data_ptr(), ldg, B * ldg, c0d.data_ptr(), B * ldh, 4, cd.data_ptr(),
                                         hd_.data_ptr(), tcd.data_ptr(), ldh, B * ldh, B, Hd, Z, None, hip.stream()), "lf")
    assert rel(hd_[:, :, :Hd], h) < 1e-5 and rel(cd[:, :, :Hd], c) < 1e-5
    assert float(hd_[:, :, Hd:].abs().max()) == 0.0
    dhd = torch.zeros(Z, B, ldh, device="cuda"); dhd[:, :, :Hd] = dh.cuda()
    dcd = torch.zeros(Z, B, ldh, device="cuda"); dcd[:, :, :Hd] = dc_in.cuda()
    dG = torch.zeros(Z, B, ldg, device="cuda")
    hip.check(L.cadre_lstm_pointwise_bwd(Gd.data_ptr(), dG.data_ptr(), ldg, B * ldg, dhd.data_ptr(), dcd.data_ptr(),
                                         B * ldh, tcd.data_ptr(), c0d.data_ptr(), B * ldh, 4, ldh, B * ldh, B, Hd, Z,
                                         None, 4, None, hip.stream()), "lb")
    # Gr.grad accumulated both backward calls: second call's contribution = full; subtract first
    Gr2 = G.clone().requires_grad_(True); c0r2 = c0.clone().requires_grad_(True)
    i, f, gg, o = Gr2.chunk(4, -1)
    c2 = torch.sigmoid(f) * c0r2[torch.arange(Z) // 4] + torch.sigmoid(i) * torch.tanh(gg)
    h2 = torch.sigmoid(o) * torch.tanh(c2)
    (h2 * dh + c2 * dc_in).sum().backward()
    assert rel(dG, Gr2.grad) < 1e-5
    want_dcprev = (dc_in + dh * torch.sigmoid(o) * (1 - torch.tanh(c2) ** 2)) * torch.sigmoid(f)
    assert rel(dcd[:, :, :Hd], want_dcprev.detach()) < 1e-5


def _lstm_step_case(Z, B, seed, sorted_rows):
    """Operands of one fused LSTM step (ppo_update.hip) in the update workspace's layout: hidden 530 padded to 544,
    gate rows [i f g o] x 530 at pitch 2176, optional row segments (one run of rows per net of a head)."""
    g = torch.Generator().manual_seed(seed)
    D, DP, H4, H4P = 530, 544, 2120, 2176
    W = torch.zeros(Z, H4, DP); W[:, :, :D] = torch.randn(Z, H4, D, generator=g) * 0.04
    b = torch.randn(Z, H4, generator=g) * 0.1
    Gx = torch.randn(Z, B, H4, generator=g) * 0.5
    hp = torch.zeros(Z, B, DP); hp[:, :, :D] = torch.randn(Z, B, D, generator=g) * 0.5
    cp = torch.zeros(Z, B, DP); cp[:, :, :D] = torch.randn(Z, B, D, generator=g)
    seg = None
    if sorted_rows:
        cuts = sorted(torch.randint(0, B + 1, (3,), generator=g).tolist())
        run = [(0, cuts[0]), (cuts[0], cuts[1] - cuts[0]), (cuts[1], cuts[2] - cuts[1]), (cuts[2], B - cuts[2])]
        seg = torch.tensor((run + run)[:Z], dtype=torch.int32)
    return D, DP, H4, H4P, W, b, Gx, hp, cp, seg


@pytest.mark.parametrize("Z,B,sorted_rows", [(8, 64, True), (8, 256, True), (8, 24, False), (2, 1, False), (8, 96, True)])
def test_lstm_step_fwd_fused(hip, Z, B, sorted_rows):
    """cadre_lstm_step_fwd: gates = x-projection + h W_hh^T + b_hh, nn.LSTMCell cell math (models.py:139-152), in one
    launch, against float64 torch; rows outside a net's run untouched, padding columns never written."""
    D, DP, H4, H4P, W, b, Gx, hp, cp, seg = _lstm_step_case(Z, B, 100 + B, sorted_rows)
    pre = Gx.double() + torch.bmm(hp[:, :, :D].double(), W[:, :, :D].double().transpose(1, 2)) + b.double()[:, None]
    i, f, gg, o = pre.chunk(4, -1)
    c = torch.sigmoid(f) * cp[:, :, :D].double() + torch.sigmoid(i) * torch.tanh(gg)
    h = torch.sigmoid(o) * torch.tanh(c)
    act = torch.cat([torch.sigmoid(i), torch.sigmoid(f), torch.tanh(gg), torch.sigmoid(o)], -1)
    # arena-like strides: W and bias of net z at z * w_str inside one buffer
    w_str = H4 * DP + H4
    buf = torch.zeros(Z * w_str)
    for z in range(Z):
        buf[z * w_str: z * w_str + H4 * DP] = W[z].reshape(-1)
        buf[z * w_str + H4 * DP: (z + 1) * w_str] = b[z]
    bufd = dev(buf)
    L = hip.lib()
    NP = 34 * 4 * 34 * 256                                 # floats of one net's packed copy
    packed = torch.full((2, Z, NP), 7.0, device="cuda")
    hip.check(L.cadre_pack_lstm_weights(bufd.data_ptr(), w_str, DP, D, Z, packed[0].data_ptr(), packed[1].data_ptr(), NP,
                                        hip.stream()), "cadre_pack_lstm_weights")
    # fragment order, forward: [slice][gate][k-block][q][c][i] = W[g*D + 16*slice + c][16j + 4q + i]
    Wz = torch.zeros(Z, 4, 34 * 16, DP); Wz[:, :, :D] = W.view(Z, 4, D, DP)
    want_f = Wz.view(Z, 4, 34, 16, 34, 4, 4).permute(0, 2, 1, 4, 5, 3, 6).reshape(Z, NP)
    assert torch.equal(packed[0].cpu(), want_f)
    # backward: [slice][quarter w][k-block j][q][c][i] = W[n = 16*(34w + j) + 4q + i][16*slice + c]
    Wn = torch.zeros(Z, 2176, 34 * 16); Wn[:, :H4, :D] = W[:, :, :D]
    want_b = Wn.view(Z, 4, 34, 4, 4, 34, 16).permute(0, 5, 1, 2, 3, 6, 4).reshape(Z, NP)
    assert torch.equal(packed[1].cpu(), want_b)
    Gd = torch.full((Z, B, H4P), 7.0, device="cuda"); Gd[:, :, :H4] = Gx.cuda()
    hpd, cpd = dev(hp), dev(cp)
    ho = torch.full((Z, B, DP), 7.0, device="cuda"); co = torch.full_like(ho, 7.0); tco = torch.full_like(ho, 7.0)
    segd = None if seg is None else dev(seg)
    hip.check(L.cadre_lstm_step_fwd(packed[0].data_ptr(), NP, bufd.data_ptr() + 4 * H4 * DP, w_str, Gd.data_ptr(), H4P, B * H4P,
                                    hpd.data_ptr(), cpd.data_ptr(), ho.data_ptr(), co.data_ptr(), tco.data_ptr(), DP, B * DP,
                                    B, D, Z, None if segd is None else segd.data_ptr(), B & 1, hip.stream()), "cadre_lstm_step_fwd")
    torch.cuda.synchronize()
    worst = 0.0
    for z in range(Z):
        lo, hi = (0, B) if seg is None else (int(seg[z, 0]), min(B, int(seg[z, 0]) + int(seg[z, 1])))      # exactly the net's run
        if hi > lo:
            worst = max(worst, rel(ho[z, lo:hi, :D], h[z, lo:hi]), rel(co[z, lo:hi, :D], c[z, lo:hi]),
                        rel(tco[z, lo:hi, :D], torch.tanh(c[z, lo:hi])), rel(Gd[z, lo:hi, :H4], act[z, lo:hi]))
        rows_out = torch.ones(B, dtype=torch.bool); rows_out[lo:hi] = False
        assert bool((ho[z][rows_out.cuda()] == 7.0).all()) and bool((Gd[z][rows_out.cuda()][:, :H4] == Gx[z][rows_out].cuda()).all())
        assert bool((ho[z, :, D:] == 7.0).all()) and bool((Gd[z, :, H4:] == 7.0).all())        # padding never written
    print("lstm_step_fwd Z=%d B=%d sorted=%s: rel-max-err %.2e" % (Z, B, sorted_rows, worst))
    assert worst < 1e-5


@pytest.mark.parametrize("Z,B,sorted_rows", [(8, 64, True), (8, 256, True), (8, 24, False), (8, 96, True)])
def test_lstm_step_bwd_fused(hip, Z, B, sorted_rows):
    """cadre_lstm_step_bwd: dh_{t-1} = dG_t W_hh (+ dh_in) on the transposed weights, then the cell backward of step t-1
    in the same launch — against the autograd of the cell; ownership mask (rows of other command nets: exact zeros);
    the product-free first step; the fragment-order copy of dG_{t-1} for the next step."""
    D, DP, H4, H4P, W, b, Gx, hp, cp, seg = _lstm_step_case(Z, B, 300 + B, sorted_rows)
    g = torch.Generator().manual_seed(B)
    C = 4
    L = hip.lib()
    NP = 34 * 4 * 34 * 256
    Wd = dev(W)
    packed = torch.zeros(2, Z, NP, device="cuda")
    hip.check(L.cadre_pack_lstm_weights(Wd.data_ptr(), H4 * DP, DP, D, Z, packed[0].data_ptr(), packed[1].data_ptr(), NP,
                                        hip.stream()), "cadre_pack_lstm_weights")

    def frag(t):                                          # [Z][B][H4] -> fragment order [Z][tile][k-block][q][r16][i], B padded to 16;
        Bp = (B + 15) // 16 * 16                          # tiles count from the first row of the net's run (NaN where no row lives)
        full = torch.full((Z, Bp, H4P), float("nan")); full[:, :, H4:] = 0
        for z in range(Z):
            lo, n = (0, B) if seg is None else (int(seg[z, 0]), int(seg[z, 1]))
            full[z, :n, :H4] = t[z, lo:lo + n]
        return full.view(Z, Bp // 16, 16, 136, 4, 4).permute(0, 1, 3, 4, 2, 5).contiguous()
    # cell of step t-1 (activated gates, tanh c, c_prev) and incoming gradients
    pre = torch.randn(Z, B, H4, generator=g)
    i, f, gg, o = pre.chunk(4, -1)
    act = torch.cat([torch.sigmoid(i), torch.sigmoid(f), torch.tanh(gg), torch.sigmoid(o)], -1)
    cprev = torch.randn(Z, B, D, generator=g)
    cn = torch.sigmoid(f) * cprev + torch.sigmoid(i) * torch.tanh(gg)
    dG_t = torch.randn(Z, B, H4, generator=g) * 0.3
    dh_up = torch.randn(Z, B, D, generator=g) * 0.3
    dc_in = torch.randn(Z, B, D, generator=g) * 0.3
    cmds = torch.randint(0, C, (2, B), generator=g, dtype=torch.int32)
    if seg is not None:                                   # rows sorted by command: consistent with the runs
        for hd in range(2):
            for c in range(C):
                b0, n = int(seg[hd * C + c, 0]), int(seg[hd * C + c, 1])
                cmds[hd, b0:b0 + n] = c
    pad = lambda t, w: torch.cat([t, torch.zeros(*t.shape[:-1], w - t.shape[-1])], -1)
    for with_product in (True, False):
        dh = (torch.bmm(dG_t.double(), W[:, :, :D].double()) if with_product else 0) + dh_up.double()
        tc = torch.tanh(cn.double())
        si, sf, tg, so = (t.double() for t in act.chunk(4, -1))
        dct = dc_in.double() + dh * so * (1 - tc * tc)
        want = torch.cat([dct * tg * si * (1 - si), dct * cprev.double() * sf * (1 - sf), dct * si * (1 - tg * tg),
                          dh * tc * so * (1 - so)], -1)
        want_dc = dct * sf
        own = torch.stack([cmds[z // C] == z % C for z in range(Z)])          # [Z][B]
        want = want * own[:, :, None]; want_dc = want_dc * own[:, :, None]
        dGt_d = dev(frag(dG_t)); act_d = dev(pad(act, H4P))
        gps = dGt_d[0].numel()
        dGo = torch.full((Z, B, H4P), 7.0, device="cuda")
        dGpo = torch.full_like(dGt_d, 7.0)
        dCd = dev(pad(dc_in, DP)); dhd = dev(pad(dh_up, DP))
        tcd = dev(pad(torch.tanh(cn), DP)); cpd = dev(pad(cprev, DP))
        segd = None if seg is None else dev(seg)
        cm = dev(cmds)
        hip.check(L.cadre_lstm_step_bwd(packed[1].data_ptr(), NP, dGt_d.data_ptr() if with_product else None, dGpo.data_ptr(), gps,
                                        dGo.data_ptr(), act_d.data_ptr(), H4P, B * H4P, dhd.data_ptr(), dCd.data_ptr(), B * DP, tcd.data_ptr(),
                                        cpd.data_ptr(), DP, B * DP, B, D, Z, cm.data_ptr(), C,
                                        None if segd is None else segd.data_ptr(), int(with_product), hip.stream()), "cadre_lstm_step_bwd")
        torch.cuda.synchronize()
        worst = 0.0
        for z in range(Z):
            lo, hi = (0, B) if seg is None else (int(seg[z, 0]), min(B, int(seg[z, 0]) + int(seg[z, 1])))      # exactly the net's run
            if hi > lo:
                scale = float(want[z].abs().max()) or 1.0
                worst = max(worst, float((dGo[z, lo:hi, :H4].double().cpu() - want[z, lo:hi]).abs().max()) / scale,
                            float((dCd[z, lo:hi, :D].double().cpu() - want_dc[z, lo:hi]).abs().max()) / max(float(want_dc[z].abs().max()), 1e-30))
                foreign = ~own[z, lo:hi]
                if bool(foreign.any()):
                    assert float(dGo[z, lo:hi, :H4][foreign.cuda()].abs().max()) == 0.0          # exact zeros
            rows_out = torch.ones(B, dtype=torch.bool); rows_out[lo:hi] = False
            assert bool((dGo[z][rows_out.cuda()] == 7.0).all())
            assert bool((dGo[z, :, H4:] == 7.0).all())
            # the fragment-order copy holds the same values as the row-major one, rows counted from the run's first; 7.0 elsewhere
            got_p = dGpo[z].permute(0, 3, 1, 2, 4).reshape(-1, H4P)[:B]
            if hi > lo:
                assert torch.equal(got_p[:hi - lo, :H4], dGo[z, lo:hi, :H4])
            assert bool((got_p[hi - lo:] == 7.0).all()) and bool((got_p[:, H4:] == 7.0).all())
        print("lstm_step_bwd Z=%d B=%d sorted=%s product=%s: rel-max-err %.2e" % (Z, B, sorted_rows, with_product, worst))
        assert worst < 2e-5


@pytest.mark.parametrize("B,sorted_rows", [(64, True), (256, True), (24, False), (6, False), (96, True)])
def test_lstm_dw_fused(hip, B, sorted_rows):
    """cadre_lstm_dw: dW_hh = sum_t dG_t^T h_{t-1}, dW_ih = sum_t dG_t^T x_t, db = column sums of dG for 8 nets in one
    launch (autograd of models.py:139-152) against float64; only a net's run of rows is multiplied (the others are stale
    memory — NaN here); padding columns of the gradient rows come out zero; bit-identical on repeat."""
    g = torch.Generator().manual_seed(900 + B)
    Z, S, C, D, DP, H4, H4P = 8, 8, 4, 530, 544, 2120, 2176
    seg = None
    own = torch.ones(Z, B, dtype=torch.bool)
    if sorted_rows:
        cuts = sorted(torch.randint(0, B + 1, (3,), generator=g).tolist())
        run = [(0, cuts[0]), (cuts[0], cuts[1] - cuts[0]), (cuts[1], cuts[2] - cuts[1]), (cuts[2], B - cuts[2])]
        seg = torch.tensor(run + run, dtype=torch.int32)
        own = torch.zeros(Z, B, dtype=torch.bool)
        for z in range(Z):
            own[z, int(seg[z, 0]):int(seg[z, 0]) + int(seg[z, 1])] = True
    dG = torch.zeros(Z, S, B, H4P); dG[..., :H4] = torch.randn(Z, S, B, H4, generator=g) * own[:, None, :, None]
    Hs = torch.zeros(Z, S + 1, B, DP); Hs[..., :D] = torch.randn(Z, S + 1, B, D, generator=g)
    X = torch.zeros(2, S, B, DP); X[..., :D] = torch.randn(2, S, B, D, generator=g)
    want_hh = torch.einsum("zsbm,zsbn->zmn", dG[..., :H4].double(), Hs[:, :S].double())
    want_ih = torch.einsum("zsbm,zsbn->zmn", dG[..., :H4].double(), X[torch.arange(Z) // C].double())
    want_b = dG[..., :H4].double().sum((1, 2))
    sL = 2 * H4 * DP + 2 * H4
    grads = torch.full((Z * sL,), 7.0, device="cuda")
    dGd, Hsd, Xd = dev(dG), dev(Hs), dev(X)
    dGd[(~own).cuda()[:, None].expand(Z, S, B)] = float("nan")          # rows of other nets: never written by the backward, never read
    Hsd[:, :S][(~own).cuda()[:, None].expand(Z, S, B)] = float("nan")   # (their h rows hold whatever an earlier minibatch left)
    segd = None if seg is None else dev(seg)
    L = hip.lib()

    def run():
        hip.check(L.cadre_lstm_dw(dGd.data_ptr(), H4P, S * B * H4P, Hsd.data_ptr(), Xd.data_ptr(), DP, (S + 1) * B * DP,
                                  S * B * DP, C, grads.data_ptr() + 4 * H4 * DP, grads.data_ptr(), grads.data_ptr() + 4 * 2 * H4 * DP,
                                  grads.data_ptr() + 4 * (2 * H4 * DP + H4), DP, sL, B, S, H4, DP, Z,
                                  None if segd is None else segd.data_ptr(), hip.stream()), "cadre_lstm_dw")
        torch.cuda.synchronize()
        return grads.clone()
    out = run().view(Z, sL)
    ih = out[:, :H4 * DP].view(Z, H4, DP); hh = out[:, H4 * DP:2 * H4 * DP].view(Z, H4, DP)
    b_ih = out[:, 2 * H4 * DP:2 * H4 * DP + H4]; b_hh = out[:, 2 * H4 * DP + H4:]
    e = (rel(hh, want_hh), rel(ih, want_ih), rel(b_ih, want_b))
    print("lstm_dw B=%d sorted=%s: rel-max-err dWhh %.2e dWih %.2e db %.2e" % ((B, sorted_rows) + e))
    assert max(e) < 2e-5
    assert torch.equal(b_ih, b_hh)
    assert float(hh[:, :, D:].abs().max()) == 0.0 and float(ih[:, :, D:].abs().max()) == 0.0
    assert torch.equal(run(), out.view(-1))


@pytest.mark.parametrize("B,segs", [(64, [[0, 13], [13, 22], [35, 11], [46, 18]]), (256, [[0, 70], [70, 58], [128, 49], [177, 79]]),
                                    (24, None), (5, None), (96, [[0, 0], [0, 96], [96, 0], [96, 0]])])
def test_mlp_towers_fused(hip, B, segs):
    """cadre_mlp_fwd / cadre_mlp_bwd / cadre_mlp_dw: the three layers of the 16 actor / critic towers (models.py:171-177,
    distributions.py:34-40) forward, the dX chain backward (dh summed over a net's two towers) and the six parameter
    gradients per tower, each on exactly its net's run of rows — against float64 torch autograd; rows outside a run are
    neither read (NaN there) nor written, an empty run writes zero gradients."""
    import ctypes
    g = torch.Generator().manual_seed(1234 + B)
    Z, Z2, D, DP, hid, NP = 8, 16, 530, 544, 128, 64
    n_out = [33, 1] * 4 + [3, 1] * 4                       # actor / critic rows of W3 that exist
    t_w1, t_b1 = 0, hid * DP
    t_w2, t_b2 = t_b1 + hid, t_b1 + hid + hid * hid
    t_w3, t_b3 = t_b2 + hid, t_b2 + hid + NP * hid
    sT = t_b3 + NP
    offs = (ctypes.c_int32 * 6)(t_w1, t_b1, t_w2, t_b2, t_w3, t_b3)
    W1 = torch.zeros(Z2, hid, DP); W1[:, :, :D] = torch.randn(Z2, hid, D, generator=g) * 0.05
    W2 = torch.randn(Z2, hid, hid, generator=g) * 0.1
    W3 = torch.zeros(Z2, NP, hid)
    b1, b2, b3 = torch.randn(Z2, hid, generator=g) * 0.1, torch.randn(Z2, hid, generator=g) * 0.1, torch.zeros(Z2, NP)
    for z2 in range(Z2):
        W3[z2, :n_out[z2]] = torch.randn(n_out[z2], hid, generator=g) * 0.1
        b3[z2, :n_out[z2]] = torch.randn(n_out[z2], generator=g) * 0.1
    P = torch.cat([torch.cat([W1[z].reshape(-1), b1[z], W2[z].reshape(-1), b2[z], W3[z].reshape(-1), b3[z]]) for z in range(Z2)])
    assert P.numel() == Z2 * sT
    seg = None if segs is None else torch.tensor(segs + segs, dtype=torch.int32)
    own = torch.ones(Z, B, dtype=torch.bool)
    if seg is not None:
        own[:] = False
        for z in range(Z):
            own[z, int(seg[z, 0]):int(seg[z, 0]) + int(seg[z, 1])] = True
    own2 = own.repeat_interleave(2, 0)                      # [Z2][B]
    H = torch.zeros(Z, B, DP); H[..., :D] = torch.randn(Z, B, D, generator=g)
    dO3 = torch.zeros(Z2, B, NP)
    for z2 in range(Z2):
        dO3[z2, :, :n_out[z2]] = torch.randn(B, n_out[z2], generator=g) * 0.3
    # float64 reference through autograd, per tower on its own rows
    Wd = [t.double().requires_grad_() for t in (W1, b1, W2, b2, W3, b3)]
    Hd = H.double().requires_grad_()
    Hin2 = Hd.repeat_interleave(2, 0)
    a1 = torch.relu(torch.einsum("zbk,znk->zbn", Hin2, Wd[0]) + Wd[1][:, None])
    a2 = torch.relu(torch.einsum("zbk,znk->zbn", a1, Wd[2]) + Wd[3][:, None])
    o3 = torch.einsum("zbk,znk->zbn", a2, Wd[4]) + Wd[5][:, None]
    (o3 * (dO3.double() * own2[:, :, None])).sum().backward()
    # device run: NaN in every row a tower does not own
    nan_rows = lambda t, o: torch.where(o[..., None].expand_as(t), t, torch.full_like(t, float("nan")))
    Pd, Hdv = dev(P), dev(nan_rows(H, own) if seg is not None else H)
    A1 = torch.full((Z2, B, hid), 7.0, device="cuda"); A2 = torch.full_like(A1, 7.0); O3 = torch.full((Z2, B, NP), 7.0, device="cuda")
    L = hip.lib()
    sp = None if seg is None else dev(seg)
    spp = None if sp is None else sp.data_ptr()
    hip.check(L.cadre_mlp_fwd(Pd.data_ptr(), sT, offs, Hdv.data_ptr(), DP, B * DP, A1.data_ptr(), A2.data_ptr(), O3.data_ptr(), B, Z2,
                              spp, hip.stream()), "cadre_mlp_fwd")
    torch.cuda.synchronize()
    m2 = own2.cuda()
    e_f = max(rel(A1[m2], a1.detach()[own2]), rel(A2[m2], a2.detach()[own2]), rel(O3[m2], o3.detach()[own2])) if bool(own2.any()) else 0.0
    assert bool((A1[~m2] == 7.0).all()) and bool((A2[~m2] == 7.0).all()) and bool((O3[~m2] == 7.0).all())
    dO3d = dev(nan_rows(dO3, own2) if seg is not None else dO3)
    dA1 = torch.full((Z2, B, hid), 7.0, device="cuda"); dA2 = torch.full_like(dA1, 7.0); dH = torch.full((Z, B, DP), 7.0, device="cuda")
    hip.check(L.cadre_mlp_bwd(Pd.data_ptr(), sT, offs, dO3d.data_ptr(), A1.data_ptr(), A2.data_ptr(), dA1.data_ptr(), dA2.data_ptr(),
                              dH.data_ptr(), DP, B * DP, B, Z2, spp, hip.stream()), "cadre_mlp_bwd")
    G = torch.full((Z2 * sT,), 7.0, device="cuda")
    hip.check(L.cadre_mlp_dw(dO3d.data_ptr(), dA2.data_ptr(), dA1.data_ptr(), A2.data_ptr(), A1.data_ptr(), Hdv.data_ptr(), DP, B * DP,
                             G.data_ptr(), sT, offs, B, Z2, spp, hip.stream()), "cadre_mlp_dw")
    torch.cuda.synchronize()
    mo = own.cuda()
    e_h = rel(dH[mo], Hd.grad[own]) if bool(own.any()) else 0.0
    assert bool((dH[~mo] == 7.0).all()) and bool((dA1[~m2] == 7.0).all()) and bool((dA2[~m2] == 7.0).all())
    Gv = G.view(Z2, sT).cpu()
    got = [Gv[:, t_w1:t_b1].view(Z2, hid, DP), Gv[:, t_b1:t_w2], Gv[:, t_w2:t_b2].view(Z2, hid, hid), Gv[:, t_b2:t_w3],
           Gv[:, t_w3:t_b3].view(Z2, NP, hid), Gv[:, t_b3:]]
    e_w = max(rel(gt, wt.grad) for gt, wt in zip(got, Wd))
    print("mlp towers B=%d sorted=%s: rel-max-err forward %.2e, dh %.2e, parameter gradients %.2e" % (B, seg is not None, e_f, e_h, e_w))
    assert max(e_f, e_h, e_w) < 2e-5
    for z2 in range(Z2):
        if not bool(own2[z2].any()):
            assert float(Gv[z2].abs().max()) == 0.0        # a net without rows: exact zero gradients
    assert float(got[0][:, :, D:].abs().max()) == 0.0       # padding columns of W1


@needs_ab
@pytest.mark.parametrize("B,sorted_rows", [(64, True), (256, True), (24, False), (96, True)])
def test_lstm_seq_fwd_persistent_equals_per_step(hip, B, sorted_rows):
    """cadre_lstm_seq_fwd — all S steps of the 8 nets in one persistent launch, the h rows exchanged between workgroups
    through L2 (agent-scope publish / consume) — gives the bits of S launches of cadre_lstm_step_fwd, on every one of 30
    launches with every CU busy, and never reports a timed-out wait."""
    Z, S = 8, 8
    D, DP, H4, H4P, W, b, Gx, hp, cp, seg = _lstm_step_case(Z, B, 500 + B, sorted_rows)
    g = torch.Generator().manual_seed(B)
    Gall = torch.zeros(Z, S, B, H4P); Gall[..., :H4] = torch.randn(Z, S, B, H4, generator=g) * 0.5
    w_str = H4 * DP + H4
    buf = torch.zeros(Z * w_str)
    for z in range(Z):
        buf[z * w_str: z * w_str + H4 * DP] = W[z].reshape(-1)
        buf[z * w_str + H4 * DP: (z + 1) * w_str] = b[z]
    bufd = dev(buf)
    L = hip.lib()
    NP = 34 * 4 * 34 * 256
    packed = torch.zeros(2, Z, NP, device="cuda")
    hip.check(L.cadre_pack_lstm_weights(bufd.data_ptr(), w_str, DP, D, Z, packed[0].data_ptr(), packed[1].data_ptr(), NP, hip.stream()), "pack")
    segd = None if seg is None else dev(seg)
    sp = None if segd is None else segd.data_ptr()

    def state():
        Hs = torch.full((Z, S + 1, B, DP), 0.0, device="cuda"); Cs = torch.zeros_like(Hs); TC = torch.zeros_like(Hs)
        Hs[:, 0] = hp.cuda(); Cs[:, 0] = cp.cuda()
        return dev(Gall.clone()), Hs, Cs, TC
    G1, H1, C1, T1 = state()
    for t in range(S):
        hip.check(L.cadre_lstm_step_fwd(packed[0].data_ptr(), NP, bufd.data_ptr() + 4 * H4 * DP, w_str, G1[:, t].data_ptr(), H4P,
                                        S * B * H4P, H1[:, t].data_ptr(), C1[:, t].data_ptr(), H1[:, t + 1].data_ptr(),
                                        C1[:, t + 1].data_ptr(), T1[:, t + 1].data_ptr(), DP, (S + 1) * B * DP, B, D, Z, sp, t & 1,
                                        hip.stream()), "step")
    torch.cuda.synchronize()
    sync = torch.zeros(Z * S + 1, dtype=torch.int32, device="cuda")
    for rep in range(30):
        G2, H2, C2, T2 = state()
        hip.check(L.cadre_lstm_seq_fwd(packed[0].data_ptr(), NP, bufd.data_ptr() + 4 * H4 * DP, w_str, G2.data_ptr(), H4P, S * B * H4P,
                                       H2.data_ptr(), C2.data_ptr(), T2.data_ptr(), DP, (S + 1) * B * DP, B, D, S, Z, sp,
                                       sync.data_ptr(), hip.stream()), "seq")
        torch.cuda.synchronize()
        assert int(sync[-1]) == 0, "a wait timed out"
        for z in range(Z):          # (the persistent kernel works on the 32-row tiles around a run, the step kernel on the run itself)
            lo, hi = (0, B) if seg is None else (int(seg[z, 0]), min(B, int(seg[z, 0]) + int(seg[z, 1])))
            assert all(torch.equal(x[z, :, lo:hi], y[z, :, lo:hi]) for x, y in ((H2, H1), (C2, C1), (T2, T1), (G2, G1))), (rep, z)
    assert bool(torch.isfinite(H1).all()) and float(H1[:, S].abs().max()) > 0


@needs_ab
def test_lstm_pointwise_and_colsum2_row_segments(hip):
    """Rows sorted by command: the pointwise LSTM passes and the bias-gradient column sum touch only the 32-row tiles
    that intersect each net's run of rows — same values there as the unrestricted kernels, bit for bit; rows outside
    untouched; the segment-aware column sum equals the full one when the rows outside hold the zeros the backward
    writes."""
    g = torch.Generator().manual_seed(77)
    Z, B, S, Hd, ldh = 8, 128, 2, 530, 544
    ldg = 4 * Hd
    run = [(0, 40), (40, 0), (40, 70), (110, 18)]
    seg = torch.tensor(run + run, dtype=torch.int32, device="cuda")
    hull = [(b & ~31, ((b + c + 31) & ~31) if c else (b & ~31)) for b, c in run + run]
    G = torch.randn(Z, B, ldg, generator=g).cuda()
    c0 = torch.zeros(Z, B, ldh, device="cuda"); c0[:, :, :Hd] = torch.randn(Z, B, Hd, generator=g).cuda()
    L = hip.lib()
    outs = []
    for sg in (None, seg):
        Gd = G.clone()
        cd = torch.full((Z, B, ldh), 5.0, device="cuda"); hd_ = torch.full_like(cd, 5.0); tcd = torch.full_like(cd, 5.0)
        hip.check(L.cadre_lstm_pointwise_fwd(Gd.data_ptr(), ldg, B * ldg, c0.data_ptr(), B * ldh, 1, cd.data_ptr(), hd_.data_ptr(),
                                             tcd.data_ptr(), ldh, B * ldh, B, Hd, Z, None if sg is None else sg.data_ptr(),
                                             hip.stream()), "lf")
        outs.append((Gd, cd, hd_, tcd))
    for z in range(Z):
        lo, hi = hull[z]
        for full, part in zip(outs[0], outs[1]):
            assert torch.equal(full[z, lo:hi], part[z, lo:hi])
        assert torch.equal(outs[1][0][z, :lo], G[z, :lo]) and torch.equal(outs[1][0][z, hi:], G[z, hi:])     # gates untouched
        assert float((outs[1][2][z, :lo, :Hd] - 5.0).abs().max() if lo else 0.0) == 0.0
    # backward: foreign rows inside the tiles get exact zeros, rows outside keep what they held
    cmds = torch.zeros(2, B, dtype=torch.int32)
    for c, (b0, cnt) in enumerate(run):
        cmds[:, b0:b0 + cnt] = c
    cmds = cmds.cuda()
    dh = torch.zeros(Z, B, ldh, device="cuda"); dh[:, :, :Hd] = torch.randn(Z, B, Hd, generator=g).cuda()
    Ga, _cd, _hd, tca = outs[0]
    res = []
    for sg in (None, seg):
        dG = torch.full((Z, B, ldg), 9.0, device="cuda"); dc = torch.zeros(Z, B, ldh, device="cuda")
        hip.check(L.cadre_lstm_pointwise_bwd(Ga.data_ptr(), dG.data_ptr(), ldg, B * ldg, dh.data_ptr(), dc.data_ptr(), B * ldh,
                                             tca.data_ptr(), c0.data_ptr(), B * ldh, 1, ldh, B * ldh, B, Hd, Z, cmds.data_ptr(), 4,
                                             None if sg is None else sg.data_ptr(), hip.stream()), "lb")
        res.append((dG, dc))
    for z in range(Z):
        lo, hi = hull[z]
        assert torch.equal(res[0][0][z, lo:hi], res[1][0][z, lo:hi]) and torch.equal(res[0][1][z, lo:hi], res[1][1][z, lo:hi])
        outside = torch.cat([res[1][0][z, :lo], res[1][0][z, hi:]])
        assert outside.numel() == 0 or float((outside - 9.0).abs().max()) == 0.0
        b0, cnt = (run + run)[z]
        inside_foreign = torch.cat([res[1][0][z, lo:min(b0, hi)], res[1][0][z, max(b0 + cnt, lo):hi]])
        assert inside_foreign.numel() == 0 or float(inside_foreign.abs().max()) == 0.0
    # bias gradients: dG as the unrestricted backward leaves it (zeros on every foreign row), S time steps of B rows
    dGs = torch.stack([res[0][0], res[0][0].flip(1) * 0 + res[0][0]], 1).contiguous()        # [Z][S][B][ldg]
    o1 = torch.zeros(Z, ldg, device="cuda"); o2 = torch.zeros_like(o1); p1 = torch.zeros_like(o1); p2 = torch.zeros_like(o1)
    hip.check(L.cadre_colsum2(dGs.data_ptr(), ldg, S * B * ldg, o1.data_ptr(), o2.data_ptr(), ldg, S * B, ldg, Z, None, 0, hip.stream()), "cs")
    hip.check(L.cadre_colsum2(dGs.data_ptr(), ldg, S * B * ldg, p1.data_ptr(), p2.data_ptr(), ldg, S * B, ldg, Z, seg.data_ptr(), B,
                              hip.stream()), "cs")
    assert torch.equal(o1, p1) and torch.equal(o2, p2) and torch.equal(o1, o2)
    assert rel(o1, dGs.double().sum((1, 2))) < 1e-5


def test_colsum_relu_bwd(hip):
    X = torch.randn(3, 70, 130)
    out = torch.ones(3, 130, device="cuda")
    Xd = dev(X)
    hip.check(hip.lib().cadre_colsum(Xd.data_ptr(), 130, 70 * 130, out.data_ptr(), 130, 70, 130, 3, 1,
                                     hip.stream()), "colsum")
    assert rel(out, X.sum(1) + 1) < 1e-5
    a = torch.randn(1000); dy = torch.randn(1000); dyd = dev(dy.clone())
    ad = dev(a)
    hip.check(hip.lib().cadre_relu_bwd(ad.data_ptr(), dyd.data_ptr(), 1000, None, 0, 0, 0, hip.stream()), "relu_bwd")
    assert torch.equal(dyd.cpu(), dy * (a > 0))


# ----------------------------------------------------------------------------- PPO loss
@pytest.mark.parametrize("B,scale", [(16, 1.0), (64, 4.0), (200, 1.0), (7, 2.0)])
def test_ppo_loss_fwd_bwd(hip, B, scale):
    """vs autograd through the oracle formulas (agent.py:166-229); `scale` widens ratios so
    both clip branches and both value branches are exercised."""
    g = torch.Generator().manual_seed(B)
    ldl, nS, nT = 64, 33, 3
    logits = torch.zeros(8, B, ldl); values = torch.randn(8, B, generator=g)
    logits[:4, :, :nS] = torch.randn(4, B, nS, generator=g) * scale
    logits[4:, :, :nT] = torch.randn(4, B, nT, generator=g) * scale
    actions = torch.stack([torch.randint(0, nS, (B,), generator=g), torch.randint(0, nT, (B,), generator=g)])
    cmds = torch.randint(0, 4, (2, B), generator=g, dtype=torch.int32)
    old_v = torch.randn(2, B, generator=g); rets = torch.randn(2, B, generator=g)
    adv = torch.randn(2, B, generator=g)
    old_lp = torch.stack([torch.full((B,), -np.log(nS)), torch.full((B,), -np.log(nT))]) + 0.3 * torch.randn(2, B, generator=g)
    clip, vc, cc, ec = 0.1, 0.1, 1.0, 0.01
    lg = logits.clone().requires_grad_(True); vv = values.clone().requires_grad_(True)
    tot_v = tot_a = tot_e = 0
    for hd, K in ((0, nS), (1, nT)):
        cur_v = cur_lp = ent = 0
        for c in range(4):
            raw = lg[hd * 4 + c, :, :K]
            lgn = raw - raw.logsumexp(-1, keepdim=True)
            lp = lgn.gather(1, actions[hd].view(-1, 1))
            p = torch.softmax(lgn, -1)
            e = -(lgn * p).sum(-1, keepdim=True)
            msk = (cmds[hd] == c).view(-1, 1)
            cur_v = cur_v + vv[hd * 4 + c].view(-1, 1) * msk
            cur_lp = cur_lp + lp * msk
            ent = ent + e * msk
        ratio = torch.exp(cur_lp - old_lp[hd].view(-1, 1)); A = adv[hd].view(-1, 1)
        tot_a = tot_a - torch.min(ratio * A, torch.clamp(ratio, 1 - clip, 1 + clip) * A).mean()
        ov, R = old_v[hd].view(-1, 1), rets[hd].view(-1, 1)
        vpc = ov + (cur_v - ov).clamp(-clip, clip)
        tot_v = tot_v + 0.5 * torch.max((cur_v - R).pow(2), (vpc - R).pow(2)).mean()
        tot_e = tot_e + ent.mean()
    total = tot_v * vc + tot_a * cc - tot_e * ec
    total.backward()
    losses = torch.zeros(3, device="cuda"); dl = torch.full((8, B, ldl), 9.0, device="cuda")
    dv = torch.full((8, B), 9.0, device="cuda")
    scratch = torch.full((4 + 6 * ((B + 15) // 16),), 3.0, device="cuda")
    scratch[0] = 0.0                                     # the arrival counter: zero on first use, reset by every launch
    d_ = [dev(t) for t in (logits, values, actions, cmds, old_v, rets, old_lp, adv)]
    hip.check(hip.lib().cadre_ppo_loss(d_[0].data_ptr(), ldl, B * ldl, d_[1].data_ptr(), 1, B, d_[2].data_ptr(),
                                       d_[3].data_ptr(), d_[4].data_ptr(), d_[5].data_ptr(),
                                       d_[6].data_ptr(), d_[7].data_ptr(), B, 4, nS, nT, clip, vc, cc, ec,
                                       1.0 / B, losses.data_ptr(), dl.data_ptr(), dv.data_ptr(), scratch.data_ptr(),
                                       None, hip.stream()), "loss")
    want = torch.tensor([float(tot_v * vc), float(tot_a * cc), float(tot_e * ec)])
    assert rel(losses, want) < 1e-5
    # the loss sums are combined in a fixed order (last-arriving workgroup): bit-identical over repeated launches
    first = losses.clone()
    for _ in range(5):
        hip.check(hip.lib().cadre_ppo_loss(d_[0].data_ptr(), ldl, B * ldl, d_[1].data_ptr(), 1, B, d_[2].data_ptr(),
                                           d_[3].data_ptr(), d_[4].data_ptr(), d_[5].data_ptr(),
                                           d_[6].data_ptr(), d_[7].data_ptr(), B, 4, nS, nT, clip, vc, cc, ec,
                                           1.0 / B, losses.data_ptr(), dl.data_ptr(), dv.data_ptr(), scratch.data_ptr(),
                                           None, hip.stream()), "loss")
        assert torch.equal(losses, first)
    poison = torch.ones(1, dtype=torch.int32, device="cuda")                 # a reported forward-pass timeout: NaN losses
    hip.check(hip.lib().cadre_ppo_loss(d_[0].data_ptr(), ldl, B * ldl, d_[1].data_ptr(), 1, B, d_[2].data_ptr(),
                                       d_[3].data_ptr(), d_[4].data_ptr(), d_[5].data_ptr(),
                                       d_[6].data_ptr(), d_[7].data_ptr(), B, 4, nS, nT, clip, vc, cc, ec,
                                       1.0 / B, losses.data_ptr(), dl.data_ptr(), dv.data_ptr(), scratch.data_ptr(),
                                       poison.data_ptr(), hip.stream()), "loss")
    assert bool(torch.isnan(losses).all())
    assert rel(dv, vv.grad) < 1e-5
    assert rel(dl, lg.grad) < 2e-5


def test_sample_matches_golden_rule(hip, golden):
    from oracle import ppo_ref
    g = torch.Generator().manual_seed(1)
    for K in (33, 3):
        R = 64
        logits = torch.randn(R, K, generator=g) * 2
        q = torch.empty(R, K).exponential_(1, generator=g)
        lgn = logits - logits.logsumexp(-1, keepdim=True)
        want = ppo_ref.sample_from_logits(lgn, q)
        act = torch.empty(R, dtype=torch.int64, device="cuda"); lp = torch.empty(R, device="cuda")
        ld = torch.zeros(R, 64); ld[:, :K] = logits
        ld_d, q_d = dev(ld), dev(q)
        hip.check(hip.lib().cadre_sample(ld_d.data_ptr(), 64, q_d.data_ptr(), K, R, K, act.data_ptr(),
                                         lp.data_ptr(), hip.stream()), "sample")
        assert torch.equal(act.cpu(), want)                                   # bit-exact indices
        assert rel(lp, lgn.gather(1, want.view(-1, 1)).view(-1)) < 1e-5


def test_clip_adam(hip):
    from oracle import ppo_ref
    g = torch.Generator().manual_seed(2)
    sizes = [1000, 257, 4096, 33]
    off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    n = int(off[-1])
    p0 = torch.randn(n, generator=g)
    params = {"m%d" % i: {"w": p0[off[i]:off[i + 1]].clone()} for i in range(4)}
    adam = {k: {"w": (torch.zeros_like(v["w"]), torch.zeros_like(v["w"]))} for k, v in params.items()}
    pd = dev(p0.clone()); m = torch.zeros(n, device="cuda"); v = torch.zeros(n, device="cuda")
    nrm = torch.zeros(4, dtype=torch.float64, device="cuda"); offd = dev(off)
    for step in (1, 2, 3):
        gr = torch.randn(n, generator=g) * torch.repeat_interleave(torch.tensor([30.0, 0.01, 5.0, 100.0]),
                                                                   torch.tensor(sizes))
        grads = {"m%d" % i: {"w": gr[off[i]:off[i + 1]].clone()} for i in range(4)}
        ppo_ref.chief_step(params, grads, adam, step, lr=3e-4, max_grad_norm=250.0)
        gr_d = dev(gr)
        hip.check(hip.lib().cadre_clip_adam(pd.data_ptr(), gr_d.data_ptr(), m.data_ptr(), v.data_ptr(),
                                            offd.data_ptr(), 4, nrm.data_ptr(), 250.0, 3e-4, 0.9, 0.999, 1e-8, step,
                                            hip.stream()), "adam")
        want = torch.cat([params["m%d" % i]["w"] for i in range(4)])
        assert float((pd.cpu() - want).abs().max()) < 2e-7


def test_clip_adam_with_weight_packing_equals_adam_then_pack(hip):
    """cadre_clip_adam_pack_graph (the optimiser step writes the fragment-order copies of W_hh itself) against
    cadre_clip_adam_graph followed by cadre_pack_lstm_weights on the real arena layout: parameters, both Adam moments and
    both packed copies bit-identical over three steps (chief.py:13-21; models.py:139-152's weights of the next update)."""
    from cadre_amd.arena import PPOArena
    L = hip.lib()
    arenas = [PPOArena("cuda:0", 530, {"steer": 33, "throttle": 3}, 4) for _ in range(2)]
    g = torch.Generator(device="cuda").manual_seed(5)
    p0 = torch.randn(arenas[0].total, device="cuda", generator=g) * 0.05
    a0 = arenas[0]
    # zero padding of the arena stays zero: columns D..DP-1 of the LSTM matrices
    for z in range(a0.Z):
        for o in (a0.o_wih, a0.o_whh):
            p0[z * a0.size_L + o: z * a0.size_L + o + a0.H4 * a0.DP].view(a0.H4, a0.DP)[:, a0.D:] = 0
    n_pack = ((a0.D + 15) // 16) * 4 * (a0.DP // 16) * 256
    packs = [torch.zeros(2, a0.Z, n_pack, device="cuda") for _ in range(2)]
    for a in arenas:
        a.params.copy_(p0)
        a.ensure_adam()
    for step in range(3):
        gr = torch.randn(a0.total, device="cuda", generator=g) * (40.0 if step == 1 else 0.3)      # step 1: the clip bites
        for z in range(a0.Z):
            for o in (a0.o_wih, a0.o_whh):
                gr[z * a0.size_L + o: z * a0.size_L + o + a0.H4 * a0.DP].view(a0.H4, a0.DP)[:, a0.D:] = 0
        for a in arenas:
            a.grads.copy_(gr)
        a, wp = arenas[0], packs[0]
        hip.check(L.cadre_clip_adam_graph(hip.ptr(a.params), hip.ptr(a.grads), hip.ptr(a.exp_avg), hip.ptr(a.exp_avg_sq), hip.ptr(a.seg_off),
                                          2 * a.Z, hip.ptr(a.norms2), 250.0, 3e-4, 0.9, 0.999, 1e-8, hip.ptr(a.step_dev), hip.stream()), "adam")
        hip.check(L.cadre_pack_lstm_weights(hip.ptr(a.params[a.o_whh:]), a.size_L, a.DP, a.D, a.Z, hip.ptr(wp[0]), hip.ptr(wp[1]),
                                            wp.stride(1), hip.stream()), "pack")
        a, wp = arenas[1], packs[1]
        hip.check(L.cadre_clip_adam_pack_graph(hip.ptr(a.params), hip.ptr(a.grads), hip.ptr(a.exp_avg), hip.ptr(a.exp_avg_sq),
                                               hip.ptr(a.seg_off), 2 * a.Z, hip.ptr(a.norms2), 250.0, 3e-4, 0.9, 0.999, 1e-8,
                                               hip.ptr(a.step_dev), a.Z, a.size_L, a.o_whh, a.H4, a.DP, a.D, hip.ptr(wp[0]), hip.ptr(wp[1]),
                                               wp.stride(1), hip.stream()), "adam+pack")
        torch.cuda.synchronize()
        assert torch.equal(arenas[0].params, arenas[1].params), step
        assert torch.equal(arenas[0].exp_avg, arenas[1].exp_avg) and torch.equal(arenas[0].exp_avg_sq, arenas[1].exp_avg_sq)
        assert torch.equal(packs[0], packs[1]), step
    assert float(packs[1].abs().max()) > 0 and int(arenas[1].step_dev.item()) == 3


# ----------------------------------------------------------------------------- bf16 GEMM (config C3)
def _bf(x):
    return x.to(torch.bfloat16)


@pytest.mark.parametrize("M,N,K,tile", [(300, 256, 512, 1), (70, 64, 192, 3), (600, 128, 4608, 4), (200, 64, 576, 2),
                                        (300, 64, 576, 10), (600, 128, 320, 11),
                                        (700, 512, 1152, 7), (256, 256, 64, 7)])
def test_gemm_bf16_dense(hip, M, N, K, tile):
    g = torch.Generator().manual_seed(M + N)
    A, B = _bf(torch.randn(M, K, generator=g)), _bf(torch.randn(N, K, generator=g) * 0.1)
    scale = torch.rand(N, generator=g) + 0.5; shift = torch.randn(N, generator=g)
    resid = _bf(torch.randn(M, N, generator=g))
    want = F.relu((A.float() @ B.float().t()) * scale + shift + resid.float())
    out32 = torch.zeros(M, N, device="cuda")
    Ad, Bd, rd, sd_, hd_ = dev(A), dev(B), dev(resid), dev(scale), dev(shift)
    hip.gemm(Ad, Bd, out32, M, N, K, K, K, N, scale=sd_, shift=hd_, resid=rd, ldr=N, act=1, tile=tile, bf16=True, flags=4)
    assert rel(out32, want) < 2e-5                      # fp32 accumulate of exact bf16 products
    out16 = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)
    hip.gemm(Ad, Bd, out16, M, N, K, K, K, N, scale=sd_, shift=hd_, resid=rd, ldr=N, act=1, tile=tile, bf16=True, flags=6)
    assert rel(out16.float(), want) < 5e-3              # one bf16 rounding of the result


@pytest.mark.parametrize("M,N,K,split", [(300, 256, 4608, 2), (2048, 1536, 8192, 4), (77, 256, 128, 1), (513, 512, 4608, 3),
                                         (256, 768, 320, 4), (1000, 1536, 41472, 16)])
def test_gemm_bf16_w128_partial_sums_equal_the_tile_kernels(hip, M, N, K, split):
    """cadre_gemm_bf16_w128 (gemm_bf16_w128.hip, round 5: 256 x 256 tiles, one wave per SIMD, B streamed to registers in fragment
    order — the inter-task first layers, intertask_att.py:39-80) writes the SAME raw split-K partial sums, bit for bit, as
    cadre_gemm_bf16 with split_k (same slices, same k order): the encoder's results must not depend on which kernel a frame batch
    selects.  Their sum agrees with torch-CPU fp32.  Shapes: M tiles with a tail, one and several N tiles, slices of an odd and an
    even number of 64-element k-tiles, a slice of ONE k-tile and an EMPTY slice (K = 320, split 4: 2 + 2 + 1 + 0), the 288 x 288
    model's K = 41472 in 16 slices (41 k-tiles each, the last 33)."""
    from cadre_amd.encoder import _w128_dense_b
    g = torch.Generator().manual_seed(M + N + K)
    A, B = _bf(torch.randn(M, K, generator=g)), _bf(torch.randn(N, K, generator=g) * 0.1)
    assert hip.lib().cadre_gemm_bf16_w128_supported(M, N, K, K, N, split) == 1
    Ad, Bd = dev(A), dev(B)
    Bf = dev(_w128_dense_b(B.float())).to(torch.bfloat16)
    got = torch.full((split, M, N), float("nan"), device="cuda")
    hip.gemm_bf16_w128(Ad, Bf, got, M, N, K, K, N, split)
    ref = torch.full((split, M, N), float("nan"), device="cuda")
    if split > 1:
        hip.gemm(Ad, Bd, ref, M, N, K, K, K, N, split_k=split, bf16=True)
    else:
        hip.gemm(Ad, Bd, ref[0], M, N, K, K, K, N, bf16=True)
    torch.cuda.synchronize()
    assert not torch.isnan(got).any()
    assert torch.equal(got, ref)
    rows = slice(0, min(M, 96))
    want = A[rows].float() @ B.float().t()
    assert rel(got.sum(0)[rows], want) < 2e-5


def test_gemm_bf16_w128_repeatable_under_load(hip):
    """The inter-task shape of the 288 x 288 model at 2048 frames: 10 launches agree bit for bit (counted vmcnt + one raw barrier
    per k-tile of 64: a race shows up as run-to-run differences)."""
    from cadre_amd.encoder import _w128_dense_b
    M, N, K, split = 2048, 1536, 41472, 16
    g = torch.Generator(device="cuda").manual_seed(7)
    A = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
    B = (torch.randn(N, K, device="cuda", generator=g) * 0.05).to(torch.bfloat16)
    Bf = _w128_dense_b(B.float().cpu()).to(torch.bfloat16).cuda()
    outs = []
    for rep in range(10):
        o = torch.empty(split, M, N, device="cuda")
        hip.gemm_bf16_w128(A, Bf, o, M, N, K, K, N, split)
        outs.append(o.sum(0))
    torch.cuda.synchronize()
    assert all(torch.equal(outs[0], o) for o in outs[1:])
    want = A[:32].float().cpu() @ B.float().cpu().t()
    assert rel(outs[0][:32], want) < 2e-5


@pytest.mark.parametrize("Cin,Cout,H,W,k,s,p,tile", [(64, 64, 18, 22, 3, 1, 1, 0), (64, 128, 18, 22, 3, 2, 1, 0),
                                                      (64, 128, 17, 21, 1, 2, 0, 0), (512, 128, 9, 9, 3, 1, 1, 0),
                                                      (256, 256, 18, 18, 3, 1, 1, 7), (128, 512, 9, 9, 1, 1, 0, 7),
                                                      ab(64, 64, 18, 22, 3, 1, 1, 12), ab(128, 192, 11, 9, 3, 2, 1, 12)])
def test_conv_bf16(hip, Cin, Cout, H, W, k, s, p, tile):
    g = torch.Generator().manual_seed(Cin + Cout + k + 1)
    Nimg = 3
    x = _bf(torch.randn(Nimg, Cin, H, W, generator=g))
    w = _bf(torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5)
    scale = torch.rand(Cout, generator=g) + 0.5; shift = torch.randn(Cout, generator=g)
    y = F.conv2d(x.float(), w.float(), None, s, p)
    Ho, Wo = y.shape[2], y.shape[3]
    want = F.relu(y * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1))
    xd = dev(x.permute(0, 2, 3, 1).contiguous()); wd = dev(w.permute(0, 2, 3, 1).reshape(Cout, -1).contiguous())
    out = torch.empty(Nimg, Ho, Wo, Cout, device="cuda")
    K = k * k * Cin
    hip.gemm(xd, wd, out, Nimg * Ho * Wo, Cout, K, 0, K, Cout, a_mode=2, scale=dev(scale), shift=dev(shift), act=1,
             conv=(H, W, Cin, Ho, Wo, k, k, s, p), bf16=True, tile=tile)
    torch.cuda.synchronize()
    assert rel(out.permute(0, 3, 1, 2), want) < 2e-5


@pytest.mark.parametrize("tile", [3, 10, ab(12)])
def test_bf16_padded_stem(hip, tile):
    """cadre_gemm_bf16 a_mode 4: 7x7/s2 stem on the zero-padded bf16 NHWC4 image (encoder C3 path), every tile
    that serves N = 64 incl. the streamed kernel (several M-tiles per workgroup)."""
    from cadre_amd.encoder import _stem_rows_bf16
    g = torch.Generator().manual_seed(9)
    Nimg, H, W = 5, 46, 58
    x = _bf(torch.randn(Nimg, 4, H, W, generator=g)); x[:, 3] = 0
    w = _bf(torch.randn(64, 4, 7, 7, generator=g) / 14.0)
    scale = torch.rand(64, generator=g) + 0.5; shift = torch.randn(64, generator=g)
    y = F.conv2d(x.float(), w.float(), None, 2, 3)
    Ho, Wo = y.shape[2], y.shape[3]
    want = F.relu(y * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1))
    Hp, Wp = max(H + 6, (Ho - 1) * 2 + 8), max(W + 6, (Wo - 1) * 2 + 8)
    Wp += Wp & 1
    xp = torch.zeros(Nimg, Hp, Wp, 4)
    xp[:, 3:3 + H, 3:3 + W] = x.permute(0, 2, 3, 1)
    xd = dev(xp).to(torch.bfloat16)
    wd = dev(_stem_rows_bf16(w)).to(torch.bfloat16)
    out = torch.empty(Nimg, Ho, Wo, 64, device="cuda", dtype=torch.bfloat16)
    K = wd.shape[1]
    hip.gemm(xd, wd, out, Nimg * Ho * Wo, 64, K, 0, K, 64, a_mode=4, scale=dev(scale), shift=dev(shift), act=1,
             conv=(Hp, Wp, 4, Ho, Wo, 7, 7, 2, 0), bf16=True, flags=2, tile=tile)
    torch.cuda.synchronize()
    assert rel(out.float().permute(0, 3, 1, 2), want) < 6e-3          # bf16 output rounding


def test_f32_stem_bf16_output_and_bf16_pool(hip):
    g = torch.Generator().manual_seed(4)
    x = torch.randn(2, 4, 30, 36, generator=g); w = torch.randn(64, 4, 7, 7, generator=g) / 14.0
    y = F.relu(F.conv2d(x, w, None, 2, 3))
    from cadre_amd.encoder import _khwc
    xd = dev(x.permute(0, 2, 3, 1).contiguous()); wd = dev(_khwc(w))
    Ho, Wo = y.shape[2], y.shape[3]
    out = torch.empty(2, Ho, Wo, 64, device="cuda", dtype=torch.bfloat16)
    hip.gemm(xd, wd, out, 2 * Ho * Wo, 64, 224, 0, 224, 64, a_mode=3, act=1, conv=(30, 36, 4, Ho, Wo, 7, 7, 2, 3), flags=2)
    assert rel(out.float().permute(0, 3, 1, 2), y) < 5e-3
    want = F.max_pool2d(out.float().permute(0, 3, 1, 2), 3, 2, 1)
    p = torch.empty(2, want.shape[2], want.shape[3], 64, device="cuda", dtype=torch.bfloat16)
    hip.check(hip.lib().cadre_maxpool3x3s2_bf16(out.data_ptr(), p.data_ptr(), 2, Ho, Wo, 64, hip.stream()), "pool")
    assert torch.equal(p.float().permute(0, 3, 1, 2).cpu(), want.cpu())


# ----------------------------------------------------------------------------- row-sorted update support
def test_sort_rows_and_permute(hip):
    g = torch.Generator().manual_seed(12)
    B, C, S, ld = 96, 4, 3, 16
    cmds = torch.randint(0, C, (2, B), generator=g, dtype=torch.int32)
    cmds[1, :50] = 3
    cd = dev(cmds); pos = torch.empty(2, B, dtype=torch.int32, device="cuda"); seg = torch.empty(2 * C, 2, dtype=torch.int32, device="cuda")
    L = hip.lib()
    hip.check(L.cadre_sort_rows_by_command(cd.data_ptr(), B, C, pos.data_ptr(), seg.data_ptr(), hip.stream()), "sort")
    for hd in range(2):
        order = torch.sort(cmds[hd], stable=True).indices            # order[d] = source row of sorted slot d
        want_pos = torch.empty(B, dtype=torch.int64); want_pos[order] = torch.arange(B)
        assert torch.equal(pos[hd].cpu().long(), want_pos)
        cnt = torch.bincount(cmds[hd].long(), minlength=C)
        off = torch.cumsum(cnt, 0) - cnt
        assert torch.equal(seg[hd * C:(hd + 1) * C].cpu().long(), torch.stack([off, cnt], 1))
    # both heads in one launch: every per-head array is [2][...]
    X = torch.randn(2, S, B, ld, generator=g); h0 = torch.randn(2, B, ld, generator=g); c0 = torch.randn(2, B, ld, generator=g)
    sc = [torch.randint(0, 9, (2, B), generator=g), cmds.clone(), torch.randn(2, B, generator=g), torch.randn(2, B, generator=g),
          torch.randn(2, B, generator=g), torch.randn(2, B, generator=g)]
    ins = [dev(t) for t in [X, h0, c0] + sc]
    outs = [torch.zeros_like(t) for t in ins]
    hip.check(L.cadre_permute_minibatch(pos.data_ptr(), B, S, ins[0].data_ptr(), outs[0].data_ptr(), ld,
                                        ins[1].data_ptr(), ins[2].data_ptr(), outs[1].data_ptr(), outs[2].data_ptr(), ld,
                                        *[t.data_ptr() for t in ins[3:]], *[t.data_ptr() for t in outs[3:]], 2, S * B * ld, B * ld,
                                        hip.stream()), "permute")
    for hd in range(2):
        order = torch.sort(cmds[hd], stable=True).indices
        assert torch.equal(outs[0][hd].cpu(), X[hd][:, order]) and torch.equal(outs[1][hd].cpu(), h0[hd][order])
        assert torch.equal(outs[2][hd].cpu(), c0[hd][order]) and torch.equal(outs[3][hd].cpu(), sc[0][hd][order])
        assert torch.equal(outs[4][hd].cpu(), cmds[hd][order]) and torch.equal(outs[8][hd].cpu(), sc[5][hd][order])


@pytest.mark.parametrize("P,tile,BM,segs", [(128, 3, 64, [[0, 40], [40, 0], [40, 70], [110, 18]]),
                                            (128, 9, 32, [[0, 40], [40, 0], [40, 70], [110, 18]]),
                                            ab(128, 11, 32, [[0, 40], [40, 0], [40, 70], [110, 18]]),
                                            (96, 0, 32, [[0, 10], [10, 30], [40, 0], [40, 56]]),     # 32-row periods: auto -> tile 9
                                            (64, 0, 32, [[0, 16], [16, 16], [32, 31], [63, 1]])])
def test_gemm_row_segments(hip, P, tile, BM, segs):
    """seg_mode 1: only M tiles that intersect the batch entry's row segment are written;
    seg_mode 2: only k tiles (rows) of the segment are multiplied."""
    g = torch.Generator().manual_seed(21)
    Z, S, N, K = 4, 2, 96, 64
    seg = torch.tensor(segs, dtype=torch.int32)       # an empty net, ragged bounds
    A = torch.randn(Z, S * P, K, generator=g); W = torch.randn(Z, N, K, generator=g)
    out = torch.full((Z, S * P, N), 7.0, device="cuda")
    Ad, Wd, sd_ = dev(A), dev(W), dev(seg)
    hip.gemm(Ad, Wd, out, S * P, N, K, K, K, N, batch=Z, a_z=(1, 0, S * P * K), b_z=(1, 0, N * K), c_z=(1, 0, S * P * N),
             seg=(1, sd_, P, 1), tile=tile)
    full = torch.bmm(A, W.transpose(1, 2))
    o = out.cpu()
    for z in range(Z):
        b, c = int(seg[z, 0]), int(seg[z, 1])
        for t in range(S):
            for m0 in range(0, P, BM):
                rows = slice(t * P + m0, t * P + m0 + BM)
                owned = c > 0 and not (m0 + BM <= b or m0 >= b + c)
                if owned:
                    assert rel(o[z, rows], full[z, rows]) < 2e-5
                else:
                    assert float((o[z, rows] - 7.0).abs().max()) == 0.0               # skipped tile: untouched
    # K mode: dW[z] = sum over the segment's rows (k-tile granularity: rows outside must be zero in dY)
    dY = torch.randn(Z, S * P, N, generator=g)
    for z in range(Z):
        b, c = int(seg[z, 0]), int(seg[z, 1])
        mask = torch.zeros(P); mask[b:b + c] = 1
        dY[z] *= mask.repeat(S).view(-1, 1)                                           # callers keep foreign rows at 0
    X = torch.randn(S * P, K, generator=g)
    dW = torch.full((Z, N, K), 3.0, device="cuda")
    dYd, Xd = dev(dY), dev(X)
    hip.gemm(dYd, Xd, dW, N, K, S * P, N, K, K, a_mode=1, b_mode=1, batch=Z, a_z=(1, 0, S * P * N), b_z=(1, 1, 0),
             c_z=(1, 0, N * K), seg=(2, sd_, P, 1))
    want = torch.stack([dY[z].t() @ X for z in range(Z)])
    assert rel(dW, want) < 2e-5
    for z in range(Z):
        if int(seg[z, 1]) == 0:
            assert float(dW[z].abs().max()) == 0.0                                        # empty net: exact zeros


@pytest.mark.parametrize("P,tile,N,segs", [(64, 0, 2120, [[0, 13], [13, 22], [35, 11], [46, 18]]),
                                          (64, 3, 96, [[0, 0], [0, 64], [64, 0], [64, 0]]),
                                          (100, 9, 128, [[3, 1], [4, 37], [41, 59], [100, 0]]),       # period not a multiple of 32
                                          (256, 8, 2120, [[0, 70], [70, 58], [128, 49], [177, 79]])])
def test_gemm_compact_row_segments(hip, P, tile, N, segs):
    """seg_mode 3: the M index runs over the batch entry's own rows only, period after period (the update's LSTM
    input projection, models.py:148-151: rows [S][B] sorted by command): own rows = the full product (+ bias), every
    other row of C untouched, empty runs write nothing."""
    g = torch.Generator().manual_seed(23)
    Z, S, K = 4, 8, 544
    seg = torch.tensor(segs, dtype=torch.int32)
    A = torch.randn(S * P, K, generator=g); W = torch.randn(Z, N, K, generator=g) * 0.05; b = torch.randn(Z, N, generator=g)
    ldc = (N + 63) // 64 * 64
    own = torch.zeros(Z, P, dtype=torch.bool)
    for z in range(Z):
        own[z, int(seg[z, 0]):int(seg[z, 0]) + int(seg[z, 1])] = True
    out = torch.full((Z, S * P, ldc), 7.0, device="cuda")
    Ad, Wd, bd, sd_ = dev(A), dev(W), dev(b), dev(seg)
    hip.gemm(Ad, Wd, out, S * P, N, K, K, K, ldc, shift=bd, batch=Z, a_z=(1, 1, 0), b_z=(1, 0, N * K), c_z=(1, 0, S * P * ldc),
             s_z=(1, 0, N), seg=(3, sd_, P, 1), tile=tile)
    want = torch.einsum("mk,znk->zmn", A.double(), W.double()) + b.double()[:, None]
    o = out.cpu()
    for z in range(Z):
        rows = own[z].repeat(S)
        if bool(rows.any()):
            assert rel(o[z][rows][:, :N], want[z][rows]) < 2e-5, z
        assert float((o[z][~rows] - 7.0).abs().max() if bool((~rows).any()) else 0.0) == 0.0, z       # foreign rows: untouched
        assert float((o[z][:, N:] - 7.0).abs().max() if ldc > N else 0.0) == 0.0                       # padding columns too


@needs_ab
@pytest.mark.parametrize("B", [1, 24, 64, 256])
def test_gemm_skinny_update_shapes(hip, B):
    """Tile 11 (gemm_f32_skinny.hip: fragments straight from global memory, four K slices per workgroup summed in LDS)
    on the products of one LSTM step of update_policy (models.py:139-152): the recurrent forward
    gates = h W_hh^T + b_hh + gates (in place, row segments), the backward dh = dG W_hh (k-major B, K = 2120), and a
    ReLU tower layer — against float64, and against the tile kernel it replaces."""
    g = torch.Generator().manual_seed(5 + B)
    Z, D, H4, hid = 8, 544, 2120, 128
    # rows sorted by command: net z = head*4 + c owns one run of its head's B rows
    cuts = sorted(torch.randint(0, B + 1, (3,), generator=g).tolist())
    run = [(0, cuts[0]), (cuts[0], cuts[1] - cuts[0]), (cuts[1], cuts[2] - cuts[1]), (cuts[2], B - cuts[2])]
    seg = torch.tensor(run + run, dtype=torch.int32)
    sd_ = dev(seg)
    use_seg = B % 32 == 0
    sg = (1, sd_, B, 1) if use_seg else None
    h = torch.randn(Z, B, D, generator=g); W = torch.randn(Z, H4, D, generator=g) * 0.05
    b = torch.randn(Z, H4, generator=g); G0 = torch.randn(Z, B, H4, generator=g)
    hd, Wd, bd = dev(h), dev(W), dev(b)
    outs = {}
    for tile in (11, 9):
        G = dev(G0)
        hip.gemm(hd, Wd, G, B, H4, D, D, D, H4, shift=bd, resid=G, ldr=H4, batch=Z, a_z=(1, 0, B * D), b_z=(1, 0, H4 * D),
                 c_z=(1, 0, B * H4), s_z=(1, 0, H4), r_z=(1, 0, B * H4), seg=sg, tile=tile)
        outs[tile] = G.cpu()
    want = (torch.bmm(h.double(), W.double().transpose(1, 2)) + b.double()[:, None] + G0.double()).float()
    for z in range(Z):
        b0, c = (int(seg[z, 0]), int(seg[z, 1])) if use_seg else (0, B)
        rows = slice(b0, b0 + c)
        if c:
            assert rel(outs[11][z, rows], want[z, rows]) < 2e-5
            assert rel(outs[11][z, rows], outs[9][z, rows]) < 2e-5
        for m0 in range(0, B, 32):                                                  # tiles outside the run: untouched
            if use_seg and (c == 0 or m0 + 32 <= b0 or m0 >= b0 + c):
                assert torch.equal(outs[11][z, m0:m0 + 32], G0[z, m0:m0 + 32])
    # backward: dh = dG W_hh  ([B, 2120] x [2120, 544], k-major B)
    dG = torch.randn(Z, B, H4, generator=g)
    dGd = dev(dG)
    dH = torch.full((Z, B, D), 3.0, device="cuda")
    hip.gemm(dGd, Wd, dH, B, D, H4, H4, D, D, b_mode=1, batch=Z, a_z=(1, 0, B * H4), b_z=(1, 0, H4 * D), c_z=(1, 0, B * D),
             seg=sg, tile=11)
    wantH = torch.bmm(dG.double(), W.double()).float()
    for z in range(Z):
        b0, c = (int(seg[z, 0]), int(seg[z, 1])) if use_seg else (0, B)
        if c:
            assert rel(dH[z, b0:b0 + c].cpu(), wantH[z, b0:b0 + c]) < 2e-5
    # a tower layer: relu(x W1^T + b1), 16 towers reading 8 inputs (a_div 2)
    W1 = torch.randn(2 * Z, hid, D, generator=g) * 0.05; b1 = torch.randn(2 * Z, hid, generator=g)
    A1 = torch.empty(2 * Z, B, hid, device="cuda")
    hip.gemm(hd, dev(W1), A1, B, hid, D, D, D, hid, shift=dev(b1), act=1, batch=2 * Z, a_z=(2, 0, B * D), b_z=(1, 0, hid * D),
             c_z=(1, 0, B * hid), s_z=(1, 0, hid), seg=None if sg is None else (1, sd_, B, 2), tile=11)
    wantA = torch.relu(torch.bmm(h.double().repeat_interleave(2, 0), W1.double().transpose(1, 2)) + b1.double()[:, None]).float()
    for z in range(2 * Z):
        b0, c = (int(seg[z // 2, 0]), int(seg[z // 2, 1])) if use_seg else (0, B)
        if c:
            assert rel(A1[z, b0:b0 + c].cpu(), wantA[z, b0:b0 + c]) < 2e-5


def test_conv_decode_random_geometries(hip):
    """The im2col row decode (scalar division of the tile's first row + float-reciprocal carries) against
    F.conv2d on seeded random geometries: odd sizes, strides, 1-pixel-high / very wide maps, frame counts that
    leave partial tiles, every conv-capable tile; fp32 and bf16 kernels."""
    from cadre_amd.encoder import _khwc
    r = np.random.RandomState(1234)
    cases = [(1, 64, 64, 1, 700, 3, 1, 1), (5, 64, 128, 3, 2, 3, 1, 1), (2, 128, 64, 31, 17, 3, 2, 1), (7, 64, 64, 9, 9, 1, 1, 0)]
    for _ in range(14):
        k = int(r.choice([1, 3]))
        cases.append((int(r.randint(1, 7)), int(r.choice([64, 128])), int(r.choice([64, 96, 128])), int(r.randint(1, 40)),
                      int(r.randint(1, 80)), k, int(r.choice([1, 2])), k // 2))
    for ci, (Nimg, Cin, Cout, H, W, k, s, p) in enumerate(cases):
        g = torch.Generator().manual_seed(ci)
        x = torch.randn(Nimg, Cin, H, W, generator=g)
        w = torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5
        shift = torch.randn(Cout, generator=g)
        want = F.conv2d(x, w, shift, s, p)
        Ho, Wo = want.shape[2], want.shape[3]
        xd, wd, sh = dev(x.permute(0, 2, 3, 1).contiguous()), dev(_khwc(w)), dev(shift)
        K = wd.shape[1]
        for tile in (0, 2, 3, 8, 9, 10) + ((12,) if _has_ab() else ()):
            out = torch.full((Nimg, Ho, Wo, Cout), 9.0, device="cuda")
            hip.gemm(xd, wd, out, Nimg * Ho * Wo, Cout, K, 0, K, Cout, a_mode=2, shift=sh, conv=(H, W, Cin, Ho, Wo, k, k, s, p),
                     tile=tile)
            assert rel(out.permute(0, 3, 1, 2), want) < 2e-5, (cases[ci], tile)
        x16, w16 = xd.to(torch.bfloat16), wd.to(torch.bfloat16)
        want16 = F.conv2d(x16.float().permute(0, 3, 1, 2).cpu(), w.to(torch.bfloat16).float(), shift, s, p)
        for tile in (0, 1, 3) + ((12,) if K >= 128 and _has_ab() else ()):          # the streamed kernel needs two k-tiles
            out = torch.full((Nimg, Ho, Wo, Cout), 9.0, device="cuda")
            hip.gemm(x16, w16, out, Nimg * Ho * Wo, Cout, K, 0, K, Cout, a_mode=2, shift=sh, conv=(H, W, Cin, Ho, Wo, k, k, s, p),
                     tile=tile, bf16=True)
            assert rel(out.permute(0, 3, 1, 2), want16) < 1e-4, (cases[ci], tile, "bf16")


@pytest.mark.parametrize("F,H,W,use_resid,act", [(3, 72, 72, True, 1), (2, 21, 21, False, 1), (5, 7, 10, True, 0), (1, 1, 1, False, 1),
                                                 (40, 9, 13, True, 1), (300, 6, 6, False, 1)])
def test_winograd_c64_fused_matches_torch(hip, F, H, W, use_resid, act):
    """cadre_winograd_c64 (fused F(2x2,3x3): transforms + 16 plane products in one kernel, layer1 of the fp32 model) vs torch
    conv2d + folded BN + residual + ReLU (resnet.py:26-55) on maps the 2x2 tiles divide and do not divide, fewer tiles
    than one workgroup takes and many items per workgroup; same frames in a larger batch: same bits."""
    from cadre_amd.encoder import _winograd_u_c64
    g = torch.Generator().manual_seed(F * 100 + H)
    x = torch.randn(F, H, W, 64, generator=g)
    w = torch.randn(64, 64, 3, 3, generator=g) / 24.0
    sc, sh = 0.5 + torch.rand(64, generator=g), torch.randn(64, generator=g)
    res = torch.randn(F, H, W, 64, generator=g) if use_resid else None
    want = torch.nn.functional.conv2d(x.permute(0, 3, 1, 2), w, padding=1).permute(0, 2, 3, 1) * sc + sh
    if res is not None:
        want = want + res
    if act == 1:
        want = torch.relu(want)
    u, scd, shd = dev(_winograd_u_c64(w)), dev(sc), dev(sh)

    def run(xd, rd, Fb):
        out = torch.full((Fb, H, W, 64), 7.0, device="cuda")
        hip.check(hip.lib().cadre_winograd_c64(hip.ptr(xd), hip.ptr(u), hip.ptr(scd), hip.ptr(shd), hip.ptr(rd), hip.ptr(out), Fb, H, W, act,
                                               hip.stream()), "cadre_winograd_c64")
        return out
    got = run(dev(x), None if res is None else dev(res), F)
    assert rel(got.cpu(), want) < 2e-5, float((got.cpu() - want).abs().max())
    xb = torch.randn(F + 3, H, W, 64, generator=g)
    rb = torch.randn(F + 3, H, W, 64, generator=g) if use_resid else None
    xb[2:2 + F] = x
    if rb is not None:
        rb[2:2 + F] = res
    gb = run(dev(xb), None if rb is None else dev(rb), F + 3)
    assert torch.equal(gb[2:2 + F], got)


@pytest.mark.parametrize("use_resid", [False, True])
def test_winograd_c64_fused_repeatable_under_load(hip, use_resid):
    """The fused layer-1 kernel orders its weight DMA and patch loads with hand-counted s_waitcnt vmcnt(N) + raw barriers, issues its
    MFMAs as inline asm (the hazard recognizer does not see them: the wait states in front of the epilogue's accumulator reads are
    written by hand) and reads residuals requested inside the last MFMA block.  A miscount is a race: 20 launches with every CU
    walking several 64-tile items (both forms of the item's last counted wait) must agree bit for bit, and with torch."""
    from cadre_amd.encoder import _winograd_u_c64
    F, H, W = 320, 72, 72                                       # 6480 items over 256 workgroups: 25-26 items each
    g = torch.Generator(device="cuda").manual_seed(11 + int(use_resid))
    x = torch.randn(F, H, W, 64, device="cuda", generator=g)
    w = torch.randn(64, 64, 3, 3, device="cuda", generator=g) / 24.0
    sc = 0.5 + torch.rand(64, device="cuda", generator=g)
    sh = torch.randn(64, device="cuda", generator=g)
    res = torch.randn(F, H, W, 64, device="cuda", generator=g) if use_resid else None
    u = _winograd_u_c64(w.cpu()).cuda()
    outs = []
    for rep in range(20):
        out = torch.empty(F, H, W, 64, device="cuda")
        hip.check(hip.lib().cadre_winograd_c64(hip.ptr(x), hip.ptr(u), hip.ptr(sc), hip.ptr(sh), hip.ptr(res), hip.ptr(out), F, H, W, 1,
                                               hip.stream()), "cadre_winograd_c64")
        outs.append(out)
    torch.cuda.synchronize()
    assert all(torch.equal(outs[0], o) for o in outs[1:])
    for sl in (slice(0, 2), slice(F - 2, F)):                     # first and last items of the launch
        # reference = torch on the CPU in fp32, like every other kernel test (not MIOpen on the device: HIP vs HIP)
        ref = torch.nn.functional.conv2d(x[sl].cpu().permute(0, 3, 1, 2), w.cpu(), padding=1).permute(0, 2, 3, 1) * sc.cpu() + sh.cpu()
        if use_resid:
            ref = ref + res[sl].cpu()
        ref = torch.relu(ref)
        assert float((outs[0][sl].cpu() - ref).abs().max() / ref.abs().max()) < 2e-5


@pytest.mark.parametrize("F,H,W,Cin,N,use_resid,act", [(3, 9, 9, 64, 128, True, 1), (2, 18, 18, 32, 64, False, 1), (2, 7, 10, 16, 32, True, 17),
                                                     (1, 1, 1, 8, 8, False, 0), (5, 6, 5, 12, 20, True, 0)])
@pytest.mark.parametrize("m", [2, 3, 4, 6])
def test_winograd_conv3x3_matches_torch(hip, F, H, W, Cin, N, use_resid, act, m):
    """cadre_winograd_in -> batched cadre_gemm_f32 over the (m+2)^2 transform planes -> cadre_winograd_out, F(2x2,3x3) and
    F(3x3,3x3), vs torch conv2d fp32 on the reference layer's formulation (resnet.py:26-55: conv3x3 / s1 / p1 + folded BN +
    residual + ReLU), sizes the tiles do not divide (the last tile row / column is partly outside the map) and a 1x1 map;
    and the same frames inside a larger batch give the same bits."""
    from cadre_amd.encoder import _winograd_u
    g = torch.Generator().manual_seed(F * 100 + H)
    x = torch.randn(F, H, W, Cin, generator=g)
    w = torch.randn(N, Cin, 3, 3, generator=g) / (9 * Cin) ** 0.5
    sc, sh = 0.5 + torch.rand(N, generator=g), torch.randn(N, generator=g)
    res = torch.randn(F, H, W, N, generator=g) if use_resid else None
    want = torch.nn.functional.conv2d(x.permute(0, 3, 1, 2), w, padding=1).permute(0, 2, 3, 1) * sc + sh
    if res is not None and not (act & 16):
        want = want + res
    if (act & 15) == 1:
        want = torch.relu(want)
    if res is not None and (act & 16):
        want = want + res

    def run(xd, rd, Fb):
        P, T = (m + 2) ** 2, Fb * -(-H // m) * -(-W // m)
        V = torch.full((P, T, Cin), 7.0, device="cuda")
        Mx = torch.full((P, T, N), 7.0, device="cuda")
        out = torch.full((Fb, H, W, N), 7.0, device="cuda")
        L = hip.lib()
        hip.check(L.cadre_winograd_in(hip.ptr(xd), hip.ptr(V), Fb, H, W, Cin, m, hip.stream()), "in")
        hip.gemm(V, u, Mx, T, N, Cin, Cin, Cin, N, batch=P, a_z=(1, P, T * Cin), b_z=(1, P, N * Cin), c_z=(1, P, T * N))
        hip.check(L.cadre_winograd_out(hip.ptr(Mx), hip.ptr(scd), hip.ptr(shd), hip.ptr(rd), hip.ptr(out), Fb, H, W, N, act, m, hip.stream()), "out")
        return out
    u, scd, shd = dev(_winograd_u(w, m)), dev(sc), dev(sh)
    got = run(dev(x), None if res is None else dev(res), F)
    # (F(6x6): 4-5x the rounding error of F(4x4) — tools/dbg/wino_points.py; its own bar)
    assert rel(got.cpu(), want) < (1e-4 if m == 6 else 2e-5), float((got.cpu() - want).abs().max())
    big = 7
    xb = torch.randn(big, H, W, Cin, generator=g)
    rb = torch.randn(big, H, W, N, generator=g) if use_resid else None
    xb[2:2 + F] = x
    if rb is not None:
        rb[2:2 + F] = res
    gb = run(dev(xb), None if rb is None else dev(rb), big)
    assert torch.equal(gb[2:2 + F], got) or F + 2 > big


def _wino_fused_run(hip, xd, u, scd, shd, rd, Fb, H, W, Cin, N, act, m):
    L = hip.lib()
    assert L.cadre_winograd_fused_capable(Fb, H, W, Cin, N, m) == 1
    V = torch.full((int(L.cadre_winograd_frag_elems(Fb, H, W, Cin, m)),), float("nan"), device="cuda")
    out = torch.full((Fb, H, W, N), 7.0, device="cuda")
    hip.winograd_fused(xd, V, u, scd, shd, rd, out, Fb, H, W, Cin, N, act, m)
    return out


@pytest.mark.parametrize("F,H,W,Cin,N,use_resid,act", [(3, 9, 9, 64, 128, True, 1), (2, 18, 18, 32, 64, False, 1), (2, 7, 10, 32, 32, True, 17),
                                                     (1, 1, 1, 32, 32, False, 0), (5, 6, 5, 96, 160, True, 0), (9, 36, 36, 128, 128, True, 1)])
@pytest.mark.parametrize("m", [2, 3, 4])
def test_winograd_fused_matches_torch(hip, F, H, W, Cin, N, use_resid, act, m):
    """cadre_winograd_in_frag -> cadre_winograd_gemm_out (all (m+2)^2 plane products and the inverse transform in one kernel,
    csrc/winograd_fused.hip) vs torch conv2d fp32 on the reference layer's formulation (resnet.py:26-55: conv3x3 / s1 / p1 + folded
    BN + residual + ReLU): maps the tiles do not divide, a 1x1 map, tile counts that are no multiple of 64 (V is padded with NaN
    here: a padding tile must never reach a stored output), several items per tile block; the same frames inside a larger batch
    give the same bits (the latent cache rests on it)."""
    from cadre_amd.encoder import _winograd_u_frag
    g = torch.Generator().manual_seed(F * 100 + H + m)
    x = torch.randn(F, H, W, Cin, generator=g)
    w = torch.randn(N, Cin, 3, 3, generator=g) / (9 * Cin) ** 0.5
    sc, sh = 0.5 + torch.rand(N, generator=g), torch.randn(N, generator=g)
    res = torch.randn(F, H, W, N, generator=g) if use_resid else None
    want = torch.nn.functional.conv2d(x.permute(0, 3, 1, 2), w, padding=1).permute(0, 2, 3, 1) * sc + sh
    if res is not None and not (act & 16):
        want = want + res
    if (act & 15) == 1:
        want = torch.relu(want)
    if res is not None and (act & 16):
        want = want + res
    u, scd, shd = dev(_winograd_u_frag(w, m)), dev(sc), dev(sh)
    got = _wino_fused_run(hip, dev(x), u, scd, shd, None if res is None else dev(res), F, H, W, Cin, N, act, m)
    assert torch.isfinite(got).all()
    assert rel(got.cpu(), want) < 2e-5, float((got.cpu() - want).abs().max())
    big = F + 5
    xb = torch.randn(big, H, W, Cin, generator=g)
    rb = torch.randn(big, H, W, N, generator=g) if use_resid else None
    xb[2:2 + F] = x
    if rb is not None:
        rb[2:2 + F] = res
    gb = _wino_fused_run(hip, dev(xb), u, scd, shd, None if rb is None else dev(rb), big, H, W, Cin, N, act, m)
    assert torch.equal(gb[2:2 + F], got)


@pytest.mark.parametrize("m,H,Cin,N,F", [(4, 36, 128, 128, 80), (3, 18, 256, 256, 150), (3, 9, 512, 128, 500)])
def test_winograd_fused_persistent_items_and_repeatable_under_load(hip, m, H, Cin, N, F):
    """More items than workgroups (every workgroup walks several (tile block, channel block) items: requests run on across item
    boundaries, the epilogue's stores and residual loads share the queue with them) on the encoder's own layer shapes, against
    torch-CPU fp32 on sampled frames; ten launches with every CU loaded give the same bits (hand-counted vmcnt + raw barriers)."""
    from cadre_amd.encoder import _winograd_u_frag
    g = torch.Generator().manual_seed(m * 1000 + H)
    x = torch.randn(F, H, H, Cin, generator=g)
    w = torch.randn(N, Cin, 3, 3, generator=g) / (9 * Cin) ** 0.5
    sc, sh = 0.5 + torch.rand(N, generator=g), torch.randn(N, generator=g)
    res = torch.randn(F, H, H, N, generator=g)
    T = F * (-(-H // m)) ** 2
    assert -(-T // 64) * (N // 32) > 256
    u, scd, shd, xd, rd = dev(_winograd_u_frag(w, m)), dev(sc), dev(sh), dev(x), dev(res)
    outs = [_wino_fused_run(hip, xd, u, scd, shd, rd, F, H, H, Cin, N, 1, m) for _ in range(10)]
    torch.cuda.synchronize()
    for o in outs[1:]:
        assert torch.equal(o, outs[0])
    # the same two frames alone: few tiles -> the 16-tile / 128-thread item shape (NTB = 1) instead of 64 tiles / 512 threads;
    # every (tile, channel) is summed in the same order by either shape: the bits must not depend on the batch
    alone = _wino_fused_run(hip, xd[F - 2:].contiguous(), u, scd, shd, rd[F - 2:].contiguous(), 2, H, H, Cin, N, 1, m)
    assert torch.equal(alone, outs[0][F - 2:])
    for sl in (slice(0, 2), slice(F // 2, F // 2 + 2), slice(F - 2, F)):
        ref = torch.nn.functional.conv2d(x[sl].permute(0, 3, 1, 2), w, padding=1).permute(0, 2, 3, 1) * sc + sh + res[sl]
        ref = torch.relu(ref)
        assert float((outs[0][sl].cpu() - ref).abs().max() / ref.abs().max()) < 2e-5


def test_winograd_fused_rejects_bad_arguments(hip):
    L = hip.lib()
    assert L.cadre_winograd_fused_capable(4, 36, 36, 128, 128, 4) == 1
    assert L.cadre_winograd_fused_capable(4, 36, 36, 100, 128, 4) == 0      # Cin % 32
    assert L.cadre_winograd_fused_capable(4, 36, 36, 128, 48, 4) == 0       # N % 32
    assert L.cadre_winograd_fused_capable(4, 36, 36, 128, 128, 5) == 0
    assert L.cadre_winograd_fused_capable(4096, 36, 36, 128, 128, 4) == 0    # V past 2 GiB
    assert L.cadre_winograd_fused_supported(4, 36, 36, 100, 128, 4) == 0    # (policy never says yes where the kernels cannot)
    x = torch.zeros(1, 4, 4, 32, device="cuda")
    V = torch.zeros(int(L.cadre_winograd_frag_elems(1, 4, 4, 32, 2)), device="cuda")
    with pytest.raises(hip.CadreHipError):
        hip.check(L.cadre_winograd_gemm_out(hip.ptr(V), hip.ptr(x), None, None, None, hip.ptr(x), 1, 4, 4, 32, 48, 0, 2, hip.stream()), "bad N")
    with pytest.raises(hip.CadreHipError):
        hip.check(L.cadre_winograd_in_frag(hip.ptr(x), None, 1, 4, 4, 32, 2, hip.stream()), "null V")


@pytest.mark.parametrize("H,W,F", [(84, 84, 3), (144, 256, 2), (288, 288, 2)])
def test_fused_stem_pool_exact_bf16_pieces(hip, H, W, F, monkeypatch):
    """cadre_stem_pool mode 2 (CADRE_STEM_EXACT_BF16=1): the fp32 front on the bf16 matrix cores — pixel bytes are exact in bf16, every
    fp32 weight is the exact sum of three bf16 pieces (asserted on the host), products exact, sums fp32.  Against torch-CPU fp32 at the
    fp32 kernel's bar, against the v_mfma_f32 kernel at accumulation-order distance, band decomposition without a bit of influence."""
    import torch.nn.functional as Fn
    from cadre_amd import synth
    from cadre_amd.encoder import DANetEncoderHIP, _stem_taps_x3
    sd = synth.encoder_state(*synth.feat_hw(H, W), 7)
    pcs = _stem_taps_x3(torch.as_tensor(sd["backbone.conv1.weight"]).float())
    flat = torch.as_tensor(sd["backbone.conv1.weight"]).float().permute(0, 2, 3, 1).reshape(64, -1)
    assert torch.equal(pcs.float().sum(0)[:, :196], flat) and float(pcs.float().sum(0)[:, 196:].abs().max()) == 0.0
    r = np.random.RandomState(H + W)
    rgb = r.randint(0, 256, (F, H, W, 3)).astype(np.uint8)
    route = ((r.rand(F, W, H) < 0.15) * 255).astype(np.uint8)
    rgb_d, route_d = torch.from_numpy(rgb).cuda(), torch.from_numpy(route).cuda()
    pools = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("CADRE_STEM_EXACT_BF16", mode)
        enc = DANetEncoderHIP(sd, H, W, "cuda:0", max_frames=64)
        assert enc.fused_stem and enc.stem_x3 == (mode == "1")
        taps = {}
        enc.forward_nhwc(enc.preprocess(rgb_d, route_d), taps=taps)
        pools[mode] = taps["pool"].float().cpu().clone()
        if mode == "1":
            big = 48
            rgb_b = torch.from_numpy(r.randint(0, 256, (big, H, W, 3)).astype(np.uint8)).cuda()
            route_b = torch.from_numpy(((r.rand(big, W, H) < 0.15) * 255).astype(np.uint8)).cuda()
            rgb_b[5:5 + F], route_b[5:5 + F] = rgb_d, route_d
            taps2 = {}
            enc.forward_nhwc(enc.preprocess(rgb_b, route_b), taps=taps2)
            assert torch.equal(taps2["pool"][5:5 + F].float().cpu(), pools["1"])
    x = np.zeros((F, 4, H, W), np.float32)
    x[:, :3] = (rgb.transpose(0, 3, 1, 2) / 255.).astype(np.float32)
    rt = route.copy()
    for i in range(F):
        if rt[i].max() > 0:
            rt[i] = (1.0 * rt[i] / rt[i].max()).astype(np.uint8)
    x[:, 3] = rt.transpose(0, 2, 1).astype(np.float32)
    t = lambda k: torch.as_tensor(sd[k]).float()
    y = Fn.conv2d(torch.from_numpy(x), t("backbone.conv1.weight"), t("backbone.conv1.bias"), stride=2, padding=3)
    y = Fn.batch_norm(y, t("backbone.bn1.running_mean"), t("backbone.bn1.running_var"), t("backbone.bn1.weight"),
                      t("backbone.bn1.bias"), False, 0.0, 1e-5)
    want = Fn.max_pool2d(torch.relu(y), 3, 2, 1).permute(0, 2, 3, 1)
    e_ref = float((pools["1"] - want).abs().max() / want.abs().max())
    e_f32 = float((pools["1"] - pools["0"]).abs().max() / want.abs().max())
    print("exact-bf16 front %dx%d: rel-max-err vs torch fp32 %.2e, vs the v_mfma_f32 kernel %.2e" % (H, W, e_ref, e_f32))
    assert e_ref < 2e-5 and e_f32 < 2e-6


@pytest.mark.parametrize("H,W,F", [(84, 84, 3), (144, 256, 2), (288, 288, 2)])
@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_fused_stem_pool(hip, H, W, F, dtype):
    """cadre_pack_obs + cadre_stem_pool (LUT -> conv 7x7/s2 + BN + ReLU -> max-pool 3x3/s2 in one kernel) vs torch-CPU
    fp32 on the reference's own formulation (agent.py:46, resnet.py:111-115,168-172); the band decomposition
    (which wave computes which pooled rows) must not change a single bit."""
    import torch.nn.functional as Fn
    from cadre_amd import synth
    from cadre_amd.encoder import DANetEncoderHIP
    sd = synth.encoder_state(*synth.feat_hw(H, W), 7)
    enc = DANetEncoderHIP(sd, H, W, "cuda:0", max_frames=64, dtype=dtype)
    assert enc.fused_stem
    r = np.random.RandomState(H + W)
    rgb = r.randint(0, 256, (F, H, W, 3)).astype(np.uint8)
    route = ((r.rand(F, W, H) < 0.15) * 255).astype(np.uint8)
    route[F - 1] = 0                                                      # a frame whose route max is 0
    rgb_d, route_d = torch.from_numpy(rgb).cuda(), torch.from_numpy(route).cuda()
    rn = torch.empty_like(route_d)
    taps = {}
    enc.forward_nhwc(enc.preprocess(rgb_d, route_d, rn), taps=taps)
    got = taps["pool"].float().cpu()
    # reference
    x = np.zeros((F, 4, H, W), np.float32)
    x[:, :3] = (rgb.transpose(0, 3, 1, 2) / 255.).astype(np.float32)
    rt = route.copy()
    for i in range(F):
        if rt[i].max() > 0:
            rt[i] = (1.0 * rt[i] / rt[i].max()).astype(np.uint8)          # agent.py:51-54 quirk
    x[:, 3] = rt.transpose(0, 2, 1).astype(np.float32)
    assert np.array_equal(rn.cpu().numpy(), rt)
    w = torch.from_numpy(sd["backbone.conv1.weight"]) if not isinstance(sd["backbone.conv1.weight"], torch.Tensor) else sd["backbone.conv1.weight"]
    t = lambda k: torch.as_tensor(sd[k]).float()
    y = Fn.conv2d(torch.from_numpy(x), t("backbone.conv1.weight"), t("backbone.conv1.bias"), stride=2, padding=3)
    y = Fn.batch_norm(y, t("backbone.bn1.running_mean"), t("backbone.bn1.running_var"), t("backbone.bn1.weight"),
                      t("backbone.bn1.bias"), False, 0.0, 1e-5)
    want = Fn.max_pool2d(torch.relu(y), 3, 2, 1).permute(0, 2, 3, 1)
    err = float((got - want).abs().max() / want.abs().max())
    print("fused front %dx%d %s: rel-max-err %.2e" % (H, W, dtype, err))
    assert tuple(got.shape) == tuple(want.shape)
    assert err < (2e-5 if dtype == "f32" else 2e-2)
    # band invariance: the same frames inside a larger batch (fewer bands per frame) -> identical bits
    big = 48
    rgb_b = torch.from_numpy(r.randint(0, 256, (big, H, W, 3)).astype(np.uint8)).cuda()
    route_b = torch.from_numpy(((r.rand(big, W, H) < 0.15) * 255).astype(np.uint8)).cuda()
    rgb_b[5:5 + F], route_b[5:5 + F] = rgb_d, route_d
    first = taps["pool"].clone()
    taps2 = {}
    enc.forward_nhwc(enc.preprocess(rgb_b, route_b), taps=taps2)
    assert torch.equal(taps2["pool"][5:5 + F], first)


@needs_ab
@pytest.mark.parametrize("F,H,W,use_resid,relu", [(3, 18, 22, False, 1), (2, 72, 72, True, 1), (5, 9, 9, True, 0), (70, 21, 21, True, 1)])
def test_conv3x3_c64_bf16(hip, F, H, W, use_resid, relu):
    """Autonomous-wave stage-1 conv (LDS-DMA ring, resident weights) vs torch fp32 on the same bf16 operands and vs the
    generic cadre_gemm_bf16 implicit-GEMM path (same k order and epilogue arithmetic: expected bit-identical)."""
    r = np.random.RandomState(F * 1000 + H)
    x = torch.from_numpy(r.standard_normal((F, H, W, 64)).astype(np.float32)).cuda().bfloat16()
    w = torch.from_numpy((r.standard_normal((64, 3, 3, 64)) * 0.05).astype(np.float32)).cuda().bfloat16()
    sc = torch.from_numpy((0.5 + r.rand(64)).astype(np.float32)).cuda()
    sh = torch.from_numpy(r.standard_normal(64).astype(np.float32)).cuda()
    res = torch.from_numpy(r.standard_normal((F, H, W, 64)).astype(np.float32)).cuda().bfloat16() if use_resid else None
    out = torch.full((F, H, W, 64), 7.0, device="cuda", dtype=torch.bfloat16)
    hip.conv3x3_c64_bf16(x, w.reshape(64, 576), sc, sh, res, out, F, H, W, relu)
    ref = torch.nn.functional.conv2d(x.float().permute(0, 3, 1, 2).cpu(), w.float().permute(0, 3, 1, 2).cpu(), padding=1)
    ref = ref.permute(0, 2, 3, 1) * sc.cpu() + sh.cpu()
    if use_resid:
        ref = ref + res.float().cpu()
    if relu:
        ref = torch.relu(ref)
    err = float((out.float().cpu() - ref).abs().max() / ref.abs().max())
    out2 = torch.empty_like(out)
    M = F * H * W
    hip.gemm(x, w.reshape(64, 576), out2, M, 64, 576, 0, 576, 64, a_mode=2, scale=sc, shift=sh, resid=res, ldr=64,
             act=relu, conv=(H, W, 64, H, W, 3, 3, 1, 1), bf16=True, flags=2 | (4 if use_resid else 0))
    same = bool(torch.equal(out, out2))
    print("conv3x3_c64 F=%d %dx%d: rel-max-err vs torch %.2e, bit-identical to cadre_gemm_bf16: %s" % (F, H, W, err, same))
    assert err < 1.5e-2          # one bf16 rounding of the output
    assert float((out.float() - out2.float()).abs().max()) <= 2.0 ** -7 * float(ref.abs().max())


def test_div255_arithmetic_is_the_reference_table(hip):
    """agent.py:46 `rgb / 255.` (double division, float32 store): the fused front's multiply + Newton correction
    must reproduce it for every byte."""
    lut = torch.from_numpy((np.arange(256) / 255.).astype(np.float32)).cuda()
    bad = torch.full((1,), -1, dtype=torch.int32, device="cuda")
    hip.check(hip.lib().cadre_div255_selfcheck(hip.ptr(lut), hip.ptr(bad), hip.stream()), "cadre_div255_selfcheck")
    assert int(bad.item()) == 0


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("F,H,W,Cin,N,use_resid,act", [
    (2, 18, 22, 64, 64, False, 1), (1, 72, 72, 64, 64, True, 1), (2, 36, 36, 128, 128, True, 1), (3, 18, 18, 256, 256, True, 1),
    (5, 9, 9, 512, 512, False, 1), (5, 9, 9, 512, 128, False, 1), (7, 9, 9, 128, 128, True, 1 | 16), (2, 21, 21, 64, 96, True, 0),
    (40, 9, 9, 128, 256, True, 1), (260, 9, 9, 256, 128, True, 1), (9, 30, 26, 64, 64, True, 1), (3, 50, 50, 128, 128, False, 1),
    (3, 60, 60, 128, 128, True, 1), (2, 36, 36, 64, 128, True, 1), (2, 30, 30, 128, 64, True, 1), (2, 20, 20, 128, 192, False, 0)])
def test_conv3x3_ring(hip, dtype, F, H, W, Cin, N, use_resid, act):
    """cadre_conv3x3_ring (window of pixels resident in LDS, nine taps as row offsets, weights streamed by LDS-DMA)
    vs torch fp32 conv2d on the same operands: every trunk / head shape class, residual before and after the ReLU,
    an N that is not a multiple of the tile, several M tiles and work items per workgroup."""
    from cadre_amd.encoder import _ring_w
    bf = dtype == "bf16"
    td = torch.bfloat16 if bf else torch.float32
    r = np.random.RandomState(F * 131 + H * 7 + Cin + N)
    x = torch.from_numpy(r.standard_normal((F, H, W, Cin)).astype(np.float32)).cuda().to(td)
    w = torch.from_numpy((r.standard_normal((N, Cin, 3, 3)) * (1.5 / np.sqrt(9 * Cin))).astype(np.float32)).to(td)
    sc = torch.from_numpy((0.5 + r.rand(N)).astype(np.float32)).cuda()
    sh = torch.from_numpy(r.standard_normal(N).astype(np.float32)).cuda()
    res = torch.from_numpy(r.standard_normal((F, H, W, N)).astype(np.float32)).cuda().to(td) if use_resid else None
    out = torch.full((F, H, W, N), 7.0, device="cuda", dtype=td)
    wr = _ring_w(w.float(), 64 if bf else 32).to(td).cuda()
    hip.conv3x3_ring(x, wr, sc, sh, res, out, F, H, W, Cin, N, act)
    ref = torch.nn.functional.conv2d(x.float().permute(0, 3, 1, 2).cpu(), w.float(), padding=1).permute(0, 2, 3, 1) * sc.cpu() + sh.cpu()
    if use_resid and not (act & 16):
        ref = ref + res.float().cpu()
    if act & 1:
        ref = torch.relu(ref)
    if use_resid and (act & 16):
        ref = ref + res.float().cpu()
    err = float((out.float().cpu() - ref).abs().max() / ref.abs().max())
    print("conv3x3_ring %s F=%d %dx%d %d->%d: rel-max-err %.2e" % (dtype, F, H, W, Cin, N, err))
    assert err < (1.5e-2 if bf else 2e-5)
    if not bf:                                   # fp32 output of the bf16 model's head convs (conv5a / conv5c feed PAM / CAM in fp32)
        return
    out32 = torch.empty((F, H, W, N), device="cuda", dtype=torch.float32)
    if not use_resid:
        hip.conv3x3_ring(x, wr, sc, sh, None, out32, F, H, W, Cin, N, act)
        assert float((out32.cpu() - ref).abs().max() / ref.abs().max()) < 1.5e-2


@pytest.mark.parametrize("dtype,F,H,W,Cin,N", [("bf16", 192, 36, 36, 128, 128), ("bf16", 256, 18, 18, 256, 256),
                                               ("f32", 48, 72, 72, 64, 64), ("bf16", 48, 72, 72, 64, 64)])
def test_conv3x3_ring_and_c64_repeatable_under_load(hip, dtype, F, H, W, Cin, N):
    """The LDS-DMA kernels order their staging with hand-counted s_waitcnt vmcnt(N) and raw barriers: a miscount is a
    race, and a race shows up as run-to-run differences once every CU is busy with several tiles.  20 launches on the
    same operands (many persistent items per workgroup, all CUs loaded) must agree bit for bit, and with torch."""
    from cadre_amd.encoder import _ring_w
    bf = dtype == "bf16"
    td = torch.bfloat16 if bf else torch.float32
    g = torch.Generator(device="cuda").manual_seed(F + Cin)
    x = torch.randn(F, H, W, Cin, device="cuda", generator=g).to(td)
    w = (torch.randn(N, Cin, 3, 3, device="cuda", generator=g) * (1.5 / np.sqrt(9 * Cin))).to(td)
    sc = 0.5 + torch.rand(N, device="cuda", generator=g)
    sh = torch.randn(N, device="cuda", generator=g)
    res = torch.randn(F, H, W, N, device="cuda", generator=g).to(td)
    wr = _ring_w(w.float().cpu(), 64 if bf else 32).to(td).cuda()
    outs = []
    for rep in range(20):
        out = torch.empty(F, H, W, N, device="cuda", dtype=td)
        hip.conv3x3_ring(x, wr, sc, sh, res, out, F, H, W, Cin, N, 1)
        outs.append(out)
    torch.cuda.synchronize()
    assert all(torch.equal(outs[0], o) for o in outs[1:])
    # reference = torch on the CPU in fp32 (not MIOpen on the device)
    ref = torch.nn.functional.conv2d(x[:4].float().cpu().permute(0, 3, 1, 2), w.float().cpu(), padding=1).permute(0, 2, 3, 1) * sc.cpu() + sh.cpu()
    ref = torch.relu(ref + res[:4].float().cpu())
    assert float((outs[0][:4].float().cpu() - ref).abs().max() / ref.abs().max()) < (1.5e-2 if bf else 2e-5)
    if bf and Cin == 64 and N == 64 and _has_ab():
        w_khwc = w.permute(0, 2, 3, 1).reshape(N, 576).contiguous()
        outs2 = []
        for rep in range(20):
            out = torch.empty(F, H, W, N, device="cuda", dtype=td)
            hip.conv3x3_c64_bf16(x, w_khwc, sc, sh, res, out, F, H, W, 1)
            outs2.append(out)
        torch.cuda.synchronize()
        assert all(torch.equal(outs2[0], o) for o in outs2[1:])
        assert float((outs2[0][:4].float().cpu() - ref).abs().max() / ref.abs().max()) < 1.5e-2


# ---------------------------------------------------------------------------------------------------------------
# stride-2 plane-window conv (csrc/conv3x3_s2.hip, round 5): resnet.py:26-55 conv1 of the down-sampling blocks
@pytest.mark.parametrize("Nimg,H,W,Cin,Cout,act", [(3, 18, 18, 64, 128, 1), (2, 36, 36, 128, 256, 1), (5, 8, 12, 64, 64, 0),
                                                   (1, 2, 2, 64, 32, 1), (7, 10, 6, 256, 160, 1), (2, 72, 72, 64, 128, 1),
                                                   (9, 6, 4, 192, 512, 0)])
def test_conv3x3_s2_matches_torch(hip, Nimg, H, W, Cin, Cout, act):
    """cadre_conv3x3_s2 vs torch-CPU fp32 conv2d(stride 2, pad 1) + folded BN + ReLU on bf16-rounded operands: maps whose
    rows are shorter / longer than an 8-pixel DMA piece, M tiles that straddle rows and frames (top / left halo masks inside
    a tile), 1-4 channel chunks, N tiles that are not full (160 = 128 + 32), a single 1 x 1 output map."""
    from cadre_amd.encoder import _s2_w
    g = torch.Generator().manual_seed(Nimg * 1000 + H * 10 + Cin + Cout)
    x = _bf(torch.randn(Nimg, Cin, H, W, generator=g))
    scale = torch.rand(Cout, generator=g) + 0.5; shift = torch.randn(Cout, generator=g)
    # folded BN: the scale goes into the weight rows (one rounding to bf16, as cadre_amd/encoder.py does), the kernel adds the shift
    w = _bf(torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5 * scale.view(-1, 1, 1, 1))
    y = F.conv2d(x.float(), w.float(), None, 2, 1)
    want = y + shift.view(1, -1, 1, 1)
    if act:
        want = F.relu(want)
    Ho, Wo = H // 2, W // 2
    assert tuple(y.shape[2:]) == (Ho, Wo)
    assert hip.lib().cadre_conv3x3_s2_supported(Nimg, H, W, Cin, Cout) == 1
    xd = dev(x.permute(0, 2, 3, 1).contiguous())
    wd = dev(_s2_w(w.float())).to(torch.bfloat16)
    out = torch.full((Nimg, Ho, Wo, Cout), float("nan"), device="cuda", dtype=torch.bfloat16)
    hip.conv3x3_s2(xd, wd, None, dev(shift), out, Nimg, H, W, Cin, Cout, act)
    torch.cuda.synchronize()
    assert not torch.isnan(out.float()).any()
    assert rel(out.float().permute(0, 3, 1, 2), want) < 6e-3           # bf16 output rounding (2^-9 relative) on exact-input sums


def test_conv3x3_s2_full_size_repeatable_under_load(hip):
    """The three trunk shapes at many frames (several persistent items per workgroup, every CU loaded): 10 launches agree
    bit for bit (hand-counted vmcnt + raw barriers: a race shows up as run-to-run differences), first and last frames agree
    with torch-CPU fp32."""
    from cadre_amd.encoder import _s2_w
    for F_, H, Cin, Cout in ((192, 72, 64, 128), (256, 36, 128, 256), (512, 18, 256, 512)):
        g = torch.Generator(device="cuda").manual_seed(H + Cin)
        x = torch.randn(F_, H, H, Cin, device="cuda", generator=g).to(torch.bfloat16)
        w = (torch.randn(Cout, Cin, 3, 3, device="cuda", generator=g) * (1.5 / np.sqrt(9 * Cin))).to(torch.bfloat16)
        sh = torch.randn(Cout, device="cuda", generator=g)
        wd = _s2_w(w.float().cpu()).to(torch.bfloat16).cuda()
        outs = []
        for rep in range(10):
            out = torch.empty(F_, H // 2, H // 2, Cout, device="cuda", dtype=torch.bfloat16)
            hip.conv3x3_s2(x, wd, None, sh, out, F_, H, H, Cin, Cout, 1)
            outs.append(out)
        torch.cuda.synchronize()
        assert all(torch.equal(outs[0], o) for o in outs[1:]), (H, Cin)
        for sl in (slice(0, 2), slice(F_ - 2, F_)):
            ref = F.conv2d(x[sl].float().cpu().permute(0, 3, 1, 2), w.float().cpu(), None, 2, 1).permute(0, 2, 3, 1) + sh.cpu()
            ref = torch.relu(ref)
            assert float((outs[0][sl].float().cpu() - ref).abs().max() / ref.abs().max()) < 6e-3, (H, Cin)


# ---------------------------------------------------------------------------------------------------------------
# conv2 of a down-sampling block with the shortcut as K-extension (csrc/conv3x3_s1x.hip, round 5)
@pytest.mark.parametrize("Nimg,H,W,C1,Cd,Cout", [(3, 9, 9, 128, 64, 128), (2, 18, 18, 256, 128, 256), (5, 4, 6, 64, 64, 64),
                                                 (1, 1, 2, 128, 64, 32), (7, 5, 3, 256, 128, 160), (2, 36, 36, 128, 64, 128),
                                                 (3, 9, 9, 512, 256, 512), (4, 7, 46, 128, 128, 96), (3, 18, 18, 128, 0, 128),
                                                 (2, 36, 36, 64, 0, 64)])
def test_conv3x3_s1x_matches_torch(hip, Nimg, H, W, C1, Cd, Cout):
    """cadre_conv3x3_s1x vs torch-CPU fp32: relu(conv2d(t, W2, pad 1) + conv2d(x, Wd, stride 2) + shift) on bf16-rounded operands
    (resnet.py:40-55 with the downsample of resnet.py:152-158; both BN scales folded into the weights): 1-8 chunks with 1-4
    shortcut k-tiles (incl. one behind EVERY chunk, Cd == C1), maps narrower than a DMA piece and as wide as the kernel takes,
    M tiles that straddle rows and frames, partial N tiles."""
    from cadre_amd.encoder import _s1x_w
    g = torch.Generator().manual_seed(Nimg * 1000 + H * 10 + C1 + Cout)
    t = _bf(torch.randn(Nimg, C1, H, W, generator=g))
    x = _bf(torch.randn(Nimg, max(Cd, 1), 2 * H, 2 * W, generator=g))
    w2 = _bf(torch.randn(Cout, C1, 3, 3, generator=g) / (C1 * 9) ** 0.5)
    wd = _bf(torch.randn(Cout, Cd, 1, 1, generator=g) / max(Cd, 1) ** 0.5)
    shift = torch.randn(Cout, generator=g)
    want = F.conv2d(t.float(), w2.float(), None, 1, 1) + shift.view(1, -1, 1, 1)
    if Cd:                                                    # (Cd == 0: no shortcut — the kernel as a plain 3x3 / s1 conv)
        want = want + F.conv2d(x.float(), wd.float(), None, 2, 0)
    want = F.relu(want)
    assert hip.lib().cadre_conv3x3_s1x_supported(Nimg, H, W, C1, Cd, Cout) == 1
    td = dev(t.permute(0, 2, 3, 1).contiguous()); xd = dev(x.permute(0, 2, 3, 1).contiguous()) if Cd else None
    wf = dev(_s1x_w(w2.float(), wd.float())).to(torch.bfloat16)
    out = torch.full((Nimg, H, W, Cout), float("nan"), device="cuda", dtype=torch.bfloat16)
    hip.conv3x3_s1x(td, xd, wf, dev(shift), out, Nimg, H, W, C1, Cd, Cout, 1)
    torch.cuda.synchronize()
    assert not torch.isnan(out.float()).any()
    assert rel(out.float().permute(0, 3, 1, 2), want) < 6e-3


def test_conv3x3_s1x_full_size_repeatable_under_load(hip):
    """The three trunk shapes at many frames: 10 launches agree bit for bit (hand-counted vmcnt + raw barriers), first and last
    frames agree with torch-CPU fp32."""
    from cadre_amd.encoder import _s1x_w
    for F_, H, C1, Cd in ((160, 36, 128, 64), (256, 18, 256, 128), (512, 9, 512, 256)):
        g = torch.Generator(device="cuda").manual_seed(H + C1)
        t = torch.randn(F_, H, H, C1, device="cuda", generator=g).to(torch.bfloat16)
        x = torch.randn(F_, 2 * H, 2 * H, Cd, device="cuda", generator=g).to(torch.bfloat16)
        w2 = (torch.randn(C1, C1, 3, 3, device="cuda", generator=g) / np.sqrt(9 * C1)).to(torch.bfloat16)
        wd = (torch.randn(C1, Cd, 1, 1, device="cuda", generator=g) / np.sqrt(Cd)).to(torch.bfloat16)
        sh = torch.randn(C1, device="cuda", generator=g)
        wf = _s1x_w(w2.float().cpu(), wd.float().cpu()).to(torch.bfloat16).cuda()
        outs = []
        for rep in range(10):
            out = torch.empty(F_, H, H, C1, device="cuda", dtype=torch.bfloat16)
            hip.conv3x3_s1x(t, x, wf, sh, out, F_, H, H, C1, Cd, C1, 1)
            outs.append(out)
        torch.cuda.synchronize()
        assert all(torch.equal(outs[0], o) for o in outs[1:]), (H, C1)
        for sl in (slice(0, 2), slice(F_ - 2, F_)):
            ref = (F.conv2d(t[sl].float().cpu().permute(0, 3, 1, 2), w2.float().cpu(), None, 1, 1)
                   + F.conv2d(x[sl].float().cpu().permute(0, 3, 1, 2), wd.float().cpu(), None, 2, 0)).permute(0, 2, 3, 1) + sh.cpu()
            ref = torch.relu(ref)
            assert float((outs[0][sl].float().cpu() - ref).abs().max() / ref.abs().max()) < 6e-3, (H, C1)


# ---------------------------------------------------------------------------------------------------------------
# 3x3 / s1 window conv with a 128 x 128 wave tile, weights streamed to registers (csrc/ab/conv3x3_w128.hip, round 5; A/B build:
# a tie with the ping-pong kernel, profiles/r05_w128_wave_tile_vs_ping_pong.txt)
@pytest.mark.parametrize("Nimg,H,W,Cin,Cout,res,act", [(3, 9, 9, 128, 256, False, 1), (2, 18, 18, 256, 256, True, 1),
                                                       (5, 4, 6, 128, 128, True, 1), (1, 1, 2, 128, 128, False, 0),
                                                       (2, 36, 36, 128, 128, True, 1), (3, 9, 9, 512, 512, True, 1),
                                                       (4, 7, 46, 192, 256, False, 1), (2, 5, 47, 128, 384, True, 0),
                                                       (9, 9, 9, 512, 128, False, 1), (40, 18, 18, 128, 256, True, 1),
                                                       (2, 36, 36, 64, 128, True, 1)])
@needs_ab
def test_conv3x3_w128_matches_torch(hip, Nimg, H, W, Cin, Cout, res, act):
    """cadre_conv3x3_w128 vs torch-CPU fp32: act(conv2d(x, w, pad 1) + shift (+ residual)) on bf16-rounded operands
    (resnet.py:26-55, BN scale folded into the weights): both tile shapes (N % 256 == 0: 256 x 256, else 512 x 128), 2-8 chunks,
    maps narrower than a DMA piece and as wide as the kernel takes, M tiles that straddle rows and frames, several items per
    workgroup (the last case: more items than CUs would need many frames; 40 frames give every persistent workgroup one item and
    the residual prefetch its full pattern)."""
    from cadre_amd.encoder import _w128_w
    g = torch.Generator().manual_seed(Nimg * 1000 + H * 10 + Cin + Cout)
    x = _bf(torch.randn(Nimg, Cin, H, W, generator=g))
    w = _bf(torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5)
    r = _bf(torch.randn(Nimg, Cout, H, W, generator=g))
    shift = torch.randn(Cout, generator=g)
    want = F.conv2d(x.float(), w.float(), None, 1, 1) + shift.view(1, -1, 1, 1)
    if res:
        want = want + r.float()
    if act:
        want = F.relu(want)
    assert hip.lib().cadre_conv3x3_w128_supported(Nimg, H, W, Cin, Cout) == 1
    xd = dev(x.permute(0, 2, 3, 1).contiguous())
    rd = dev(r.permute(0, 2, 3, 1).contiguous()) if res else None
    wf = dev(_w128_w(w.float())).to(torch.bfloat16)
    out = torch.full((Nimg, H, W, Cout), float("nan"), device="cuda", dtype=torch.bfloat16)
    hip.conv3x3_w128(xd, wf, dev(shift), rd, out, Nimg, H, W, Cin, Cout, act)
    torch.cuda.synchronize()
    assert not torch.isnan(out.float()).any()
    assert rel(out.float().permute(0, 3, 1, 2), want) < 6e-3


@needs_ab
def test_conv3x3_w128_full_size_repeatable_under_load(hip):
    """The trunk shapes at many frames (several persistent items per workgroup): 10 launches agree bit for bit (counted vmcnt,
    one raw barrier per chunk: a race shows up as run-to-run differences), first and last frames agree with torch-CPU fp32."""
    from cadre_amd.encoder import _w128_w
    for F_, H, C, res in ((160, 36, 128, True), (256, 18, 256, True), (512, 9, 512, False)):
        g = torch.Generator(device="cuda").manual_seed(H + C)
        x = torch.randn(F_, H, H, C, device="cuda", generator=g).to(torch.bfloat16)
        r = torch.randn(F_, H, H, C, device="cuda", generator=g).to(torch.bfloat16)
        w = (torch.randn(C, C, 3, 3, device="cuda", generator=g) / np.sqrt(9 * C)).to(torch.bfloat16)
        sh = torch.randn(C, device="cuda", generator=g)
        wf = _w128_w(w.float().cpu()).to(torch.bfloat16).cuda()
        outs = []
        for rep in range(10):
            out = torch.empty(F_, H, H, C, device="cuda", dtype=torch.bfloat16)
            hip.conv3x3_w128(x, wf, sh, r if res else None, out, F_, H, H, C, C, 1)
            outs.append(out)
        torch.cuda.synchronize()
        assert all(torch.equal(outs[0], o) for o in outs[1:]), (H, C)
        for sl in (slice(0, 2), slice(F_ - 2, F_)):
            ref = F.conv2d(x[sl].float().cpu().permute(0, 3, 1, 2), w.float().cpu(), None, 1, 1).permute(0, 2, 3, 1) + sh.cpu()
            if res:
                ref = ref + r[sl].float().cpu()
            ref = torch.relu(ref)
            assert float((outs[0][sl].float().cpu() - ref).abs().max() / ref.abs().max()) < 6e-3, (H, C)
