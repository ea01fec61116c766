#!/usr/bin/env python3
"""Cook-Toom matrices of Winograd F(m x m, 3x3) for a point set and their fp32 rounding error on one conv layer, emulated on
the CPU in float32 (transforms in the kernels' order of operations, products as an fp32 matmul) against float64 — the search
that picked the F(3x3) / F(4x4) sets of csrc/winograd_mats.h, extended to F(6x6) (8 x 8 tiles, 64 planes, 5.06x fewer multiplies,
1.78x the input in transform-domain traffic).
    python tools/dbg/wino_points.py 6            # search
    python tools/dbg/wino_points.py 6 0.5 1 2    # one set: points 0, +-0.5, +-1, +-2, inf"""
import itertools
import sys
from fractions import Fraction as Fr

import numpy as np
import torch


def matrices(m, pts):
    """A^T (m x n), G (n x 3), B^T (n x n) as Fractions for finite points pts (n - 1 of them) + infinity, n = m + 2."""
    r, n = 3, m + 2
    pts = [Fr(p) for p in pts]
    assert len(pts) == n - 1 and len(set(pts)) == n - 1
    AT = [[(p ** i if not (p == 0 and i == 0) else Fr(1)) for p in pts] + [Fr(1 if i == m - 1 else 0)] for i in range(m)]
    Nj = []
    for j, p in enumerate(pts):
        d = Fr(1)
        for l, q in enumerate(pts):
            if l != j:
                d *= (p - q)
        Nj.append(d)
    G = [[(p ** k if not (p == 0 and k == 0) else Fr(1)) / Nj[j] for k in range(r)] for j, p in enumerate(pts)] + [[Fr(0), Fr(0), Fr(1)]]

    def polymul(a, b):
        out = [Fr(0)] * (len(a) + len(b) - 1)
        for i, x in enumerate(a):
            for j, y in enumerate(b):
                out[i + j] += x * y
        return out
    Mx = [Fr(1)]
    for p in pts:
        Mx = polymul(Mx, [-p, Fr(1)])                    # prod (x - p): degree n - 1, n coefficients
    BT = []
    for j, p in enumerate(pts):
        q = [Fr(1)]
        for l, pl in enumerate(pts):
            if l != j:
                q = polymul(q, [-pl, Fr(1)])             # M(x) / (x - p_j): degree n - 2
        BT.append(q + [Fr(0)])
    BT.append(Mx)
    return AT, G, BT


def dyadic_scale(AT, G, BT):
    """Move a power-of-two-free factor from each B^T row into G so that B^T's entries are dyadic where possible: scale row j of
    B^T by s_j and row j of G by 1 / s_j (the product is unchanged)."""
    n = len(BT)
    for j in range(n):
        dens = [x.denominator for x in BT[j] if x != 0]
        s = Fr(1)
        lcm = 1
        for d in dens:
            lcm = lcm * d // np.gcd(lcm, d)
        s = Fr(lcm)
        # also normalise the largest |entry| to about 1..4 by a power of two
        mx = max(abs(x * s) for x in BT[j])
        while mx > 4:
            s /= 2; mx /= 2
        BT[j] = [x * s for x in BT[j]]
        G[j] = [x / s for x in G[j]]
    return AT, G, BT


def f32(mat):
    return torch.tensor([[float(x) for x in row] for row in mat], dtype=torch.float32)


def error(m, pts, C=256, N=64, H=18, F=2, seed=0):
    AT, G, BT = dyadic_scale(*matrices(m, pts))
    g = torch.Generator().manual_seed(seed)
    x = torch.relu(torch.randn(F, C, H, H, generator=g))
    w = torch.randn(N, C, 3, 3, generator=g) / (9 * C) ** 0.5
    ref = torch.nn.functional.conv2d(x.double(), w.double(), padding=1)
    n = m + 2
    TH = -(-H // m)
    xp = torch.zeros(F, C, TH * m + 2, TH * m + 2)
    xp[:, :, 1:H + 1, 1:H + 1] = x
    # tiles [F][C][TH][TW][n][n]
    d = xp.unfold(2, n, m).unfold(3, n, m).contiguous()
    BTf, ATf = f32(BT), f32(AT)
    Gd = torch.tensor([[float(x) for x in row] for row in G], dtype=torch.float64)
    U = torch.einsum("ik,ockl,jl->ocij", Gd, w.double(), Gd).float()                     # one rounding (host, float64)
    V = torch.einsum("ik,fcabkl->fcabil", BTf, d)                                         # B^T d   (fp32)
    V = torch.einsum("fcabil,jl->fcabij", V, BTf)                                         # (B^T d) B
    M = torch.einsum("fcabij,ocij->foabij", V, U)                                         # fp32 products and sums
    Y = torch.einsum("ik,foabkl->foabil", ATf, M)
    Y = torch.einsum("foabil,jl->foabij", Y, ATf)                                         # [F][N][TH][TW][m][m]
    out = Y.permute(0, 1, 2, 4, 3, 5).reshape(F, N, TH * m, TH * m)[:, :, :H, :H]
    e = (out.double() - ref)
    s = float(ref.abs().max())
    return float(e.abs().max()) / s, float((e ** 2).mean().sqrt()) / s


if __name__ == "__main__":
    m = int(sys.argv[1])
    if len(sys.argv) > 2:
        half = [Fr(a) for a in sys.argv[2:]]
        pts = [Fr(0)] + [s * a for a in half for s in (1, -1)]
        AT, G, BT = dyadic_scale(*matrices(m, pts))
        print("points", pts, "-> max %.2e rms %.2e" % error(m, pts))
        for nm, M_ in (("BT", BT), ("AT", AT), ("G", G)):
            print(nm, "= [" + ", ".join("[" + ", ".join(str(x) for x in row) + "]" for row in M_) + "]")
        sys.exit(0)
    nh = (m + 1) // 2
    cand = [Fr(1, 2), Fr(3, 4), Fr(1), Fr(5, 4), Fr(3, 2), Fr(2), Fr(5, 2), Fr(3)] if m >= 6 else [Fr(1, 2), Fr(3, 4), Fr(1), Fr(3, 2), Fr(2)]
    res = []
    for half in itertools.combinations(cand, nh):
        pts = [Fr(0)] + [s * a for a in half for s in (1, -1)]
        if len(pts) != m + 1:
            pts = pts[:m + 1]
        mx, rms = error(m, pts)
        res.append((rms, mx, half))
    for rms, mx, half in sorted(res)[:8]:
        print("points 0, +-%s, inf: rms %.2e max %.2e" % (", +-".join(str(h) for h in half), rms, mx))
    for mm, pp in ((4, [0, Fr(3, 4), Fr(-3, 4), Fr(3, 2), Fr(-3, 2)]), (3, [0, Fr(3, 4), Fr(-3, 4), 2]), (2, [0, 1, -1])):
        print("for comparison F(%dx%d) on the product's points: max %.2e rms %.2e" % ((mm, mm) + error(mm, pp)))
