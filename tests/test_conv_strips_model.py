"""CPU: the column-strip geometry of the 5x5 kernels for maps wider than their linear LDS slab (conv_f16x3.hip STRIP mode,
conv3_f16x3.hip, conv_mfma.hip; reference head.py:622-629 has no width limit) restated in numpy, index for index as the
kernels compute it: strip-plane index -> map cell (os2d_strip_cell), the slab a tile loads, the shift-and-accumulate over the flat
strip-plane index with the strip pitch, the epilogue's mapping back.  Checked against a direct zero-padded correlation: every
data cell written exactly once with the right value, every pad cell of the rows written as zero, nothing else touched."""
import numpy as np
import pytest

PAD, NT = 3, 256


def ws(W):
    return W + PAD


def base(W):
    return (PAD * ws(W) + PAD + 3) // 4 * 4


def plane(H, W):
    return (base(W) + (H + PAD) * ws(W) + PAD + 63) // 64 * 64


def conv_strips(W, R):          # os2d_conv_strips
    NS = (ws(W) + 255) // 256
    return NS, (ws(W) + NS - 1) // NS + 2 * R


def strip_cell(n, SP, c0mR, H, W):        # os2d_strip_cell (C division truncates towards zero: np >= 0 is tested first)
    if n < 0:
        return 0
    h, j = divmod(n, SP)
    c = c0mR + j
    return base(W) + h * ws(W) + c if (h < H and 0 <= c < W) else 0


def run_strips(x, w, halo_round=1):
    """x [H, W], w [KS, KS] -> (plane-shaped output, write counts) the way a STRIP launch produces them."""
    H, W = x.shape
    KS = w.shape[0]
    R = KS // 2
    Ws, BASE, PLANE = ws(W), base(W), plane(H, W)
    xin = np.zeros(PLANE)
    for h in range(H):
        xin[BASE + h * Ws:BASE + h * Ws + W] = x[h]
    NS, SP = conv_strips(W, R)
    HALO = (R * SP + R + halo_round - 1) // halo_round * halo_round
    SLAB = NT + 2 * HALO
    assert SLAB <= 1536
    TPS = (H * SP + NT - 1) // NT
    out = np.full(PLANE, np.nan)
    writes = np.zeros(PLANE, dtype=int)
    for tile in range(NS * TPS):
        strip = tile // TPS
        c0mR = strip * (SP - 2 * R) - R
        n0 = (tile - strip * TPS) * NT
        slab = np.array([xin[strip_cell(n0 - HALO + i, SP, c0mR, H, W)] for i in range(SLAB)])
        for p in range(NT):
            n = n0 + p
            hr, j = divmod(n, SP)
            wc = c0mR + j
            if j < R or j >= SP - R or hr >= H or wc >= Ws:
                continue
            acc = 0.0
            for dy in range(KS):
                for dx in range(KS):
                    acc += w[dy, dx] * slab[p + HALO - R * SP - R + dy * SP + dx]
            cell = BASE + hr * Ws + wc
            out[cell] = acc if wc < W else 0.0
            writes[cell] += 1
    return out, writes


@pytest.mark.parametrize("H,W,KS,halo_round", [(3, 317, 5, 1), (5, 400, 5, 1), (2, 640, 5, 4), (7, 509, 5, 4), (1, 1030, 5, 1)])
def test_strip_geometry_reproduces_the_zero_padded_correlation(H, W, KS, halo_round):
    rs = np.random.RandomState(H * 1000 + W)
    x, w = rs.randn(H, W), rs.randn(KS, KS)
    R = KS // 2
    ref = np.zeros((H, W))
    xp = np.pad(x, R)
    for dy in range(KS):
        for dx in range(KS):
            ref += w[dy, dx] * xp[dy:dy + H, dx:dx + W]
    out, writes = run_strips(x, w, halo_round)
    Ws, BASE = ws(W), base(W)
    rows = np.arange(H)[:, None] * Ws + BASE
    data = rows + np.arange(W)[None, :]
    pads = rows + np.arange(W, Ws)[None, :]
    assert (writes[data] == 1).all() and (writes[pads] == 1).all() and writes.sum() == H * Ws
    np.testing.assert_allclose(out[data], ref, rtol=0, atol=1e-12)
    assert (out[pads] == 0).all()


def test_strip_widths_fit_every_kernel_slab():
    """SLAB = 256 + 2 * HALO <= 1536 units for every width the head accepts (the slab prefetch of conv_f16x3 / conv_mfma) and
    the strips cover the W data columns + 3 pad columns."""
    for W in range(317, 3601):
        NS, SP = conv_strips(W, 2)
        assert NS * (SP - 4) >= ws(W) and SP - 4 <= 256
        assert NT + 2 * ((2 * SP + 2 + 3) // 4 * 4) <= 1536
