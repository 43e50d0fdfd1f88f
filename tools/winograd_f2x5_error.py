#!/usr/bin/env python3
"""Paper study for VERDICT r4 item 6(i): would Winograd F(2x2, 5x5) keep the 5x5 128->64 layer (reference head.py:622-625) inside the
pinned parity (loc 3e-5, cls 1e-6)?  CPU only.  The layer is evaluated on realistic operands (post-ReLU activations of a 7x7 layer,
BatchNorm-folded weights) three ways: float64 (truth), direct convolution in fp32 (what the f16x3 kernel is equivalent to: 22-bit
products, fp32 accumulation), and Winograd F(2x2, 5x5) with every transform and the 36 element-wise GEMMs in fp32.  Points
{0, +-1, +-2, inf} (the best-conditioned integer set for 6 points) and {0, +-1, +-1/2, inf}.
    python tools/winograd_f2x5_error.py"""
import numpy as np


def ct_matrices(points, m=2, r=5):
    n = m + r - 1

    def E(k):
        M = np.zeros((n, k))
        for i, a in enumerate(points):
            M[i] = [0.0] * (k - 1) + [1.0] if a is None else [a ** j for j in range(k)]
        return M
    AT, G, BT = E(m).T, E(r), np.linalg.inv(E(n)).T
    return AT, G, BT


def conv_direct(x, w, dtype):
    C, H, W = x.shape
    O = w.shape[0]
    y = np.zeros((O, H - 4, W - 4), dtype=dtype)
    xs, ws = x.astype(dtype), w.astype(dtype)
    for dy in range(5):
        for dx in range(5):
            y += np.tensordot(ws[:, :, dy, dx], xs[:, dy:dy + H - 4, dx:dx + W - 4], axes=(1, 0)).astype(dtype)
    return y


def conv_winograd(x, w, points, dtype):
    AT, G, BT = [M.astype(dtype) for M in ct_matrices(points)]
    C, H, W = x.shape
    O = w.shape[0]
    U = np.einsum("ij,ocjk,lk->ocil", G, w.astype(dtype), G).astype(dtype)          # [O, C, 6, 6]
    th, tw = (H - 4) // 2, (W - 4) // 2
    y = np.zeros((O, 2 * th, 2 * tw), dtype=dtype)
    xs = x.astype(dtype)
    for ty in range(th):
        d = np.stack([xs[:, 2 * ty:2 * ty + 6, 2 * tx:2 * tx + 6] for tx in range(tw)], 0)        # [T, C, 6, 6]
        V = np.einsum("ij,tcjk,lk->tcil", BT, d, BT).astype(dtype)
        M = np.einsum("ocil,tcil->toil", U, V).astype(dtype)                                        # 36 GEMMs over c
        Y = np.einsum("ij,tojk,lk->toil", AT, M, AT).astype(dtype)                                  # [T, O, 2, 2]
        for tx in range(tw):
            y[:, 2 * ty:2 * ty + 2, 2 * tx:2 * tx + 2] = Y[tx]
    return y


def main():
    rs = np.random.RandomState(0)
    C, O, H, W = 128, 64, 28, 36
    x = np.maximum(rs.randn(C, H, W) * 0.7 + 0.2, 0.0)                  # post-ReLU activations
    w = rs.randn(O, C, 5, 5) / np.sqrt(C * 25.0) * (1.0 + 0.3 * rs.randn(O, 1, 1, 1))
    ref = conv_direct(x, w, np.float64)
    scale = np.abs(ref).max()
    d32 = conv_direct(x, w, np.float32)
    print("outputs up to {:.3f}; direct fp32 vs float64: max abs {:.3e} (relative to the largest output {:.3e})".format(
        scale, np.abs(d32 - ref).max(), np.abs(d32 - ref).max() / scale))
    for name, pts in (("0, +-1, +-2, inf", [0.0, 1.0, -1.0, 2.0, -2.0, None]), ("0, +-1, +-1/2, inf", [0.0, 1.0, -1.0, 0.5, -0.5, None])):
        AT, G, BT = ct_matrices(pts)
        wg = conv_winograd(x, w, pts, np.float32)
        w64 = conv_winograd(x, w, pts, np.float64)
        e = np.abs(wg - ref[:, :wg.shape[1], :wg.shape[2]]).max()
        print("points {{{}}}: max |B^T| {:.2f}, max |G| {:.2f};  Winograd fp32 vs float64: max abs {:.3e} (relative {:.3e}) = {:.0f}x the direct fp32 error; "
              "float64 Winograd self-check {:.1e}".format(name, np.abs(BT).max(), np.abs(G).max(), e, e / scale, e / np.abs(d32 - ref).max(),
                                                          np.abs(w64 - ref[:, :wg.shape[1], :wg.shape[2]]).max()))


if __name__ == "__main__":
    main()
