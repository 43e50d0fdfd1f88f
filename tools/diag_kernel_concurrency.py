#!/usr/bin/env python3
"""Diagnostic: run ONE kernel of the frequency-domain layer concurrently on several streams (own buffers each) and compare
with the same launches run one after the other."""
import os, sys, ctypes
import torch
REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
from os2d_amd import _lib
from test_spectral_gpu import twiddles, fft_sizes
lib = _lib.load(); dev = torch.device("cuda:0")
H, W, NB, C, Cout, NS = 48, 64, 128, 225, 128, 7
P, Q, nbins = fft_sizes(H, W)
tq, tp = twiddles(Q, dev), twiddles(P, dev)
g = torch.Generator().manual_seed(0)
streams = [torch.cuda.Stream(device=dev) for _ in range(NS)]
corr = [torch.randn(NB, C, H * W, generator=g).to(dev) for _ in range(NS)]
inv = [torch.rand(NB, H * W, generator=g).to(dev) + 0.5 for _ in range(NS)]
Wsp = torch.randn(lib.os2d_spectral_weight_bytes(C, Cout, nbins) // 4, generator=g).to(dev)
bp = torch.ones(3 * 128, device=dev)
status = torch.zeros(1, dtype=torch.int32, device=dev)
def bufs():
    return dict(X=[torch.zeros(C, NB, nbins, 2, device=dev) for _ in range(NS)], Y=[torch.zeros(NB, Cout, nbins, 2, device=dev) for _ in range(NS)],
                out=[torch.zeros(NB * lib.os2d_shb_bytes(Cout, H, W), dtype=torch.uint8, device=dev) for _ in range(NS)])
def launch(kind, i, b, st):
    s = ctypes.c_void_p(st.cuda_stream)
    if kind == "fwd":
        _lib.check(lib.os2d_fft_forward(_lib.ptr(corr[i]), _lib.ptr(inv[i]), _lib.ptr(b["X"][i]), _lib.ptr(tq), _lib.ptr(tp), NB, C, H, W, s), "f")
    elif kind == "gemm":
        _lib.check(lib.os2d_spectral_gemm(_lib.ptr(Wsp), _lib.ptr(b["X"][i]), _lib.ptr(b["Y"][i]), NB, C, Cout, nbins, s), "g")
    else:
        _lib.check(lib.os2d_fft_inverse(_lib.ptr(b["Y"][i]), _lib.ptr(bp), _lib.ptr(b["out"][i]), _lib.ptr(tq), _lib.ptr(tp), NB, Cout, H, W, _lib.ptr(status), s), "i")
ref = bufs()
main = torch.cuda.current_stream(dev)
for kind in ("fwd", "gemm", "inv"):
    for i in range(NS):
        launch(kind, i, ref, main)
    torch.cuda.synchronize()
key = {"fwd": "X", "gemm": "Y", "inv": "out"}
for kind in ("fwd", "gemm", "inv"):
    nbad = 0
    for it in range(6):
        b = bufs()
        for k2 in ("X", "Y"):                      # inputs of the stage under test = the reference's
            for i in range(NS):
                b[k2][i].copy_(ref[k2][i])
        b[key[kind]] = [torch.zeros_like(t) for t in ref[key[kind]]]
        torch.cuda.synchronize()
        for i in range(NS):
            launch(kind, i, b, streams[i])
        torch.cuda.synchronize()
        for i in range(NS):
            if not torch.equal(b[key[kind]][i], ref[key[kind]][i]):
                nbad += 1
                d = (b[key[kind]][i] != ref[key[kind]][i])
                print(kind, "iteration", it, "stream", i, "differing elements", int(d.sum()), "first index", d.nonzero()[0].tolist())
    print(kind, "bad:", nbad, "of", 6 * NS)

# ---- the chain forward -> GEMM -> inverse on every stream, all streams concurrently
for it in range(6):
    b = bufs()
    torch.cuda.synchronize()
    for i in range(NS):
        for kind in ("fwd", "gemm", "inv"):
            launch(kind, i, b, streams[i])
    torch.cuda.synchronize()
    for i in range(NS):
        msg = []
        for kind in ("fwd", "gemm", "inv"):
            d = b[key[kind]][i] != ref[key[kind]][i]
            if d.any():
                idx = d.nonzero()
                msg.append("{}: {} elements, dim0 values {} dim1 values {}".format(key[kind], int(d.sum()), sorted(set(idx[:, 0].tolist()))[:6],
                                                                                   sorted(set(idx[:, 1].tolist()))[:6] if idx.size(1) > 1 else ""))
        if msg:
            print("chain iteration", it, "stream", i, "|", " | ".join(msg))
print("chain done")

# ---- sustained: 12 chained repetitions per stream without host synchronisation, every repetition into its own buffers
REPS = 4
for it in range(3):
    bb = [bufs() for _ in range(REPS)]
    torch.cuda.synchronize()
    for r in range(REPS):
        for i in range(NS):
            for kind in ("fwd", "gemm", "inv"):
                launch(kind, i, bb[r], streams[i])
    torch.cuda.synchronize()
    nb = 0
    for r in range(REPS):
        for i in range(NS):
            for kind in ("fwd", "gemm", "inv"):
                if not torch.equal(bb[r][key[kind]][i], ref[key[kind]][i]):
                    nb += 1
                    d = bb[r][key[kind]][i] != ref[key[kind]][i]
                    print("sustained", it, "rep", r, "stream", i, key[kind], "differing", int(d.sum()), "first", d.nonzero()[0].tolist())
    print("sustained iteration", it, "bad buffers:", nb, "of", REPS * NS * 3)
    del bb
