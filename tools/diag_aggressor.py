#!/usr/bin/env python3
"""Diagnostic: forward / inverse FFT kernels on 4 streams while 3 other streams run ONE other kernel of the head
(corr | conv2 | conv3 | gemm16) in a loop; are the transforms' outputs still what they are alone?"""
import os, sys, ctypes
import torch
REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
from os2d_amd import _lib
from os2d_amd.modeling import head as head_mod
from os2d_amd.utils import synthetic
from test_spectral_gpu import twiddles, fft_sizes
lib = _lib.load(); dev = torch.device("cuda:0")
H, W, NB, C, Cout = 48, 64, 128, 225, 128
P, Q, nbins = fft_sizes(H, W)
tq, tp = twiddles(Q, dev), twiddles(P, dev)
g = torch.Generator().manual_seed(0)
NF, NA = 4, 3
fstreams = [torch.cuda.Stream(device=dev) for _ in range(NF)]
astreams = [torch.cuda.Stream(device=dev) for _ in range(NA)]
corr = [torch.randn(NB, C, H * W, generator=g).to(dev) for _ in range(NF)]
inv = [torch.rand(NB, H * W, generator=g).to(dev) + 0.5 for _ in range(NF)]
Yin = [torch.randn(NB, Cout, nbins, 2, generator=g).to(dev) for _ in range(NF)]
bp = torch.ones(3 * 128, device=dev)
status = torch.zeros(1, dtype=torch.int32, device=dev)
shb = lib.os2d_shb_bytes(Cout, H, W)
def run_fft(i, X, out, st):
    s = ctypes.c_void_p(st.cuda_stream)
    _lib.check(lib.os2d_fft_forward(_lib.ptr(corr[i]), _lib.ptr(inv[i]), _lib.ptr(X), _lib.ptr(tq), _lib.ptr(tp), NB, C, H, W, s), "f")
    _lib.check(lib.os2d_fft_inverse(_lib.ptr(Yin[i]), _lib.ptr(bp), _lib.ptr(out), _lib.ptr(tq), _lib.ptr(tp), NB, Cout, H, W, _lib.ptr(status), s), "i")
refX = [torch.zeros(C, NB, nbins, 2, device=dev) for _ in range(NF)]
refO = [torch.zeros(NB * shb, dtype=torch.uint8, device=dev) for _ in range(NF)]
main = torch.cuda.current_stream(dev)
for i in range(NF):
    run_fft(i, refX[i], refO[i], main)
torch.cuda.synchronize()
# aggressors
A_, Cf = 1, 1024
fm = synthetic.make_feature_map(Cf, H, W, seed=1).to(dev)
qs = torch.randn(lib.os2d_class_split_bytes(NB, Cf) // 2, generator=g).to(dev).to(torch.float16).view(torch.uint8)
cws = [torch.zeros(lib.os2d_corr_f16x3_workspace_bytes(A_, Cf, H, W), dtype=torch.uint8, device=dev) for _ in range(NA)]
ccorr = [torch.zeros(NB, 225, H * W, device=dev) for _ in range(NA)]
crshb = [torch.zeros(NB * lib.os2d_shb_bytes(225, H, W), dtype=torch.uint8, device=dev) for _ in range(NA)]
net = head_mod.TransformationNet(output_dim=6)
net.load_state_dict(synthetic.make_transform_net_state(6, seed=3)); net.to(dev).eval()
w1, b1, w2, b2, w3, b3 = net.packed("f16x3")
h1 = [torch.zeros(NB * lib.os2d_shb_bytes(128, H, W), dtype=torch.uint8, device=dev) for _ in range(NA)]
h2 = [torch.zeros(NB * lib.os2d_shb_bytes(64, H, W), dtype=torch.uint8, device=dev) for _ in range(NA)]
par = [torch.zeros(NB, 6, H * W, device=dev) for _ in range(NA)]
def aggress(kind, j, st):
    s = ctypes.c_void_p(st.cuda_stream)
    if kind == "corr":
        _lib.check(lib.os2d_corr_f16x3(_lib.ptr(fm), _lib.ptr(qs), _lib.ptr(ccorr[j]), _lib.ptr(crshb[j]), A_, NB, Cf, H, W, _lib.ptr(cws[j]), cws[j].numel(), s), "corr")
    elif kind == "conv1":
        _lib.check(lib.os2d_transform_conv_f16x3(1, _lib.ptr(crshb[j]), _lib.ptr(w1), _lib.ptr(b1), _lib.ptr(h1[j]), NB, 6, H, W, 3, _lib.ptr(status), s), "c1")
    elif kind == "conv2":
        _lib.check(lib.os2d_transform_conv_f16x3(2, _lib.ptr(h1[j]), _lib.ptr(w2), _lib.ptr(b2), _lib.ptr(h2[j]), NB, 6, H, W, 3, _lib.ptr(status), s), "c2")
    elif kind == "conv3":
        _lib.check(lib.os2d_transform_conv_f16x3(3, _lib.ptr(h2[j]), _lib.ptr(w3), _lib.ptr(b3), _lib.ptr(par[j]), NB, 6, H, W, 3, _lib.ptr(status), s), "c3")
for kind in sys.argv[1:] or ["none", "corr", "conv1", "conv2", "conv3"]:
    nbad = 0
    for it in range(4):
        X = [torch.zeros_like(t) for t in refX]; O = [torch.zeros_like(t) for t in refO]
        torch.cuda.synchronize()
        for rep in range(3):
            for j in range(NA):
                if kind != "none":
                    aggress(kind, j, astreams[j])
            for i in range(NF):
                run_fft(i, X[i], O[i], fstreams[i])
        torch.cuda.synchronize()
        for i in range(NF):
            bx, bo = not torch.equal(X[i], refX[i]), not torch.equal(O[i], refO[i])
            if bx or bo:
                nbad += 1
                print("aggressor", kind, "iteration", it, "fft stream", i, "X differs" if bx else "", "out differs" if bo else "")
                if bx:
                    d = (X[i] != refX[i]).any(dim=-1)                     # [C, NB, nbins]
                    imgs = d.any(dim=-1).nonzero().tolist()
                    print("   images (c, n):", imgs[:8], "count", len(imgs))
                    c0, n0 = imgs[0]
                    bins = d[c0, n0].nonzero().flatten().tolist()
                    print("   bins of the first image:", len(bins), bins[:12], "..", bins[-4:], "P,Q,V =", P, Q, Q // 2 + 1)
                    b0 = bins[0]
                    print("   got", X[i][c0, n0, b0].tolist(), "want", refX[i][c0, n0, b0].tolist(),
                          "| max abs diff in image", float((X[i][c0, n0] - refX[i][c0, n0]).abs().max()), "max abs value", float(refX[i][c0, n0].abs().max()))
                if bo:
                    d = (O[i] != refO[i]).view(NB, -1)
                    rows = d.any(dim=-1).nonzero().flatten().tolist()
                    off = d[rows[0]].nonzero().flatten()
                    print("   pairs with differing output bytes:", rows[:8], "bytes in first:", int(off.numel()), "offsets", off[:6].tolist(), "..", off[-3:].tolist(), "per pair bytes", d.size(1))
    print("aggressor", kind, "bad:", nbad, "of", 4 * NF)
