#!/usr/bin/env python3
"""Time the building blocks of the frequency-domain 7x7 layer at the benchmark size (64 classes, 60x80)."""
import os, sys, time
import torch
REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
from os2d_amd import _lib
from test_spectral_gpu import twiddles, fft_sizes
lib = _lib.load(); dev = torch.device("cuda:0")
NB = int(sys.argv[1]) if len(sys.argv) > 1 else 64
H, W, C, Cout = 60, 80, 225, 128
P, Q, nbins = fft_sizes(H, W)
tq, tp = twiddles(Q, dev), twiddles(P, dev)
corr = torch.randn(NB, C, H * W, device=dev); inv = torch.rand(NB, H * W, device=dev)
X = torch.empty(C, NB, nbins, 2, device=dev); Y = torch.randn(NB, Cout, nbins, 2, device=dev)
Wsp = torch.randn(lib.os2d_spectral_weight_bytes(C, Cout, nbins) // 4, device=dev)
bp = torch.ones(3 * 128, device=dev)
out = torch.empty(NB * lib.os2d_shb_bytes(Cout, H, W), dtype=torch.uint8, device=dev)
status = torch.zeros(1, dtype=torch.int32, device=dev)
st = _lib.current_stream(dev)
def t(f, n=10):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
fwd = lambda: _lib.check(lib.os2d_fft_forward(_lib.ptr(corr), _lib.ptr(inv), _lib.ptr(X), _lib.ptr(tq), _lib.ptr(tp), NB, C, H, W, st), "f")
invq = lambda: _lib.check(lib.os2d_fft_inverse_ex(_lib.ptr(Y), _lib.ptr(bp), _lib.ptr(out), _lib.ptr(tq), _lib.ptr(tp), NB, Cout, H, W, _lib.ptr(status), 1, st), "iq")
gem = lambda: _lib.check(lib.os2d_spectral_gemm(_lib.ptr(Wsp), _lib.ptr(X), _lib.ptr(Y), NB, C, Cout, nbins, st), "g")
inv_ = lambda: _lib.check(lib.os2d_fft_inverse(_lib.ptr(Y), _lib.ptr(bp), _lib.ptr(out), _lib.ptr(tq), _lib.ptr(tp), NB, Cout, H, W, _lib.ptr(status), st), "i")
W16 = torch.randn(lib.os2d_spectral_weight16_bytes(C, nbins) // 2, device=dev).to(torch.float16).view(torch.uint8)
W16.view(torch.float32)[-128:] = 1.0
xs = lib.os2d_spectral_xscale(H, W)
gem16 = lambda: _lib.check(lib.os2d_spectral_gemm_f16(_lib.ptr(W16), _lib.ptr(X), _lib.ptr(Y), NB, C, Cout, nbins, xs, st), "g16")
X.normal_()
print("TIME lib={} NB={} P={} Q={} bins={}: forward {:.3f} ms, spectral GEMM {:.3f} ms (split-half: {:.3f} ms), inverse {:.3f} ms (quad layout {:.3f})".format(
    os.environ.get("OS2D_HIP_LIB", "product"), NB, P, Q, nbins, t(fwd), t(gem), t(gem16), t(inv_), t(invq)))
