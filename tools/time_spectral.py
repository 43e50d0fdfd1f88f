#!/usr/bin/env python3
"""Time os2d_spectral_gemm at the real problem size (no correctness check: for diagnostic builds)."""
import os, sys, time
import torch
REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, REPO)
from os2d_amd import _lib
lib = _lib.load()
dev = torch.device("cuda:0")
NB = int(sys.argv[1]) if len(sys.argv) > 1 else 64
C, Cout, nbins = 225, 128, 72 * 49
W = torch.randn(lib.os2d_spectral_weight_bytes(C, Cout, nbins) // 4, device=dev)
X = torch.randn(NB, C, nbins, 2, device=dev)
Y = torch.empty(NB, Cout, nbins, 2, device=dev)
st = _lib.current_stream(dev)
def run():
    _lib.check(lib.os2d_spectral_gemm(_lib.ptr(W), _lib.ptr(X), _lib.ptr(Y), NB, C, Cout, nbins, st), "gemm")
run(); torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): run()
torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 10 * 1e3
print("spectral GEMM NB={}: {:.3f} ms = {:.1f} TFLOP/s".format(NB, ms, 8.0 * Cout * C * NB * nbins / ms / 1e9))
