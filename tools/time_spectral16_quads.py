#!/usr/bin/env python3
"""Time os2d_spectral_gemm_f16_quads (the default mode's per-bin GEMM: both spectra in quads of bins, blocks of 64 pairs) at the
benchmark's transform size (64 x 84: 2752 bins) for the given pair counts; no correctness check (for diagnostic builds:
OS2D_HIP_LIB=tools/diag_libs/<tag>/libos2d_hip.so)."""
import os, sys, time
import torch
REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, REPO)
from os2d_amd import _lib
from os2d_amd.modeling import head as head_mod
from os2d_amd.utils import synthetic
lib = _lib.load()
dev = torch.device("cuda:0")
net = head_mod.TransformationNet(output_dim=6)
net.load_state_dict(synthetic.make_transform_net_state(6, seed=3)); net.to(dev).eval()
H, W = 60, 80
w16, _, _, nbins = net.spectra(H, W, split=True)
xs = lib.os2d_dft_xscale(H, W)
cpad = lib.os2d_dft_channel_stride(225)
st = _lib.current_stream(dev)
for NB in [int(a) for a in sys.argv[1:]] or [64, 1024]:
    X = (torch.rand(nbins // 4, NB, cpad, 4, 2, device=dev) * 40.0 - 20.0)
    Y = torch.empty(nbins // 4, NB, 128, 4, 2, device=dev)
    def run():
        _lib.check(lib.os2d_spectral_gemm_f16_quads(_lib.ptr(w16), _lib.ptr(X), _lib.ptr(Y), NB, 225, 128, nbins, xs, st), "gemm16 quads")
    for _ in range(3): run()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 20 if NB <= 128 else 6
    for _ in range(n): run()
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / n * 1e3
    gb = 8.0 * nbins * (128 * 225 + 225 * NB + 128 * NB) / 1e9
    print("TIME lib={} spectral_gemm_f16_quads NB={}: {:.4f} ms  ({:.2f} GB algorithmic -> {:.2f} TB/s)".format(
        os.environ.get("OS2D_HIP_LIB", "product").split("/")[-2] if "/" in os.environ.get("OS2D_HIP_LIB", "") else "product", NB, ms, gb, gb / ms))
    del X, Y
