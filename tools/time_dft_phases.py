#!/usr/bin/env python3
"""NEEDS A DIAGNOSTIC BUILD: python -m os2d_amd.build --variant stamps -DOS2D_DIAG_DFT_STAMPS, run with
OS2D_HIP_LIB=tools/diag_libs/stamps/libos2d_hip.so.  Phase breakdown of the matrix-product transforms (dft_mfma.h): wall-clock
ticks of thread 0 of every work-group between the phase barriers, averaged per iteration (4 images).
    python tools/time_dft_phases.py [pairs=64] [H=60] [W=80]"""
import ctypes
import os
import sys
import time

import torch

REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
from os2d_amd import _lib  # noqa: E402
from test_dft_gpu import matrices  # noqa: E402
from test_spectral_gpu import dft_sizes  # noqa: E402

NB = int(sys.argv[1]) if len(sys.argv) > 1 else 64
H = int(sys.argv[2]) if len(sys.argv) > 2 else 60
W = int(sys.argv[3]) if len(sys.argv) > 3 else 80
lib = _lib.load()
raw = ctypes.CDLL(_lib.lib_path())
dev = torch.device("cuda:0")
P, Q, nbins, tiles = dft_sizes(H, W)
T = tiles[0] * tiles[1]
cpad = lib.os2d_dft_channel_stride(225)
mats = matrices(P, Q, dev)
corr = torch.rand(NB, 225, H, W, device=dev)
inv = torch.rand(NB, H, W, device=dev) * 0.1
X = torch.empty(nbins // 4, NB * T, cpad, 4, 2, device=dev)
Yq = torch.randn(nbins // 4, NB * T, 128, 4, 2, device=dev)
bp = torch.ones(3 * 128, device=dev)
out = torch.zeros(NB * lib.os2d_shb_bytes(128, H, W), dtype=torch.uint8, device=dev)
status = torch.zeros(1, dtype=torch.int32, device=dev)
st = _lib.current_stream(dev)


def fwd():
    _lib.check(lib.os2d_dft_forward(_lib.ptr(corr), _lib.ptr(inv), _lib.ptr(X), _lib.ptr(mats), NB, 225, H, W, st), "fwd")


def invt():
    _lib.check(lib.os2d_dft_inverse(_lib.ptr(Yq), _lib.ptr(bp), _lib.ptr(out), _lib.ptr(mats), NB, 128, H, W, _lib.ptr(status), st), "inv")


stamps = (ctypes.c_ulonglong * 16)()
for name, fn, base, phases in (("forward", fwd, 0, ("W write x", "step 1", "R write", "step 2", "XS stage", "ST store")),
                               ("inverse", invt, 8, ("max", "WY write", "step A", "WT write", "step B", "epilogue"))):
    fn()
    torch.cuda.synchronize()
    has = raw.os2d_debug_dft_stamps(stamps, 1) == 0
    t0 = time.perf_counter()
    n = 10
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / n * 1e3
    line = "dft {} {} pairs {}x{} (P={} Q={} T={}): {:.3f} ms".format(name, NB, H, W, P, Q, T, ms)
    if has:
        raw.os2d_debug_dft_stamps(stamps, 1)
        iters = max(stamps[base + 6], 1)
        per = [stamps[base + k] / iters * 10.0 for k in range(6)]           # 100 MHz ticks -> ns
        line += " | per iteration (us): " + ", ".join("{} {:.2f}".format(p, v / 1e3) for p, v in zip(phases, per))
        line += " | sum {:.2f} us x {:.1f} iterations per work-group".format(sum(per) / 1e3, iters / n / min(256, (iters // n + 7) // 8 * 8))
    print(line)
