#!/usr/bin/env python3
"""PROTOTYPE (not product): the 7x7 TransformNet layer in the frequency domain with library kernels (torch.fft + bmm),
to validate the numerics on the GPU and to see what the transform / spectral-GEMM split costs before any hand-written
kernel exists.  Usage: tools/proto_fft_conv.py [B] [P] [Q]"""
import os, sys, time
import torch
import torch.nn.functional as F
REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
from os2d_amd.utils import synthetic
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
P = int(sys.argv[2]) if len(sys.argv) > 2 else 72
Q = int(sys.argv[3]) if len(sys.argv) > 3 else 96
H, W = 60, 80
state = synthetic.make_transform_net_state(6, seed=1)
s = state["conv.1.weight"].double() / torch.sqrt(state["conv.1.running_var"].double() + 1e-5)
w1 = (state["conv.0.weight"].double() * s.view(-1, 1, 1, 1))
b1 = ((state["conv.0.bias"].double() - state["conv.1.running_mean"].double()) * s + state["conv.1.bias"].double())
g = torch.Generator().manual_seed(0)
r = torch.rand(B, 225, H, W, generator=g).clamp(min=0.5) - 0.5           # relu-like
rn = (r / (r.norm(dim=1, keepdim=True) + 1e-6)).to(dev)
w1d, b1d = w1.float().to(dev), b1.float().to(dev)

def sync(): torch.cuda.synchronize()
def timeit(f, n=5):
    f(); sync(); t0 = time.perf_counter()
    for _ in range(n): f()
    sync(); return (time.perf_counter() - t0) / n * 1e3

# weight spectra (once per weights): correlation = convolution with the flipped kernel; fold the (3,3) centre shift in
k = torch.zeros(128, 225, P, Q, device=dev)
k[:, :, :7, :7] = w1d.flip(2, 3)
K = torch.fft.rfft2(k)                                                     # [128,225,P,V]
V = Q // 2 + 1
Kb = K.permute(2, 3, 0, 1).reshape(P * V, 128, 225).contiguous()          # [bins, o, c]
print("weight spectra: {:.0f} MB".format(Kb.numel() * 8 / 1e6))

def fwd():
    return torch.fft.rfft2(rn, s=(P, Q))                                  # [B,225,P,V]
def gemm(X):
    Xb = X.permute(2, 3, 1, 0).reshape(P * V, 225, B)                      # [bins, c, b]
    return torch.bmm(Kb, Xb)                                              # [bins, o, b]
def inv(Yb):
    Y = Yb.reshape(P, V, 128, B).permute(3, 2, 0, 1)
    y = torch.fft.irfft2(Y, s=(P, Q))
    return F.relu(y[:, :, 6 - 3:6 - 3 + H, 6 - 3:6 - 3 + W] + b1d.view(1, -1, 1, 1))
X = fwd(); Yb = gemm(X); y = inv(Yb)
ref = F.relu(F.conv2d(rn[:4].double().cpu(), w1, b1, padding=3))
print("max abs err vs fp64 direct (4 classes):", float((y[:4].double().cpu() - ref).abs().max()), " |y| max", float(ref.max()))
y32 = F.relu(F.conv2d(rn[:4], w1d, b1d, padding=3))
print("fp32 direct (MIOpen) err vs fp64:", float((y32.double().cpu() - ref).abs().max()))
print("B={} P={} Q={} bins={}: rfft2 {:.3f} ms, bmm {:.3f} ms, irfft2+bias+relu {:.3f} ms".format(
    B, P, Q, P * V, timeit(fwd), timeit(lambda: gemm(X)), timeit(lambda: inv(Yb))))
