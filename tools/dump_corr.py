#!/usr/bin/env python3
"""Debug: read the fp32 correlation tensor the f16x3 head left in its workspace and compare with the oracle's."""
import os, sys
import torch
REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import util
from os2d_amd.modeling import head as head_mod
dev = torch.device("cuda:0")
fx = util.load_head_fixture(sys.argv[1] if len(sys.argv) > 1 else "affine_noinv")
creator = util.make_head_creator(fx["P"], fx["inverse"], fx["state"], dev)
with torch.no_grad():
    head = creator.create_os2d_head([c.to(dev) for c in fx["class_fms"]])
    head(fx["fm"].to(dev), precision="f16x3")
torch.cuda.synchronize()
A, C, H, W = fx["fm"].shape
B = len(fx["class_fms"]); HW = H * W
al = lambda x: (x + 255) // 256 * 256
off = al(A * HW * 4)
CGP = ((C + 7) // 8 + 3) // 4 * 4
off = al(off + A * CGP * 2 * HW * 4 * 4)
ws = list(head_mod._WORKSPACES.values())[0]
corr = ws[off:off + A * B * 225 * HW * 4].view(torch.float32).view(A * B, 225, HW).cpu()
ref = fx["ref_corr"].reshape(A * B, 225, HW)
d = (corr - ref).abs()
print("max diff", float(d.max()))
bad = (d > 1e-5)
print("bad rows (m) of class 0:", sorted(set(bad[0].nonzero()[:, 0].tolist()))[:40])
print("bad cols (n) of class 0:", sorted(set(bad[0].nonzero()[:, 1].tolist()))[:40])
# hypothesis: 4x4 blocks transposed
m0, n0 = 8, 4
print("got block\n", corr[0, m0:m0 + 4, n0:n0 + 4]); print("ref block\n", ref[0, m0:m0 + 4, n0:n0 + 4])
