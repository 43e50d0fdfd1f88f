#!/usr/bin/env python3
"""Fixed cost per work-group of the correlation kernel: time os2d_corr_f16x3_packed (padded form) for several channel counts at the
benchmark's shape (64 classes, 60 x 80); the K loop scales with C, launch / prologue / epilogue do not."""
import os, sys, time
import torch
REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import util
from os2d_amd import _lib
from os2d_amd.utils import synthetic
lib = _lib.load()
dev = torch.device("cuda:0")
H, W, B = 60, 80, int(sys.argv[1]) if len(sys.argv) > 1 else 64
st = _lib.current_stream(dev)
res = []
for C in (256, 512, 1024, 2048):
    creator = util.make_head_creator(6, True, synthetic.make_transform_net_state(6, seed=1), dev)
    fm = synthetic.make_feature_map(C, H, W, seed=3).to(dev)
    with torch.no_grad():
        head = creator.create_os2d_head([c.to(dev) for c in synthetic.make_class_feature_maps(B, C, seed=5)])
    qs = head._split_class_operand()
    ws = torch.empty(lib.os2d_corr_f16x3_packed_workspace_bytes(1, B, C, H, W), dtype=torch.uint8, device=dev)
    corr = torch.empty(B, 225, H * W, device=dev); invn = torch.empty(B, H * W, device=dev)
    for form in (0, 1):
        def run():
            _lib.check(lib.os2d_corr_f16x3_packed(_lib.ptr(fm), _lib.ptr(qs), _lib.ptr(corr), _lib.ptr(invn), 1, B, C, H, W, form, _lib.ptr(ws), ws.numel(), st), "corr")
        for _ in range(3): run()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        n = 30
        for _ in range(n): run()
        torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / n * 1e3
        print("TIME corr stage (sumsq + split + corr{}) B={} C={}: {:.4f} ms".format(" packed" if form else " padded", B, C, ms))
        res.append((form, C, ms))
for form in (0, 1):
    r = [(c, ms) for f, c, ms in res if f == form]
    slope = (r[3][1] - r[1][1]) / (r[3][0] - r[1][0])
    print("form {}: {:.4f} ms per 1024 channels, intercept {:.4f} ms (from C = 512 and 2048)".format(form, slope * 1024, r[1][1] - slope * r[1][0]))
