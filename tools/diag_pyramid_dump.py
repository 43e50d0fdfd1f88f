#!/usr/bin/env python3
"""NEEDS A DIAGNOSTIC BUILD: python -m os2d_amd.build --variant dump -DOS2D_DIAG_DUMP, then run with
OS2D_HIP_LIB=tools/diag_libs/dump/libos2d_hip.so (the product library compiles the dump hook out).
Diagnostic: 7 identical levels on 7 streams, intermediate buffers dumped per stream; which stage differs first?"""
import os, sys, ctypes
import torch
REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import util
from os2d_amd import _lib
from os2d_amd.engine.pyramid import level_stream
from os2d_amd.utils import synthetic
lib = _lib.load(); dev = torch.device("cuda:0")
H, W, NS, B = 48, 64, 7, 128
state = synthetic.make_transform_net_state(6, seed=1)
fm = synthetic.make_feature_map(1024, H, W, seed=102).to(dev)
base = [c.to(dev) for c in synthetic.make_class_feature_maps(8, 1024, seed=7000)]
creator = util.make_head_creator(6, True, state, dev)
P_, Q_, nb_ = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
lib.os2d_fft_sizes(H, W, ctypes.byref(P_), ctypes.byref(Q_), ctypes.byref(nb_))
plane = lib.os2d_plane_floats(H, W)
sizes = {0: B * 225 * H * W * 4, 1: B * H * W * 4, 2: B * 225 * nb_.value * 8, 3: B * 128 * nb_.value * 8, 4: B * 128 * plane * 4, 5: B * 64 * plane * 4, 6: B * 6 * H * W * 4}
names = {0: "corr", 1: "invn", 2: "xspec", 3: "yspec", 4: "h1", 5: "h2", 6: "params"}
with torch.no_grad():
    head = creator.create_os2d_head([base[b % 8] for b in range(B)])
    head.precision = sys.argv[1] if len(sys.argv) > 1 else "fft"
    streams = [level_stream(dev, i) for i in range(NS)]
    dumps = [{s: torch.zeros(n, dtype=torch.uint8, device=dev) for s, n in sizes.items()} for _ in range(NS + 1)]
    main = torch.cuda.current_stream(dev)
    for s, t in dumps[NS].items():
        lib.os2d_debug_set_dump(ctypes.c_void_p(main.cuda_stream), s, _lib.ptr(t), t.numel())
    ref_out = head(fm)[1].clone()
    torch.cuda.synchronize()
    ref = {s: t.clone() for s, t in dumps[NS].items()}
    for i in range(NS):
        for s, t in dumps[i].items():
            lib.os2d_debug_set_dump(ctypes.c_void_p(streams[i].cuda_stream), s, _lib.ptr(t), t.numel())
    for it in range(6):
        outs = []
        for i in range(NS):
            with torch.cuda.stream(streams[i]):
                outs.append(head(fm)[1])
        torch.cuda.synchronize()
        for i in range(NS):
            bad = [names[s] + ":" + str(int((dumps[i][s] != ref[s]).sum())) for s in sorted(sizes) if not torch.equal(dumps[i][s], ref[s])]
            if bad or not torch.equal(outs[i], ref_out):
                first = None
                for s in sorted(sizes):
                    d = dumps[i][s] != ref[s]
                    if d.any():
                        first = (names[s], d.nonzero()[0].item(), d.nonzero()[-1].item())
                        break
                print("iteration", it, "stream", i, "differing bytes:", " ".join(bad), "| out differs:", not torch.equal(outs[i], ref_out), "| first:", first)
    print("done")
