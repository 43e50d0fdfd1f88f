#!/usr/bin/env python3
"""Per-level stage times of the 7-level pyramid (BASELINE.json configs[4]: 128 classes per GPU), one level after the other on one
stream with the library's stage events: where do the 25 ms of the pyramid step go, level by level, against the 60 x 80 level's
per-location rate?   python tools/time_pyramid_levels.py [classes=128] [steps=5]"""
import ctypes
import json
import os
import sys

import torch

REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, REPO)
import bench  # noqa: E402
from os2d_amd.utils import synthetic  # noqa: E402

classes = int(sys.argv[1]) if len(sys.argv) > 1 else 128
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = torch.device("cuda:0")
w = bench.Workload(dev, 0, 1, classes, "v2", False, False, "all")
lib, L = w.lib, w._lib_mod
names = ["corr", "conv1", "conv2", "conv3", "sample", "fwd", "gemm", "inv"]
rows = []
for i, (h, wd) in enumerate(bench.LEVEL_HW):
    fm = synthetic.make_feature_map(bench.C_FEAT, h, wd, seed=100 + i).to(dev)
    P, Q, nb = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    t6 = (ctypes.c_int * 6)()
    lib.os2d_dft_sizes(h, wd, ctypes.byref(P), ctypes.byref(Q), ctypes.byref(nb), t6)
    with torch.no_grad():
        for _ in range(2):
            w.head(fm)
        sets = [w._new_event_set() for _ in range(steps)]
        torch.cuda.synchronize()
        for evs in sets:
            w.head(fm, stage_events=evs)
        torch.cuda.synchronize()
    ms = ctypes.c_float()
    acc = [0.0] * 8
    for evs in sets:
        pairs = [(0, 1), (2, 3), (4, 5), (6, 7), (8, 9), (10, 11), (11, 12), (12, 3)]
        for k, (a, b) in enumerate(pairs):
            L.check(lib.os2d_prof_event_elapsed_ms(evs[a], evs[b], ctypes.byref(ms)), "elapsed")
            acc[k] += ms.value / steps
    total = sum(acc[:5])
    rows.append({"level": [h, wd], "locations": h * wd, "transform": [P.value, Q.value], "tiles": [t6[0], t6[1]], "bins_total": nb.value * t6[0] * t6[1],
                 "ms": {n: round(v, 4) for n, v in zip(names, acc)}, "total_ms": round(total, 4)})
ref = next(r for r in rows if r["level"] == [60, 80])
print("level      loc  transform tiles   total | corr   fwd    gemm   inv    conv2  conv3  sample | per-location cost relative to 60x80: total corr fwd gemm inv conv2")
for r in rows:
    rel = lambda k: (r["ms"][k] / r["locations"]) / (ref["ms"][k] / ref["locations"])      # noqa: E731
    print("{:>3}x{:<4} {:>6} {:>3}x{:<3} {}x{}  {:7.3f} | {:.3f} {:.3f} {:.3f} {:.3f} {:.3f} {:.3f} {:.3f} | {:.2f}  {:.2f} {:.2f} {:.2f} {:.2f} {:.2f}".format(
        r["level"][0], r["level"][1], r["locations"], r["transform"][0], r["transform"][1], r["tiles"][0], r["tiles"][1], r["total_ms"],
        r["ms"]["corr"], r["ms"]["fwd"], r["ms"]["gemm"], r["ms"]["inv"], r["ms"]["conv2"], r["ms"]["conv3"], r["ms"]["sample"],
        (r["total_ms"] / r["locations"]) / (ref["total_ms"] / ref["locations"]), rel("corr"), rel("fwd"), rel("gemm"), rel("inv"), rel("conv2")))
# wall clock of the same 7 levels back to back on one stream (what bench.py's pyramid line times): launch gaps = wall - sum
import time  # noqa: E402
from os2d_amd.engine.pyramid import PyramidHeadRunner  # noqa: E402
fms = [synthetic.make_feature_map(bench.C_FEAT, h, wd, seed=100 + i).to(dev) for i, (h, wd) in enumerate(bench.LEVEL_HW)]
runner = PyramidHeadRunner(w.head, num_streams=1, device=dev)
with torch.no_grad():
    for _ in range(2):
        runner.run(fms, inputs_are_features=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        runner.run(fms, inputs_are_features=True)
    torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / steps * 1e3
tot = sum(r["total_ms"] for r in rows)
print("wall clock of the 7 levels back to back: {:.3f} ms per image (sum of the levels' stage times below: {:.3f} ms -> {:.3f} ms between the stages' kernels)".format(wall, tot, wall - tot))
ideal = ref["total_ms"] / ref["locations"] * sum(r["locations"] for r in rows)
print("sum of the levels {:.3f} ms; at the 60x80 level's per-location rate {:.3f} ms".format(tot, ideal))
print(json.dumps({"classes": classes, "levels": rows}))
