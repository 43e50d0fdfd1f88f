#!/usr/bin/env python3
"""Experiment: the nine launches of a 64-class head call replayed from a captured graph (torch.cuda.CUDAGraph = hipGraph) against the
same launches issued one by one - do the dependent-launch gaps shrink?  Timing only (a replay re-uses the captured call's epoch)."""
import os, sys, time
import torch
REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, REPO)
import bench
dev = torch.device("cuda:0")
classes = int(sys.argv[1]) if len(sys.argv) > 1 else 64
w = bench.Workload(dev, 0, 1, classes, "v2", False, False, "all")
fm = w.fm
with torch.no_grad():
    for _ in range(5):
        out = w.head(fm, precision="fftx3")
    torch.cuda.synchronize()
    def timed(fn, n=40):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): fn()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
    eager = timed(lambda: w.head(fm, precision="fftx3"))
    s = torch.cuda.Stream(device=dev)
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3): w.head(fm, precision="fftx3")
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    try:
        with torch.cuda.graph(g):
            out_g = w.head(fm, precision="fftx3")
        replay = timed(g.replay)
        eager2 = timed(lambda: w.head(fm, precision="fftx3"))
        print("classes {}: eager {:.4f} ms, graph replay {:.4f} ms, eager again {:.4f} ms".format(classes, eager, replay, eager2))
        ok = all(torch.equal(a, b) for a, b in zip(out, out_g) if a is not None and b is not None)
        print("replayed outputs equal the eager call's:", ok)
    except Exception as e:   # noqa: BLE001
        print("capture failed:", type(e).__name__, str(e)[:300])
        print("eager {:.4f} ms".format(eager))
