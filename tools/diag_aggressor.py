#!/usr/bin/env python3
"""Diagnostic (docs/DESIGN_HISTORY_r1-r3.md section 8, the packed-FP32 finding): a victim kernel on four streams while the direct 7x7 kernel
(half-precision MFMA at full rate) runs on three others - are the victim's outputs still the bytes it produces alone?

    OS2D_HIP_LIB=tools/diag_libs/<tag>/libos2d_hip.so python tools/diag_aggressor.py [--rounds 200] [--victims fft,gemm16,corr] [--quiet-control]

Prints one summary line per victim: rounds with a difference / rounds, for the control (no aggressor) and the contended run,
and for the first few differing outputs where they differ (the transforms: image, bins -> lanes of the last register stage).
"""
import argparse
import os
import sys

import torch

REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
from concurrency_victims import Harness, NF  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--rounds", type=int, default=200)
ap.add_argument("--victims", default="fft,gemm16,corr")
ap.add_argument("--control-rounds", type=int, default=20)
args = ap.parse_args()
dev = torch.device("cuda:0")
h = Harness(dev, NB=128)
tag = os.environ.get("OS2D_HIP_LIB", "product library")
print("library:", tag, "| device:", torch.cuda.get_device_name(0), flush=True)
for kind in args.victims.split(","):
    ref = h.reference(kind)
    counts = {}
    for label, aggr, rounds in (("control", False, args.control_rounds), ("contended", True, args.rounds)):
        nbad, shown = 0, 0
        for r in range(rounds):
            bad, out = h.contend(kind, ref, aggressor=aggr)
            if bad:
                nbad += 1
                if shown < 3:
                    shown += 1
                    i, k = bad[0]
                    a, b = out[i][k], ref[i][k]
                    if a.dtype == torch.uint8:
                        d = (a != b).nonzero().flatten()
                        print("  {} {} round {} stream {} output {}: {} bytes differ, offsets {} .. {}".format(
                            kind, label, r, i, k, int(d.numel()), d[:4].tolist(), d[-2:].tolist()))
                    else:
                        d = (a != b).view(-1, a.shape[-2], a.shape[-1]) if a.dim() > 2 else (a != b)
                        rows = d.flatten(1).any(dim=1).nonzero().flatten().tolist()
                        first = d[rows[0]].nonzero()
                        av, bv = a.view(d.shape)[rows[0]], b.view(d.shape)[rows[0]]
                        p = tuple(first[0].tolist())
                        print("  {} {} round {} stream {} output {}: {} rows differ (first {}), {} words in it, positions {} .. {}; got {} want {}; "
                              "max abs diff {:.3g} (max abs value {:.3g})".format(
                                  kind, label, r, i, k, len(rows), rows[:4], int(first.size(0)), first[0].tolist(), first[-1].tolist(),
                                  float(av[p]), float(bv[p]), float((av - bv).abs().max()), float(bv.abs().max())))
        counts[label] = (nbad, rounds)
    print("RESULT victim={} lib={} control {}/{} contended {}/{} rounds differ ({} victim launches per round)".format(
        kind, tag, counts["control"][0], counts["control"][1], counts["contended"][0], counts["contended"][1], 3 * NF), flush=True)
