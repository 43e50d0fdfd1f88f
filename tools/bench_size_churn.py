#!/usr/bin/env python3
"""Transform-size churn of the frequency-domain 7x7 layer over a dataset of many image sizes.

The reference's evaluation feeds every image at its own aspect ratio: the dataloader resizes an image of h x w pixels to the
area of target^2 (reference os2d/utils/utils.py:32-37, called from os2d/data/dataloader.py with eval.dataset_scales) and builds
the 7-scale pyramid of that (os2d/data/dataloader.py:326, os2d/config.py:194).  The weight spectra of the 7x7 layer are cached
per TRANSFORM size (P, Q); this tool draws N image sizes with Grozi-like aspect ratios, runs the class-batched head on
synthetic features of every pyramid level and reports: distinct map sizes, distinct transform sizes, cache hits / misses,
the cost of a miss, and the amortised time per image with a cold and with a warm cache.

    python tools/bench_size_churn.py [--images 48] [--classes 64] [--target 1280] [--cache-gb 16]
"""
import argparse
import json
import math
import os
import sys
import time

import torch

REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, REPO)
from os2d_amd.modeling import head as head_mod  # noqa: E402
from os2d_amd.modeling.box_coder import feature_map_size_c4  # noqa: E402
from os2d_amd.structures.feature_map import FeatureMapSize  # noqa: E402
from os2d_amd.utils import synthetic  # noqa: E402

SCALES = (0.5, 0.625, 0.8, 1.0, 1.2, 1.4, 1.6)     # reference os2d/config.py:194


def image_sizes(n, target, seed=0):
    """n (h, w) pairs: photo aspect ratios 3:4 / 4:3 / 2:3 / 16:9 ... with +-8 % jitter (crops), resized like the reference."""
    g = torch.Generator().manual_seed(seed)
    base = [4 / 3, 3 / 4, 3 / 2, 2 / 3, 16 / 9, 9 / 16, 1.0, 5 / 4]
    out = []
    for i in range(n):
        ar = base[i % len(base)] * (1.0 + 0.16 * (float(torch.rand(1, generator=g)) - 0.5))      # h / w
        w = int(target / math.sqrt(ar))
        h = int(target * math.sqrt(ar))
        out.append((max(h, 1), max(w, 1)))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--images", type=int, default=48)
    ap.add_argument("--classes", type=int, default=64)
    ap.add_argument("--target", type=int, default=1280)
    ap.add_argument("--cache-gb", type=float, default=None)
    ap.add_argument("--channels", type=int, default=1024)
    args = ap.parse_args()
    if args.cache_gb is not None:
        os.environ["OS2D_FFT_CACHE_BYTES"] = str(int(args.cache_gb * (1 << 30)))
    dev = torch.device("cuda:0")
    P, inverse = 6, True
    creator = head_mod.build_os2d_head_creator(False, False, inverse, FeatureMapSize(w=16, h=16), FeatureMapSize(w=16, h=16))
    creator.aligner.parameter_regressor.load_state_dict(synthetic.make_transform_net_state(P, seed=1))
    creator.to(dev).eval()
    net = creator.aligner.parameter_regressor
    class_fms = [c.to(dev) for c in synthetic.make_class_feature_maps(args.classes, args.channels, seed=1000)]
    sizes = image_sizes(args.images, args.target)
    maps = []
    for h, w in sizes:
        lv = []
        for s in SCALES:
            fm = feature_map_size_c4(FeatureMapSize(w=int(w * s), h=int(h * s)))
            lv.append((fm.h, fm.w))
        maps.append(lv)
    lib = head_mod._lib.load()
    import ctypes
    tsizes = set()
    for lv in maps:
        for h, w in lv:
            p, q, nb = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
            lib.os2d_dft_sizes(h, w, ctypes.byref(p), ctypes.byref(q), ctypes.byref(nb), None)    # the default precision's planner
            tsizes.add((p.value, q.value))
    # features per distinct map size (the content does not matter for the timing)
    feats = {}
    for lv in maps:
        for hw in lv:
            if hw not in feats:
                feats[hw] = synthetic.make_feature_map(args.channels, hw[0], hw[1], seed=hw[0] * 1000 + hw[1]).to(dev)
    misses = {"n": 0, "s": 0.0, "sync_before": False}
    real = net.spectra.__func__

    def counting(self, H, W, split=False):
        before = len(self._spectra_cache), [id(v) for v in self._spectra_cache.values()]
        if misses.get("sync_before"):
            torch.cuda.synchronize()          # so that a miss is timed on its own, not together with the kernels queued before it
        t0 = time.perf_counter()
        r = real(self, H, W, split)
        after = [id(v) for v in self._spectra_cache.values()]
        if set(after) - set(before[1]):
            torch.cuda.synchronize()
            misses["n"] += 1
            misses["s"] += time.perf_counter() - t0
        return r
    net.spectra = counting.__get__(net)
    res = {}
    with torch.no_grad():
        head = creator.create_os2d_head(class_fms)
        head(feats[maps[0][3]])
        torch.cuda.synchronize()
        net._spectra_cache.clear()
        for label in ("cold", "warm"):
            misses["n"], misses["s"] = 0, 0.0
            calls = 0
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for lv in maps:
                for hw in lv:
                    head(feats[hw])
                    calls += 1
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            res[label] = {"ms_per_image": round(dt / len(maps) * 1e3, 2), "head_calls": calls, "cache_misses": misses["n"],
                          "hit_rate": round(1.0 - misses["n"] / calls, 4), "ms_per_miss": round(misses["s"] / max(misses["n"], 1) * 1e3, 2),
                          "miss_ms_per_image": round(misses["s"] / len(maps) * 1e3, 2)}
    # the cost of a miss without the pipeline drain: a third pass over an emptied cache with a synchronisation BEFORE every call
    net._spectra_cache.clear()
    misses["n"], misses["s"], misses["sync_before"] = 0, 0.0, True
    with torch.no_grad():
        for lv in maps:
            for hw in lv:
                head(feats[hw])
    torch.cuda.synchronize()
    isolated = {"cache_misses": misses["n"], "ms_per_miss": round(misses["s"] / max(misses["n"], 1) * 1e3, 2)}
    out = {"images": len(maps), "classes": args.classes, "levels_per_image": len(SCALES), "distinct_map_sizes": len(feats),
           "distinct_transform_sizes": len(tsizes), "cache_entries_at_end": len(net._spectra_cache),
           "cache_bytes_at_end": sum(c.nbytes() for c in net._spectra_cache.values()),
           "cache_cap_bytes": head_mod.spectra_cache_cap_bytes(), "size_policy": os.environ.get("OS2D_DFT_SIZES", "canonical"),
           "transform_sizes": sorted(tsizes), "cold": res["cold"], "warm": res["warm"], "miss_isolated": isolated,
           "miss_ms_derived": round((res["cold"]["ms_per_image"] - res["warm"]["ms_per_image"]) * len(maps) / max(res["cold"]["cache_misses"], 1), 2),
           "note": "cold.ms_per_miss includes draining the kernels queued before the miss (the head is asynchronous); miss_isolated times a "
                   "miss on its own, miss_ms_derived = (cold - warm) / misses",
           "aspect_ratios": "3:4, 4:3, 2:3, 3:2, 9:16, 16:9, 1:1, 4:5 with +-8 % jitter, resized to area target^2 like reference utils.py:32-37"}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
