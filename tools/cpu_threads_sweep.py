#!/usr/bin/env python3
"""Host-side sweep behind bench.py's cpu_baseline thread count: the same looped one-class-at-a-time call pattern at
several torch thread counts (the GPU is not used)."""
import os, sys
REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, REPO)
import bench  # noqa: E402
from os2d_amd.utils import synthetic  # noqa: E402

state = synthetic.make_transform_net_state(6, seed=1)
fm = synthetic.make_feature_map(1024, 60, 80, seed=0)
cf = synthetic.make_class_feature_maps(16, 1024, sizes=[(15, 15)], seed=1000)
for n in [int(a) for a in sys.argv[1:]] or [1, 8, 16, 32, 64, 128, 256]:
    os.environ["OS2D_CPU_THREADS"] = str(n)
    r = bench.cpu_baseline(fm, cf, state, True, 6.0)
    print(n, "threads:", r["value"], "pairs/s", flush=True)
