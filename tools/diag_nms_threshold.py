#!/usr/bin/env python3
import os, sys
import numpy as np, torch
REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
from os2d_amd.modeling.box_coder import Os2dBoxCoder
dev = torch.device("cuda:0")
rs = np.random.RandomState(3)
pairs = []
for k in range(1500):
    s = float(rs.uniform(0.5, 40.0))
    up = np.float32(np.inf if k % 2 else -np.inf)
    y = np.float32(3.0 * s)
    for _ in range(int(rs.randint(0, 4)) if k % 7 else 0):
        y = np.nextafter(y, up, dtype=np.float32)
    pairs.append([[0.0, 0.0, 10.0 * s, 10.0 * s], [0.0, 0.0, 10.0 * s, float(y)]])
b = torch.tensor(pairs, dtype=torch.float32)
a1 = (b[:, 0, 2] - b[:, 0, 0]) * (b[:, 0, 3] - b[:, 0, 1])
a2 = (b[:, 1, 2] - b[:, 1, 0]) * (b[:, 1, 3] - b[:, 1, 1])
w = torch.minimum(b[:, 0, 2], b[:, 1, 2]) - torch.maximum(b[:, 0, 0], b[:, 1, 0])
h = torch.minimum(b[:, 0, 3], b[:, 1, 3]) - torch.maximum(b[:, 0, 1], b[:, 1, 1])
inter = w.clamp(min=0) * h.clamp(min=0)
iou = inter / (a1 + a2 - inter)
ref_keep2 = ~(iou > 0.3)
keep = Os2dBoxCoder.nms_sorted(b.to(dev), torch.full((b.size(0),), 2), 0.3).cpu()
bad = (keep[:, 1] != ref_keep2).nonzero().flatten()
print("mismatches:", bad.numel(), "of", b.size(0))
for i in bad[:8].tolist():
    print(i, "boxes", b[i].tolist(), "iou %.9g" % float(iou[i]), "iou-0.3 %.3e" % (float(iou[i]) - float(np.float32(0.3))), "gpu keep2", bool(keep[i, 1]), "ref keep2", bool(ref_keep2[i]),
          "inter %.9g a1 %.9g a2 %.9g" % (float(inter[i]), float(a1[i]), float(a2[i])))
