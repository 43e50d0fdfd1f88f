#!/usr/bin/env python3
"""GPU timing of the decode + NMS step that follows the head (Os2dBoxCoder.decode_pyramid) on head outputs."""
import os, sys, time
import torch
REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, REPO)
from os2d_amd.modeling.head import build_os2d_head_creator
from os2d_amd.modeling.box_coder import Os2dBoxCoder
from os2d_amd.structures.feature_map import FeatureMapSize
from os2d_amd.utils import synthetic

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
thr = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
P, inverse = 6, True
state = synthetic.make_transform_net_state(P, seed=1)
creator = build_os2d_head_creator(False, False, inverse, FeatureMapSize(w=16, h=16), FeatureMapSize(w=16, h=16))
creator.aligner.parameter_regressor.load_state_dict(state)
creator.to(dev).eval()
fm = synthetic.make_feature_map(1024, 60, 80, seed=0).to(dev)
with torch.no_grad():
    head = creator.create_os2d_head([c.to(dev) for c in synthetic.make_class_feature_maps(B, 1024, seed=1000)])
    loc, cls, _, corners = head(fm)
coder = Os2dBoxCoder(output_box_grid_generator=creator.box_grid_generator_image_level)
img = FeatureMapSize(w=1280, h=960)
loc_l, cls_l, cor_l = loc[0].reshape(B, 4, -1), cls[0].reshape(B, -1), corners[0].reshape(B, 8, -1)
for it in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    res = coder.decode_pyramid([loc_l], [cls_l], [img], class_ids=list(range(B)), nms_score_threshold=thr,
                               nms_iou_threshold=0.3, transform_corners_pyramid=[cor_l])
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("decode_pyramid B={} thr={}: {:.2f} ms, {} detections".format(B, thr, dt * 1e3, len(res)))

# ---- breakdown with the torch profiler (kernel self times)
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    res = coder.decode_pyramid([loc_l], [cls_l], [img], class_ids=list(range(B)), nms_score_threshold=thr,
                               nms_iou_threshold=0.3, transform_corners_pyramid=[cor_l])
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=12, max_name_column_width=60))

# ---- 7-level pyramid (BASELINE.json configs[4]): all levels of a class are merged before NMS (generic path)
if "--pyramid" in sys.argv:
    from os2d_amd.engine.pyramid import PyramidHeadRunner
    from os2d_amd.modeling.box_coder import ResizeBoxes
    level_hw = [(30, 40), (38, 50), (48, 64), (60, 80), (72, 96), (84, 112), (96, 128)]
    fms = [synthetic.make_feature_map(1024, h, w, seed=100 + i).to(dev) for i, (h, w) in enumerate(level_hw)]
    sizes = [FeatureMapSize(w=16 * w, h=16 * h) for h, w in level_hw]
    with torch.no_grad():
        loc_p, cls_p, cor_p, _ = PyramidHeadRunner(head, device=dev).run(fms, inputs_are_features=True)
    loc_p = [l[0] for l in loc_p]; cls_p = [c[0] for c in cls_p]; cor_p = [k[0] for k in cor_p]
    inverse = [ResizeBoxes(FeatureMapSize(w=3264, h=2448)) for _ in sizes]
    for fused in (False, True):
        coder.use_fused_level_kernel = fused
        for it in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            res = coder.decode_pyramid(loc_p, cls_p, sizes, class_ids=list(range(B)), nms_score_threshold=thr,
                                       nms_iou_threshold=0.3, inverse_box_transforms=inverse, transform_corners_pyramid=cor_p)
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
            print("pyramid decode ({}) B={} thr={}: {:.2f} ms, {} detections".format(
                "os2d_detect_pyramid" if fused else "generic chain", B, thr, dt * 1e3, len(res)))
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        res = coder.decode_pyramid(loc_p, cls_p, sizes, class_ids=list(range(B)), nms_score_threshold=thr,
                                   nms_iou_threshold=0.3, inverse_box_transforms=inverse, transform_corners_pyramid=cor_p)
        torch.cuda.synchronize()
    print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=10, max_name_column_width=60))
    # merged labels (class-image views, reference evaluate.py:241-269): B head rows = B / V labels x V views, 7 levels
    if "--views" in sys.argv:
        V = int(sys.argv[sys.argv.index("--views") + 1])
        ids = [b // V for b in range(B)]
        for fused in (False, True):
            coder.use_fused_level_kernel = fused
            for it in range(3):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                res = coder.decode_pyramid(loc_p, cls_p, sizes, class_ids=ids, nms_score_threshold=thr,
                                           nms_iou_threshold=0.3, inverse_box_transforms=inverse, transform_corners_pyramid=cor_p)
                torch.cuda.synchronize(); dt = time.perf_counter() - t0
                print("pyramid decode, merged labels ({}) {} labels x {} views thr={}: {:.2f} ms, {} detections".format(
                    "os2d_detect_pyramid_merged" if fused else "generic chain", B // V, V, thr, dt * 1e3, len(res)))
