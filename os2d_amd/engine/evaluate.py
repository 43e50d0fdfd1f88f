"""The caller of the hot path, rebuilt for the class-batched HIP head: what the reference's
``make_iterator_extract_scores_from_images_batched`` (os2d/engine/evaluate.py:177-371) and the decode step of
``evaluate`` (:99-117) do for one image, minus datasets / dataloaders / mAP (out of scope).

    class_image_views  reference evaluate.py:241-269  class-image augmentation (rotation90 / horflip / both): every
                       view becomes one more row of the class-batched head, NMS merges the views of a class
    build_class_head   reference evaluate.py:232-274  one backbone pass per class image -> one class-batched head
                       (same-size class images go through the backbone as ONE batch instead of one call each)
    extract_scores     reference evaluate.py:306-361  per pyramid level: backbone + heads of all classes
    detect             reference evaluate.py:99-117   + Os2dBoxCoder.decode_pyramid (decode, clip, NMS, merge levels)
    detect_images      reference evaluate.py:278-371  the per-image loop, with the next image's host-to-device copy
                       prefetched on a side stream
"""
from collections import OrderedDict

import torch

from ..modeling.box_coder import ResizeBoxes
from ..structures.feature_map import FeatureMapSize
from .pyramid import PyramidHeadRunner


CLASS_IMAGE_AUGMENTATIONS = ("", "rotation90", "horflip", "horflip_rotation90")


def class_image_views(class_images, class_ids, class_image_augmentation=""):
    """Views of every class image in the reference's order (evaluate.py:241-269): rotation90 -> [im, 90, 180, 270]
    (``rot90`` over the two spatial dims), horflip -> [im, flipped], horflip_rotation90 -> the four rotations followed
    by the horizontal flip of each.  Returns (views, view_class_ids, num_class_views); the id of view l is
    ``class_ids[l // num_class_views]`` (evaluate.py:294), so ``decode_pyramid`` merges the views of a class before NMS."""
    if class_image_augmentation not in CLASS_IMAGE_AUGMENTATIONS and class_image_augmentation is not None:
        raise RuntimeError("Unknown value of class_image_augmentation: {}".format(class_image_augmentation))
    views, view_ids = [], []
    num_class_views = 1
    for im, cid in zip(class_images, class_ids):
        if im.dim() == 4:
            im = im.squeeze(0)
        mine = [im]
        if class_image_augmentation in ("rotation90", "horflip_rotation90"):
            for _ in range(3):
                mine.append(mine[-1].rot90(1, [1, 2]))
        if class_image_augmentation == "horflip":
            mine.append(im.flip(2))
        elif class_image_augmentation == "horflip_rotation90":
            mine += [v.flip(2) for v in mine[:4]]
        num_class_views = len(mine)
        views += [v.contiguous() for v in mine]
        view_ids += [cid] * len(mine)
    return views, view_ids, num_class_views


def build_class_head(net, class_images, batch_same_size=True):
    """class_images: list of [3,h,w] device tensors (already normalised).  Returns the ``Os2dHead`` of all classes in
    the given order.  The reference extracts class features one image at a time (model.py:80-88); images of equal size
    are stacked here so the backbone sees a few large batches (identical arithmetic per image).
    With class-image augmentation pass the views of ``class_image_views``."""
    feats = [None] * len(class_images)
    with torch.no_grad():
        if batch_same_size:
            groups = OrderedDict()
            for i, img in enumerate(class_images):
                groups.setdefault(tuple(img.shape), []).append(i)
            for idx in groups.values():
                out = net.net_label_features.net_class_features(torch.stack([class_images[i] for i in idx], 0))
                for k, i in enumerate(idx):
                    feats[i] = out[k:k + 1]
        else:
            feats = net.net_label_features(class_images)
        return net.os2d_head_creator.create_os2d_head(feats)


def extract_scores(net, image_levels, class_head, per_level_streams=False):
    """image_levels: list of [A,3,h_l,w_l] tensors (the image pyramid, reference dataloader.py:326-347).
    Returns dict(loc, cls, corners, fm_sizes, img_sizes) with one entry per level, shaped like ``Os2dModel.forward``:
    loc [A,B,4,HW], cls [A,B,HW], corners [A,B,8,HW]."""
    runner = PyramidHeadRunner(class_head, features=net.net_feature_maps,
                               num_streams=len(image_levels) if per_level_streams else 1, device=image_levels[0].device)
    loc, cls, corners, fm_sizes = runner.run(image_levels)
    return dict(loc=loc, cls=cls, corners=corners, fm_sizes=fm_sizes,
                img_sizes=[FeatureMapSize(img=x) for x in image_levels])


def detect(net, box_coder, image_levels, class_head, class_ids, orig_size=None, nms_score_threshold=float("-inf"),
           nms_iou_threshold=0.3, image_index=0, per_level_streams=False):
    """Detections of ONE image (``image_index`` of the batch) over the whole pyramid as a ``BoxList`` in the
    coordinates of ``orig_size`` (default: the first level's size), fields scores / labels / default_boxes /
    transform_corners.  Thresholds default to the reference's eval config (config.py:198-200)."""
    s = extract_scores(net, image_levels, class_head, per_level_streams)
    a = image_index
    inverse = None
    if orig_size is None and len({(z.w, z.h) for z in s["img_sizes"]}) > 1:
        orig_size = s["img_sizes"][0]       # levels of different sizes can only be merged in a common frame: the first level's
    if orig_size is not None:
        inverse = [ResizeBoxes(orig_size) for _ in image_levels]
    return box_coder.decode_pyramid([l[a] for l in s["loc"]], [c[a] for c in s["cls"]], s["img_sizes"], class_ids,
                                    nms_score_threshold=nms_score_threshold, nms_iou_threshold=nms_iou_threshold,
                                    inverse_box_transforms=inverse,
                                    transform_corners_pyramid=[k[a] for k in s["corners"]])


def detect_images(net, box_coder, image_pyramids, class_head, class_ids, orig_sizes=None, device=None, **detect_kwargs):
    """Generator over images: the build's counterpart of the reference's evaluation iterator (evaluate.py:278-371 +
    :99-117) for host-resident inputs.  ``image_pyramids`` is an iterable of lists of CPU tensors [1,3,h_l,w_l] (one
    list per image, e.g. from a dataloader); the levels of image i+1 are copied to the device on a side stream
    (pinned staging buffers, ``non_blocking``) while image i is in the backbone / head / decode, so the PCIe transfer
    (4.9 MB at 1280x960, ~0.1 ms) never sits in front of the compute.  Yields one ``BoxList`` per image, identical to
    calling ``detect`` on that image."""
    device = device or class_head.class_feature_maps.device
    copy_stream = torch.cuda.Stream(device=device)

    def upload(levels):
        with torch.cuda.stream(copy_stream):
            out = [(x if x.is_pinned() else x.pin_memory()).to(device, non_blocking=True) for x in levels]
        ready = torch.cuda.Event()
        ready.record(copy_stream)
        return out, ready

    it = iter(image_pyramids)
    try:
        nxt = upload(next(it))
    except StopIteration:
        return
    index = 0
    while nxt is not None:
        levels, ready = nxt
        try:
            nxt = upload(next(it))          # in flight while this image is processed
        except StopIteration:
            nxt = None
        main = torch.cuda.current_stream(device)
        main.wait_event(ready)
        for x in levels:
            x.record_stream(main)           # allocated on the copy stream, consumed on the compute stream
        orig = orig_sizes[index] if orig_sizes is not None else None
        yield detect(net, box_coder, levels, class_head, class_ids, orig_size=orig, **detect_kwargs)
        index += 1
