"""Image-pyramid driver of the head: the build's counterpart of the hot loop of the reference's evaluation iterator
(reference os2d/engine/evaluate.py:278-371: for each pyramid level one backbone pass, then the heads of ALL
classes; os2d/data/dataloader.py:326 for the level sizes; os2d/config.py:194 for the default 7 scales).

MI355X-first differences from the reference loop:
  * all classes go through ONE class-batched head call per level (the reference loops B=1 heads);
  * the levels run back to back on the caller's stream by default; ``num_streams=7`` (or $OS2D_PYRAMID_STREAMS=7) gives every
    level its own HIP stream and workspace.  The streams were built so that the small levels fill the CUs left idle by the tail
    of the large ones - and measured SLOWER on every box of rounds 3 - 5 (128 classes x 7 levels: 25.03 ms on 7 streams against
    24.85 ms serial in round 4, 24.04 against 23.12 ms in round 5): every large kernel of the head owns its CU (>= 130 KB of LDS),
    nothing co-resides, and the interleaved streams cost the L2 locality of a level's kernel chain.  The per-level all-gather of
    the class-sharded variant runs on its own stream either way and overlaps the next level's compute;
  * nothing synchronises with the host until the caller asks for the results (the reference calls
    ``torch.cuda.synchronize()`` twice per level, evaluate.py:312,332).
"""
import os

import torch

from ..structures.feature_map import FeatureMapSize

DEFAULT_SCALES = (0.5, 0.625, 0.8, 1.0, 1.2, 1.4, 1.6)   # reference os2d/config.py:194


# One pool of side streams per device, shared by every runner: level slot i always maps to the same HIP stream, so the
# per-stream head workspaces (os2d_amd/modeling/head.py: one grow-only buffer per stream) stay at one per slot however
# many runners / images come and go.  (torch hands out fresh Stream objects round-robin from 32 native streams; a new
# set per image would eventually grow 32 workspaces to the size of the largest level.)
_STREAM_POOL = {}


def level_stream(device, slot):
    index = device.index if device.index is not None else torch.cuda.current_device()
    pool = _STREAM_POOL.setdefault(index, [])
    while len(pool) <= slot:
        pool.append(torch.cuda.Stream(device=device))
    return pool[slot]


def pyramid_sizes(img_size, scales=DEFAULT_SCALES):
    """reference dataloader.py:326: level size = (int(w*s), int(h*s))."""
    return [FeatureMapSize(w=int(img_size.w * s), h=int(img_size.h * s)) for s in scales]


class PyramidHeadRunner(object):
    """Runs ``head`` (an ``Os2dHead`` / ``ClassShardedHead``) on the feature maps of every pyramid level - back to back on the
    caller's stream (default) or on ``num_streams`` HIP streams, level i on stream i % num_streams.  ``features`` may be the backbone (a module mapping [1,3,h,w] -> [1,C,H,W]) or None when the
    caller passes feature maps directly."""

    def __init__(self, head, features=None, num_streams=None, device=None):
        self.head = head
        self.features = features
        self.device = device or (head.class_feature_maps.device if hasattr(head, "class_feature_maps") else torch.device("cuda"))
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        if num_streams is None:
            try:        # (an empty or malformed value means "not set")
                num_streams = max(1, int(os.environ.get("OS2D_PYRAMID_STREAMS", "1").strip() or "1"))
            except ValueError:
                num_streams = 1
        self._num_streams = num_streams
        self.largest_first = os.environ.get("OS2D_PYRAMID_ORDER", "given") == "largest"     # queue order of the levels

    def _stream(self, i):
        n = self._num_streams
        if n is not None and n <= 1:
            return torch.cuda.current_stream(self.device)
        return level_stream(self.device, i % n)

    def run(self, level_inputs, inputs_are_features=False):
        """level_inputs: list of image tensors [A,3,h_l,w_l] (or feature maps [A,C,H_l,W_l] if
        ``inputs_are_features``).  Returns per level lists (loc [A,B,4,HW], cls [A,B,HW], corners [A,B,8,HW],
        FeatureMapSize), laid out like ``Os2dModel.forward``; loc / corners entries are None when the head gathers score
        maps only (``ClassShardedHead(gather="scores")``).  The calling stream waits for all levels on return
        (stream-ordered, no host synchronisation)."""
        main = torch.cuda.current_stream(self.device)
        if hasattr(self.head, "prepare"):
            self.head.prepare()       # cached operands (packed weights, fp16 class split) are built on the caller's
                                      # stream, BEFORE the event every level stream waits on
        ready = torch.cuda.Event()
        ready.record(main)
        n = len(level_inputs)
        locs, clss, corners_l, sizes = [None] * n, [None] * n, [None] * n, [None] * n
        done = []
        # queue order of the levels: the caller's (default), or the largest level first (OS2D_PYRAMID_ORDER=largest: its kernel
        # chain is the longest - 31 % of the locations of the 7-scale pyramid sit in the 96 x 128 level).  Measured at 128
        # classes x 7 levels (round 3): 29.1 ms in the given (ascending) order, 29.6 ms largest first - the hardware queues
        # interleave the streams' kernels either way.  Results are returned in the caller's level order.
        order = sorted(range(n), key=lambda i: -(level_inputs[i].size(-1) * level_inputs[i].size(-2))) if self.largest_first else range(n)
        with torch.no_grad():
            for i in order:
                x = level_inputs[i]
                st = self._stream(i)
                st.wait_event(ready)
                with torch.cuda.stream(st):
                    fm = x if inputs_are_features else self.features(x)
                    loc, cls, _, corners = self.head(fm)
                    A, B = cls.size(0), cls.size(1)
                    locs[i] = loc.reshape(A, B, 4, -1) if loc is not None else None
                    clss[i] = cls.reshape(A, B, -1)
                    corners_l[i] = corners.reshape(A, B, 8, -1) if corners is not None else None
                    sizes[i] = FeatureMapSize(img=fm)
                    # caching-allocator bookkeeping: the input was allocated on the caller's stream and is read on
                    # `st`; the outputs are allocated on `st` and will be read on the caller's stream
                    x.record_stream(st)
                    for t in (loc, cls, corners):
                        if t is not None:
                            t.record_stream(main)
                    ev = torch.cuda.Event()
                    ev.record(st)
                    done.append(ev)
        for ev in done:
            main.wait_event(ev)
        return locs, clss, corners_l, sizes
