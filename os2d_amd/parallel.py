"""Class-parallel execution of the OS2D head: one process per GPU, classes sharded, RCCL all-gather over xGMI.

The reference has no multi-GPU path for the head.  Classes are independent units inside the
head (reference os2d/engine/evaluate.py:323-331 loops them with no cross-class state) and, with the default
``eval.nms_across_classes = False`` (reference os2d/config.py:202), in NMS too.  So the B class maps are split into
contiguous blocks over the R ranks, the image feature map and the TransformNet are replicated, every rank runs the
HIP head on its block, and one all-gather per output tensor (loc, cls, corners = 13 floats per class-location, 250 KB per
class at 60x80) assembles the full result on every rank before decode / NMS - written by RCCL straight into the final
[1, B, k, H, W] layout (rank r's block IS rows r*b .. r*b+b-1 of it), so nothing is copied after the collective.

xGMI is point-to-point (7 links x ~153 GB/s per GPU): a ring all-gather of the 32 MB/rank at B=1024 is per-link
bound at ~1.5 ms, against ~3 ms of kernels per rank (128 classes), so it is issued per image (three large collectives, not
per class) and asynchronously - waited for after the next image's kernels are queued; ``gather="scores"`` shrinks it 13x
when the caller only needs score maps before NMS.

``torch.distributed`` backend "nccl" is RCCL on ROCm; the pure tensor logic below also runs on the ``gloo`` backend
(CPU), which is how the N>1 path is tested without GPUs (tests/test_parallel_gloo.py).
"""
import torch
import torch.distributed as dist

OUT_CHANNELS = (4, 1, 8)      # loc, cls, corners


def shard_bounds(num_classes, world_size):
    """Contiguous block partition: the first (num_classes % world_size) ranks get one extra class.
    Returns a list of (start, end) per rank."""
    base, rem = divmod(int(num_classes), int(world_size))
    bounds, start = [], 0
    for r in range(world_size):
        n = base + (1 if r < rem else 0)
        bounds.append((start, start + n))
        start += n
    return bounds


def all_gather_detections(detections, group=None):
    """Class-sharded decode: every rank has decoded + NMS-ed ITS classes (``Os2dBoxCoder.decode_pyramid`` on the local
    block; NMS is per class, reference config.py:202), and only the surviving detections cross xGMI - a few thousand
    boxes of 18 floats instead of B x H x W score maps.  ``detections`` is the rank's ``BoxList`` (fields scores,
    labels, default_boxes, optionally transform_corners); returns the union on every rank, ordered by label
    (stable: rank order, then the rank's own order, i.e. decreasing score within a label).
    Two collectives: the per-rank counts, then ONE padded all-gather of the packed rows."""
    from .structures.bounding_box import BoxList
    world = dist.get_world_size(group)
    dev = detections.bbox_xyxy.device
    has_corners = detections.has_field("transform_corners")
    n = len(detections)
    cols = [detections.bbox_xyxy.view(n, 4), detections.get_field("scores").view(n, 1).float(),
            detections.get_field("default_boxes").bbox_xyxy.view(n, 4)]
    if has_corners:
        cols.append(detections.get_field("transform_corners").view(n, 8).float())
    packed = torch.cat(cols, dim=1)                                   # [n, 9 or 17] float32
    labels = detections.get_field("labels").view(n).to(torch.int64)
    counts = torch.empty(world, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(counts, torch.tensor([n], dtype=torch.int64, device=dev), group=group)
    counts = counts.tolist()
    n_max = max(counts)
    width = packed.size(1)
    send = torch.zeros(n_max, width + 2, dtype=torch.float32, device=dev)   # labels ride along as two exact fp32 halves
    if n:
        send[:n, :width] = packed
        send[:n, width] = (labels >> 20).float()
        send[:n, width + 1] = (labels & 0xFFFFF).float()
    recv = torch.empty(world * n_max, width + 2, dtype=torch.float32, device=dev)
    if n_max:
        dist.all_gather_into_tensor(recv, send, group=group)
    rows = torch.cat([recv[r * n_max:r * n_max + counts[r]] for r in range(world)], dim=0)
    lab = (rows[:, width].long() << 20) | rows[:, width + 1].long()
    order = torch.argsort(lab, stable=True)
    rows, lab = rows[order], lab[order]
    out = BoxList(rows[:, 0:4].contiguous(), detections.image_size)
    out.add_field("scores", rows[:, 4].contiguous())
    out.add_field("labels", lab)
    out.add_field("default_boxes", BoxList(rows[:, 5:9].contiguous(), detections.image_size))
    if has_corners:
        out.add_field("transform_corners", rows[:, 9:17].contiguous())
    return out


class _GatherBuffers(object):
    """Send / receive buffers of one (A, H, W) shape.  One tensor per output (loc, cls, corners) instead of one flat block:
    with one image per call (A = 1, the evaluation's batch size) and equal class counts, rank r's block [1, b, k, H, W] IS
    rows r*b .. r*b+b-1 of the full [1, B, k, H, W] tensor, so ``all_gather_into_tensor`` writes the final layout and
    nothing is re-copied afterwards (round 2 re-copied the whole gathered result: 256 MB per rank per image at 1024
    classes).  A > 1 or ragged counts still need the permute / trim copy."""

    def __init__(self, A, b_max, H, W, world, device, channels, ragged=False):
        make = torch.zeros if ragged else torch.empty        # ragged: the padding rows of a short rank are sent too (as zeros)
        self.send = [make(A, b_max, k, H, W, dtype=torch.float32, device=device) for k in channels]
        self.recv = [torch.empty(world, A, b_max, k, H, W, dtype=torch.float32, device=device) for k in channels]


class ClassShardedHead(object):
    """Class-parallel wrapper around ``Os2dHead``: build it on every rank with the SAME global list of class feature
    maps (or with ``local_head`` prebuilt for the rank's block); ``forward`` returns the full-size outputs on every
    rank.  Mirrors ``Os2dHead.forward``'s return signature.

    Per image and rank (B classes in total, b = B / world per rank, HW locations; ``gather="all"``): the head writes its
    13 * b * HW floats straight into the send tensors, three collectives (loc, cls, corners; one for ``gather="scores"``)
    move 13 * B * HW * 4 bytes into the receive tensors, and with A = 1 and equal counts those ARE the result - no further
    copy.  ``reuse_buffers=n`` keeps a ring of n buffer sets per map shape instead of drawing fresh ones from the caching
    allocator per image: results then alias a ring slot and stay valid until ``forward`` has been called n more times for
    that shape (n >= the number of results the caller holds at once: 1 + the gathers in flight)."""

    def __init__(self, head_creator, class_feature_maps=None, group=None, gather="all", num_classes=None,
                 local_head=None, reuse_buffers=0):
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised (launch with torchrun, one process per GPU)")
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        if gather not in ("all", "scores"):
            raise ValueError("gather must be 'all' or 'scores'")
        self.gather = gather
        if local_head is None:
            self.num_classes = len(class_feature_maps)
            self.bounds = shard_bounds(self.num_classes, self.world)
            s, e = self.bounds[self.rank]
            if e == s:
                raise RuntimeError("rank {} has no classes: need num_classes >= world_size".format(self.rank))
            self.local_head = head_creator.create_os2d_head(class_feature_maps[s:e])
        else:
            if num_classes is None:
                raise ValueError("num_classes is required with a prebuilt local_head")
            self.num_classes = num_classes
            self.bounds = shard_bounds(num_classes, self.world)
            self.local_head = local_head
            s, e = self.bounds[self.rank]
            assert local_head.class_batch_size == e - s, "local head holds {} classes, shard is {}".format(local_head.class_batch_size, e - s)
        self.counts = [e - s for s, e in self.bounds]
        self.reuse_buffers = int(reuse_buffers)
        self._rings = {}            # (A, H, W, device) -> [list of _GatherBuffers, next slot]
        self.copies_last_call = None    # how many result tensors the last forward had to re-copy after the gather (tests)
        self.wait_events = None         # a list: every wait for asynchronous gathers appends (before, after) timing events recorded
                                        # on the waiting stream - how long that stream stood still for the collective (bench.py)

    def prepare(self, precision=None):
        """Build the local head's cached operands on the current stream (see ``Os2dHead.prepare``)."""
        if hasattr(self.local_head, "prepare"):
            self.local_head.prepare(precision)
        return self

    def _buffers(self, A, H, W, device):
        channels = (1,) if self.gather == "scores" else OUT_CHANNELS
        ragged = len(set(self.counts)) > 1
        if self.reuse_buffers <= 0:
            return _GatherBuffers(A, max(self.counts), H, W, self.world, device, channels, ragged)
        key = (A, H, W, str(device))
        ring = self._rings.setdefault(key, [[], 0])
        if len(ring[0]) < self.reuse_buffers:
            ring[0].append(_GatherBuffers(A, max(self.counts), H, W, self.world, device, channels, ragged))
            return ring[0][-1]
        buf = ring[0][ring[1] % self.reuse_buffers]
        ring[1] += 1
        return buf

    def route_pairs(self, A):
        """The pair count the arithmetic ROUTE of the frequency-domain modes is decided on: the GLOBAL one, A * all classes.
        A ragged tail rank holding fewer than ``FFT_MIN_PAIRS`` classes then takes the same route as the other ranks - and
        as the unsharded head - so the sharded result is bit-equal to the unsharded one."""
        return A * self.num_classes

    def _assemble(self, recv, A, H, W):
        """Receive tensors [world, A, b_max, k, H, W] -> [A, B, k, H, W], classes in global order; a view when possible."""
        outs, copies = [], 0
        b_max = max(self.counts)
        equal = all(c == b_max for c in self.counts)
        for t in recv:
            k = t.size(3)
            if equal and A == 1:
                outs.append(t.view(1, self.world * b_max, k, H, W))              # the gather already wrote the final layout
            elif equal:
                outs.append(t.permute(1, 0, 2, 3, 4, 5).reshape(A, self.world * b_max, k, H, W))
                copies += 1
            else:
                outs.append(torch.cat([t[r, :, :self.counts[r]] for r in range(self.world)], dim=1))
                copies += 1
        self.copies_last_call = copies
        return outs

    def forward(self, feature_maps, async_gather=False):
        """Full-size (loc, cls, cls, corners) on every rank; with ``async_gather`` a zero-argument callable is returned
        instead that waits for the collectives and yields that tuple (call it after queueing more work)."""
        A, _, H, W = feature_maps.shape
        b_loc = self.counts[self.rank]
        b_max = max(self.counts)
        buf = self._buffers(A, H, W, feature_maps.device)
        scores_only = self.gather == "scores"
        dev = feature_maps.device

        def local(t):       # where the head writes this rank's b_loc classes: the send tensor itself whenever that is contiguous
            v = t[:, :b_loc]
            return v if v.is_contiguous() else torch.empty(A, b_loc, t.size(2), H, W, dtype=torch.float32, device=dev)
        if scores_only:
            # the head still computes all three outputs (one fused kernel); loc / corners stay local scratch
            out = (torch.empty(A, b_loc, 4, H, W, dtype=torch.float32, device=dev), local(buf.send[0]),
                   torch.empty(A, b_loc, 8, H, W, dtype=torch.float32, device=dev))
            sent = (out[1],)
        else:
            out = sent = tuple(local(t) for t in buf.send)
        self.local_head(feature_maps, out=out, route_pairs=self.route_pairs(A))
        for dst, src in zip(buf.send, sent):       # ragged tail rank with A > 1: exact-size results into the padded send tensors
            if src.data_ptr() != dst.data_ptr():
                dst[:, :b_loc] = src
        works = [dist.all_gather_into_tensor(r.view(-1), s.view(-1), group=self.group, async_op=async_gather)
                 for r, s in zip(buf.recv, buf.send)]

        def finish(_keep=(buf, out, sent)):
            if async_gather:
                timed = self.wait_events is not None and dev.type == "cuda"
                if timed:
                    before, after = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    before.record()
                for w in works:
                    w.wait()
                if timed:
                    after.record()
                    self.wait_events.append((before, after))
            r = self._assemble(buf.recv, A, H, W)
            return (None, r[0], r[0], None) if scores_only else (r[0], r[1], r[1], r[2])
        return finish if async_gather else finish()

    __call__ = forward
