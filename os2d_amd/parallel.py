"""Class-parallel execution of the OS2D head: one process per GPU, classes sharded, RCCL all-gather over xGMI.

The reference has no multi-GPU path for the head.  Classes are independent units inside the
head (reference os2d/engine/evaluate.py:323-331 loops them with no cross-class state) and, with the default
``eval.nms_across_classes = False`` (reference os2d/config.py:202), in NMS too.  So the B class maps are split into
contiguous blocks over the R ranks, the image feature map and the TransformNet are replicated, every rank runs the
HIP head on its block, and ONE all-gather of the per-class output maps (cls | loc | corners = 13 floats per
class-location, 250 KB per class at 60x80) assembles the full result on every rank before decode / NMS.

xGMI is point-to-point (7 links x ~153 GB/s per GPU): a ring all-gather of the 32 MB/rank at B=1024 is per-link
bound at ~1.5 ms, against >= 15 ms of MFMA work per rank, so a single fused collective per image is the right
granularity; ``gather="scores"`` shrinks it 13x when the caller only needs score maps before NMS.

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


def alloc_gather_buffer(A, b_max, H, W, device, channels=OUT_CHANNELS):
    """One flat send buffer holding the rank's [A,b_max,k,H,W] blocks back to back (k = 4, 1, 8), so that the
    kernels write their outputs straight into what the collective sends (no packing copy)."""
    sizes = [A * b_max * k * H * W for k in channels]
    flat = torch.empty(sum(sizes), dtype=torch.float32, device=device)
    views, off = [], 0
    for k, n in zip(channels, sizes):
        views.append(flat[off:off + n].view(A, b_max, k, H, W))
        off += n
    return flat, views


def _assemble(gathered, counts, A, H, W, channels):
    """[world, flat] gathered buffers -> full tensors [A, sum(counts), k, H, W] per k, classes in global order."""
    world = len(counts)
    b_max = max(counts)
    gathered = gathered.view(world, -1)
    outs, off = [], 0
    for k in channels:
        n = A * b_max * k * H * W
        block = gathered[:, off:off + n].view(world, A, b_max, k, H, W)
        off += n
        if all(c == b_max for c in counts):
            full = block.permute(1, 0, 2, 3, 4, 5).reshape(A, world * b_max, k, H, W)
        else:
            full = torch.cat([block[r, :, :counts[r]] for r in range(world)], dim=1)
        outs.append(full.contiguous())
    return outs


class PendingGather(object):
    """Handle of an all-gather in flight (``async_op=True``): ``wait()`` makes the current stream wait for the
    collective and returns the assembled tensors.  Lets the caller launch the next image / pyramid level before the
    previous gather has finished, so the xGMI transfer hides behind compute."""

    def __init__(self, work, gathered, keep_alive, counts, A, H, W, channels):
        self._work, self._gathered, self._keep = work, gathered, keep_alive
        self._args = (counts, A, H, W, channels)
        self._result = None

    def wait(self):
        if self._result is None:
            self._work.wait()
            self._result = _assemble(self._gathered, *self._args)
            self._keep = None
        return self._result


def all_gather_class_outputs(flat_local, counts, A, H, W, group=None, channels=OUT_CHANNELS, async_op=False):
    """All-gather the per-rank flat buffers (``alloc_gather_buffer`` layout, padded to b_max = max(counts) classes)
    and return the full tensors [A, sum(counts), k, H, W] for each k in ``channels``, classes in global order
    (or a ``PendingGather`` when ``async_op``)."""
    world = dist.get_world_size(group)
    assert len(counts) == world
    gathered = torch.empty(world * flat_local.numel(), dtype=flat_local.dtype, device=flat_local.device)
    work = dist.all_gather_into_tensor(gathered, flat_local, group=group, async_op=async_op)
    if async_op:
        return PendingGather(work, gathered, flat_local, list(counts), A, H, W, tuple(channels))
    return _assemble(gathered, list(counts), A, H, W, tuple(channels))


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


class ClassShardedHead(object):
    """Class-parallel wrapper around ``Os2dHead``: build it on every rank with the SAME global list of class feature
    maps (or with ``local_head`` prebuilt for the rank's block); ``forward`` returns the full-size outputs on every
    rank.  Mirrors ``Os2dHead.forward``'s return signature."""

    def __init__(self, head_creator, class_feature_maps=None, group=None, gather="all", num_classes=None,
                 local_head=None):
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

    def prepare(self, precision=None):
        """Build the local head's cached operands on the current stream (see ``Os2dHead.prepare``)."""
        if hasattr(self.local_head, "prepare"):
            self.local_head.prepare(precision)
        return self

    def forward(self, feature_maps, async_gather=False):
        """Full-size (loc, cls, cls, corners) on every rank; with ``async_gather`` a zero-argument callable is returned
        instead that waits for the collective and yields that tuple (call it after queueing more work)."""
        A, _, H, W = feature_maps.shape
        b_loc = self.counts[self.rank]
        b_max = max(self.counts)
        flat, (loc, cls, corners) = alloc_gather_buffer(A, b_max, H, W, feature_maps.device)
        if b_loc == b_max:
            out = (loc, cls, corners)
            self.local_head(feature_maps, out=out)
        else:   # ragged tail rank: compute into exact-size tensors, then place into the padded send buffer
            l, c, _, k = self.local_head(feature_maps)
            flat.zero_()
            loc[:, :b_loc], cls[:, :b_loc], corners[:, :b_loc] = l, c, k
        if self.gather == "scores":
            res = all_gather_class_outputs(cls.reshape(-1), self.counts, A, H, W, self.group, channels=(1,),
                                           async_op=async_gather)
            finish = lambda r: (None, r[0], r[0], None)
        else:
            res = all_gather_class_outputs(flat, self.counts, A, H, W, self.group, async_op=async_gather)
            finish = lambda r: (r[0], r[1], r[1], r[2])
        if async_gather:
            # the closure keeps ``flat`` (the send buffer; for scores a view of it is sent) alive until the wait
            return lambda _keep=flat: finish(res.wait())
        return finish(res)

    __call__ = forward
