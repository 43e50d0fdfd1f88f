"""CPU, world_size 2 over gloo: the class-sharded path (shard plan, gather buffer layout, all-gather reassembly,
ClassShardedHead.forward incl. the ragged last rank and the scores-only mode).  The per-rank compute is supplied by
the ORACLE here (test infrastructure standing in for the HIP head, which needs a GPU); what is under test is the
distributed logic of os2d_amd/parallel.py."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from os2d_amd.parallel import shard_bounds


def test_shard_bounds():
    assert shard_bounds(1024, 8) == [(i * 128, (i + 1) * 128) for i in range(8)]
    assert shard_bounds(5, 2) == [(0, 3), (3, 5)]
    assert shard_bounds(3, 4) == [(0, 1), (1, 2), (2, 3), (3, 3)]
    b = shard_bounds(1000, 7)
    assert b[0][0] == 0 and b[-1][1] == 1000 and all(b[i][1] == b[i + 1][0] for i in range(6))
    assert max(e - s for s, e in b) - min(e - s for s, e in b) <= 1


class _OracleHead(object):
    """Stand-in with the Os2dHead calling convention, computing with the CPU oracle (test only)."""

    def __init__(self, q_hat, state, inverse):
        self.q, self.state, self.inverse = q_hat, state, inverse
        self.class_batch_size = q_hat.size(0)

    def __call__(self, fm, out=None, stage_events=None, route_pairs=None):
        from oracle import head_oracle as O
        self.route_pairs_seen = route_pairs
        with torch.no_grad():
            loc, cls, _, corners = O.head_forward(fm, self.q, self.state, self.inverse)
        if out is not None:
            out[0].copy_(loc), out[1].copy_(cls), out[2].copy_(corners)
            return out[0], out[1], out[1], out[2]
        return loc, cls, cls, corners


def _worker(rank, world, port, n_classes, gather, result_queue):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.set_num_threads(2)
        from oracle import head_oracle as O
        from os2d_amd.parallel import ClassShardedHead
        from os2d_amd.utils import synthetic
        P, inverse = 6, True
        state = synthetic.make_transform_net_state(P, seed=9)
        fm = synthetic.make_feature_map(16, 7, 9, seed=2, A=2)
        class_fms = synthetic.make_class_feature_maps(n_classes, 16, sizes=[(15, 15), (13, 16)], seed=40)
        q = O.prepare_class_maps(class_fms)
        s, e = shard_bounds(n_classes, world)[rank]
        sharded = ClassShardedHead(None, gather=gather, num_classes=n_classes, local_head=_OracleHead(q[s:e], state, inverse))
        loc, cls, cls_det, corners = sharded(fm)
        # asynchronous variant: two gathers in flight, waited for afterwards, same results
        h1, h2 = sharded(fm, async_gather=True), sharded(fm, async_gather=True)
        r2, r1 = h2(), h1()
        assert torch.equal(r1[1], cls) and torch.equal(r2[1], cls)
        with torch.no_grad():
            ref = O.head_forward(fm, q, state, inverse)
        ok = torch.equal(cls, ref[1]) and cls_det is cls
        if gather == "all":
            ok = ok and torch.equal(loc, ref[0]) and torch.equal(corners, ref[3])
        else:
            ok = ok and loc is None and corners is None
        result_queue.put((rank, bool(ok), tuple(cls.shape)))
    finally:
        dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("n_classes,gather", [(4, "all"), (5, "all"), (5, "scores")])
def test_class_sharded_head_world2(n_classes, gather):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_classes, gather, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok, shape in sorted(results):
        assert ok, "rank {} assembled a wrong result".format(rank)
        assert shape == (2, n_classes, 1, 7, 9)


def _det_worker(rank, world, port, result_queue):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from os2d_amd.parallel import all_gather_detections
        from os2d_amd.structures.bounding_box import BoxList
        from os2d_amd.structures.feature_map import FeatureMapSize
        size = FeatureMapSize(w=640, h=480)

        def make(rank_):
            g = torch.Generator().manual_seed(100 + rank_)
            n = [5, 0, 3][rank_]                      # rank 1 has found nothing
            det = BoxList(torch.rand(n, 4, generator=g) * 100, size)
            det.add_field("scores", torch.rand(n, generator=g))
            # rank 0 holds labels 0,1 ; rank 2 holds a huge label id (exact transport) and label 7
            labels = [torch.tensor([0, 0, 1, 1, 1]), torch.zeros(0, dtype=torch.long), torch.tensor([7, 7, (1 << 33) + 5])][rank_]
            det.add_field("labels", labels)
            det.add_field("default_boxes", BoxList(torch.rand(n, 4, generator=g) * 50, size))
            det.add_field("transform_corners", torch.rand(n, 8, generator=g))
            return det
        mine = make(rank)
        got = all_gather_detections(mine)
        parts = [make(r) for r in range(world)]
        ok = len(got) == 8 and got.image_size == size
        ok = ok and got.get_field("labels").tolist() == [0, 0, 1, 1, 1, 7, 7, (1 << 33) + 5]
        ok = ok and torch.equal(got.bbox_xyxy, torch.cat([p.bbox_xyxy for p in parts]))
        ok = ok and torch.equal(got.get_field("scores"), torch.cat([p.get_field("scores") for p in parts]))
        ok = ok and torch.equal(got.get_field("default_boxes").bbox_xyxy, torch.cat([p.get_field("default_boxes").bbox_xyxy for p in parts]))
        ok = ok and torch.equal(got.get_field("transform_corners"), torch.cat([p.get_field("transform_corners") for p in parts]))
        result_queue.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


def test_all_gather_detections_world3():
    """Class-sharded decode: variable-length per-rank detections (incl. an empty rank) are unioned on every rank."""
    world = 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_det_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok in results)


class _TableHead(object):
    """Stand-in head whose outputs are a closed-form function of (global class id, image, channel, location): lets the
    distributed logic be checked at BASELINE.json's class counts without computing anything."""

    def __init__(self, first_class, n_local):
        self.first, self.class_batch_size = first_class, n_local
        self.route_pairs_seen = None

    @staticmethod
    def expected(class_ids, A, k, H, W):
        c = torch.tensor(list(class_ids), dtype=torch.float32).view(1, -1, 1, 1, 1)
        a = torch.arange(A, dtype=torch.float32).view(A, 1, 1, 1, 1)
        ch = torch.arange(k, dtype=torch.float32).view(1, 1, k, 1, 1)
        hw = torch.arange(H * W, dtype=torch.float32).view(1, 1, 1, H, W)
        return c * 1000.0 + a * 100.0 + ch * 10.0 + hw / 64.0

    def __call__(self, fm, out=None, stage_events=None, route_pairs=None):
        self.route_pairs_seen = route_pairs
        A, _, H, W = fm.shape
        ids = range(self.first, self.first + self.class_batch_size)
        for t, k in zip(out, (4, 1, 8)):
            assert tuple(t.shape) == (A, self.class_batch_size, k, H, W) and t.is_contiguous()
            t.copy_(self.expected(ids, A, k, H, W))
        return out[0], out[1], out[1], out[2]


def _table_worker(rank, world, port, n_classes, A, reuse, result_queue):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.set_num_threads(1)
        from os2d_amd.modeling.head import FFT_MIN_PAIRS
        from os2d_amd.parallel import ClassShardedHead
        H, W = 3, 4
        s, e = shard_bounds(n_classes, world)[rank]
        head = _TableHead(s, e - s)
        sharded = ClassShardedHead(None, gather="all", num_classes=n_classes, local_head=head, reuse_buffers=reuse)
        fm = torch.zeros(A, 8, H, W)
        ok = True
        held = []
        for it in range(3):
            loc, cls, _, corners = sharded(fm)
            held.append(cls)
            for t, k in ((loc, 4), (cls, 1), (corners, 8)):
                ok = ok and torch.equal(t, _TableHead.expected(range(n_classes), A, k, H, W))
        pending = [sharded(fm, async_gather=True) for _ in range(2)]
        for h in reversed(pending):
            r = h()
            ok = ok and torch.equal(r[0], _TableHead.expected(range(n_classes), A, 4, H, W))
        equal = len({b - a for a, b in shard_bounds(n_classes, world)}) == 1
        # one image per call and equal class counts: the collectives write the final layout, nothing is copied afterwards
        copies_ok = sharded.copies_last_call == (0 if (A == 1 and equal) else 3)
        # every rank - also a tail rank below the frequency-domain threshold - decides its arithmetic route on the GLOBAL count
        route_ok = head.route_pairs_seen == A * n_classes
        tail_below = (e - s) * A < FFT_MIN_PAIRS
        if reuse:       # ring of `reuse` buffer sets: the result of the call `reuse` calls ago has been overwritten in place
            ok = ok and held[0].data_ptr() == held[reuse].data_ptr() if len(held) > reuse else ok
        result_queue.put((rank, bool(ok), bool(copies_ok), bool(route_ok), bool(tail_below), e - s))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n_classes,A,reuse", [(8, 1024, 1, 0), (8, 1024, 1, 2), (3, 16, 1, 0), (3, 16, 2, 0), (2, 9, 1, 0)])
def test_class_sharded_head_layout_route_and_buffers(world, n_classes, A, reuse):
    """BASELINE.json configs[2] over 8 ranks (1024 classes -> 128 per rank: the gather lands in the final
    layout, zero copies afterwards, also with a ring of reused buffers), ragged splits over 3 and 2 ranks (16 -> 6 + 5 + 5,
    9 -> 5 + 4: trimmed correctly; the tail rank holds fewer classes than FFT_MIN_PAIRS and still routes on the global count)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_table_worker, args=(r, world, port, n_classes, A, reuse, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted(q.get(timeout=240) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [r[5] for r in results] == [e - s for s, e in shard_bounds(n_classes, world)]
    for rank, ok, copies_ok, route_ok, _, _ in results:
        assert ok, "rank {} assembled a wrong result".format(rank)
        assert copies_ok, "rank {}: unexpected number of post-gather copies".format(rank)
        assert route_ok, "rank {} decided its route on a local pair count".format(rank)
    if (world, n_classes, A) in ((3, 16, 1), (2, 9, 1)):
        assert any(r[4] for r in results), "the case is meant to have a tail rank below FFT_MIN_PAIRS"
