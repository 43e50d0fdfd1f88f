"""torchrun worker of tests/test_parallel_gpu.py (not collected by pytest): one process per GPU over RCCL
(``torch.distributed`` backend "nccl"), world size from the launcher.  Every rank checks the class-sharded path
(os2d_amd/parallel.py) around the REAL HIP head against the unsharded head on the same device, bit for bit:
gather = all / scores, synchronous and asynchronous, alone and inside the per-level-stream pyramid runner, and the
class-sharded decode + detection gather.  Prints "DIST_WORKER_OK rank=<r>" on success."""
import os
import sys

import torch
import torch.distributed as dist

REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    try:
        import util
        from os2d_amd.engine.pyramid import PyramidHeadRunner
        from os2d_amd.modeling.box_coder import Os2dBoxCoder
        from os2d_amd.parallel import ClassShardedHead, all_gather_detections, shard_bounds
        from os2d_amd.structures.feature_map import FeatureMapSize
        from os2d_amd.utils import synthetic
        P, inverse, C = 6, True, 64
        n_classes = 11                                  # ragged over 2 ranks (6 + 5), 3 ranks (4 + 4 + 3), ...
        state = synthetic.make_transform_net_state(P, seed=9)
        creator = util.make_head_creator(P, inverse, state, dev)
        class_fms = [c.to(dev) for c in synthetic.make_class_feature_maps(n_classes, C, sizes=[(15, 15), (13, 16)], seed=40)]
        levels = [synthetic.make_feature_map(C, h, w, seed=70 + i, A=2).to(dev) for i, (h, w) in enumerate([(9, 12), (20, 27), (14, 18)])]
        fm = levels[1]
        with torch.no_grad():
            full_head = creator.create_os2d_head(class_fms)
            full = full_head(fm)
            for gather in ("all", "scores"):
                sharded = ClassShardedHead(creator, class_fms, gather=gather, reuse_buffers=4 if gather == "all" else 0)
                assert sharded.counts == [e - s for s, e in shard_bounds(n_classes, world)]
                loc, cls, cls_det, corners = sharded(fm)
                assert cls_det is cls and torch.equal(cls, full[1]), "scores differ ({})".format(gather)
                if gather == "all":
                    assert torch.equal(loc, full[0]) and torch.equal(corners, full[3])
                else:
                    assert loc is None and corners is None
                # asynchronous gathers: several in flight, waited for out of order
                h1, h2, h3 = (sharded(fm, async_gather=True) for _ in range(3))
                for h in (h3, h1, h2):
                    r = h()
                    assert torch.equal(r[1], full[1])
                    if gather == "all":
                        assert torch.equal(r[0], full[0]) and torch.equal(r[3], full[3])
                # inside the pyramid runner (one HIP stream per level; with gather="scores" loc / corners are None)
                serial = PyramidHeadRunner(full_head, num_streams=1, device=dev).run(levels, inputs_are_features=True)
                par = PyramidHeadRunner(sharded, num_streams=len(levels), device=dev).run(levels, inputs_are_features=True)
                torch.cuda.synchronize(dev)
                for lvl in range(len(levels)):
                    assert torch.equal(par[1][lvl], serial[1][lvl]), "pyramid scores, level {}".format(lvl)
                    if gather == "all":
                        assert torch.equal(par[0][lvl], serial[0][lvl]) and torch.equal(par[2][lvl], serial[2][lvl])
                    else:
                        assert par[0][lvl] is None and par[2][lvl] is None
            # class-sharded decode: NMS on the rank's own classes, then the union of the surviving detections
            coder = Os2dBoxCoder(output_box_grid_generator=creator.box_grid_generator_image_level)
            H, W = fm.shape[2:]
            img_size = FeatureMapSize(w=16 * W, h=16 * H)
            ids = list(range(n_classes))
            s, e = shard_bounds(n_classes, world)[rank]
            local = creator.create_os2d_head(class_fms[s:e])(fm, route_pairs=fm.size(0) * n_classes)   # same arithmetic route as the full head
            mine = coder.decode_pyramid([local[0][0].flatten(2)], [local[1][0].flatten(1)], [img_size], ids[s:e],
                                        nms_score_threshold=0.3, transform_corners_pyramid=[local[3][0].flatten(2)])
            union = all_gather_detections(mine)
            ref = coder.decode_pyramid([full[0][0].flatten(2)], [full[1][0].flatten(1)], [img_size], ids,
                                       nms_score_threshold=0.3, transform_corners_pyramid=[full[3][0].flatten(2)])
            assert len(union) == len(ref) and len(ref) > 0
            assert torch.equal(union.get_field("labels"), ref.get_field("labels"))
            assert torch.equal(union.bbox_xyxy, ref.bbox_xyxy) and torch.equal(union.get_field("scores"), ref.get_field("scores"))
            assert torch.equal(union.get_field("transform_corners"), ref.get_field("transform_corners"))
        dist.barrier()
        print("DIST_WORKER_OK rank={} world={}".format(rank, world), flush=True)
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
