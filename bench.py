#!/usr/bin/env python3
"""Benchmark of the MI355X OS2D head: query-image-pairs/s (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--classes B_per_gpu] [--precision f16x3|f16x2|f32] [--pyramid]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = the whole head (correlation -> TransformNet -> resample/pool -> box encode) for ONE 1280x960 image
feature map [1,1024,60,80] against B classes per GPU (default 64 = BASELINE.json configs[1]); with N GPUs the classes
are sharded (weak scaling: N*B classes in total) and every step ends with the RCCL all-gather of the per-class SCORE
maps (north_star: "all-gather ... of per-class score maps before NMS"), as ``os2d_amd.parallel.ClassShardedHead``
does it.  Inputs are synthetic (post-ReLU Gaussian features, perturbed TransformNet) and resident
in HBM before the timed region.

Arithmetic (``--precision``):
  f16x3 (default)  every fp32 operand of the four GEMM-shaped stages is split into fp16 hi + lo and each product is
                   evaluated with three v_mfma_f32_32x32x16_f16 (fp32 accumulation): outputs agree with the reference
                   to the same 2.4e-7 as the fp32 mode (tests/test_head_gpu.py runs every parity case in all modes);
  f16x2            as f16x3, except that the dominant 7x7 layer takes its WEIGHTS as fp16 roundings only (two MFMAs per
                   product, activations still split): scores within 1e-6 and box regression within 5e-5 of the fp32
                   result - inside the 1e-4 parity bound of BASELINE.json, but no longer fp32-equivalent;
  f32              v_mfma_f32_32x32x2_f32, exact fp32.
The primary line is measured in the selected mode; the other modes are timed right after and reported under
"other_precisions" so all are always on record.

Prints ONE JSON line on rank 0 with the driver's contract fields plus
  roofline     - the dominant kernel (conv 7x7 225->128 MFMA implicit GEMM): ALGORITHMIC FLOPs per launch divided by its
                 mean launch duration, measured with HIP events recorded on the launch stream inside the timed steps,
                 against the dense MFMA peak of the instruction it runs on; `traffic` / `hbm_gbps` / `mfma_pipe_busy` are the
                 HBM bytes, HBM rate and matrix-pipe utilisation of that kernel from the committed rocprofv3 PMC passes
  stages_ms    - mean duration of every stage of the step (same events)
  end_to_end   - secondary: backbone + head + decode/NMS per image, and the one-off class-head construction (N=1 only)
  cpu_baseline - the oracle (torch-CPU restatement of the reference head, driven one class at a time like the
                 reference's evaluation) timed on the host cores, rank 0 / N=1 only, on a bounded class sample.
"""
import argparse
import ctypes
import json
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

C_FEAT, H_FM, W_FM = 1024, 60, 80          # ResNet50-C4 features of a 1280x960 input
FLOP_PER_LOC = {"corr": 2 * 225 * 1024, "conv1": 2 * 128 * 225 * 49, "conv2": 2 * 64 * 128 * 25}
PEAK = {"f32": 157.3e12, "f16x3": 2.5e15, "f16x2": 2.5e15}  # dense MFMA peaks (MI355X_MICROARCH.md): fp32-input MFMA; fp16/bf16 MFMA
STAGES = ("corr", "conv1", "conv2", "conv3", "sample")


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--classes", type=int, default=64, help="classes per GPU")
    ap.add_argument("--variant", default="v2", choices=["v2", "v1"], help="v2: affine+inverse (P=6); v1: simplified (P=4)")
    ap.add_argument("--precision", default=os.environ.get("OS2D_PRECISION", "f16x3"), choices=["f32", "f16x3", "f16x2"])
    ap.add_argument("--pyramid", action="store_true",
                    help="BASELINE configs[4]: 7-scale pyramid (0.5-1.6) of the 1280x960 image, one HIP stream per level; "
                         "a pair then means one (image, class) over all 7 levels")
    ap.add_argument("--gather", default="scores", choices=["scores", "all"], help="what the N>1 all-gather moves")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise torch.distributed (RCCL) and use the class-sharded path even with one rank "
                         "(smoke test of the N>1 code path on a single-GPU box)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-precision", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-end-to-end", action="store_true",
                    help="skip the secondary end-to-end leg (backbone + class-head build + head + decode/NMS)")
    return ap.parse_args()


def cpu_baseline(fm_cpu, class_fms_cpu, state, inverse, budget_s):
    """Time the oracle the way the reference evaluates (one head per class, looped) on the host cores.  torch's intra-op
    scaling of this op mix peaks far below the 256 hardware threads of the box (tools/cpu_threads_sweep.py: 7 / 22 / 29 /
    21 / 10 / 4 pairs/s at 1 / 8 / 16 / 32 / 64 / 128 threads), so a few thread counts share the time budget and the
    best one is reported."""
    from oracle import head_oracle as O
    ncores = os.cpu_count() or 1
    forced = int(os.environ.get("OS2D_CPU_THREADS", "0"))
    candidates = [forced] if forced else sorted(set(max(1, min(ncores, n)) for n in (8, 16, 32)))
    q = O.prepare_class_maps(class_fms_cpu)
    results = []
    with torch.no_grad():
        for threads in candidates:
            torch.set_num_threads(threads)
            O.head_forward(fm_cpu, q[:1], state, inverse)        # warm-up
            done, t0 = 0, time.perf_counter()
            while (time.perf_counter() - t0) < budget_s / len(candidates):   # cycles over the class sample
                b = done % q.size(0)
                O.head_forward(fm_cpu, q[b:b + 1], state, inverse)
                done += 1
            dt = time.perf_counter() - t0
            results.append((done / dt, threads, done, dt))
    best = max(results)
    return {"value": round(best[0], 3), "unit": "query-image-pairs/s", "cores": best[1], "kind": "port",
            "sample": "{} class calls looped one at a time (reference evaluate.py:323-331 call pattern) on one "
                      "60x80x1024 feature map, {:.1f} s, torch CPU fp32, best of {} threads ({}) on {} hw threads"
                      .format(best[2], best[3], "/".join(str(r[1]) for r in results),
                              ", ".join("{}: {:.1f} pairs/s".format(r[1], r[0]) for r in results), ncores)}


def measured_mfma_busy(precision):
    """Matrix-pipe utilisation of the conv 7x7 kernel from the committed rocprofv3 SQ pass
    (SQ_VALU_MFMA_BUSY_CYCLES x 32 / (1024 SIMDs x GRBM_GUI_ACTIVE)); None if not recorded."""
    path = os.path.join(REPO, "profiles", "conv1_traffic_{}.json".format(precision))
    if not os.path.exists(path):
        return None
    with open(path) as f:
        return json.load(f).get("mfma_pipe_busy")


def measured_traffic(B, precision):
    """HBM bytes per conv1 launch from the committed rocprofv3 PMC passes (profiles/conv1_traffic_<precision>.json,
    written by tools/summarize_prof.py --traffic: FETCH_SIZE doubled per the gfx950 correction + WRITE_SIZE, separate
    passes), scaled to the class count of this run; None if no profile has been recorded."""
    path = os.path.join(REPO, "profiles", "conv1_traffic_{}.json".format(precision))
    if not os.path.exists(path):
        return None
    with open(path) as f:
        t = json.load(f)
    return int(t["bytes_per_class"] * B)


def end_to_end(dev, B, P, inverse, state, precision, steps=5, warmup=2):
    """Secondary number: one 1280x960 image through the PyTorch-ROCm ResNet50-C4 backbone, the
    HIP head against B classes and the HIP decode + per-class NMS of all 4800 boxes per class (score threshold -inf,
    the reference's eval default), random-init weights; plus the one-off construction of the class head from B
    240x240 class images (one batched backbone pass)."""
    from os2d_amd.engine.evaluate import build_class_head
    from os2d_amd.modeling.model import Os2dModel
    from os2d_amd.structures.feature_map import FeatureMapSize
    torch.manual_seed(0)
    net = Os2dModel(is_cuda=False, merge_branch_parameters=True, backbone_arch="resnet50",
                    use_inverse_geom_model=inverse, simplify_affine=(P == 4))
    net.os2d_head_creator.aligner.parameter_regressor.load_state_dict(state)
    net.to(dev).eval()
    g = torch.Generator().manual_seed(0)
    img = torch.randn(1, 3, 960, 1280, generator=g).to(dev)
    class_images = [torch.randn(3, 240, 240, generator=g).to(dev) for _ in range(B)]
    coder = net.build_box_coder()
    img_size = FeatureMapSize(img=img)
    ids = list(range(B))
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    with torch.no_grad():
        build_class_head(net, class_images)                      # warm-up (MIOpen picks its kernels here)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        head = build_class_head(net, class_images)
        torch.cuda.synchronize(dev)
        t_build = time.perf_counter() - t0
        head.precision = precision
        phases = [0.0, 0.0, 0.0]
        n_det = 0
        for i in range(warmup + steps):
            if i == warmup:
                torch.cuda.synchronize(dev)
                t0 = time.perf_counter()
            ev[0].record()
            fm = net.net_feature_maps(img)
            ev[1].record()
            loc, cls, _, _ = head(fm)
            ev[2].record()
            dets = coder.decode_pyramid([loc[0].flatten(2)], [cls[0].flatten(1)], [img_size], ids,
                                        nms_score_threshold=float("-inf"))
            ev[3].record()
            if i >= warmup:
                torch.cuda.synchronize(dev)
                for k in range(3):
                    phases[k] += ev[k].elapsed_time(ev[k + 1])
                n_det = len(dets)
        torch.cuda.synchronize(dev)
        dt = time.perf_counter() - t0
    return {"value": round(B * steps / dt, 2), "unit": "query-image-pairs/s", "ms_per_image": round(dt / steps * 1e3, 3),
            "backbone_ms": round(phases[0] / steps, 3), "head_ms": round(phases[1] / steps, 3),
            "decode_nms_ms": round(phases[2] / steps, 3), "detections_per_image": n_det,
            "class_head_build_ms": round(t_build * 1e3, 2), "precision": precision, "steps": steps,
            "what": "ResNet50-C4 (PyTorch-ROCm/MIOpen fp32, random init) on 1x3x960x1280 + HIP head x {} classes + HIP "
                    "decode and per-class NMS of all 4800 boxes per class; class_head_build_ms = {} class images 240x240 "
                    "-> backbone (one batch) -> 15x15 class maps (once per class set)".format(B, B)}


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus {} needs a torchrun launch with one process per GPU".format(args.gpus))
        args.gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (the OS2D head has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    import torch.distributed as dist
    use_dist = world > 1 or args.force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from os2d_amd import _lib
    from os2d_amd.modeling.head import build_os2d_head_creator
    from os2d_amd.parallel import ClassShardedHead
    from os2d_amd.structures.feature_map import FeatureMapSize
    from os2d_amd.utils import synthetic
    lib = _lib.load()

    P, inverse = (6, True) if args.variant == "v2" else (4, False)
    B = args.classes
    state = synthetic.make_transform_net_state(P, seed=1)
    fm_cpu = synthetic.make_feature_map(C_FEAT, H_FM, W_FM, seed=0)
    # this rank's classes: global class ids rank*B .. rank*B+B-1 (seeds 1000+id)
    class_fms_cpu = synthetic.make_class_feature_maps(B, C_FEAT, sizes=[(15, 15)], seed=1000 + rank * B)
    creator = build_os2d_head_creator(P == 4, False, inverse, FeatureMapSize(w=16, h=16), FeatureMapSize(w=16, h=16))
    creator.aligner.parameter_regressor.load_state_dict(state)
    creator.to(dev).eval()
    fm = fm_cpu.to(dev)
    with torch.no_grad():
        head = creator.create_os2d_head([c.to(dev) for c in class_fms_cpu])
    sharded = ClassShardedHead(creator, group=None, gather=args.gather, num_classes=B * world, local_head=head) if use_dist else None

    def new_event_set():
        arr = (ctypes.c_void_p * 10)()
        for i in range(10):
            ev = ctypes.c_void_p()
            _lib.check(lib.os2d_prof_event_create(ctypes.byref(ev)), "os2d_prof_event_create")
            arr[i] = ev.value
        return arr

    runner, level_fms = None, None
    if args.pyramid:
        from os2d_amd.engine.pyramid import PyramidHeadRunner
        level_hw = [(30, 40), (38, 50), (48, 64), (60, 80), (72, 96), (84, 112), (96, 128)]
        level_fms = [synthetic.make_feature_map(C_FEAT, h, w, seed=100 + i).to(dev) for i, (h, w) in enumerate(level_hw)]
        runner = PyramidHeadRunner(sharded if sharded is not None else head, device=dev)

    def sync_all():
        torch.cuda.synchronize(dev)
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize(dev)

    def run_mode(precision, steps, warmup):
        """W warm-up + exactly K timed steps in one arithmetic mode; returns (seconds, per-stage mean ms or None)."""
        head.precision = precision
        # one set of 10 stage events per timed step, so nothing has to be read back inside the timed region
        event_sets = [new_event_set() for _ in range(steps)] if (sharded is None and runner is None) else []

        pending = []

        def step(events):
            with torch.no_grad():
                if runner is not None:
                    return runner.run(level_fms, inputs_are_features=True)
                if sharded is not None:
                    # the all-gather of this step runs asynchronously (RCCL stream) and is waited for only after the
                    # NEXT step's kernels have been queued, so the xGMI transfer hides behind compute
                    pending.append(sharded(fm, async_gather=True))
                    if len(pending) > 1:
                        pending.pop(0)()
                    return None
                return head(fm, stage_events=events)

        for _ in range(warmup):
            step(None)
        while pending:
            pending.pop(0)()
        sync_all()
        # ---- timed region: exactly K steps; stage events are recorded on the launch stream inside these steps
        t0 = time.perf_counter()
        for i in range(steps):
            step(event_sets[i] if event_sets else None)
        while pending:
            pending.pop(0)()          # every gather of the timed steps completes inside the timed region
        sync_all()
        dt = time.perf_counter() - t0
        if use_dist:
            t = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        stage_ms = None
        if event_sets:
            ms = ctypes.c_float()
            rows = []
            for evs in event_sets:
                row = []
                for st in range(5):
                    _lib.check(lib.os2d_prof_event_elapsed_ms(evs[2 * st], evs[2 * st + 1], ctypes.byref(ms)), "elapsed")
                    row.append(ms.value)
                rows.append(row)
                for ev in evs:
                    lib.os2d_prof_event_destroy(ev)
            stage_ms = [sum(r[st] for r in rows) / len(rows) for st in range(5)]
        return dt, stage_ms

    def roofline(precision, stage_ms):
        flops = FLOP_PER_LOC["conv1"] * H_FM * W_FM * B            # algorithmic FLOPs of ONE conv1 launch
        achieved = flops / (stage_ms[1] * 1e-3)
        peak = PEAK[precision]
        r = {"kernel": "TransformNet conv 7x7 225->128 ({})".format(
                 "conv_mfma_kernel<7,...>, v_mfma_f32_32x32x2_f32" if precision == "f32"
                 else "conv_f16x3_kernel<7,...>, v_mfma_f32_32x32x16_f16 x{} per product".format(precision[-1])),
             "bound": "mfma", "achieved": round(achieved / 1e12, 3), "peak": peak / 1e12, "unit": "TFLOP/s",
             "frac": round(achieved / peak, 4), "traffic": measured_traffic(B, precision),
             "flops_per_launch": flops, "avg_launch_ms": round(stage_ms[1], 4)}
        # algorithmic HBM bytes of the same launch: every input plane read once (226 fp32 planes, or 29 groups x 8
        # channels x (hi|lo) halves) + the 128 output planes written once + the packed weights once
        plane = int(lib.os2d_plane_floats(H_FM, W_FM))
        in_planes = 226 if precision == "f32" else 232
        r["mfma_pipe_busy"] = measured_mfma_busy(precision)
        if r["traffic"]:
            r["hbm_gbps"] = round(r["traffic"] / (stage_ms[1] * 1e-3) / 1e9, 1)      # 8000 GB/s peak: far from HBM-bound
        r["algorithmic_bytes"] = int(B * (in_planes + 128) * plane * 4 + lib.os2d_packed_conv_bytes(1, {"f32": 0, "f16x3": 1, "f16x2": 2}[precision]))
        if precision != "f32":
            # every algorithmic product costs three (f16x2: two) half-precision MFMA products: the ceiling for
            # algorithmic FLOP/s on this instruction is peak/3 (peak/2); the executed rate also includes the tile /
            # channel-group padding (x1.118)
            terms = int(precision[-1])
            r["algorithmic_ceiling"] = round(peak / terms / 1e12, 1)
            r["frac_of_algorithmic_ceiling"] = round(achieved / (peak / terms), 4)
            r["executed_mfma_tflops"] = round(terms * achieved * 1.118 / 1e12, 1)
            r["executed_frac_of_peak"] = round(terms * achieved * 1.118 / peak, 4)
        return r

    whole = (FLOP_PER_LOC["corr"] + FLOP_PER_LOC["conv1"] + FLOP_PER_LOC["conv2"] + 2 * P * 64 * 25) * H_FM * W_FM
    pairs_per_step = B * world
    dt, stage_ms = run_mode(args.precision, args.steps, args.warmup)
    value = pairs_per_step * args.steps / dt
    dtype = {"f32": "f32", "f16x3": "f16x3 (fp32 operands split into fp16 hi+lo, 3 half MFMAs per product, fp32 accumulate)",
             "f16x2": "f16x2 (as f16x3; the 7x7 layer's weights enter as fp16 roundings only, 2 half MFMAs per product)"}
    result = {
        "metric": "query-image-pairs/s (1280-px input, ResNet50, N-class)",
        "value": round(value, 2),
        "unit": "query-image-pairs/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": dtype[args.precision], "data": "synthetic",
        "config": {"workload": "OS2D head, ResNet50-C4 features of one 1280x960 image ({}), {} classes per GPU "
                               "({} total), {}, {} (P={}, inverse={}), head only, features resident in HBM"
                               .format("7-level pyramid 30x40..96x128, 39580 locations" if args.pyramid else "1x1024x60x80",
                                       B, B * world, "7 scales 0.5-1.6, one HIP stream per level" if args.pyramid else "single scale",
                                       args.variant.upper(), P, int(inverse)),
                   "classes_per_gpu": B, "classes_total": B * world, "feature_map": [C_FEAT, H_FM, W_FM],
                   "precision": args.precision,
                   "parallelism": "class-sharded x{} + all-gather of {}".format(world, "score maps" if args.gather == "scores" else "loc|cls|corners")
                                  if world > 1 else "single GPU"},
    }
    if stage_ms:
        result["stages_ms"] = {k: round(v, 4) for k, v in zip(STAGES, stage_ms)}
        result["roofline"] = roofline(args.precision, stage_ms)
    if not args.pyramid:
        result["head_tflops_algorithmic"] = round(whole * value / 1e12, 3)
    if not args.no_other_precision:
        result["other_precisions"] = []
        for other in ("f16x3", "f16x2", "f32"):
            if other == args.precision:
                continue
            dt2, stage2 = run_mode(other, args.steps, 1)
            o = {"precision": other, "value": round(pairs_per_step * args.steps / dt2, 2), "ms_per_step": round(dt2 / args.steps * 1e3, 4)}
            if stage2:
                o["stages_ms"] = {k: round(v, 4) for k, v in zip(STAGES, stage2)}
                o["roofline"] = roofline(other, stage2)
            result["other_precisions"].append(o)
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not args.pyramid:
        result["cpu_baseline"] = cpu_baseline(fm_cpu, class_fms_cpu, state, inverse, args.cpu_seconds)
        result["speedup_vs_cpu_baseline"] = round(result["value"] / result["cpu_baseline"]["value"], 1)
    if rank == 0 and world == 1 and not args.no_end_to_end and not args.pyramid:
        result["end_to_end"] = end_to_end(dev, B, P, inverse, state, args.precision)
    if rank == 0:
        print(json.dumps(result), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
