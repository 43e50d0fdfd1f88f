#!/usr/bin/env python3
"""Benchmark of the MI355X OS2D head: query-image-pairs/s (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--classes B_per_gpu | --classes-total B] [--variant v2|v1]
                    [--precision fftx3|fft|f16x3|f16x2|f32] [--pyramid] [--gather all|scores|detections]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = the whole head (correlation -> TransformNet -> resample/pool -> box encode) for ONE 1280x960 image
feature map [1,1024,60,80] against the classes of the run.  Inputs are synthetic (post-ReLU Gaussian features, perturbed
TransformNet) and resident in HBM before the timed region.

Workloads
  N = 1 (default)  BASELINE.json configs[1]: 64 classes, V2 head.  The same run also times, with the driver's clock
                   running, the other single-GPU readings of BASELINE.json's configs ("sweep" in bench_details.json, the
                   headline figures inside `config` of the line): 256 classes V1 head (configs[3]), all 1024 classes on one
                   GPU (configs[2], N = 1) and the 7-level pyramid at 128 classes (the per-GPU share of configs[4]: levels
                   back to back - the runner's default - and on one HIP stream per level); each entry carries its own
                   roofline object.
  N > 1 (default)  BASELINE.json configs[2]: STRONG scaling of 1024 classes, block-sharded over the N ranks (128 per GPU
                   at N = 8); every step ends with the RCCL all-gather of the per-class output maps so that every rank
                   holds a result it can decode (--gather all: loc | cls | corners, 250 KB per class; issued
                   asynchronously and waited for after the next step's kernels are queued, every gather completing inside
                   the timed region).  The lighter exchanges are timed right after and reported under "other_gathers":
                   "scores" (north_star: "all-gather ... of per-class score maps before NMS", 19 KB per class) and
                   "detections" (every rank decodes + NMS-es its own classes, only surviving boxes cross xGMI).
                   --classes B keeps B classes per GPU instead (weak scaling).

Arithmetic (``--precision``):
  fftx3 (default)  as fft, with the per-bin complex GEMM on v_mfma_f32_32x32x16_f16: spectra split into fp16 hi + lo (three
                   MFMAs per product, fp32 accumulation; the weight spectra pre-split on the host with per-row scales, the
                   input spectra - bounded by H*W - split on the fly): the launch is bound by the HBM stream of its operands,
                   so the roofline object of this mode is an HBM one (algorithmic bytes / live launch time against 8 TB/s);
  fft              as f16x3, except that the dominant 7x7 layer (225 -> 128 channels) runs in the frequency domain in fp32:
                   in-LDS real FFT of the normalised correlation maps, one complex GEMM per frequency bin on
                   v_mfma_f32_32x32x2_f32, inverse FFT with the bias / ReLU / fp16-split epilogue fused.  16.7x fewer
                   multiply-adds than the direct layer, fp32 arithmetic throughout (closer to an fp64 evaluation than the
                   direct fp32 kernel); class batches below 7 pairs and maps that do not fit the in-LDS transform take
                   the f16x3 kernel;
  f16x3            every fp32 operand of the four GEMM-shaped stages is split into fp16 hi + lo and each product is
                   evaluated with three v_mfma_f32_32x32x16_f16 (fp32 accumulation); per-channel power-of-two scales
                   derived from rigorous bounds make fp16 overflow impossible for finite inputs (a sticky status flag
                   reports anything else).  Outputs agree with the reference to the same 2.4e-7 as the fp32 mode
                   (tests/test_head_gpu.py runs every parity case, incl. the hostile-range networks, in all modes);
                   "max_abs_diff_vs_f32" in the JSON line is measured on the benchmarked tensors in this very run;
  f16x2            as f16x3, except that the dominant 7x7 layer takes its WEIGHTS as fp16 roundings only (two MFMAs per
                   product): scores within 1e-6 and box regression within 5e-5 of fp32 - inside the 1e-4 parity bound of
                   BASELINE.json, but no longer fp32-equivalent;
  fft32            strictly fp32 and fast: fp32-MFMA correlation and 5x5 layers (as f32), the 7x7 layer in the frequency domain
                   in fp32 (as fft) - no fp16 value anywhere;
  f32              v_mfma_f32_32x32x2_f32, exact fp32 (the reference's direct arithmetic).
The primary line is measured in the selected mode; the other modes are timed right after and reported under
"other_precisions" so all are always on record.

Output (rank 0).  stdout carries ONE compact JSON line (<= 6000 bytes, strict JSON; round 4's 22 KB line could not be parsed by
the driver): the driver's contract fields, `config` (incl. the same-run figures of the strict-fp32 modes, 256 classes V1, 1024 classes
on one GPU with the longest kernel's live traffic, and the 7-level pyramid), and
  roofline       - the LONGEST kernel of the step by live HIP-event time: ALGORITHMIC FLOPs (or bytes) per launch / its mean launch
                   duration, measured with HIP events recorded on the launch stream inside the timed steps, against the dense MFMA
                   peak of the instruction (or the 8 TB/s HBM peak); `traffic` / `hbm_gbps` / `mfma_pipe_busy` /
                   `effective_clock_ghz` are LIVE at N = 1: the run spawns rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE, SQ / GRBM:
                   one pass each) over a 3-step child run of the same workload and merges them
  roofline_other - {kernel: [frac of its bound, avg launch ms]} of the other kernels of the step
  stages_ms      - mean duration of every stage of the step (same events)
  cpu_baseline   - the oracle (torch-CPU restatement of the reference head, driven one class at a time like the reference's
                   evaluation) timed on the host cores, rank 0 / N=1 only, on a bounded class sample
  end_to_end     - secondary: backbone + head + decode/NMS per image (N=1 only)
The FULL record - every kernel's roofline object, the other arithmetic modes, the sweep entries with their own roofline objects, the
raw live counters at 64 and 1024 classes, both end-to-end legs - goes to bench_details.json (repo root, and gpurun_out/ when present)
and to stderr (one line prefixed "[bench_details] ").
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
LEVEL_HW = [(30, 40), (38, 50), (48, 64), (60, 80), (72, 96), (84, 112), (96, 128)]   # 7 scales 0.5-1.6
FLOP_PER_LOC = {"corr": 2 * 225 * 1024, "conv1": 2 * 128 * 225 * 49, "conv2": 2 * 64 * 128 * 25}
PEAK = {"f32": 157.3e12, "f16x3": 2.5e15, "f16x2": 2.5e15, "fft": 157.3e12, "fftx3": 2.5e15, "fft32": 157.3e12}  # dense MFMA peaks (MI355X_MICROARCH.md): fp32-input MFMA; fp16/bf16 MFMA
STAGES = ("corr", "conv1", "conv2", "conv3", "sample")
FFT_MODES = ("fft", "fftx3", "fft32")
PREC_ID = {"f32": 0, "f16x3": 1, "f16x2": 2, "fft": 3, "fftx3": 4, "fft32": 5}
DTYPE = {"f32": "f32",
         "f16x3": "f16x3 (fp32 operands split into fp16 hi+lo, 3 half MFMAs per product, fp32 accumulate)",
         "f16x2": "f16x2 (as f16x3; the 7x7 layer's weights enter as fp16 roundings only, 2 half MFMAs per product)",
         "fft": "fft (as f16x3; the 7x7 layer in the frequency domain in fp32: real FFT, complex GEMM per bin on the fp32 MFMA, inverse FFT)",
         "fft32": "fft32 (strictly fp32: fp32-MFMA correlation and 5x5 layers as f32, the 7x7 layer in the frequency domain in fp32 as fft; no fp16 value anywhere)",
         "fftx3": "fftx3 (as fft; the per-bin complex GEMM on the fp16 MFMA with spectra split into fp16 hi+lo, 3 MFMAs per product, fp32 accumulate)"}
DISTINCT_CLASS_MAPS = 64     # synthetic class maps are generated for 64 seeds and repeated (separate device copies)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--classes", type=int, default=None, help="classes PER GPU (weak scaling); default 64 at N=1")
    ap.add_argument("--classes-total", type=int, default=None,
                    help="classes in total, block-sharded over the ranks (strong scaling); default 1024 at N>1")
    ap.add_argument("--variant", default="v2", choices=["v2", "v1"], help="v2: affine+inverse (P=6); v1: simplified (P=4)")
    ap.add_argument("--precision", default=os.environ.get("OS2D_PRECISION", "fftx3"), choices=["f32", "f16x3", "f16x2", "fft", "fftx3", "fft32"])
    ap.add_argument("--pyramid", action="store_true",
                    help="BASELINE configs[4]: 7-scale pyramid (0.5-1.6) of the 1280x960 image, one HIP stream per level; "
                         "a pair then means one (image, class) over all 7 levels")
    ap.add_argument("--gather", default="all", choices=["all", "scores", "detections"],
                    help="what the N>1 step exchanges (see the module docstring)")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise torch.distributed (RCCL) and use the class-sharded path even with one rank "
                         "(smoke test of the N>1 code path on a single-GPU box)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-precision", action="store_true")
    ap.add_argument("--no-other-gather", action="store_true")
    ap.add_argument("--no-one-gpu-reference", action="store_true",
                    help="N > 1: skip rank 0's single-GPU timing of the same workload (one_gpu_same_workload / speedup_vs_one_gpu)")
    ap.add_argument("--no-sweep", action="store_true", help="skip the 256-class V1 / 1024-class / pyramid lines (N=1)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-live-counters", action="store_true",
                    help="skip the rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE / SQ) the N=1 run spawns over a short child run")
    ap.add_argument("--no-end-to-end", action="store_true",
                    help="skip the secondary end-to-end leg (backbone + class-head build + head + decode/NMS)")
    return ap.parse_args()


def host_cpu_topology():
    """{"logical_cpus", "physical_cores", "sockets"} of the host from /proc/cpuinfo (state the physical core count, not only
    the hardware threads); None for what cannot be read."""
    logical = os.cpu_count()
    cores, sockets = set(), set()
    try:
        with open("/proc/cpuinfo") as f:
            phys = core = None
            for line in f:
                if line.startswith("physical id"):
                    phys = line.split(":")[1].strip()
                elif line.startswith("core id"):
                    core = line.split(":")[1].strip()
                elif not line.strip():
                    if phys is not None and core is not None:
                        cores.add((phys, core))
                        sockets.add(phys)
                    phys = core = None
            if phys is not None and core is not None:
                cores.add((phys, core))
                sockets.add(phys)
    except OSError:
        pass
    usable = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else logical
    return {"logical_cpus": logical, "physical_cores": len(cores) or None, "sockets": len(sockets) or None, "usable_by_this_process": usable}


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
    topo = host_cpu_topology()
    return {"value": round(best[0], 3), "unit": "query-image-pairs/s", "cores": best[1], "kind": "port",
            "host_physical_cores": topo["physical_cores"], "host_logical_cpus": topo["logical_cpus"], "host_sockets": topo["sockets"],
            "sample": "{} class calls looped one at a time (reference evaluate.py:323-331 call pattern) on one "
                      "60x80x1024 feature map, {:.1f} s, torch CPU fp32, best of {} threads ({}) on {} hw threads"
                      .format(best[2], best[3], "/".join(str(r[1]) for r in results),
                              ", ".join("{}: {:.1f} pairs/s".format(r[1], r[0]) for r in results), ncores)}


def replayed_counters(precision):
    """The committed rocprofv3 PMC passes of the conv 7x7 kernel (profiles/conv1_traffic_<precision>.json, written by
    tools/summarize_prof.py --traffic: FETCH_SIZE doubled per the gfx950 correction + WRITE_SIZE, SQ / GRBM pass;
    separate passes).  These are NOT measured in this run."""
    path = os.path.join(REPO, "profiles", ("spectral_traffic_{}.json" if precision in FFT_MODES else "conv1_traffic_{}.json").format(precision))
    if not os.path.exists(path):
        return None
    with open(path) as f:
        t = json.load(f)
    t["path"] = os.path.relpath(path, REPO)
    return t


PMC_PASSES = (("FETCH_SIZE",), ("WRITE_SIZE",), ("SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE"))
PMC_KERNELS = {"spectral_gemm_f16_kernel": "%spectral_gemm_f16_kernel%", "spectral_gemm_kernel": "%spectral_gemm_kernel%",
               "corr_f16x3_kernel": "%corr_f16x3_kernel%", "conv_f16x3_kernel<5>": "%conv_f16x3_kernelILi5%",
               "conv_f16x3_kernel<7>": "%conv_f16x3_kernelILi7%", "fft_forward_kernel": "%fft_forward_kernel%",
               "fft_inverse_kernel": "%fft_inverse_kernel%", "sample_decode_kernel": "%sample_decode_kernel%",
               "dft_forward_kernel": "%dft_forward_kernel%", "dft_inverse_kernel": "%dft_inverse_kernel%",
               "conv3_f16x3_kernel": "%conv3_f16x3_kernel%"}


def live_counters(precision, classes, keep_dir=None, timeout_s=150, passes=None, tag=""):
    """HBM traffic and matrix-pipe counters of the step's kernels, measured NOW: rocprofv3 PMC passes over a short child run
    of this very script (same workload, 3 steps), one pass per counter group as the HBM / rocprofv3 section of
    MI355X_MICROARCH.md prescribes (FETCH_SIZE and WRITE_SIZE never share a pass; --kernel-trace only, no other trace
    domain next to --pmc).  Returns {"kernels": {name: {counter: mean per launch, "avg_launch_us": .., "launches": n}},
    "passes": [...], "seconds": t} or {"error": ...} - the bench line never depends on it."""
    import glob
    import shutil
    import sqlite3
    import subprocess
    import tempfile
    rocprof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(rocprof):
        return {"error": "rocprofv3 not found"}
    root = tempfile.mkdtemp(prefix="os2d_pmc_", dir="/tmp")
    child = [sys.executable, os.path.join(REPO, "bench.py"), "--steps", "3", "--warmup", "1", "--precision", precision,
             "--classes", str(classes), "--no-cpu-baseline", "--no-other-precision", "--no-end-to-end", "--no-sweep", "--no-live-counters"]
    env = dict(os.environ, TMPDIR="/tmp")
    out = {"kernels": {}, "passes": [], "child": " ".join(child[1:])}
    t0 = time.perf_counter()
    try:
        for group in (passes or PMC_PASSES):
            d = os.path.join(root, "pmc_" + "_".join(group))
            cmd = [rocprof, "--pmc"] + list(group) + ["--kernel-trace", "-d", d, "-o", "pmc", "--"] + child
            r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=timeout_s)
            dbs = glob.glob(os.path.join(d, "**", "*.db"), recursive=True)
            out["passes"].append({"counters": list(group), "rc": r.returncode, "databases": len(dbs)})
            if r.returncode != 0 or not dbs:
                out["passes"][-1]["tail"] = r.stdout.decode("utf-8", "replace")[-400:]
                continue
            cur = sqlite3.connect(dbs[0]).cursor()
            for name, like in PMC_KERNELS.items():
                row = cur.execute("select count(*), avg(d.end - d.start) from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s "
                                  "on d.kernel_id = s.id where s.kernel_name like ?", (like,)).fetchone()
                if not row or not row[0]:
                    continue
                k = out["kernels"].setdefault(name, {})
                k["launches"], k["avg_launch_us_" + group[0]] = int(row[0]), round(row[1] / 1e3, 2)
                for c in group:
                    v = cur.execute("select avg(e.value) from rocpd_pmc_event e join rocpd_info_pmc p on e.pmc_id = p.id "
                                    "join rocpd_kernel_dispatch d on e.event_id = d.event_id join rocpd_info_kernel_symbol s "
                                    "on d.kernel_id = s.id where p.name = ? and s.kernel_name like ?", (c, like)).fetchone()
                    if v and v[0] is not None:
                        k[c] = float(v[0])
        for k in out["kernels"].values():
            if "FETCH_SIZE" in k and "WRITE_SIZE" in k:
                # KiB per launch; gfx950: FETCH_SIZE counts wide coalesced reads at half their bytes (the guide's correction)
                k["hbm_bytes_per_launch"] = int((2.0 * k["FETCH_SIZE"] + k["WRITE_SIZE"]) * 1024.0)
            if k.get("SQ_VALU_MFMA_BUSY_CYCLES") and k.get("GRBM_GUI_ACTIVE"):
                # SQ_VALU_MFMA_BUSY_CYCLES counts units of 32 cycles summed over the 1024 SIMDs of the chip
                k["mfma_pipe_busy"] = round(k["SQ_VALU_MFMA_BUSY_CYCLES"] * 32 / (1024 * k["GRBM_GUI_ACTIVE"]), 4)
                us = k.get("avg_launch_us_SQ_VALU_MFMA_BUSY_CYCLES")
                if us:
                    k["effective_clock_ghz"] = round(k["GRBM_GUI_ACTIVE"] / us / 1e3, 3)
    except Exception as e:   # noqa: BLE001 - a profiler problem must never cost the bench line
        out["error"] = "{}: {}".format(type(e).__name__, e)
    out["seconds"] = round(time.perf_counter() - t0, 1)
    if keep_dir:
        try:
            os.makedirs(keep_dir, exist_ok=True)
            with open(os.path.join(keep_dir, "live_counters_{}{}.json".format(precision, tag)), "w") as f:
                json.dump(out, f, indent=1)
        except OSError:
            pass
    shutil.rmtree(root, ignore_errors=True)
    return out


def end_to_end(dev, B, P, inverse, state, precision, arch="resnet50", steps=5, warmup=2):
    """Secondary number: one 1280x960 image through the PyTorch-ROCm ResNet-C4 backbone, the
    HIP head against B classes and the HIP decode + per-class NMS of all 4800 boxes per class (score threshold -inf,
    the reference's eval default), random-init weights; plus the one-off construction of the class head from B
    240x240 class images (one batched backbone pass)."""
    from os2d_amd.engine.evaluate import build_class_head
    from os2d_amd.modeling.model import Os2dModel
    from os2d_amd.structures.feature_map import FeatureMapSize
    torch.manual_seed(0)
    net = Os2dModel(is_cuda=False, merge_branch_parameters=True, backbone_arch=arch,
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
            "backbone": arch, "backbone_ms": round(phases[0] / steps, 3), "head_ms": round(phases[1] / steps, 3),
            "decode_nms_ms": round(phases[2] / steps, 3), "detections_per_image": n_det,
            "class_head_build_ms": round(t_build * 1e3, 2), "precision": precision, "steps": steps,
            "what": "{}-C4 (PyTorch-ROCm/MIOpen fp32, random init) on 1x3x960x1280 + HIP head x {} classes + HIP "
                    "decode and per-class NMS of all 4800 boxes per class; class_head_build_ms = {} class images 240x240 "
                    "-> backbone (one batch) -> 15x15 class maps (once per class set)".format(arch, B, B)}


class Workload(object):
    """One benchmark configuration resident in HBM: feature map(s), the class head of this rank's classes and, with
    several ranks, the class-sharded wrapper."""

    def __init__(self, dev, rank, world, classes_total, variant, pyramid, use_dist, gather):
        from os2d_amd import _lib
        from os2d_amd.modeling.head import build_os2d_head_creator
        from os2d_amd.parallel import ClassShardedHead, shard_bounds
        from os2d_amd.structures.feature_map import FeatureMapSize
        from os2d_amd.utils import synthetic
        self.lib = _lib.load()
        self._lib_mod = _lib
        self.dev, self.rank, self.world, self.use_dist, self.gather = dev, rank, world, use_dist, gather
        self.P, self.inverse = (6, True) if variant == "v2" else (4, False)
        self.variant, self.pyramid = variant, pyramid
        self.classes_total = classes_total
        self.bounds = shard_bounds(classes_total, world)
        s, e = self.bounds[rank]
        self.B_local = e - s
        self.state = synthetic.make_transform_net_state(self.P, seed=1)
        self.fm_cpu = synthetic.make_feature_map(C_FEAT, H_FM, W_FM, seed=0)
        # global class id c uses seed 1000 + (c % 64): 64 distinct synthetic class maps, every class its own device copy
        distinct = {}
        for c in range(s, e):
            distinct.setdefault(c % DISTINCT_CLASS_MAPS, None)
        for k in distinct:
            distinct[k] = synthetic.make_class_feature_maps(1, C_FEAT, sizes=[(15, 15)], seed=1000 + k)[0]
        self.class_fms_cpu = [distinct[c % DISTINCT_CLASS_MAPS] for c in range(s, min(e, s + DISTINCT_CLASS_MAPS))]
        self.creator = build_os2d_head_creator(self.P == 4, False, self.inverse, FeatureMapSize(w=16, h=16), FeatureMapSize(w=16, h=16))
        self.creator.aligner.parameter_regressor.load_state_dict(self.state)
        self.creator.to(dev).eval()
        self.fm = self.fm_cpu.to(dev)
        with torch.no_grad():
            self.head = self.creator.create_os2d_head([distinct[c % DISTINCT_CLASS_MAPS].to(dev) for c in range(s, e)])
        self.sharded = None
        if use_dist:
            self.sharded = ClassShardedHead(self.creator, group=None, gather="scores" if gather == "scores" else "all",
                                            num_classes=classes_total, local_head=self.head, reuse_buffers=3)
        self.gather_wait_ms = None     # mean time per step the compute stream waited for the step's gathers (set by run())
        self.runner, self.level_fms = None, None
        if pyramid:
            from os2d_amd.engine.pyramid import PyramidHeadRunner
            self.level_fms = [synthetic.make_feature_map(C_FEAT, h, w, seed=100 + i).to(dev) for i, (h, w) in enumerate(LEVEL_HW)]
            self.runner = PyramidHeadRunner(self.sharded if (self.sharded is not None and gather != "detections") else self.head, device=dev)
        self.coder = None
        if gather == "detections" and use_dist:
            from os2d_amd.modeling.box_coder import Os2dBoxCoder
            self.coder = Os2dBoxCoder(output_box_grid_generator=self.creator.box_grid_generator_image_level)
            self.img_sizes = [FeatureMapSize(w=16 * w, h=16 * h) for h, w in (LEVEL_HW if pyramid else [(H_FM, W_FM)])]
            self.local_ids = list(range(s, e))

    @property
    def locations(self):
        return sum(h * w for h, w in LEVEL_HW) if self.pyramid else H_FM * W_FM

    def set_gather(self, gather):
        """Switch what the N>1 step exchanges (same resident data)."""
        w = Workload.__new__(Workload)
        w.__dict__.update(self.__dict__)
        w.gather = gather
        from os2d_amd.parallel import ClassShardedHead
        w.sharded = ClassShardedHead(self.creator, group=None, gather="scores" if gather == "scores" else "all",
                                     num_classes=self.classes_total, local_head=self.head, reuse_buffers=3)
        w.coder = None
        if self.pyramid:
            from os2d_amd.engine.pyramid import PyramidHeadRunner
            w.runner = PyramidHeadRunner(w.sharded if gather != "detections" else self.head, device=self.dev)
        if gather == "detections":
            from os2d_amd.modeling.box_coder import Os2dBoxCoder
            from os2d_amd.structures.feature_map import FeatureMapSize
            w.coder = Os2dBoxCoder(output_box_grid_generator=self.creator.box_grid_generator_image_level)
            w.img_sizes = [FeatureMapSize(w=16 * ww, h=16 * hh) for hh, ww in (LEVEL_HW if self.pyramid else [(H_FM, W_FM)])]
            s, e = self.bounds[self.rank]
            w.local_ids = list(range(s, e))
        return w

    def _new_event_set(self):
        arr = (ctypes.c_void_p * 13)()
        for i in range(13):
            ev = ctypes.c_void_p()
            self._lib_mod.check(self.lib.os2d_prof_event_create(ctypes.byref(ev)), "os2d_prof_event_create")
            arr[i] = ev.value
        return arr

    def sync_all(self):
        import torch.distributed as dist
        torch.cuda.synchronize(self.dev)
        if self.use_dist:
            dist.barrier()
            torch.cuda.synchronize(self.dev)

    def run(self, precision, steps, warmup):
        """W warm-up + exactly K timed steps in one arithmetic mode; returns (seconds (max over ranks), per-stage mean
        ms or None)."""
        import torch.distributed as dist
        head, sharded, runner = self.head, self.sharded, self.runner
        head.precision = precision
        staged = sharded is None and runner is None
        # one set of 13 stage events per timed step, so nothing has to be read back inside the timed region
        event_sets = [self._new_event_set() for _ in range(steps)] if staged else []
        pending = []

        def step(events):
            with torch.no_grad():
                if self.coder is not None:
                    # class-sharded decode: the rank's own classes through head + decode + per-class NMS, then the union of
                    # the surviving detections (two small collectives)
                    from os2d_amd.parallel import all_gather_detections
                    if runner is not None:
                        loc, cls, _, _ = runner.run(self.level_fms, inputs_are_features=True)
                        loc, cls = [l[0] for l in loc], [c[0] for c in cls]
                    else:
                        l, c, _, _ = head(self.fm)
                        loc, cls = [l[0].flatten(2)], [c[0].flatten(1)]
                    inverse = None
                    if len(self.img_sizes) > 1:          # the levels of a pyramid are merged in the frame of the 1280x960 image
                        from os2d_amd.modeling.box_coder import ResizeBoxes
                        from os2d_amd.structures.feature_map import FeatureMapSize
                        inverse = [ResizeBoxes(FeatureMapSize(w=16 * W_FM, h=16 * H_FM)) for _ in self.img_sizes]
                    dets = self.coder.decode_pyramid(loc, cls, self.img_sizes, self.local_ids, inverse_box_transforms=inverse,
                                                     nms_score_threshold=float("-inf"))      # reference eval default (config.py:198)
                    return all_gather_detections(dets)
                if runner is not None:
                    return runner.run(self.level_fms, inputs_are_features=True)
                if sharded is not None:
                    # the all-gather of this step runs asynchronously (RCCL stream) and is waited for only after the
                    # NEXT step's kernels have been queued, so the xGMI transfer hides behind compute
                    pending.append(sharded(self.fm, async_gather=True))
                    if len(pending) > 1:
                        pending.pop(0)()
                    return None
                return head(self.fm, stage_events=events)

        for _ in range(warmup):
            step(None)
        while pending:
            pending.pop(0)()
        self.sync_all()
        if sharded is not None:
            sharded.wait_events = []       # (before, after) HIP events around every gather wait of the timed steps
        # ---- timed region: exactly K steps; stage events are recorded on the launch stream inside these steps
        t0 = time.perf_counter()
        for i in range(steps):
            step(event_sets[i] if event_sets else None)
        while pending:
            pending.pop(0)()          # every gather of the timed steps completes inside the timed region
        self.sync_all()
        dt = time.perf_counter() - t0
        if sharded is not None and sharded.wait_events:
            self.gather_wait_ms = sum(a.elapsed_time(b) for a, b in sharded.wait_events) / steps
        if sharded is not None:
            sharded.wait_events = None
        if self.use_dist:
            t = torch.tensor([dt], dtype=torch.float64, device=self.dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        stage_ms = None
        if event_sets:
            ms = ctypes.c_float()
            rows = []
            for evs in event_sets:
                row = []
                for st in range(5):
                    self._lib_mod.check(self.lib.os2d_prof_event_elapsed_ms(evs[2 * st], evs[2 * st + 1], ctypes.byref(ms)), "elapsed")
                    row.append(ms.value)
                if head.last_precision in FFT_MODES:   # sub-stages of the 7x7 layer: forward FFT | spectral GEMM | inverse FFT
                    for a, b in ((10, 11), (11, 12), (12, 3)):
                        self._lib_mod.check(self.lib.os2d_prof_event_elapsed_ms(evs[a], evs[b], ctypes.byref(ms)), "elapsed")
                        row.append(ms.value)
                rows.append(row)
                for ev in evs:
                    self.lib.os2d_prof_event_destroy(ev)
            stage_ms = [sum(r[st] for r in rows) / len(rows) for st in range(min(len(r) for r in rows))]
        return dt, stage_ms

    def whole_head_flops_per_class(self):
        per_loc = FLOP_PER_LOC["corr"] + FLOP_PER_LOC["conv1"] + FLOP_PER_LOC["conv2"] + 2 * self.P * 64 * 25
        return per_loc * self.locations

    def roofline(self, precision, stage_ms=None, seconds_per_step=None):
        """Roofline object.  With stage events: the conv 7x7 kernel alone (algorithmic FLOPs of one launch / its mean
        duration).  Without (several streams / ranks): the whole head of this rank (algorithmic FLOPs of a step / step
        time) - `kernel` says which."""
        B = self.B_local
        if stage_ms is not None and precision in FFT_MODES and len(stage_ms) >= 8:
            return self.roofline_fft(stage_ms, precision)
        fell_back = precision in FFT_MODES and stage_ms is not None     # staged run without the sub-stage events: the class batch
        if fell_back:                                               # is below FFT_MIN_PAIRS and the direct kernel ran
            precision = "f32" if precision == "fft32" else "f16x3"
        peak = PEAK[precision]
        if stage_ms is not None:
            flops = FLOP_PER_LOC["conv1"] * H_FM * W_FM * B            # algorithmic FLOPs of ONE conv1 launch
            seconds = stage_ms[1] * 1e-3
            kernel = "TransformNet conv 7x7 225->128 ({})".format(
                "conv_mfma_kernel<7,...>, v_mfma_f32_32x32x2_f32" if precision == "f32"
                else "frequency domain: fft_forward + spectral_gemm (v_mfma_f32_32x32x2_f32) + fft_inverse; the FLOPs are the "
                     "DIRECT layer's (the transform route executes 16.7x fewer), so frac may exceed 1" if precision == "fft"
                else "conv_f16x3_kernel<7,...>, v_mfma_f32_32x32x16_f16 x{} per product".format(precision[-1]))
        elif precision in FFT_MODES:
            return self.roofline_fft_whole(seconds_per_step, precision)
        else:
            flops = self.whole_head_flops_per_class() * B
            seconds = seconds_per_step
            kernel = "whole head of one rank (correlation + 3 TransformNet convolutions, all MFMA kernels of a step{})".format(
                ", 7 levels" if self.pyramid else "")
        achieved = flops / seconds
        r = {"kernel": kernel, "bound": "mfma", "achieved": round(achieved / 1e12, 3), "peak": peak / 1e12, "unit": "TFLOP/s",
             "frac": round(achieved / peak, 4), "traffic": None, "flops_per_launch": flops,
             "avg_launch_ms": round(seconds * 1e3, 4), "timing": "HIP events on the launch stream, this run" if stage_ms is not None
             else "wall clock of the timed steps, this run"}
        if stage_ms is not None and precision != "fft":
            c = replayed_counters(precision)
            if c:
                r["traffic"] = int(c["bytes_per_class"] * B)
                r["hbm_gbps"] = round(r["traffic"] / seconds / 1e9, 1)      # 8000 GB/s peak: far from HBM-bound
                r["mfma_pipe_busy"] = c.get("mfma_pipe_busy")
                if c.get("grbm_gui_active") and c.get("avg_launch_us"):
                    clk = c["grbm_gui_active"] / (c["avg_launch_us"] * 1e-6) / 1e9
                    r["effective_clock_ghz"] = round(clk, 3)
                    r["frac_clock_adjusted"] = round(achieved / (peak * clk / 2.4), 4)   # against the peak at the sustained clock
                r["counters_source"] = "REPLAYED from {} (rocprofv3 PMC passes recorded at {} classes, {}); not measured in this run".format(
                    c["path"], c.get("classes_profiled"), c.get("source"))
            # algorithmic HBM bytes of the same launch: every input plane read once (226 fp32 planes, or 29 groups x 8
            # channels x (hi|lo) halves) + the 128 output planes written once + the packed weights once
            plane = int(self.lib.os2d_plane_floats(H_FM, W_FM))
            in_planes = 226 if precision == "f32" else 232
            r["algorithmic_bytes"] = int(B * (in_planes + 128) * plane * 4 + self.lib.os2d_packed_conv_bytes(1, PREC_ID["f16x3" if precision == "fft" else precision]))
        if precision not in ("f32", "fft"):
            # every algorithmic product costs three (f16x2: two, 7x7 layer only) half-precision MFMA products: the ceiling for
            # algorithmic FLOP/s on this instruction is peak/3 (peak/2); the executed rate also includes the tile /
            # channel-group padding (x1.118 for the 7x7 kernel)
            terms = int(precision[-1]) if stage_ms is not None else 3
            r["algorithmic_ceiling"] = round(peak / terms / 1e12, 1)
            r["frac_of_algorithmic_ceiling"] = round(achieved / (peak / terms), 4)
            if stage_ms is not None:
                r["executed_mfma_tflops"] = round(terms * achieved * 1.118 / 1e12, 1)
                r["executed_frac_of_peak"] = round(terms * achieved * 1.118 / peak, 4)
        if fell_back:
            r["note"] = "a frequency-domain mode was requested; with fewer than 7 image-class pairs the head runs the direct f16x3 7x7 kernel"
        return r

    def roofline_corr(self, stage_ms, precision):
        """The correlation kernel (as long as the spectral GEMM in the default mode): algorithmic FLOPs of one launch
        (2 x 225 x 1024 per class-location) / its mean duration from the same stage events, against the MFMA peak of the
        instruction it runs on; the split-fp16 kernel executes three MFMAs per product on 256 padded rows."""
        flops = FLOP_PER_LOC["corr"] * H_FM * W_FM * self.B_local
        seconds = stage_ms[0] * 1e-3
        f16 = precision not in ("f32", "fft32")
        peak = PEAK["f16x3"] if f16 else PEAK["f32"]
        r = {"kernel": "corr_f16x3_kernel (v_mfma_f32_32x32x16_f16 x3 per product)" if f16 else "corr_mfma_kernel (v_mfma_f32_32x32x2_f32)",
             "bound": "mfma", "achieved": round(flops / seconds / 1e12, 3), "peak": peak / 1e12, "unit": "TFLOP/s",
             "frac": round(flops / seconds / peak, 4), "flops_per_launch": flops, "avg_launch_ms": round(seconds * 1e3, 4),
             "timing": "HIP events on the launch stream, this run"}
        if f16:
            r["executed_frac_of_peak"] = round(3 * (256.0 / 225.0) * flops / seconds / peak, 4)
            # operands once (fp16 hi + lo = 4 B per value: image map C x HW, class maps B x C x 256 rows) + the fp32 correlation
            # tensor B x 225 x HW and the per-location inverse norms; `traffic` (live PMC) over this = operand re-fetch
            r["algorithmic_bytes"] = 4 * C_FEAT * H_FM * W_FM + 4 * self.B_local * C_FEAT * 256 + 4 * self.B_local * 226 * H_FM * W_FM
        return r

    def roofline_fft_whole(self, seconds, precision="fft"):
        """fft mode without stage events (several streams / ranks): all MFMA work of one rank's step against the blend of
        the two instruction peaks it runs on - the correlation, the two 5x5 layers (and the 7x7 layer of maps that do not
        fit the in-LDS transform) count their ALGORITHMIC FLOPs against the fp16 MFMA peak (each costs three MFMA products:
        ceiling 1/3), the per-bin complex GEMMs their executed FLOPs against the fp32 MFMA peak."""
        B = self.B_local
        levels = LEVEL_HW if self.pyramid else [(H_FM, W_FM)]
        f16 = f32 = 0
        for h, w in levels:
            pP, pQ, nb = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
            per_loc = FLOP_PER_LOC["corr"] + FLOP_PER_LOC["conv2"] + 2 * self.P * 64 * 25
            t6 = (ctypes.c_int * 6)()       # TY, TX, ...: maps beyond one transform are cut into TY x TX overlap-save tiles
            if precision == "fftx3":
                ok = self.lib.os2d_dft_sizes(h, w, ctypes.byref(pP), ctypes.byref(pQ), ctypes.byref(nb), t6) == 0
            else:
                ok = self.lib.os2d_fft_sizes(h, w, ctypes.byref(pP), ctypes.byref(pQ), ctypes.byref(nb)) == 0
                t4 = [ctypes.c_int() for _ in range(4)]
                self.lib.os2d_fft_tiles(h, w, *[ctypes.byref(t) for t in t4])
                t6[0], t6[1] = t4[0].value, t4[1].value
            if ok:
                f32 += 8 * 128 * 225 * B * t6[0] * t6[1] * pP.value * (pQ.value // 2 + 1)
            else:
                per_loc += FLOP_PER_LOC["conv1"]
            f16 += per_loc * h * w * B
        if precision == "fftx3":       # the per-bin GEMMs run on the fp16 MFMA too: three matrix products per complex product term
            f16, f32 = f16 + f32, 0
        peak = (f16 + f32) / (f16 / PEAK["f16x3"] + f32 / PEAK["fft"])
        achieved = (f16 + f32) / seconds
        return {"kernel": "whole head of one rank: correlation + 5x5 layers on v_mfma_f32_32x32x16_f16 (algorithmic FLOPs, three "
                          "MFMA products each) + per-bin complex GEMMs of the 7x7 layer on v_mfma_f32_32x32x2_f32 (executed FLOPs)"
                          + (", 7 levels" if self.pyramid else ""),
                "bound": "mfma", "achieved": round(achieved / 1e12, 3), "peak": round(peak / 1e12, 1), "unit": "TFLOP/s",
                "frac": round(achieved / peak, 4), "traffic": None, "flops_per_launch": f16 + f32,
                "flops_fp16_mfma_algorithmic": f16, "flops_fp32_mfma_executed": f32,
                "peak_is": "FLOP-weighted harmonic blend of the fp16 (2500) and fp32 (157.3) dense MFMA peaks for this mix",
                "avg_launch_ms": round(seconds * 1e3, 4), "timing": "wall clock of the timed steps, this run"}

    def transform_size(self, precision):
        """(P, Q, nbins) as ctypes ints of the 60 x 80 map's transform: the matrix-product transforms of "fftx3" take P = 64,
        Q = 84 too, but other maps differ from the FFT-friendly sizes of "fft" / "fft32"."""
        pP, pQ, nb = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        if precision == "fftx3":
            self.lib.os2d_dft_sizes(H_FM, W_FM, ctypes.byref(pP), ctypes.byref(pQ), ctypes.byref(nb), None)
        else:
            self.lib.os2d_fft_sizes(H_FM, W_FM, ctypes.byref(pP), ctypes.byref(pQ), ctypes.byref(nb))
        return pP, pQ, nb

    def roofline_fft(self, stage_ms, precision="fft"):
        """fft mode: the spectral GEMM (dominant kernel of the step).  Per bin Y[128 x pairs] = K[128 x 225] X[225 x pairs]
        in complex fp32 = 8 real FLOPs per complex multiply-add, all of them issued as v_mfma_f32_32x32x2_f32 with
        k = {re, im}; bins = P * (Q/2 + 1) of the P x Q transform of the map.  Algorithmic HBM bytes: the weight spectra,
        the input spectra and the output spectra once each (8 B per complex value)."""
        pP, pQ, nb = self.transform_size(precision)
        bins, pairs = pP.value * (pQ.value // 2 + 1), self.B_local
        fwd, gemm, inv = stage_ms[5] * 1e-3, stage_ms[6] * 1e-3, stage_ms[7] * 1e-3
        flops = 8 * 128 * 225 * pairs * bins
        nbytes = 8 * bins * (128 * 225 + 225 * pairs + 128 * pairs)
        peak = PEAK["fft"]
        direct = FLOP_PER_LOC["conv1"] * H_FM * W_FM * pairs
        r = {"kernel": "spectral_gemm_kernel (7x7 layer in the frequency domain: complex GEMM per bin, v_mfma_f32_32x32x2_f32)",
                "bound": "mfma", "achieved": round(flops / gemm / 1e12, 3), "peak": peak / 1e12, "unit": "TFLOP/s",
                "frac": round(flops / gemm / peak, 4), "traffic": None, "flops_per_launch": flops,
                "avg_launch_ms": round(gemm * 1e3, 4), "timing": "HIP events on the launch stream, this run",
                "transform": [pP.value, pQ.value], "bins": bins,
                "algorithmic_bytes": nbytes, "hbm_view": {"achieved": round(nbytes / gemm / 1e9, 1), "peak": 8000.0, "unit": "GB/s",
                                                          "frac": round(nbytes / gemm / 8e12, 4)},
                "layer_ms": {"fft_forward": round(fwd * 1e3, 4), "spectral_gemm": round(gemm * 1e3, 4),
                             "fft_inverse": round(inv * 1e3, 4)},
                "layer_direct_equivalent_tflops": round(direct / (stage_ms[1] * 1e-3) / 1e12, 1),
                "note": "layer_direct_equivalent_tflops = FLOPs of the DIRECT 7x7 layer / time of the whole frequency-domain "
                        "layer (the transform route executes {:.1f}x fewer FLOPs); not a utilisation figure".format(direct / flops)}
        if precision == "fftx3":
            # split-half GEMM: 16x the matrix rate for 3x the matrix work - the launch is a stream of its operands
            r["kernel"] = ("spectral_gemm_f16_kernel (7x7 layer in the frequency domain: complex GEMM per bin on v_mfma_f32_32x32x16_f16, "
                           "spectra split into fp16 hi + lo)")
            r["mfma_view"] = {"achieved": r["achieved"], "executed_tflops": round(3 * flops / gemm / 1e12 * 232 / 225, 1), "peak": PEAK["fftx3"] / 1e12,
                              "unit": "TFLOP/s", "frac_executed": round(3 * flops / gemm / PEAK["fftx3"] * 232 / 225, 4)}
            hv = r.pop("hbm_view")
            r.update({"bound": "hbm", "achieved": hv["achieved"], "peak": 8000.0, "unit": "GB/s", "frac": hv["frac"]})
        c = replayed_counters(precision)
        if c and c.get("classes_profiled") == pairs:      # the weight spectra are shared by all classes: no per-class scaling
            r["traffic"] = int(c["bytes_per_launch"])
            r["hbm_gbps"] = round(r["traffic"] / gemm / 1e9, 1)
            r["mfma_pipe_busy"] = c.get("mfma_pipe_busy")
            r["effective_clock_ghz"] = c.get("effective_clock_ghz")
            if c.get("effective_clock_ghz"):
                if precision == "fft":
                    r["frac_clock_adjusted"] = round(flops / gemm / (peak * c["effective_clock_ghz"] / 2.4), 4)
            r["counters_source"] = "REPLAYED from {} (rocprofv3 PMC passes recorded at {} classes, {}); not measured in this run".format(
                c["path"], c.get("classes_profiled"), c.get("source"))
        return r

    def rooflines(self, stage_ms, precision):
        """One roofline object per kernel of the step, from the stage events of THIS run: {name: object}, every object with
        `avg_launch_ms` (so the caller can name the longest), `pmc_kernel` (its name in the live PMC passes) and the bound it is
        priced against - the dense MFMA peak of the instruction for the GEMM-shaped kernels (algorithmic FLOPs), the 8 TB/s HBM
        peak for the streaming ones (algorithmic bytes: every operand / result of the launch once)."""
        B, HW, P = self.B_local, H_FM * W_FM, self.P
        out = {}
        fft = precision in FFT_MODES and len(stage_ms) >= 8
        f16 = precision not in ("f32", "fft32")

        def mfma(kernel, pmc, flops, ms, split_terms, note=None):
            peak = PEAK["f16x3"] if f16 else PEAK["f32"]
            r = {"kernel": kernel, "pmc_kernel": pmc, "bound": "mfma", "achieved": round(flops / (ms * 1e-3) / 1e12, 3), "peak": peak / 1e12,
                 "unit": "TFLOP/s", "frac": round(flops / (ms * 1e-3) / peak, 4), "traffic": None, "flops_per_launch": flops,
                 "avg_launch_ms": round(ms, 4), "timing": "HIP events on the launch stream, this run"}
            if f16:
                r["executed_frac_of_peak"] = round(split_terms * flops / (ms * 1e-3) / peak, 4)
            if note:
                r["note"] = note
            return r

        def hbm(kernel, pmc, nbytes, ms, note=None):
            r = {"kernel": kernel, "pmc_kernel": pmc, "bound": "hbm", "achieved": round(nbytes / (ms * 1e-3) / 1e9, 1), "peak": 8000.0,
                 "unit": "GB/s", "frac": round(nbytes / (ms * 1e-3) / 8e12, 4), "traffic": None, "algorithmic_bytes": int(nbytes),
                 "avg_launch_ms": round(ms, 4), "timing": "HIP events on the launch stream, this run"}
            if note:
                r["note"] = note
            return r

        out["corr"] = self.roofline_corr(stage_ms, precision)
        out["corr"]["pmc_kernel"] = "corr_f16x3_kernel" if f16 else "corr_mfma_kernel"
        plane = int(self.lib.os2d_plane_floats(H_FM, W_FM))
        if fft:
            pP, pQ, nb = self.transform_size(precision)
            bins = nb.value
            mm = precision == "fftx3"      # the transforms as matrix products on the half-precision matrix cores (dft_mfma.hip)
            out["spectral_gemm"] = self.roofline_fft(stage_ms, precision)
            out["spectral_gemm"]["pmc_kernel"] = "spectral_gemm_f16_kernel" if precision == "fftx3" else "spectral_gemm_kernel"
            out["fft_forward"] = hbm("forward transform of the 7x7 layer's input (relu + normalisation folded into the load): reads corr + inverse "
                                     "norms, writes the input spectra" + (" (matrix products, v_mfma_f32_32x32x16_f16 x3)" if mm else " (in-LDS FFT)"),
                                     "dft_forward_kernel" if mm else "fft_forward_kernel",
                                     B * (225 * HW * 4 + HW * 4 + 225 * bins * 8), stage_ms[5])
            out["fft_inverse"] = hbm("inverse transform + bias / ReLU / fp16 split epilogue: reads the output spectra, writes the 5x5 layer's "
                                     "activations" + (" (matrix products, v_mfma_f32_32x32x16_f16 x3)" if mm else " (in-LDS FFT)"),
                                     "dft_inverse_kernel" if mm else "fft_inverse_kernel", B * 128 * (bins * 8 + HW * 4), stage_ms[7])
        else:
            r = self.roofline(precision, stage_ms[:5], None)
            r["pmc_kernel"] = "conv_f16x3_kernel<7>"
            out["conv1"] = r
        out["conv2"] = mfma("TransformNet conv 5x5 128->64 (conv_f16x3_kernel<5,...>)" if f16 else "TransformNet conv 5x5 128->64 (conv_mfma_kernel<5,...>)",
                            "conv_f16x3_kernel<5>", FLOP_PER_LOC["conv2"] * HW * B, stage_ms[2], 3)
        if stage_ms[4] < 0.02 and f16:      # (two event records back to back: ~5 us)
            # round 6: the last layer and the alignment epilogue are ONE launch on the split-fp16 route (conv3_f16x3_kernel<FUSE>): its
            # time sits in the conv3 stage events, the sample stage is empty.  Priced as the gather stream it mostly is.
            out["conv3_sample"] = hbm("TransformNet conv 5x5 64->P (v_mfma_f32_16x16x32_f16) + resample / pool / box / corner extraction in one "
                                      "launch (conv3_f16x3_kernel<true>): reads the 64-channel activations and, through L2, the correlation "
                                      "tensor; writes loc | cls | corners", "conv3_f16x3_kernel",
                                      B * (225 * HW * 4 + 64 * 4 * plane + 13 * HW * 4), stage_ms[3])
        else:
            out["conv3"] = mfma("TransformNet conv 5x5 64->P (conv3_f16x3_kernel, v_mfma_f32_16x16x32_f16)" if f16 else "TransformNet conv 5x5 64->P",
                                "conv3_f16x3_kernel", 2 * P * 64 * 25 * HW * B, stage_ms[3], 3,
                                note="P = {} output rows on a 16-row matrix instruction: the executed work is 16 / P of the algorithmic one".format(P))
            out["sample_decode"] = hbm("resample + pool + box / corner extraction (sample_decode_kernel): reads the correlation tensor through L2, "
                                       "writes loc | cls | corners", "sample_decode_kernel", B * (225 * HW * 4 + P * HW * 4 + 13 * HW * 4), stage_ms[4])
        for r in out.values():
            r.setdefault("traffic", None)
        return out

    def describe(self):
        return ("OS2D head, ResNet50-C4 features of one 1280x960 image ({}), {} classes in total ({} on this GPU), {}, {} "
                "(P={}, inverse={}), head only, features resident in HBM".format(
                    "7-level pyramid 30x40..96x128, 39580 locations" if self.pyramid else "1x1024x60x80",
                    self.classes_total, self.B_local, "7 scales 0.5-1.6" if self.pyramid else "single scale",
                    self.variant.upper(), self.P, int(self.inverse)))


def precision_deviation(w):
    """max |mode - f32| of the head outputs on the benchmarked tensors (this run, this workload)."""
    out = {}
    with torch.no_grad():
        ref = [t.clone() for t in w.head(w.fm, precision="f32")]
        keep = {}
        for p in ("f16x3", "f16x2", "fft", "fftx3", "fft32"):
            o = w.head(w.fm, precision=p)
            out[p] = {"cls": float((o[1] - ref[1]).abs().max()), "loc": float((o[0] - ref[0]).abs().max()),
                      "corners_px": float((o[3] - ref[3]).abs().max())}
            if p in ("fft", "f16x3"):
                keep[p] = o[0].clone()
        # the two fp32-equivalent modes share every stage but the 7x7 layer: their outputs differ in the last bits only
        out["fft_vs_f16x3_loc"] = float((keep["fft"] - keep["f16x3"]).abs().max())
        out["fft_vs_f16x3_identical_fraction"] = float((keep["fft"] == keep["f16x3"]).float().mean())
        out["range_flag"] = w.head.range_status(synchronize=True)
    return out


def sweep_entry(dev, name, classes, variant, pyramid, precision, steps, warmup):
    w = Workload(dev, 0, 1, classes, variant, pyramid, False, "all")
    dt, stage_ms = w.run(precision, steps, warmup)
    e = {"name": name, "workload": w.describe(), "classes": classes, "variant": variant, "pyramid": pyramid,
         "precision": precision, "steps": steps, "warmup": warmup, "value": round(classes * steps / dt, 2),
         "unit": "query-image-pairs/s", "ms_per_step": round(dt / steps * 1e3, 4)}
    if stage_ms:
        e["stages_ms"] = {k: round(v, 4) for k, v in zip(STAGES, stage_ms)}      # zip stops at the 5 stages
        per_kernel = w.rooflines(stage_ms, w.head.last_precision or precision)
        longest = max(per_kernel, key=lambda k: per_kernel[k]["avg_launch_ms"])
        e["roofline"] = dict(per_kernel[longest], stage=longest)
        e["roofline_other"] = {k: {f: v[f] for f in ("bound", "achieved", "peak", "unit", "frac", "avg_launch_ms", "pmc_kernel", "traffic", "algorithmic_bytes") if f in v}
                               for k, v in per_kernel.items() if k != longest}
    else:
        e["roofline"] = w.roofline(precision, stage_ms, dt / steps)
    if pyramid:
        # the runner's default is the levels back to back on one stream (measured faster in rounds 4 and 5); the same levels on
        # one HIP stream per level (BASELINE.json configs[4]'s wording) right after, for the record
        from os2d_amd.engine.pyramid import PyramidHeadRunner
        serial_runner = w.runner
        w.runner = PyramidHeadRunner(w.head, num_streams=len(LEVEL_HW), device=dev)
        dt7, _ = w.run(precision, steps, warmup)
        w.runner = serial_runner
        e["pyramid_serial_ms"] = e["ms_per_step"]
        e["pyramid_streams_ms"] = round(dt7 / steps * 1e3, 4)
    e["head_tflops_algorithmic"] = round(w.whole_head_flops_per_class() * classes * steps / dt / 1e12, 3)
    return e, w


_LINE_OUT = None      # the process's ORIGINAL stdout, kept for the one JSON line


def claim_stdout():
    """stdout carries ONE line - the record - and nothing else (the driver parses it).  Libraries print there too (RCCL's banner and
    warnings under --force-dist / N > 1 put five more lines in front of the record in round 6's first run): file descriptor 1 is
    pointed at stderr for the life of the process and the record is written to a duplicate of the original descriptor."""
    global _LINE_OUT
    if _LINE_OUT is None:
        sys.stdout.flush()
        _LINE_OUT = os.fdopen(os.dup(1), "w")
        os.dup2(2, 1)


def print_line(text):
    out = _LINE_OUT or sys.stdout
    out.write(text + "\n")
    out.flush()


def main():
    args = parse()
    claim_stdout()
    if os.environ.get("OS2D_BENCH_WATCHDOG"):      # debugging aid: dump the Python stacks every N seconds to stderr
        import faulthandler
        faulthandler.dump_traceback_later(float(os.environ["OS2D_BENCH_WATCHDOG"]), repeat=True)
    # the host driver of these boxes supports dmabuf IPC only: without this RCCL / cross-process tensor sharing fails with
    # "hipIpcGetMemHandle: invalid argument" (already exported on the GPU boxes; kept for any environment that launches us bare)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
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
        import datetime
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        # a first N > 1 run must be diagnosable: RCCL warnings go to one file per rank whose tail lands in
        # the JSON line if anything below raises, and every collective carries a timeout instead of hanging the box
        os.environ.setdefault("NCCL_DEBUG", "WARN")
        os.environ.setdefault("NCCL_DEBUG_FILE", "/tmp/os2d_bench_rccl_{}_%h_%p.log".format(os.environ["MASTER_PORT"]))
        os.environ.setdefault("TORCH_NCCL_ASYNC_ERROR_HANDLING", "1")
    try:
        if use_dist:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev,
                                    timeout=datetime.timedelta(seconds=float(os.environ.get("OS2D_DIST_TIMEOUT_S", "180"))))
        run_bench(args, rank, local_rank, world, dev, use_dist)
    except BaseException as e:   # noqa: BLE001 - report, then fail
        if isinstance(e, SystemExit) and not e.code:
            raise
        import traceback
        line = {"metric": "query-image-pairs/s (1280-px input, ResNet50, N-class)", "value": None, "n_gpus": world, "rank": rank,
                "error": "{}: {}".format(type(e).__name__, e), "traceback_tail": traceback.format_exc()[-1500:]}
        if use_dist:
            line["rccl_log_tail"] = rccl_log_tail()
        if rank == 0:
            print_line(json.dumps(line))
        else:
            print(json.dumps(line), file=sys.stderr, flush=True)
        raise
    finally:
        if use_dist and dist.is_initialized():
            try:
                dist.destroy_process_group()
            except Exception:   # noqa: BLE001
                pass


def rccl_log_tail(limit=1500):
    """Tail of this job's RCCL debug files (NCCL_DEBUG=WARN, one per rank)."""
    import glob
    pattern = os.environ.get("NCCL_DEBUG_FILE", "").replace("%h", "*").replace("%p", "*")
    text = []
    for path in sorted(glob.glob(pattern))[:16] if pattern else []:
        try:
            with open(path) as f:
                t = f.read()
            if t.strip():
                text.append("[{}] {}".format(os.path.basename(path), t[-limit // 4:]))
        except OSError:
            pass
    return "\n".join(text)[-limit:]


def allgather_probe(dev, world, nbytes_per_rank=32 << 20, iters=10):
    """A standalone all-gather of 32 MB per rank (the size of one rank's loc | cls | corners block at 1024 classes over 8
    GPUs): time per collective and the bus bandwidth it achieves over xGMI ((N-1)/N of the gathered bytes per rank / time) -
    what the step's gather can hope for, measured outside the head."""
    import torch.distributed as dist
    send = torch.empty(nbytes_per_rank // 4, dtype=torch.float32, device=dev).normal_()
    recv = torch.empty(world * send.numel(), dtype=torch.float32, device=dev)
    for _ in range(3):
        dist.all_gather_into_tensor(recv, send)
    torch.cuda.synchronize(dev)
    dist.barrier()
    t0 = time.perf_counter()
    for _ in range(iters):
        dist.all_gather_into_tensor(recv, send)
    torch.cuda.synchronize(dev)
    dt = (time.perf_counter() - t0) / iters
    t = torch.tensor([dt], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())
    return {"bytes_per_rank": nbytes_per_rank, "ms": round(dt * 1e3, 4),
            "busbw_gbps": round(nbytes_per_rank * (world - 1) / dt / 1e9, 2) if world > 1 else None,
            "algbw_gbps": round(nbytes_per_rank * world / dt / 1e9, 2),
            "what": "dist.all_gather_into_tensor of {} MB per rank, {} iterations, max over ranks; busbw = (N-1) x bytes_per_rank / time "
                    "(what every rank receives from its peers)".format(nbytes_per_rank >> 20, iters)}


def run_bench(args, rank, local_rank, world, dev, use_dist):
    import torch.distributed as dist

    # ---- workload of the primary line
    if args.classes is not None and args.classes_total is not None:
        raise SystemExit("give --classes (per GPU, weak scaling) or --classes-total (strong scaling), not both")
    if args.classes is not None:
        classes_total, scaling = args.classes * world, "weak"
    elif args.classes_total is not None:
        classes_total, scaling = args.classes_total, "strong"
    elif world > 1:
        classes_total, scaling = 1024, "strong"          # BASELINE.json configs[2]
    else:
        classes_total, scaling = 64, "weak"              # BASELINE.json configs[1]
    if classes_total < world:
        raise SystemExit("need at least one class per rank")
    one_gpu = None
    if use_dist and not args.no_one_gpu_reference:
        # The N = 1 point of THIS curve, in THIS run: rank 0 alone times all classes_total classes of the same
        # workload on its GPU - the others wait at the barrier - so that the line of an N > 1 run carries its own single-GPU
        # reference and speed-up (the default N = 1 command measures configs[1], 64 classes: another workload).
        if rank == 0:
            w1 = Workload(dev, 0, 1, classes_total, args.variant, args.pyramid, False, "all")
            n1 = max(2, min(args.steps, 10))
            dt1, _ = w1.run(args.precision, n1, max(1, min(args.warmup, 2)))
            one_gpu = {"pairs_per_s": round(classes_total * n1 / dt1, 2), "ms_per_step": round(dt1 / n1 * 1e3, 4), "steps": n1,
                       "what": "all {} classes on rank 0's GPU alone, same process, before the sharded steps".format(classes_total)}
            del w1
            from os2d_amd.modeling import head as head_mod
            head_mod.release_workspaces()
            torch.cuda.empty_cache()
        torch.cuda.synchronize(dev)
        dist.barrier()
    w = Workload(dev, rank, world, classes_total, args.variant, args.pyramid, use_dist, args.gather)

    dt, stage_ms = w.run(args.precision, args.steps, args.warmup)
    value = classes_total * args.steps / dt
    effective = args.precision          # below 7 image-class pairs the fft modes run the direct f16x3 kernel for the 7x7 layer
    if not args.pyramid and getattr(w.head, "last_precision", None):
        effective = w.head.last_precision
    gather_desc = {"all": "loc | cls | corners maps (every rank can decode every class)", "scores": "score maps only",
                   "detections": "local decode + per-class NMS, then the surviving detections"}
    result = {
        "metric": "query-image-pairs/s (1280-px input, ResNet50, N-class)",
        "value": round(value, 2),
        "unit": "query-image-pairs/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 4),
        "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
        "dtype": DTYPE[effective], "data": "synthetic",
        "config": {"workload": ("BASELINE.json configs[{}]: ".format(
                                    4 if args.pyramid else (2 if classes_total == 1024 and args.variant == "v2" else
                                                            (1 if classes_total == 64 and world == 1 and args.variant == "v2" else "-")))
                                + w.describe()),
                   "classes_per_gpu": w.B_local if scaling == "weak" else [e - s for s, e in w.bounds],
                   "classes_total": classes_total, "feature_map": [C_FEAT, H_FM, W_FM],
                   "precision": args.precision,
                   "parallelism": ("class-sharded x{} ({} scaling) + RCCL all-gather per step of the {}".format(world, scaling, gather_desc[args.gather])
                                   if use_dist else "single GPU")},
    }
    if one_gpu is not None:
        result["one_gpu_same_workload"] = one_gpu
        result["speedup_vs_one_gpu"] = round(value / one_gpu["pairs_per_s"], 3)
    if stage_ms:
        result["stages_ms"] = {k: round(v, 4) for k, v in zip(STAGES, stage_ms)}
    if stage_ms:
        # `roofline` = the LONGEST kernel of the step by its live stage time; every other kernel of the step
        # under `roofline_other`
        per_kernel = w.rooflines(stage_ms, effective)
        longest = max(per_kernel, key=lambda k: per_kernel[k]["avg_launch_ms"])
        result["roofline"] = dict(per_kernel[longest], stage=longest,
                                  selected_as="the longest kernel of the step by HIP-event stage time in the timed steps")
        result["roofline_other"] = {k: v for k, v in per_kernel.items() if k != longest}
    else:
        result["roofline"] = w.roofline(args.precision, stage_ms, dt / args.steps)
    result["head_tflops_algorithmic"] = round(w.whole_head_flops_per_class() * value / 1e12, 3)
    if not args.no_other_precision:
        result["other_precisions"] = []
        for other in ("fft", "fftx3", "f16x3", "f16x2", "fft32", "f32"):
            if other == args.precision:
                continue
            dt2, stage2 = w.run(other, max(2, min(args.steps, 10)), 1)
            n2 = max(2, min(args.steps, 10))
            o = {"precision": other, "value": round(classes_total * n2 / dt2, 2), "ms_per_step": round(dt2 / n2 * 1e3, 4), "steps": n2}
            if stage2:
                o["stages_ms"] = {k: round(v, 4) for k, v in zip(STAGES, stage2)}
            o["roofline"] = w.roofline(other, stage2, dt2 / n2)
            result["other_precisions"].append(o)
            if other == "fft32" and stage2 and not args.pyramid:
                # the strict reading of the precision claim (no fp16 value anywhere) with a complete record of its own: the longest
                # kernel of the fft32 step against ITS roof - 157.3 TFLOP/s of v_mfma_f32_32x32x2_f32, or 8 TB/s
                pk = w.rooflines(stage2, "fft32")
                lk = max(pk, key=lambda k: pk[k]["avg_launch_ms"])
                result["roofline_strict_fp32"] = {"precision": "fft32", "pairs_per_s": o["value"], "ms_per_step": o["ms_per_step"], "stage": lk,
                                                  "kernel": pk[lk].get("pmc_kernel"), "bound": pk[lk]["bound"], "achieved": pk[lk]["achieved"],
                                                  "peak": pk[lk]["peak"], "unit": pk[lk]["unit"], "frac": pk[lk]["frac"],
                                                  "avg_launch_ms": pk[lk]["avg_launch_ms"]}
        if not args.pyramid:
            result["max_abs_diff_vs_f32"] = precision_deviation(w)
    if use_dist:
        result["config"]["classes_per_gpu"] = [e - s for s, e in w.bounds]          # per rank, whatever the scaling mode
        result["config"]["dist_timeout_s"] = float(os.environ.get("OS2D_DIST_TIMEOUT_S", "180"))
        if w.gather_wait_ms is not None:
            # how long the compute stream actually stood still for the collectives of a step (HIP events around the wait that
            # makes the step's stream depend on RCCL's; 0 = the gather had finished behind the next step's kernels)
            t = torch.tensor([w.gather_wait_ms], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            result["gather_wait_ms"] = round(float(t.item()), 4)
        result["allgather_probe"] = allgather_probe(dev, world)
    if use_dist and not args.no_other_gather:
        result["other_gathers"] = []
        for g in ("all", "scores", "detections"):
            if g == args.gather:
                continue
            wg = w.set_gather(g)
            n2 = max(2, min(args.steps, 10))
            dt2, _ = wg.run(args.precision, n2, 1)
            result["other_gathers"].append({"gather": g, "what": gather_desc[g], "value": round(classes_total * n2 / dt2, 2),
                                            "ms_per_step": round(dt2 / n2 * 1e3, 4), "steps": n2})
    single = rank == 0 and world == 1 and not args.force_dist
    if single and not args.no_sweep and not args.pyramid:
        # the other single-GPU readings of BASELINE.json's configs, timed in this very run
        n = max(2, min(args.steps, 5))
        result["sweep"] = []
        for name, classes, variant, pyr in (("configs[3]: 256 classes, V1 simplified-affine head", 256, "v1", False),
                                            ("configs[2] on one GPU: 1024 classes", 1024, "v2", False),
                                            ("configs[4] per-GPU share: 7-level pyramid, 128 classes, levels back to back (per-level streams timed alongside)", 128, "v2", True)):
            e, ws = sweep_entry(dev, name, classes, variant, pyr, args.precision, n, 1)
            result["sweep"].append(e)
            if name.startswith("configs[2]"):
                result["scaling_reference"] = {"what": "all 1024 classes of configs[2] on ONE GPU (the N=1 point of the strong-scaling curve "
                                                       "bench.py --gpus N measures)", "value": e["value"], "ms_per_step": e["ms_per_step"]}
            del ws
            from os2d_amd.modeling import head as head_mod
            head_mod.release_workspaces()
            torch.cuda.empty_cache()
    if single and not args.no_cpu_baseline and not args.pyramid:
        result["cpu_baseline"] = cpu_baseline(w.fm_cpu, w.class_fms_cpu, w.state, w.inverse, args.cpu_seconds)
        result["speedup_vs_cpu_baseline"] = round(result["value"] / result["cpu_baseline"]["value"], 1)
    if single and not args.no_end_to_end and not args.pyramid:
        result["end_to_end"] = end_to_end(dev, min(classes_total, 256), w.P, w.inverse, w.state, args.precision)
        if not args.no_sweep:
            # configs[3] names a ResNet101 backbone: the same leg with it and the V1 head at 256 classes
            from os2d_amd.utils import synthetic
            result["end_to_end_resnet101_v1_256"] = end_to_end(dev, 256, 4, False, synthetic.make_transform_net_state(4, seed=1),
                                                               args.precision, arch="resnet101", steps=3, warmup=1)
    if single and not args.no_live_counters and not args.pyramid:
        # HBM bytes / matrix-pipe counters of THIS workload measured now (rocprofv3 PMC passes over a short child run), merged
        # into the roofline objects in place of the replayed figures
        lc = live_counters(args.precision, classes_total, keep_dir=os.path.join(REPO, "gpurun_out", "live_counters")
                           if os.path.isdir(os.path.join(REPO, "gpurun_out")) else None)
        result["live_counters"] = lc
        objs = [result.get("roofline")] + list(result.get("roofline_other", {}).values())
        for r in objs:
            kname = (r or {}).get("pmc_kernel")
            k = lc.get("kernels", {}).get(kname)
            if not (k and r and "hbm_bytes_per_launch" in k):
                continue
            seconds = r["avg_launch_ms"] * 1e-3
            r["traffic"] = k["hbm_bytes_per_launch"]
            r["hbm_gbps"] = round(k["hbm_bytes_per_launch"] / seconds / 1e9, 1)
            for c in ("mfma_pipe_busy", "effective_clock_ghz"):
                if c in k:
                    r[c] = k[c]
            r["counters_source"] = ("LIVE: rocprofv3 PMC passes of this run over a 3-step child run of the same workload ({} launches of {}; "
                                    "FETCH_SIZE x2 (gfx950 correction) + WRITE_SIZE, separate passes)".format(k.get("launches"), kname))
    if single and not args.no_live_counters and not args.no_sweep and not args.pyramid:
        # the same two traffic passes over the 1024-class workload (the correlation's operand re-fetch at scale)
        lc = live_counters(args.precision, 1024, keep_dir=os.path.join(REPO, "gpurun_out", "live_counters")
                           if os.path.isdir(os.path.join(REPO, "gpurun_out")) else None, passes=PMC_PASSES[:2], tag="_1024", timeout_s=240)
        result["live_counters_1024"] = lc
        for e in result.get("sweep", []):
            if e["name"].startswith("configs[2]"):
                for r in [e.get("roofline")] + list(e.get("roofline_other", {}).values()):
                    k = lc.get("kernels", {}).get((r or {}).get("pmc_kernel"))
                    if k and "hbm_bytes_per_launch" in k:
                        r["traffic"] = k["hbm_bytes_per_launch"]
    if "other_precisions" in result:
        by_mode = {o["precision"]: o["value"] for o in result["other_precisions"]}
        if "f32" in by_mode:
            result["f32_pairs_per_s"] = by_mode["f32"]        # strictly fp32, the reference's own arithmetic (direct kernels on v_mfma_f32_32x32x2_f32), same run
        if "fft32" in by_mode:
            result["fft32_pairs_per_s"] = by_mode["fft32"]    # strictly fp32 with the 7x7 layer in the frequency domain (no fp16 value anywhere)
        # ... and inside `config`, which survives the driver's trimming of the record
        result["config"]["same_run_pairs_per_s"] = {"f32 (strictly fp32, direct kernels: the reference's arithmetic)": by_mode.get("f32"),
                                                    "fft32 (strictly fp32, 7x7 layer in the frequency domain)": by_mode.get("fft32")}
    for e in result.get("sweep", []):
        if e["name"].startswith("configs[2]"):
            allk = dict(e.get("roofline_other", {}))
            allk[e["roofline"].get("stage", "?")] = e["roofline"]
            result["config"]["classes_1024_one_gpu"] = {"pairs_per_s": e["value"], "ms_per_step": e["ms_per_step"],
                                                        "longest_kernel": {k: e["roofline"].get(k) for k in ("stage", "bound", "frac", "avg_launch_ms", "traffic", "algorithmic_bytes")},
                                                        "layer7x7_ms": {k: v["avg_launch_ms"] for k, v in allk.items()
                                                                        if k in ("fft_forward", "spectral_gemm", "fft_inverse")},
                                                        "layer7x7_hbm_frac": {k: v["frac"] for k, v in allk.items()
                                                                              if k in ("fft_forward", "spectral_gemm", "fft_inverse")}}
        elif e["name"].startswith("configs[3]"):
            result["config"]["classes_256_v1"] = {"pairs_per_s": e["value"], "ms_per_step": e["ms_per_step"]}
        elif e["name"].startswith("configs[4]"):
            result["config"]["pyramid_7_levels_128_classes"] = {"pairs_per_s": e["value"], "streams_ms": e.get("pyramid_streams_ms"),
                                                                "serial_ms": e.get("pyramid_serial_ms")}
    if rank == 0:
        emit(result)


LINE_BUDGET = 6000      # bytes of the ONE stdout line (the 22 KB line of round 4 was not parsed by the driver)
CONTRACT_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                 "dtype", "data", "config", "roofline", "cpu_baseline")
ROOFLINE_KEYS = ("kernel", "stage", "bound", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_ms", "flops_per_launch",
                 "algorithmic_bytes", "executed_frac_of_peak", "hbm_gbps", "mfma_pipe_busy", "effective_clock_ghz", "counters")


def _finite(o):
    """Strict JSON: non-finite floats become null (json.loads of the driver must never meet NaN / Infinity)."""
    if isinstance(o, float):
        return o if o == o and o not in (float("inf"), float("-inf")) else None
    if isinstance(o, dict):
        return {str(k): _finite(v) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return [_finite(v) for v in o]
    return o


def compact_line(result, budget=LINE_BUDGET):
    """The ONE line of stdout: the driver's contract keys, `roofline` (longest kernel, with its live `traffic`), `cpu_baseline`,
    `roofline_other` as {kernel: [frac, avg_launch_ms]} and a few scalars - a few KB.  Everything else of `result` (the other
    arithmetic modes, the sweep entries with their own roofline objects, the raw live counters, the end-to-end legs) goes to
    bench_details.json and to stderr (emit()).  Optional keys are dropped, last first, should the line ever exceed `budget`."""
    result = _finite(result)
    line = {k: result[k] for k in CONTRACT_KEYS if k in result}
    if isinstance(line.get("roofline"), dict):
        r = dict(result["roofline"])
        src = r.get("counters_source")
        if src:
            r["counters"] = "live rocprofv3 PMC passes (FETCH_SIZE x2 + WRITE_SIZE, separate passes)" if src.startswith("LIVE") else "replayed (profiles/)"
        line["roofline"] = {k: r[k] for k in ROOFLINE_KEYS if k in r}
        line["roofline"].setdefault("traffic", None)
    if isinstance(line.get("cpu_baseline"), dict) and len(line["cpu_baseline"].get("sample", "")) > 240:
        line["cpu_baseline"] = dict(line["cpu_baseline"], sample=line["cpu_baseline"]["sample"][:237] + "...")
    optional = []
    if "stages_ms" in result:
        optional.append(("stages_ms", result["stages_ms"]))
    if isinstance(result.get("roofline_other"), dict):
        optional.append(("roofline_other", {k: [v.get("frac"), v.get("avg_launch_ms")] for k, v in result["roofline_other"].items()}))
        optional.append(("roofline_other_is", "{kernel: [frac of its bound (see bench_details.json), avg launch ms]}"))
    for k in ("one_gpu_same_workload", "speedup_vs_one_gpu", "roofline_strict_fp32"):      # (first = dropped last)
        if k in result:
            optional.append((k, {f: v for f, v in result[k].items() if f != "what"} if isinstance(result[k], dict) else result[k]))
    for k in ("speedup_vs_cpu_baseline", "head_tflops_algorithmic", "f32_pairs_per_s", "fft32_pairs_per_s", "gather_wait_ms"):
        if k in result:
            optional.append((k, result[k]))
    if isinstance(result.get("end_to_end"), dict):
        optional.append(("end_to_end", {k: result["end_to_end"].get(k) for k in ("value", "ms_per_image", "backbone_ms", "head_ms", "decode_nms_ms")}))
    if isinstance(result.get("allgather_probe"), dict):
        optional.append(("allgather_probe", {k: result["allgather_probe"].get(k) for k in ("bytes_per_rank", "ms", "busbw_gbps", "algbw_gbps")}))
    if isinstance(result.get("other_gathers"), list):
        optional.append(("other_gathers", {g.get("gather"): g.get("value") for g in result["other_gathers"]}))
    optional.append(("details", "bench_details.json (same directory; also on stderr)"))
    for k, v in optional:
        line[k] = v
    text = json.dumps(line, allow_nan=False, separators=(",", ":"))
    while len(text) > budget and optional:
        line.pop(optional.pop()[0], None)
        text = json.dumps(line, allow_nan=False, separators=(",", ":"))
    if len(text) > budget and isinstance(line.get("config"), dict):      # last resort: config down to its workload string
        line["config"] = {"workload": line["config"].get("workload")}
        text = json.dumps(line, allow_nan=False, separators=(",", ":"))
    return text


def emit(result):
    """Rank 0: the full record to bench_details.json (repo root; and gpurun_out/ when present, which is what comes back from
    the GPU box) and to stderr, the compact line - alone - to stdout."""
    full = json.dumps(_finite(result), allow_nan=False, indent=1)
    for d in (REPO, os.path.join(REPO, "gpurun_out")):
        if os.path.isdir(d):
            try:
                with open(os.path.join(d, "bench_details.json"), "w") as f:
                    f.write(full + "\n")
            except OSError:
                pass
    print("[bench_details] " + json.dumps(_finite(result), allow_nan=False), file=sys.stderr, flush=True)
    print_line(compact_line(result))


if __name__ == "__main__":
    main()
