"""CPU: host-side mirror of the reference API (no device compute): value types, anchors, module layout, state-dict
keys, loud failures."""
import pytest
import torch

from oracle import head_oracle as O
from os2d_amd.modeling.box_coder import BoxGridGenerator, feature_map_size_c4
from os2d_amd.modeling.head import Os2dHeadCreator, build_os2d_head_creator
from os2d_amd.structures.bounding_box import BoxList, cat_boxlist
from os2d_amd.structures.feature_map import FeatureMapSize


def test_feature_map_size_semantics():
    a = FeatureMapSize(w=80, h=60)
    assert (a.w, a.h) == (80, 60) and a == FeatureMapSize(img=torch.zeros(1, 3, 60, 80)) and hash(a) == hash(FeatureMapSize(w=80, h=60))
    with pytest.raises(AttributeError):
        a.w = 3
    with pytest.raises(RuntimeError):
        FeatureMapSize()
    assert repr(a) == "FeatureMapSize(w=80, h=60)"


def test_c4_feature_map_sizes_of_the_pyramid():
    # the 7-scale pyramid of a 1280x960 image
    inp = [(640, 480), (800, 600), (1024, 768), (1280, 960), (1536, 1152), (1792, 1344), (2048, 1536)]
    fms = [(40, 30), (50, 38), (64, 48), (80, 60), (96, 72), (112, 84), (128, 96)]
    for (w, h), (fw, fh) in zip(inp, fms):
        assert feature_map_size_c4(FeatureMapSize(w=w, h=h)) == FeatureMapSize(w=fw, h=fh)


def test_anchor_grid_matches_oracle_and_is_row_major():
    gen = BoxGridGenerator(box_size=FeatureMapSize(w=240, h=240), box_stride=FeatureMapSize(w=16, h=16))
    a = gen.create_strided_boxes_columnfirst(FeatureMapSize(w=5, h=3))
    assert torch.equal(a, O.anchor_grid(3, 5, 240.0, 16.0))
    # index h*W + w: second entry moves along x
    assert a[1, 0] - a[0, 0] == 16 and a[1, 1] == a[0, 1] and a[5, 1] - a[0, 1] == 16


def test_rec_field_and_stride_composition():
    rf, st = Os2dHeadCreator.get_rec_field_and_stride_after_concat_nets(
        FeatureMapSize(w=16, h=16), FeatureMapSize(w=16, h=16), FeatureMapSize(w=15, h=15), FeatureMapSize(w=1, h=1))
    assert rf == FeatureMapSize(w=240, h=240) and st == FeatureMapSize(w=16, h=16)


@pytest.mark.parametrize("simple,P", [(False, 6), (True, 4)])
def test_head_creator_layout_and_identity_init(simple, P):
    creator = build_os2d_head_creator(simple, False, True, FeatureMapSize(w=16, h=16), FeatureMapSize(w=16, h=16))
    keys = set(creator.state_dict().keys())
    for k in ("conv.0.weight", "conv.0.bias", "conv.1.weight", "conv.1.bias", "conv.1.running_mean", "conv.1.running_var",
              "conv.3.weight", "conv.4.running_var", "linear.weight", "linear.bias"):
        assert "aligner.parameter_regressor." + k in keys
    net = creator.aligner.parameter_regressor
    assert tuple(net.conv[0].weight.shape) == (128, 225, 7, 7) and tuple(net.conv[3].weight.shape) == (64, 128, 5, 5)
    assert tuple(net.linear.weight.shape) == (P, 64, 5, 5)
    assert float(net.linear.weight.abs().max()) == 0.0
    expect = [1, 0, 0, 0, 1, 0] if P == 6 else [1, 0, 1, 0]
    assert net.linear.bias.tolist() == expect          # identity transform (reference head.py:632-642)
    assert sum(p.numel() for p in net.parameters()) == (1626182 if P == 6 else 1626182 - 2 * (64 * 25 + 1))
    assert creator.box_grid_generator_image_level.box_size == FeatureMapSize(w=240, h=240)
    assert creator.box_grid_generator_feature_map_level.box_size == FeatureMapSize(w=15, h=15)
    assert creator.aligner.model_type == ("simple_affine" if simple else "affine")


def test_model_state_dict_keys_and_param_count():
    from os2d_amd.modeling.model import Os2dModel
    net = Os2dModel(is_cuda=False, merge_branch_parameters=True, backbone_arch="resnet50")
    sd = net.state_dict()
    assert "net_feature_maps.conv1.weight" in sd and "net_feature_maps.layer3.5.bn3.running_var" in sd
    assert "net_label_features.net_class_features.layer1.0.downsample.0.weight" in sd
    assert "os2d_head_creator.aligner.parameter_regressor.linear.bias" in sd
    assert not any(k.startswith("net_feature_maps.layer4") or ".fc." in k for k in sd)
    # the reference demo logs 10,169,478 parameters (demo.ipynb cell 5) for this configuration
    assert sum(p.numel() for p in net.parameters()) == 10169478
    assert net.get_feature_map_size(FeatureMapSize(w=1280, h=960)) == FeatureMapSize(w=80, h=60)
    assert not net.training


def test_backbone_output_shape_cpu():
    from os2d_amd.modeling.feature_extractor import build_feature_extractor
    net = build_feature_extractor("resnet50").eval()
    with torch.no_grad():
        y = net(torch.zeros(1, 3, 96, 130))
    assert tuple(y.shape) == (1, 1024, 6, 9) and float(y.min()) >= 0
    with pytest.raises(RuntimeError, match="Unknown backbone arch"):
        build_feature_extractor("vgg")
    assert build_feature_extractor("ResNet101").get_num_blocks_in_feature_extractor() == 1 + 3 + 4 + 23


def test_head_refuses_cpu_tensors_without_touching_a_device():
    creator = build_os2d_head_creator(False, False, True, FeatureMapSize(w=16, h=16), FeatureMapSize(w=16, h=16))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        creator.create_os2d_head([torch.zeros(1, 8, 15, 15)])
    with pytest.raises(RuntimeError, match="move the model to the HIP device"):
        creator.aligner.parameter_regressor.packed()
    with pytest.raises(RuntimeError):
        from os2d_amd.modeling.head import TransformationNet
        TransformationNet(kernel_sizes=[3, 3], use_cuda=False)


def test_boxlist_basics():
    b = BoxList(torch.tensor([[0., 0., 10., 10.], [5., 5., 5., 9.], [-4., 2., 30., 50.]]), FeatureMapSize(w=20, h=40))
    b.add_field("scores", torch.tensor([0.1, 0.2, 0.3]))
    assert b.get_mask_empty_boxes().tolist() == [False, True, False]
    c = b.resize(FeatureMapSize(w=40, h=40))
    assert c.bbox_xyxy[0].tolist() == [0, 0, 20, 10] and c.image_size == FeatureMapSize(w=40, h=40)
    b.clip_to_image(remove_empty=False)
    assert b.bbox_xyxy[2].tolist() == [0, 2, 20, 40]
    sel = b[torch.tensor([True, False, True])]
    assert len(sel) == 2 and sel.get_field("scores").tolist() == pytest.approx([0.1, 0.3])
    both = cat_boxlist([sel, sel])
    assert len(both) == 4 and BoxList(torch.tensor([[5., 5., 2., 2.]]), b.image_size, mode="cx_cy_w_h").bbox_xyxy.tolist() == [[4, 4, 6, 6]]


def test_class_image_views_order_and_ids():
    """reference evaluate.py:241-269,294: order of the augmented views and the id each one carries."""
    from os2d_amd.engine.evaluate import class_image_views
    im = torch.arange(3 * 2 * 3, dtype=torch.float32).view(3, 2, 3)
    im2 = -im
    v, ids, n = class_image_views([im, im2], [7, 9], "")
    assert n == 1 and ids == [7, 9] and torch.equal(v[0], im) and torch.equal(v[1], im2)
    v, ids, n = class_image_views([im, im2], [7, 9], "horflip")
    assert n == 2 and ids == [7, 7, 9, 9] and torch.equal(v[1], im.flip(2)) and torch.equal(v[3], im2.flip(2))
    v, ids, n = class_image_views([im.unsqueeze(0)], [4], "rotation90")
    r90 = im.rot90(1, [1, 2])
    assert n == 4 and ids == [4] * 4 and v[1].shape == (3, 3, 2)
    assert torch.equal(v[1], r90) and torch.equal(v[2], r90.rot90(1, [1, 2])) and torch.equal(v[3], im.rot90(3, [1, 2]))
    v, ids, n = class_image_views([im], [4], "horflip_rotation90")
    assert n == 8 and torch.equal(v[4], im.flip(2)) and torch.equal(v[5], r90.flip(2)) and torch.equal(v[7], im.rot90(3, [1, 2]).flip(2))
    assert all(ids[l] == [4][l // n] for l in range(8))
    with pytest.raises(RuntimeError):
        class_image_views([im], [4], "rot45")


def test_bench_refuses_to_run_without_a_device():
    """No CPU fallback anywhere near the measured path: without a HIP device bench.py stops with a clear message."""
    import os
    import subprocess
    import sys
    if torch.cuda.is_available():
        pytest.skip("a HIP device is present")
    repo = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    out = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--steps", "1"], cwd=repo, capture_output=True,
                         text=True, timeout=300)
    assert out.returncode != 0 and "HIP device" in (out.stderr + out.stdout)


def test_dataloader_style_inverse_transforms_are_traced_exactly():
    """the reference hands decode_pyramid a TransformList of closures per level.  On the fixture the
    reference's own dataloader functions produced (flips + mined crop + two resizes, tests/golden/make_golden.py::
    make_decode_transforms_fixture): (i) this repo's BoxList runs the closures to the reference's coordinates bit for bit,
    (ii) ``trace_box_transform`` recovers the chain of 5 operations, and (iii) applying the recorded chain reproduces the
    closures' result bit for bit - so the fused decode kernels can replace the closures."""
    import numpy as np
    import util
    from os2d_amd.modeling.box_coder import OP_HFLIP, OP_SCALE, OP_SHIFT, OP_VFLIP, ResizeBoxes, apply_box_ops, trace_box_transform
    d = np.load(util.GOLDEN + "/decode_transforms.npz")
    orig = FeatureMapSize(w=int(d["orig_size"][0]), h=int(d["orig_size"][1]))
    probe = torch.from_numpy(d["probe"])
    for l in range(int(d["n_levels"])):
        size = FeatureMapSize(w=int(d["img_sizes"][l][0]), h=int(d["img_sizes"][l][1]))
        t = util.dataloader_style_inverse(d["chain"][l])
        out = t(BoxList(probe.clone(), size))
        assert out.image_size == orig
        assert torch.equal(out.bbox_xyxy, torch.from_numpy(d["probe_out_%d" % l]))
        ops, anchor_ops, out_size = trace_box_transform(t, size)
        assert out_size == orig and [o[0] for o in ops] == [OP_SCALE, OP_SCALE, OP_SHIFT, OP_VFLIP, OP_HFLIP]
        assert torch.equal(apply_box_ops(probe, ops), torch.from_numpy(d["probe_out_%d" % l]))
        # the anchors: as a BoxList field of the boxes they are cropped / flipped (not resized) WITH the boxes, then the whole
        # transform is applied to them again (the reference's behaviour, reproduced): 3 + 5 operations
        assert [o[0] for o in anchor_ops] == [OP_SHIFT, OP_VFLIP, OP_HFLIP, OP_SCALE, OP_SCALE, OP_SHIFT, OP_VFLIP, OP_HFLIP]
        assert torch.equal(apply_box_ops(probe + 0.5, anchor_ops), torch.from_numpy(d["probe_default_out_%d" % l]))
        from os2d_amd.modeling.box_coder import transform_level_boxes
        b, dflt, c, sz = transform_level_boxes(t, probe.clone(), probe.clone() + 0.5, probe.clone(), size)
        assert sz == orig and torch.equal(b, torch.from_numpy(d["probe_out_%d" % l])) and torch.equal(c, b)
        assert torch.equal(dflt, torch.from_numpy(d["probe_default_out_%d" % l]))
    # identity / ResizeBoxes / plain lambdas trace too; anything else than the three BoxList operations does not
    size = FeatureMapSize(w=320, h=272)
    assert trace_box_transform(None, size) == ((), (), size)
    ops, anchor_ops, out_size = trace_box_transform(ResizeBoxes(orig), size)
    assert out_size == orig and ops == ((OP_SCALE, 500.0 / 320, 380.0 / 272),) and anchor_ops == ops     # resize-only: one chain
    assert trace_box_transform(lambda b: b.resize(orig), size)[0] == ops
    assert trace_box_transform(lambda b: BoxList(b.bbox_xyxy + 1.0, b.image_size), size) is None      # touches coordinates
    assert trace_box_transform(lambda b: b.resize(orig).clip_to_image(), size) is None                # an untraced method
    long_chain = util.InverseTransformList()
    for _ in range(7):
        long_chain.append(lambda b: b.resize(orig))
    assert trace_box_transform(long_chain, size) is None                                               # more than 6 operations


def test_trace_cache_sees_a_transform_list_that_grew_after_it_was_traced():
    """the reference's TransformList keeps its closures in ``_transforms`` (structures/transforms.py:18-22); a list that is
    appended to after its first trace must trace to the longer chain (round 5's identity-keyed cache returned the stale chain), and so
    must a closure whose captured state changed.  The cache keeps nothing alive."""
    import gc
    import weakref
    import util
    from os2d_amd.modeling import box_coder as bc
    size, orig = FeatureMapSize(w=320, h=272), FeatureMapSize(w=500, h=380)
    chain = util.InverseTransformList()
    assert not hasattr(chain, "transforms")            # the list lives in an attribute the round-5 key did not look at
    chain.append(lambda b: b.resize(orig))
    first = bc.trace_box_transform(chain, size)
    assert bc.trace_box_transform(chain, size) == first and len(first[0]) == 1
    chain.append(lambda b: b.transpose(bc.FLIP_LEFT_RIGHT))
    second = bc.trace_box_transform(chain, size)
    assert second == bc._trace_box_transform(chain, size) and [o[0] for o in second[0]] == [bc.OP_HFLIP, bc.OP_SCALE]      # last appended runs first
    state = {"target": orig}
    closure = lambda b: b.resize(state["target"])      # noqa: E731
    one = bc.trace_box_transform(closure, size)
    state["target"] = FeatureMapSize(w=250, h=190)
    two = bc.trace_box_transform(closure, size)
    assert one[2] == orig and two[2] == state["target"] and two == bc._trace_box_transform(closure, size)
    # probe checks are cached per traced chain (a second call does not run the CPU probe) ...
    calls = []
    real = bc._probe_check
    bc._probe_check = lambda *a: calls.append(1) or real(*a)
    try:
        assert bc.trace_box_transform(chain, size) == second and not calls
    finally:
        bc._probe_check = real
    # ... and the cache holds weak references only
    ref = weakref.ref(chain)
    del chain
    gc.collect()
    assert ref() is None


def test_fixed_point_norm_sums_of_the_correlation_kernels():
    """The arithmetic of corr_f16x3.hip's per-position sums of relu(corr)^2, restated in numpy (float32 operations as the kernel
    does them): a run of 4 rows is added in fp32, the run's sum g goes to 2^-44 fixed point as two integers
    (hi = trunc(g * 2^20), lo = trunc((g * 2^20 - hi) * 2^24)), integers are added from there on and the total becomes
    float32(total * 2^-44).  Checked here: the conversion is exact for g >= 2^-20 and drops less than 2^-44 below; a class's 57 runs
    cannot overflow the two 32-bit accumulators; any order and any split of the runs over lanes / waves / work-groups gives
    the same integer; the result is the correctly rounded float32 of the exact sum of the run sums."""
    import numpy as np
    rng = np.random.RandomState(7)
    f32 = np.float32

    def convert(g):
        ga = np.minimum(g, f32(1024.0)) * f32(1048576.0)
        hi = np.floor(ga).astype(np.uint32)                        # v_cvt_u32_f32 truncates (ga >= 0)
        lo = np.floor((ga - hi.astype(np.float32)) * f32(16777216.0)).astype(np.uint32)
        return hi, lo

    for scale in (1.0, 1e-2, 1e-5, 1e-9):
        rows = (rng.rand(228, 64).astype(np.float32) * f32(scale))                     # 57 runs of 4 rows, 64 positions
        rows[225:] = 0                                                                  # the padding rows of the stride
        sq = rows * rows                                                                # relu(v)^2, v >= 0 here
        g = np.zeros((57, 64), np.float32)
        for k in range(4):                                                              # g = fma chain over the run, from 0
            g = (sq.reshape(57, 4, 64)[:, k].astype(np.float64) + g.astype(np.float64)).astype(np.float32)
        hi, lo = convert(g)
        exact = g.astype(np.float64) * 2.0 ** 44
        fixed = hi.astype(np.float64) * 2.0 ** 24 + lo.astype(np.float64)
        assert np.all(fixed <= exact) and np.all(exact - fixed < 1.0)                   # < 2^-44 dropped per run
        big = g >= f32(2.0 ** -20)
        assert np.all(fixed[big] == exact[big])                                         # exact from 2^-20 up
        # no overflow of the 32-bit accumulators over a class's 57 runs, even at the largest values (g <= 4)
        assert int(hi.astype(np.uint64).sum(axis=0).max()) < 2 ** 32 and int(lo.astype(np.uint64).sum(axis=0).max()) < 2 ** 32
        assert 57 * (4 << 20) < 2 ** 32 and 57 * (2 ** 24 - 1) < 2 ** 32
        total = (hi.astype(np.uint64) << np.uint64(24)).sum(axis=0) + lo.astype(np.uint64).sum(axis=0)
        # any grouping / order of the runs: integer addition
        perm = rng.permutation(57)
        cut = 23
        part_a = (hi[perm[:cut]].astype(np.uint64) << np.uint64(24)).sum(axis=0) + lo[perm[:cut]].astype(np.uint64).sum(axis=0)
        part_b = (hi[perm[cut:]].astype(np.uint64) << np.uint64(24)).sum(axis=0) + lo[perm[cut:]].astype(np.uint64).sum(axis=0)
        assert np.array_equal(part_a + part_b, total)
        assert int(total.max()) < 2 ** 53                                               # exact as a double
        s = (total.astype(np.float64) * 2.0 ** -44).astype(np.float32)                  # one rounding
        want = fixed.sum(axis=0) * 2.0 ** -44
        assert np.array_equal(s, want.astype(np.float32))
        if scale >= 1e-2:                                                               # and that is the sum of the run sums
            ref = g.astype(np.float64).sum(axis=0)
            assert np.all(np.abs(s.astype(np.float64) - ref) <= np.abs(ref) * 2.0 ** -23)
    # a non-finite or absurd run sum is flagged, not converted
    assert not (f32(np.nan) < f32(1024.0)) and not (f32(np.inf) < f32(1024.0)) and (f32(4.0) < f32(1024.0))


def test_bench_compact_line_fits_the_driver():
    """bench.py's stdout line from a full record (round 4's 22 KB line, which the driver could not parse): at most 6000
    bytes, strict JSON, contract keys intact, optional keys dropped before contract ones under a smaller budget."""
    import json
    import os
    import bench
    REPO = os.path.dirname(os.path.abspath(bench.__file__))
    with open(os.path.join(REPO, "profiles", "r04_w_bench.json")) as f:
        full = json.load(f)
    assert len(json.dumps(full)) > 20000
    line = bench.compact_line(full)
    assert len(line) <= bench.LINE_BUDGET and "\n" not in line
    d = json.loads(line)
    for k in bench.CONTRACT_KEYS:
        assert k in d, k
    assert d["value"] == full["value"] and d["roofline"]["frac"] == full["roofline"]["frac"]
    assert d["roofline"]["traffic"] == full["roofline"]["traffic"] and d["cpu_baseline"]["value"] == full["cpu_baseline"]["value"]
    assert d["roofline_other"]["conv2"] == [full["roofline_other"]["conv2"]["frac"], full["roofline_other"]["conv2"]["avg_launch_ms"]]
    for k in ("other_precisions", "sweep", "live_counters"):
        assert k not in d
    small = json.loads(bench.compact_line(full, budget=2600))
    assert all(k in small for k in bench.CONTRACT_KEYS) and "end_to_end" not in small
    bad = dict(full, value=float("nan"), roofline=dict(full["roofline"], hbm_gbps=float("inf")))
    d = json.loads(bench.compact_line(bad), parse_constant=lambda c: (_ for _ in ()).throw(ValueError(c)))
    assert d["value"] is None and d["roofline"]["hbm_gbps"] is None


def test_bench_compact_line_of_a_multi_rank_record():
    """The N > 1 record (no sweep / cpu_baseline / stage events; all-gather probe, gather wait, the other exchanges) through the
    same compaction: contract keys first, the diagnosis fields of a first real multi-GPU run kept, still one short line."""
    import json
    import bench
    rec = {"metric": "query-image-pairs/s (1280-px input, ResNet50, N-class)", "value": 400000.0, "unit": "query-image-pairs/s", "n_gpus": 8,
           "steps": 20, "warmup": 5, "ms_per_step": 2.56, "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "fftx3 (...)",
           "data": "synthetic", "config": {"workload": "BASELINE.json configs[2]: ...", "classes_per_gpu": [128] * 8, "classes_total": 1024,
                                           "feature_map": [1024, 60, 80], "precision": "fftx3", "parallelism": "class-sharded x8 ...", "dist_timeout_s": 180.0},
           "roofline": {"kernel": "whole head of one rank: ...", "bound": "mfma", "achieved": 900.0, "peak": 2500.0, "unit": "TFLOP/s", "frac": 0.36,
                        "traffic": None, "flops_per_launch": 1, "avg_launch_ms": 2.56, "timing": "wall clock", "peak_is": "x" * 300},
           "head_tflops_algorithmic": 7000.0, "gather_wait_ms": 0.01,
           "allgather_probe": {"bytes_per_rank": 32 << 20, "ms": 0.9, "busbw_gbps": 260.0, "algbw_gbps": 298.0, "what": "y" * 400},
           "other_gathers": [{"gather": "scores", "what": "score maps only", "value": 410000.0, "ms_per_step": 2.5, "steps": 10},
                             {"gather": "detections", "what": "...", "value": 380000.0, "ms_per_step": 2.7, "steps": 10}]}
    line = bench.compact_line(rec)
    d = json.loads(line)
    assert len(line) < 2000 and list(d)[:len(bench.CONTRACT_KEYS) - 1] == [k for k in bench.CONTRACT_KEYS if k in rec]
    assert "cpu_baseline" not in d and d["roofline"]["traffic"] is None and "peak_is" not in d["roofline"]
    assert d["allgather_probe"] == {"bytes_per_rank": 32 << 20, "ms": 0.9, "busbw_gbps": 260.0, "algbw_gbps": 298.0}
    assert d["other_gathers"] == {"scores": 410000.0, "detections": 380000.0} and d["gather_wait_ms"] == 0.01
    assert d["config"]["classes_per_gpu"] == [128] * 8


def test_fixture_pins_cover_every_head_fixture_and_fp32_equivalent_mode():
    """tests/golden/head_fixture_pins.json (measured on an MI355X, regenerated by the golden GPU test under
    $OS2D_WRITE_FIXTURE_PINS) holds [cls, loc, corners] error ratios for every head fixture in every fp32-equivalent mode, all of them
    far inside the generic tolerance - so the per-fixture assertion of tests/test_head_gpu.py has a pin wherever it looks one up."""
    import util
    pins = util.fixture_pins()
    assert sorted(pins) == util.head_fixture_names()
    for name, modes in pins.items():
        assert sorted(modes) == sorted(util.FP32_EQUIVALENT), name
        for mode, ratios in modes.items():
            assert len(ratios) == 3 and all(0.0 <= r <= 0.25 for r in ratios), (name, mode, ratios)
