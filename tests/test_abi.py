"""CPU: the C-ABI library builds, loads and exports every symbol include/os2d_hip.h declares (no compute calls)."""
import ctypes
import os
import re
import subprocess

import pytest

REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
HEADER = os.path.join(REPO, "include", "os2d_hip.h")


def declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(os2d_[a-z0-9_]+)\s*\(", text)))


@pytest.fixture(scope="module")
def lib_path():
    from os2d_amd import build
    return build.build(verbose=False)


def test_header_and_binding_agree():
    from os2d_amd import _lib
    assert declared_functions() == sorted(_lib.SIGNATURES), "include/os2d_hip.h and os2d_amd/_lib.py list different entry points"


def test_library_exports_every_declared_symbol(lib_path):
    out = subprocess.check_output(["nm", "-D", "--defined-only", lib_path]).decode()
    exported = set(re.findall(r" T (os2d_[a-z0-9_]+)", out))
    missing = [f for f in declared_functions() if f not in exported]
    assert not missing, "not exported: {}".format(missing)


def test_library_loads_and_reports_abi(lib_path):
    from os2d_amd import _lib
    lib = _lib.load()
    assert lib.os2d_abi_version() == _lib.ABI_VERSION
    # pure host-side helpers (no device needed)
    assert lib.os2d_packed_conv_floats(1) == 113 * 49 * 2 * 128
    assert lib.os2d_packed_conv_floats(2) == 64 * 25 * 2 * 64
    assert lib.os2d_packed_conv_floats(3) == 32 * 25 * 2 * 32
    assert lib.os2d_packed_bias_floats(1) == 3 * 128      # folded bias | 2^-weight_exp | 2^out_exp per row
    assert lib.os2d_plane_floats(60, 80) % 64 == 0 and lib.os2d_plane_floats(60, 80) >= 63 * 83
    n = ctypes.c_size_t()
    assert lib.os2d_head_workspace_bytes(1, 64, 1024, 60, 80, 6, ctypes.byref(n)) == 0 and n.value > 0
    one = ctypes.c_size_t()
    assert lib.os2d_head_workspace_bytes(1, 1, 1024, 60, 80, 6, ctypes.byref(one)) == 0 and one.value < n.value
    # argument errors are reported through return codes + os2d_last_error, never exceptions
    assert lib.os2d_head_workspace_bytes(1, 1, 0, 60, 80, 6, ctypes.byref(n)) == -1 and b"C" in lib.os2d_last_error()
    assert lib.os2d_head_workspace_bytes(1, 1, 1023, 60, 80, 6, ctypes.byref(n)) == 0      # any channel count (round 6; the reference has no constraint)
    assert lib.os2d_head_workspace_bytes(1, 1, 1024, 60, 80, 5, ctypes.byref(n)) == -1
    # widest map: 3600 columns (the transform planner's 48 tiles per axis); the direct 7x7 kernels stop at 209 - beyond that the head
    # runs the layer in the frequency domain (tiled) whatever the batch - and the 5x5 kernels' linear slabs at 316: column strips
    assert lib.os2d_head_workspace_bytes(1, 1, 1024, 60, 316, 6, ctypes.byref(n)) == 0
    assert lib.os2d_head_workspace_bytes(1, 1, 1024, 60, 640, 6, ctypes.byref(n)) == 0
    assert lib.os2d_head_workspace_bytes(1, 1, 1024, 4, 3600, 6, ctypes.byref(n)) == 0
    assert lib.os2d_head_workspace_bytes(1, 1, 1024, 4, 3601, 6, ctypes.byref(n)) == -1 and b"width" in lib.os2d_last_error()
    # tallest map: 2784 rows (48 tiles of 58 rows); one more fails in the argument check, and both frequency-domain planners have a
    # plan for every height up to the limit at the benchmark's width (a taller map used to fail inside the planner)
    assert lib.os2d_head_workspace_bytes(1, 1, 1024, 2784, 4, 6, ctypes.byref(n)) == 0
    assert lib.os2d_head_workspace_bytes(1, 1, 1024, 2785, 4, 6, ctypes.byref(n)) == -1 and b"height" in lib.os2d_last_error()
    for prec in (3, 4, 5):       # OS2D_PRECISION_FFT, FFTX3, FFT32
        for h in (1, 61, 64, 65, 117, 1000, 2784):
            assert lib.os2d_head_workspace_bytes_ex(1, 1, 64, h, 80, 6, prec, ctypes.byref(n)) == 0, (prec, h, lib.os2d_last_error())


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from os2d_amd import _lib
    monkeypatch.setattr(_lib, "_LIB", None)
    monkeypatch.setenv("OS2D_HIP_LIB", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.Os2dLibraryError, match="no CPU or PyTorch fallback"):
        _lib.load()


def test_packed_fp32_setting_is_per_translation_unit():
    """docs/DESIGN_HISTORY_r1-r3.md section 8: v_pk_*_f32 results were wrong in 16-lane groups next to MFMA-heavy kernels of other streams.
    The switch is per translation unit (build.NO_PACKED_FP32); the transforms - the kernels it was found in - are always in."""
    from os2d_amd import build
    assert "fft.hip" in build.NO_PACKED_FP32
    assert "-packed-fp32-ops" in build.flags_for("fft.hip")
    for s in build.SOURCES:
        assert ("-packed-fp32-ops" in build.flags_for(s)) == (s in build.NO_PACKED_FP32)
    assert "-packed-fp32-ops" not in build.flags_for("fft.hip", packed="on")
    assert "-packed-fp32-ops" in build.flags_for("nms.hip", packed="off")
    assert "-packed-fp32-ops" not in build.flags_for("nms.hip", packed="fft")


def test_compiler_flags_are_part_of_the_build_stamp(monkeypatch):
    """Object files carry no record of their flags: a flag change must invalidate the stamp (and with it every object) -
    the first -packed-fp32-ops build left the untouched sources compiled the old way (docs/DESIGN_HISTORY_r1-r3.md section 8)."""
    from os2d_amd import build
    h0 = build.source_hash()
    monkeypatch.setattr(build, "FLAGS", build.FLAGS + ["-DOS2D_SOME_EXPERIMENT"])
    h1 = build.source_hash()
    assert h1 != h0
    monkeypatch.setattr(build, "NO_PACKED_FP32", set(build.NO_PACKED_FP32) - {"nms.hip"})      # a per-unit flag counts too
    assert build.source_hash() != h1


def test_every_included_header_is_part_of_the_build_stamp(tmp_path, monkeypatch):
    """fft_regs.h (all the register DFTs) was in neither the hash nor the object dependencies, so editing it left a
    stale library in use.  Headers are globbed now; every quoted #include of every source must resolve to one of them, and a
    header that appears later changes the hash."""
    import os
    from os2d_amd import build
    names = {os.path.basename(h) for h in build.headers()}
    assert "fft_regs.h" in names and "os2d_common.h" in names and "os2d_hip.h" in names
    for path in [os.path.join(build.CSRC, s) for s in build.SOURCES] + build.headers():
        for inc in build.local_includes(path):
            assert os.path.basename(inc) in names, (path, inc)
    h0 = build.source_hash()
    extra = tmp_path / "new_header.h"
    extra.write_text("// new\n")
    real = build.headers
    monkeypatch.setattr(build, "headers", lambda: real() + [str(extra)])
    assert build.source_hash() != h0


def test_fft_plan_host_logic():
    """os2d_fft_sizes / os2d_fft_tiles are host code: transform sizes of the frequency-domain 7x7 layer for the pyramid
    levels of BASELINE.json configs[4] (the 96 x 128 level is cut into overlap-save tiles), the invariants every plan must
    satisfy for every map (the in-LDS FFTs serve precisions "fft" / "fft32"), and the fp16 scale of the input spectra."""
    import ctypes
    from os2d_amd import _lib
    lib = _lib.load()

    def plan(h, w):
        P, Q, nb = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        ty, tx, th, tw = ctypes.c_int(), ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        rc = lib.os2d_fft_sizes(h, w, ctypes.byref(P), ctypes.byref(Q), ctypes.byref(nb))
        rc2 = lib.os2d_fft_tiles(h, w, ctypes.byref(ty), ctypes.byref(tx), ctypes.byref(th), ctypes.byref(tw))
        assert rc == rc2
        return (rc, P.value, Q.value, nb.value, ty.value, tx.value, th.value, tw.value)

    expected = {(30, 40): (36, 48), (38, 50): (42, 54), (48, 64): (54, 72), (60, 80): (64, 84), (72, 96): (84, 108), (84, 112): (96, 128)}
    for (h, w), (p, q) in expected.items():
        rc, P, Q, nb, ty, tx, th, tw = plan(h, w)
        assert rc == 0 and (P, Q) == (p, q) and nb == (p * (q // 2 + 1) + 7) // 8 * 8
        assert (ty, tx, th, tw) == (1, 1, h, w)              # these fit one in-LDS transform
    # the largest level of the 7-scale pyramid (reference os2d/config.py:194): 108 x 144 would need 177 KB of LDS -> four
    # tiles of 48 x 64 outputs, each a 54 x 72 transform (the size the 48 x 64 level uses: one set of weight spectra)
    assert plan(96, 128) == (0, 54, 72, 2000, 2, 2, 48, 64)
    for h in list(range(1, 100, 7)) + [120, 157, 300]:
        for w in list(range(1, 130, 9)) + [150, 209]:
            rc, P, Q, nb, ty, tx, th, tw = plan(h, w)
            assert rc == 0, (h, w)                           # every map the head accepts has a plan
            assert ty >= 1 and tx >= 1 and ty * th >= h and tx * tw >= w and (ty - 1) * th < h and (tx - 1) * tw < w
            assert P >= (th + 6 if ty > 1 else h + 3) and Q >= (tw + 6 if tx > 1 else w + 3)
            assert P % 2 == 0 and Q % 2 == 0 and nb % 8 == 0 and nb >= P * (Q // 2 + 1)
            for n in (P, Q):            # 2^a 3^b, or one of the sizes with a factor 7 that have a two-stage form
                m = n
                while m % 2 == 0:
                    m //= 2
                while m % 3 == 0:
                    m //= 3
                assert m == 1 or n in (42, 84)
    # |X| <= samples of one window: 60 x 80 -> 2^3 * 4800 <= 65504; a 96 x 128 map's window is a 54 x 70 tile + halo -> 2^4
    assert lib.os2d_spectral_xscale(60, 80) == 8.0 and lib.os2d_spectral_xscale(96, 128) == 16.0


def test_dft_plan_host_logic():
    """os2d_dft_sizes (precision "fftx3": the transforms as matrix products, dft_mfma.hip) is host code.  Any P % 4 == 0 <= 64
    and even Q <= 94 could be a transform size; the default policy plans every map on SIX canonical sizes - the exact transforms
    of the 7-scale pyramid of a 1280 x 960 image - through overlap-save tiles, so that the weight spectra of a whole dataset are
    2.4 GB, not one 0.2 - 0.7 GB set per map shape; every map up to the head's width limit has a plan."""
    import ctypes
    from os2d_amd import _lib
    lib = _lib.load()

    def plan(h, w):
        P, Q, nb = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        t = (ctypes.c_int * 6)()
        rc = lib.os2d_dft_sizes(h, w, ctypes.byref(P), ctypes.byref(Q), ctypes.byref(nb), t)
        return (rc, P.value, Q.value, nb.value) + tuple(t)

    canonical = {(36, 46), (44, 54), (48, 62), (52, 68), (56, 70), (64, 84)}
    assert plan(60, 80) == (0, 64, 84, 2752, 1, 1, 60, 80, 60, 80)
    assert plan(30, 40)[:4] == (0, 36, 46, 864) and plan(38, 50)[:4] == (0, 44, 54, 1232) and plan(48, 64)[:4] == (0, 52, 68, 52 * 35 + 4)
    assert plan(72, 96)[1:8] == (44, 54, 1232, 2, 2, 36, 48) and plan(84, 112)[1:8] == (48, 62, 1536, 2, 2, 42, 56)
    assert plan(96, 128)[1:8] == (56, 70, 2016, 2, 2, 48, 64)
    assert sum(p * (q // 2 + 1) for p, q in canonical) * 128 * 232 * 8 < 2.5e9          # all six sets of weight spectra
    for h in list(range(1, 100, 7)) + [120, 157, 300]:
        for w in list(range(1, 130, 9)) + [150, 209, 260, 316, 317, 400, 640, 1000, 3600]:
            rc, P, Q, nb, ty, tx, th, tw, lh, lw = plan(h, w)
            assert rc == 0, (h, w)
            assert ty >= 1 and tx >= 1 and ty * th >= h and tx * tw >= w and (ty - 1) * th < h and (tx - 1) * tw < w
            assert (lh, lw) == (th + 6 if ty > 1 else h, tw + 6 if tx > 1 else w)
            assert P >= (th + 6 if ty > 1 else h + 3) and Q >= (tw + 6 if tx > 1 else w + 3)
            assert (P, Q) in canonical and nb % 8 == 0 and nb >= P * (Q // 2 + 1)
    assert lib.os2d_dft_channel_stride(225) == 232 and lib.os2d_dft_matrices_bytes(64, 84) > 0
    assert lib.os2d_dft_xscale(60, 80) == 8.0
