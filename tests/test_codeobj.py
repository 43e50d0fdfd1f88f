"""CPU: resource figures of the kernels in the built library, read from its embedded code objects (os2d_amd/codeobj.py).

VERDICT r4 item 2: the transform kernels of round 4 shipped with 31 / 20 spilled registers (scratch reloads queued in front of
the next window's loads made the matrix phases wait for HBM).  Pinned here: no kernel the default head path launches spills a
register or touches scratch memory, and the diagnostic 2-image forward shape is not part of the product library."""
import re

import pytest

pytest.importorskip("msgpack")      # the code-object metadata is msgpack; without the package there is nothing to read

from os2d_amd import build, codeobj  # noqa: E402

# kernels of one default (fftx3) head call + the decode that follows it
DEFAULT_PATH = ("fm_sumsq_kernel", "split_fm_kernel", "split_qp_kernel", "corr_f16x3_kernel", "dft_forward_kernel", "spectral_gemm_f16_kernel",
                "dft_inverse_kernel", "conv_f16x3_kernel", "conv3_f16x3_kernel", "sample_decode_kernel", "class_resize_batch_kernel",
                "class_normalize_batch_kernel", "detect_level_kernel", "pyr_decode_kernel", "pyr_compact_kernel", "pyr_chunk_nms_kernel",
                "pyr_finalize_kernel", "dft_matrices_kernel", "spectra_pack_kernel")


@pytest.fixture(scope="module")
def kernels():
    if not build.up_to_date():
        build.build(force=False, verbose=False)
    return codeobj.kernels(build.LIB_PATH)


def test_default_path_kernels_do_not_spill(kernels):
    assert len(kernels) > 60
    seen = set()
    for name, k in kernels.items():
        base = next((b for b in DEFAULT_PATH if b in name), None)
        if base is None:
            continue
        seen.add(base)
        assert k["vgpr_spills"] == 0 and k["scratch_bytes"] == 0, (name, k)      # (scalar registers parked in vector lanes are not memory traffic)
    assert seen == set(DEFAULT_PATH), set(DEFAULT_PATH) - seen


def test_transform_kernels_fit_two_waves_per_simd(kernels):
    """8 waves per work-group, one work-group per CU: 256 registers per wave.  Every instantiation - the k-step counts 5 .. 8 of
    the canonical transform sizes and the generic one - stays below that without spilling."""
    dft = {n: k for n, k in kernels.items() if "dft_forward_kernel" in n or "dft_inverse_kernel" in n}
    assert len(dft) == 3 * 5 + 2 * 5 + 3 + 2      # 4 images per iteration: k-step counts 5 .. 8 + generic; 8 images (round 6): one each
    for name, k in dft.items():
        assert k["vgprs"] + k["agprs"] <= 256 and k["vgpr_spills"] == 0 and k["scratch_bytes"] == 0, (name, k)
        assert k["max_threads"] == 512, name
    # dft_forward_kernel<TILED, FAST, G, NW, KS>: 4 images per iteration (KS = 0 | 5 .. 8) and 8 images (KS = 0), always 8 waves
    assert all(re.search(r"dft_forward_kernelILb[01]ELb[01]E(Li4ELi8ELi[05678]|Li8ELi8ELi0)E", n) for n in dft if "forward" in n)


def test_reader_sees_lds_and_register_counts(kernels):
    corr = [k for n, k in kernels.items() if "corr_f16x3_kernel" in n]
    assert corr and all(0 < k["vgprs"] <= 256 for k in corr)
    nms = [k for n, k in kernels.items() if "nms_kernel" in n and "chunk" not in n]
    assert nms and nms[0]["lds_bytes"] > 0


def test_shipped_kernels_are_exactly_the_whitelist(kernels):
    """the retired variants live in tools/patches/, not behind compile switches in the product sources - and the
    library holds exactly the kernels of tests/golden/kernels.txt (``python -m os2d_amd.codeobj --names``): a diagnostic instantiation
    that slips into the build, or a kernel that silently disappears, fails here."""
    import os
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "kernels.txt")
    want = set(open(path).read().split())
    have = set(kernels)
    assert have == want, {"not in the whitelist": sorted(have - want), "missing from the library": sorted(want - have)}


def test_product_sources_carry_no_retired_switches():
    import glob
    import os
    import re
    csrc = os.path.join(os.path.dirname(os.path.abspath(build.__file__)), "csrc")
    hits = []
    for f in sorted(glob.glob(os.path.join(csrc, "*.hip")) + glob.glob(os.path.join(csrc, "*.h"))):
        for i, line in enumerate(open(f), 1):
            if re.search(r"OS2D_SH_|_SPREAD|OS2D_DIAG_", line):
                hits.append("{}:{}".format(os.path.basename(f), i))
    # what remains: the dump aid of abi.hip and the phase stamps of the transforms (diagnostic builds, documented in place)
    assert len(hits) <= 8 and all(h.startswith(("abi.hip", "dft_mfma.h")) for h in hits), hits
