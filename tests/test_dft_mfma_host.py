"""CPU: the kernel source of the matrix-product transforms (os2d_amd/csrc/dft_mfma.h - forward, inverse, plans and the
constant-matrix builder) compiled for the host and run on the SPMD emulator of tests/host/spmd_emu.h (work items = threads,
LDS = a buffer, v_mfma_f32_32x32x16_f16 = an operand exchange inside the wave) against float64 DFTs: the LDS layouts, fragment
addressing, tile ownership, window / tile arithmetic and the power-of-two scale bookkeeping are checked without a GPU."""
import os
import shutil
import subprocess

import pytest

REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def test_dft_mfma_kernels_on_the_host_emulator(tmp_path):
    cxx = shutil.which("clang++") or "/opt/rocm/lib/llvm/bin/clang++"      # ext_vector_type + _Float16: clang
    if not os.path.exists(cxx) and shutil.which(cxx) is None:
        pytest.skip("no clang++")
    exe = str(tmp_path / "dft_mfma_check")
    subprocess.run([cxx, "-std=c++17", "-O1", "-pthread", "-Wno-psabi", "-I", os.path.join(REPO, "os2d_amd", "csrc"),
                    "-I", os.path.join(REPO, "tests", "host"), os.path.join(REPO, "tests", "host", "dft_mfma_check.cpp"), "-o", exe],
                   check=True, timeout=300)
    # small untiled (W % 4 != 0, partial channel group), fast path with several work-groups, a tiled map with ragged tiles, a
    # level of the pyramid with P % 8 == 4
    out = subprocess.run([exe, "11", "13", "5", "1", "20", "24", "4", "2", "70", "100", "5", "1", "30", "43", "6", "2", "9", "11", "4", "66"],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), out.stdout[-3000:] + out.stderr[-2000:]
    # the same kernels planned on the canonical transform sizes (the default policy): a map much smaller than its transform, and
    # a map that needs tiles although its smallest transform would fit
    out = subprocess.run([exe, "canonical", "11", "13", "5", "1", "50", "70", "4", "1"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), out.stdout[-3000:] + out.stderr[-2000:]
    # the levels of the benchmark pyramid whose transforms take 8 / 7 / 6 k-steps in step 2 / step A: the instantiations with the
    # k-step count as a template parameter (round 5), which is what the default head launches
    out = subprocess.run([exe, "canonical", "60", "80", "4", "1", "48", "64", "8", "1", "38", "50", "4", "2"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), out.stdout[-3000:] + out.stderr[-2000:]
    assert "P=64 Q=84" in out.stdout and "P=52 Q=68" in out.stdout and "P=44 Q=54" in out.stdout
    # round 6: the 8-image shape of both kernels (row operand of step 2 / step A in LDS, whole activation units) on the transforms
    # that take it - the small levels of the pyramid and the 2 x 2 tiles of the 72 x 96 / 84 x 112 levels - next to the 4-image shape
    out = subprocess.run([exe, "canonical", "30", "40", "17", "1", "72", "96", "5", "1", "84", "112", "9", "1", "45", "61", "8", "2"],
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), out.stdout[-3000:] + out.stderr[-2000:]
    assert out.stdout.count("forward (G = 8)") == 3 and out.stdout.count("inverse (G = 8)") == 3      # 36x46, 44x54 (tiled), 48x62 (tiled)
    assert "no 8-image plan" in out.stdout                                                               # 52x68 stays with 4 images
