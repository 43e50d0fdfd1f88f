"""CPU: the register DFTs and the two-stage factorisation of os2d_amd/csrc/fft_regs.h (the header compiles for the host
too) against a direct double-precision DFT - every size fft.hip instantiates, forward and inverse."""
import os
import shutil
import subprocess

import pytest

REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def test_fft_regs_header_on_the_host(tmp_path):
    cxx = shutil.which("clang++") or "/opt/rocm/lib/llvm/bin/clang++"      # ext_vector_type + generic lambdas: clang
    if not os.path.exists(cxx) and shutil.which(cxx) is None:
        pytest.skip("no clang++")
    exe = str(tmp_path / "fft_regs_check")
    subprocess.run([cxx, "-std=c++17", "-O1", "-I", os.path.join(REPO, "os2d_amd", "csrc"),
                    os.path.join(REPO, "tests", "host", "fft_regs_check.cpp"), "-o", exe], check=True, timeout=300)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and out.stdout.startswith("ok"), out.stdout + out.stderr
