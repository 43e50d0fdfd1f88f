"""In-tree build of libos2d_hip.so for gfx950 with hipcc (no JIT cache: the .so travels with the tree).

    python -m os2d_amd.build                      # build if stale
    python -m os2d_amd.build --force
    python -m os2d_amd.build --variant TAG [--packed on|off|fft] [-DFLAG ...]
                                                  # diagnostic copy under tools/diag_libs/TAG/ (run with OS2D_HIP_LIB=...)
"""
import glob
import hashlib
import os
import re
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_DIR = os.path.join(HERE, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libos2d_hip.so")
BUILD_DIR = os.path.join(HERE, "csrc", "build")
VARIANT_DIR = os.path.join(HERE, "..", "tools", "diag_libs")
SOURCES = ["abi.hip", "prep.hip", "corr_mfma.hip", "conv_mfma.hip", "conv_f16x3.hip", "conv3_f16x3.hip", "corr_f16x3.hip", "sample_decode.hip", "nms.hip", "detect.hip", "detect_pyramid.hip", "spectral.hip", "spectral_f16.hip", "spectra_pack.hip", "fft.hip", "dft_mfma.hip"]
ARCH = "gfx950"
FLAGS = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-Wall", "-Wno-unused-function"]
# No packed-FP32 VALU instructions (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32) in the translation units listed in
# NO_PACKED_FP32 - all of them.  Measured on MI355X (docs/DESIGN_HISTORY_r1-r3.md section 8, profiles/r03_packed_fp32/): such instructions
# occasionally return wrong values in lanes 48 - 63 of a wave while waves of ANOTHER kernel on the same CU issue fp16 MFMA
# instructions at full rate (other HIP streams); a register-only victim without LDS, barriers or memory accesses reproduces
# it (tools/repro_packed_fp32.hip), scalar v_fma_f32 code never does, fp32-MFMA / plain-VALU neighbours never trigger it, and
# the library's transforms fail identically with full __syncthreads() barriers - it is not a race of ours.  Any kernel of the
# library can meet another stream's conv / correlation / spectral-GEMM waves under the per-level-stream pyramid runner.
PACKED_OFF = ["-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"]
NO_PACKED_FP32 = set(SOURCES)             # which translation units are compiled without the packed instructions
FLAGS += os.environ.get("OS2D_EXTRA_HIPCC_FLAGS", "").split()      # kernel experiments (-DOS2D_DIAG_...); part of the source hash


def headers():
    """Every header a translation unit can include: csrc/*.h + the public ABI header.  Globbed, not listed: a new header
    (fft_regs.h was missed in round 2) is part of the source hash and of every object's dependencies from the day it exists."""
    return sorted(glob.glob(os.path.join(CSRC, "*.h"))) + [os.path.join(HERE, "..", "include", "os2d_hip.h")]


def local_includes(path):
    """Names of the quoted #include files of a source (used by the tests: each must be in headers())."""
    with open(path) as f:
        return re.findall(r'^\s*#\s*include\s+"([^"]+)"', f.read(), flags=re.M)


def flags_for(source, packed=None):
    """hipcc flags of one translation unit.  packed: None = the product setting (NO_PACKED_FP32), 'on' / 'off' = every
    unit with / without packed-FP32 instructions, 'fft' = only fft.hip without them (diagnostic variants)."""
    if packed is None:
        off = source in NO_PACKED_FP32
    else:
        off = packed == "off" or (packed == "fft" and source == "fft.hip")
    return FLAGS + (PACKED_OFF if off else [])


def _hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC or install ROCm under /opt/rocm)")


STAMP_PATH = LIB_PATH + ".srchash"


def source_hash():
    """sha256 over every source, header and the compiler flags of every unit: what decides whether the library is up to
    date (mtimes do not survive a copy of the tree to another machine, contents do)."""
    h = hashlib.sha256()
    for s in SOURCES:
        h.update((s + ":" + " ".join(flags_for(s)) + "\n").encode())
    for path in [os.path.join(CSRC, s) for s in SOURCES] + headers():
        h.update(os.path.basename(path).encode())
        with open(path, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def up_to_date():
    """True when libos2d_hip.so exists and was built from exactly the sources in the tree."""
    if not (os.path.exists(LIB_PATH) and os.path.exists(STAMP_PATH)):
        return False
    with open(STAMP_PATH) as f:
        return f.read().strip() == source_hash()


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def _compile_all(hipcc, build_dir, packed, extra, force, verbose):
    os.makedirs(build_dir, exist_ok=True)
    objs, procs = [], []
    for s in [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]:
        src = os.path.join(CSRC, s)
        obj = os.path.join(build_dir, s.replace(".hip", ".o"))
        objs.append(obj)
        if force or _stale(obj, [src] + headers()):
            cmd = [hipcc] + flags_for(s, packed) + list(extra) + ["-c", src, "-o", obj]
            if verbose:
                print("[os2d_amd.build]", " ".join(cmd), flush=True)
            procs.append((cmd, subprocess.Popen(cmd)))
            if len(procs) >= (os.cpu_count() or 4):
                _wait(procs)
    _wait(procs)
    return objs


def _wait(procs):
    failed = [cmd for cmd, p in procs if p.wait() != 0]
    del procs[:]
    if failed:
        raise subprocess.CalledProcessError(1, failed[0])


def build(force=False, verbose=True):
    """Compile every HIP translation unit for gfx950 and link libos2d_hip.so. Returns the library path."""
    hipcc = _hipcc()
    os.makedirs(LIB_DIR, exist_ok=True)
    # object files carry no record of the flags they were compiled with: when the stamp (sources + headers + flags) does not
    # match, everything is recompiled - a flag change must not leave objects of the old flavour in the library
    force = force or not up_to_date()
    objs = _compile_all(hipcc, BUILD_DIR, None, [], force, verbose)
    digest = source_hash()
    if force or _stale(LIB_PATH, objs) or not up_to_date():
        tmp = LIB_PATH + ".tmp.{}".format(os.getpid())          # link aside, then rename: a concurrent loader never
        cmd = [hipcc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", tmp] + objs   # maps a half-written file
        if verbose:
            print("[os2d_amd.build]", " ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        os.replace(tmp, LIB_PATH)
        with open(STAMP_PATH + ".tmp", "w") as f:
            f.write(digest + "\n")
        os.replace(STAMP_PATH + ".tmp", STAMP_PATH)
    return LIB_PATH


def build_variant(tag, packed=None, extra=(), verbose=False):
    """A DIAGNOSTIC copy of the library (other flags, -DOS2D_DIAG_... switches) under tools/diag_libs/<tag>/; the product
    library is untouched.  Built HERE (hipcc cross-compiles) so that a GPU call spends its minutes measuring:
    OS2D_HIP_LIB=tools/diag_libs/<tag>/libos2d_hip.so selects it at run time (os2d_amd/_lib.py)."""
    hipcc = _hipcc()
    out = os.path.join(VARIANT_DIR, tag)
    objs = _compile_all(hipcc, os.path.join(out, "build"), packed, extra, True, verbose)
    lib = os.path.join(out, "libos2d_hip.so")
    subprocess.check_call([hipcc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", lib] + objs)
    shutil.rmtree(os.path.join(out, "build"))
    with open(os.path.join(out, "FLAGS.txt"), "w") as f:
        f.write("packed={} extra={}\n".format(packed, " ".join(extra)))
    return lib


if __name__ == "__main__":
    argv = sys.argv[1:]
    if "--variant" in argv:
        i = argv.index("--variant")
        tag = argv[i + 1]
        packed = argv[argv.index("--packed") + 1] if "--packed" in argv else None
        print(build_variant(tag, packed, [a for a in argv if a.startswith("-D")], verbose=True))
    else:
        print(build(force="--force" in argv))
