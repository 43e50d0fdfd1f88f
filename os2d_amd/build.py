"""In-tree build of libos2d_hip.so for gfx950 with hipcc (no JIT cache: the .so travels with the tree).

    python -m os2d_amd.build          # build if stale
    python -m os2d_amd.build --force
"""
import hashlib
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_DIR = os.path.join(HERE, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libos2d_hip.so")
BUILD_DIR = os.path.join(HERE, "csrc", "build")
SOURCES = ["abi.hip", "prep.hip", "corr_mfma.hip", "conv_mfma.hip", "conv_f16x3.hip", "conv3_f16x3.hip", "corr_f16x3.hip", "sample_decode.hip", "nms.hip", "detect.hip", "detect_pyramid.hip", "spectral.hip", "spectral_f16.hip", "fft.hip"]
HEADERS = [os.path.join(CSRC, "os2d_common.h"), os.path.join(HERE, "..", "include", "os2d_hip.h")]
ARCH = "gfx950"
FLAGS = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-Wall", "-Wno-unused-function"]
# No packed-FP32 VALU instructions (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32).  Measured on MI355X (round 2, DESIGN.md
# section 8): the FFT kernels - 19k of these instructions for their complex arithmetic - return wrong values in 16-lane
# groups of single registers, a few images in 30,000, whenever MFMA-heavy kernels of OTHER streams run at the same time
# (never alone, never on one stream; with the feature off: 0 of 32 runs against 16 of 32).  Scalar v_fma_f32 code costs
# nothing measurable in those kernels (LDS / latency bound); the resampler pays 0.017 ms per 64 classes.
FLAGS += ["-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"]
FLAGS += os.environ.get("OS2D_EXTRA_HIPCC_FLAGS", "").split()      # kernel experiments (-DOS2D_DIAG_...); part of the source hash


def _hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC or install ROCm under /opt/rocm)")


STAMP_PATH = LIB_PATH + ".srchash"


def source_hash():
    """sha256 over every source, header and the compiler flags: what decides whether the library is up to date
    (mtimes do not survive a copy of the tree to another machine, contents do)."""
    h = hashlib.sha256()
    h.update(" ".join(FLAGS).encode())
    for path in [os.path.join(CSRC, s) for s in SOURCES] + HEADERS:
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


def build(force=False, verbose=True):
    """Compile every HIP translation unit for gfx950 and link libos2d_hip.so. Returns the library path."""
    hipcc = _hipcc()
    os.makedirs(BUILD_DIR, exist_ok=True)
    os.makedirs(LIB_DIR, exist_ok=True)
    srcs = [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    objs = []
    # object files carry no record of the flags they were compiled with: when the stamp (sources + headers + FLAGS) does not
    # match, everything is recompiled - a flag change must not leave objects of the old flavour in the library
    force = force or not up_to_date()
    for s in srcs:
        src = os.path.join(CSRC, s)
        obj = os.path.join(BUILD_DIR, s.replace(".hip", ".o"))
        objs.append(obj)
        if force or _stale(obj, [src] + HEADERS):
            cmd = [hipcc] + FLAGS + ["-c", src, "-o", obj]
            if verbose:
                print("[os2d_amd.build]", " ".join(cmd), flush=True)
            subprocess.check_call(cmd)
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


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
