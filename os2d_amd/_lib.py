"""ctypes binding of libos2d_hip.so (C ABI declared in include/os2d_hip.h).

There is deliberately NO fallback: if the shared library is missing or does not export the declared ABI the
import of the compute path fails loudly (``Os2dLibraryError``).  Build it with ``python -m os2d_amd.build``
(or ``__graft_entry__.build()``).
"""
import ctypes
import os

from .build import LIB_PATH

ABI_VERSION = 9

_c_float_p = ctypes.c_void_p   # device pointers travel as raw addresses (tensor.data_ptr())
_vp = ctypes.c_void_p
_i = ctypes.c_int
_f = ctypes.c_float
_sz = ctypes.c_size_t

# name -> (restype, argtypes): must list EVERY symbol of include/os2d_hip.h (tests/test_abi.py checks the header)
SIGNATURES = {
    "os2d_abi_version": (_i, []),
    "os2d_last_error": (ctypes.c_char_p, []),
    "os2d_packed_conv_floats": (_sz, [_i]),
    "os2d_packed_bias_floats": (_sz, [_i]),
    "os2d_pack_conv": (_i, [_i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _f, _vp, _vp, _vp]),
    "os2d_class_prepare": (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _vp]),
    "os2d_head_workspace_bytes": (_i, [_i, _i, _i, _i, _i, _i, ctypes.POINTER(_sz)]),
    "os2d_head_forward": (_i, [_vp] * 8 + [_i] * 9 + [_vp, _vp, _vp, _vp, _sz, _vp]),
    "os2d_packed_conv_bytes": (_sz, [_i, _i]),
    "os2d_pack_conv_f16x3": (_i, [_i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _f, _vp, _vp, _vp, _vp, _vp, _vp]),
    "os2d_rnorm_exp": (_i, []),
    "os2d_class_prepare_batch": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "os2d_class_prepare_workspace_floats": (_sz, [_i, _i]),
    "os2d_shb_bytes": (_sz, [_i, _i, _i]),
    "os2d_corr_normalize_f16x3": (_i, [_vp, _vp, _i, _i, _i, _vp]),
    "os2d_transform_conv_f16x3": (_i, [_i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp]),
    "os2d_alignment_grids": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _vp, _vp]),
    "os2d_spectral_weight_bytes": (_sz, [_i, _i, _i]),
    "os2d_spectral_gemm": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "os2d_spectral_weight16_bytes": (_sz, [_i, _i]),
    "os2d_spectral_xscale": (_f, [_i, _i]),
    "os2d_spectral_weights_build": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp]),
    "os2d_spectral_gemm_f16": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _f, _vp]),
    "os2d_debug_set_dump": (None, [_vp, _i, _vp, _sz]),
    "os2d_fft_sizes": (_i, [_i, _i, ctypes.POINTER(_i), ctypes.POINTER(_i), ctypes.POINTER(_i)]),
    "os2d_fft_tiles": (_i, [_i, _i, ctypes.POINTER(_i), ctypes.POINTER(_i), ctypes.POINTER(_i), ctypes.POINTER(_i)]),
    "os2d_fft_forward": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "os2d_fft_inverse": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    "os2d_fft_inverse_ex": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _i, _vp]),
    "os2d_dft_sizes": (_i, [_i, _i, ctypes.POINTER(_i), ctypes.POINTER(_i), ctypes.POINTER(_i), ctypes.POINTER(_i)]),
    "os2d_dft_channel_stride": (_i, [_i]),
    "os2d_dft_matrices_bytes": (_sz, [_i, _i]),
    "os2d_dft_matrices_build": (_i, [_vp, _vp, _i, _i, _vp, _vp]),
    "os2d_dft_forward": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "os2d_dft_inverse": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    "os2d_spectral_weights_build_dft": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp]),
    "os2d_spectral_gemm_f16_quads": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _f, _vp]),
    "os2d_dft_xscale": (_f, [_i, _i]),
    "os2d_class_split": (_i, [_vp, _vp, _i, _i, _vp]),
    "os2d_class_split_bytes": (_sz, [_i, _i]),
    "os2d_head_forward_ex": (_i, [_vp] * 8 + [_i] * 9 + [_vp, _vp, _vp, _vp, _sz, _vp, _i, _vp,
                                  ctypes.POINTER(_vp), ctypes.POINTER(_i), _vp, _vp, _vp, _vp]),
    "os2d_head_workspace_bytes_ex": (_i, [_i, _i, _i, _i, _i, _i, _i, ctypes.POINTER(_sz)]),
    "os2d_prof_event_create": (_i, [ctypes.POINTER(_vp)]),
    "os2d_prof_event_destroy": (_i, [_vp]),
    "os2d_prof_event_elapsed_ms": (_i, [_vp, _vp, ctypes.POINTER(_f)]),
    "os2d_fm_sumsq": (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    "os2d_plane_floats": (_sz, [_i, _i]),
    "os2d_corr": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "os2d_corr_normalize": (_i, [_vp, _vp, _i, _i, _i, _vp]),
    "os2d_corr_f16x3_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "os2d_corr_f16x3": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _sz, _vp]),
    "os2d_corr_f16x3_packed_workspace_bytes": (_sz, [_i, _i, _i, _i, _i]),
    "os2d_corr_f16x3_packed": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _sz, _vp]),
    "os2d_transform_conv": (_i, [_i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "os2d_sample_decode": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "os2d_decode_boxes": (_i, [_vp, _i, _i, _i, _i, _i, _f, _f, _vp, _vp]),
    "os2d_nms_workspace_bytes": (_i, [_i, _i, ctypes.POINTER(_sz)]),
    "os2d_nms": (_i, [_vp, _vp, _i, _i, _f, _vp, _vp, _vp, _sz, _vp]),
    "os2d_detect_level_supported": (_i, [_i, _i]),
    "os2d_detect_level": (_i, [_vp, _vp] + [_i] * 5 + [_f] * 6 + [_vp] * 5),
    "os2d_detect_level_ops": (_i, [_vp, _vp] + [_i] * 5 + [_f, _f, _i, _vp, _vp, _f, _f] + [_vp] * 5),
    "os2d_detect_pyramid_supported": (_i, [_i, _i, _i]),
    "os2d_detect_pyramid_workspace_bytes": (_i, [_i, _i, _i, ctypes.POINTER(_sz)]),
    "os2d_detect_pyramid": (_i, [_vp, _vp, _vp, _i, _i, _vp, _i, _i, _vp, _vp, _f, _f, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp,
                                 _vp, _vp, _sz, _vp]),
    "os2d_detect_pyramid_merged": (_i, [_vp, _vp, _vp, _i, _i, _vp, _i, _i, _vp, _vp, _f, _f, _i, _i, _i, _i, _vp, _vp, _vp, _vp,
                                        _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "os2d_detect_pyramid_ops": (_i, [_vp, _vp, _vp, _i, _i, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _f, _f, _i, _i, _i, _i, _vp, _vp, _vp,
                                     _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
}


class Os2dLibraryError(RuntimeError):
    pass


_LIB = None


def lib_path():
    return os.environ.get("OS2D_HIP_LIB", LIB_PATH)


def load():
    """Load (once) and return the ctypes handle; raises Os2dLibraryError if it is not there."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = lib_path()
    if "OS2D_HIP_LIB" not in os.environ:
        # The .so is a build artefact (not tracked): compile it in-tree when it is missing OR was built from other
        # sources than the ones in the tree (content hash, os2d_amd/build.py - an edited kernel never runs stale).
        # This is the same HIP library, not a fallback implementation; if hipcc is missing the error below fires.
        from . import build as _build
        if not _build.up_to_date():
            try:
                import fcntl
                os.makedirs(os.path.dirname(path), exist_ok=True)
                with open(path + ".lock", "w") as lock:      # one builder at a time (torchrun starts N ranks at once);
                    fcntl.flock(lock, fcntl.LOCK_EX)         # the check is repeated under the lock
                    if not _build.up_to_date():
                        _build.build(verbose=False)
            except Exception as e:  # noqa: BLE001
                raise Os2dLibraryError("libos2d_hip.so is missing or stale and building it failed ({}); the OS2D head "
                                       "has no CPU or PyTorch fallback".format(e))
    if not os.path.exists(path):
        raise Os2dLibraryError(
            "libos2d_hip.so not found at {} - the OS2D head has no CPU or PyTorch fallback; build the HIP "
            "extension first: python -m os2d_amd.build".format(path))
    # torch must be imported first so that the HIP runtime already mapped in the process (same soname,
    # libamdhip64.so.7) is the one our library binds to: one runtime, shared streams and allocations.
    import torch  # noqa: F401
    try:
        handle = ctypes.CDLL(path, mode=ctypes.RTLD_LOCAL)
    except OSError as e:
        raise Os2dLibraryError("cannot load {}: {}".format(path, e))
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(handle, name)
        except AttributeError:
            raise Os2dLibraryError("{} does not export {} (stale build? run python -m os2d_amd.build --force)".format(path, name))
        fn.restype = res
        fn.argtypes = args
    if handle.os2d_abi_version() != ABI_VERSION:
        raise Os2dLibraryError("ABI version mismatch: library {} vs binding {}".format(handle.os2d_abi_version(), ABI_VERSION))
    _LIB = handle
    return _LIB


def check(rc, what):
    """Raise RuntimeError with the library's message if a call returned an error code."""
    if rc != 0:
        msg = load().os2d_last_error()
        raise RuntimeError("{} failed (code {}): {}".format(what, rc, msg.decode("utf-8", "replace") if msg else "?"))


def ptr(t):
    """Device address of a contiguous float32 CUDA/HIP tensor (None -> NULL)."""
    if t is None:
        return None
    import torch
    if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype in (torch.float32, torch.float64, torch.uint8, torch.int32, torch.int64)
            and t.is_contiguous()):
        raise ValueError("expected a contiguous device tensor, got {}".format(
            (type(t).__name__, getattr(t, "device", None), getattr(t, "dtype", None))))
    return ctypes.c_void_p(t.data_ptr())


def host_ptr(t):
    """Address of a PINNED host tensor (hipHostMalloc memory is mapped into the device address space under the same
    address: kernels may store to it); None -> NULL."""
    if t is None:
        return None
    if not t.is_pinned():
        raise ValueError("expected a pinned host tensor")
    return ctypes.c_void_p(t.data_ptr())


def current_stream(device):
    import torch
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)
