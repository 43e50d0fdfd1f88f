"""Victim / aggressor harness of the multi-stream stability checks (docs/DESIGN_HISTORY_r1-r3.md section 8, the packed-FP32 finding).

A *victim* is one kernel (or kernel pair) of the head launched through the C ABI with fixed inputs; its output must be the
same bytes whether it runs alone or on four streams while the direct 7x7 kernel (half-precision MFMA at full rate) runs on
three others.  Victims: the transforms (`fft`: the kernels the finding was made in), the two other kernels with
hand-made barriers and LDS-DMA pipelines - the split-half spectral GEMM (`gemm16`) and the correlation (`corr`) - and the
resampler (`sample`: 528 packed instructions when they are enabled).
Used by tests/test_spectral_gpu.py and tools/diag_aggressor.py.
"""
import ctypes

import torch

from os2d_amd import _lib
from os2d_amd.modeling import head as head_mod
from os2d_amd.utils import synthetic

NF, NA = 4, 3           # victim streams, aggressor streams


def _twiddles(n, device):
    import numpy as np
    m = torch.arange(n, dtype=torch.float64)
    ang = -2.0 * np.pi * m / n
    return torch.stack([torch.cos(ang), torch.sin(ang)], 1).float().to(device).contiguous()


def _fft_sizes(H, W):
    lib = _lib.load()
    P, Q, nb = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    _lib.check(lib.os2d_fft_sizes(H, W, ctypes.byref(P), ctypes.byref(Q), ctypes.byref(nb)), "os2d_fft_sizes")
    return P.value, Q.value, nb.value


def _dft_sizes(H, W):
    """Transform size of the matrix-product transforms (precision "fftx3"): P % 4 == 0, even Q, bins = v * P + u."""
    lib = _lib.load()
    P, Q, nb = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    _lib.check(lib.os2d_dft_sizes(H, W, ctypes.byref(P), ctypes.byref(Q), ctypes.byref(nb), None), "os2d_dft_sizes")
    return P.value, Q.value, nb.value


class Harness:
    def __init__(self, device, H=48, W=64, NB=64):
        self.lib = lib = _lib.load()
        self.dev, self.H, self.W, self.NB = device, H, W, NB
        self.g = torch.Generator().manual_seed(0)
        self.fstreams = [torch.cuda.Stream(device=device) for _ in range(NF)]
        self.astreams = [torch.cuda.Stream(device=device) for _ in range(NA)]
        self.status = torch.zeros(1, dtype=torch.int32, device=device)
        self.net = head_mod.TransformationNet(output_dim=6)
        self.net.load_state_dict(synthetic.make_transform_net_state(6, seed=3))
        self.net.to(device).eval()
        # the aggressor: the direct 7x7 kernel on zero activations of twice the victims' batch
        self.w1, self.b1 = self.net.packed("f16x3")[:2]
        self.a_in = [torch.zeros(2 * NB * lib.os2d_shb_bytes(225, H, W), dtype=torch.uint8, device=device) for _ in range(NA)]
        self.a_out = [torch.zeros(2 * NB * lib.os2d_shb_bytes(128, H, W), dtype=torch.uint8, device=device) for _ in range(NA)]
        self._victims = {}

    # ------------------------------------------------------------------------------------------------ aggressor
    def aggress(self, j):
        _lib.check(self.lib.os2d_transform_conv_f16x3(1, _lib.ptr(self.a_in[j]), _lib.ptr(self.w1), _lib.ptr(self.b1), _lib.ptr(self.a_out[j]),
                                                      2 * self.NB, 6, self.H, self.W, 3, _lib.ptr(self.status),
                                                      ctypes.c_void_p(self.astreams[j].cuda_stream)), "conv1 (aggressor)")

    # ------------------------------------------------------------------------------------------------ victims
    def victim(self, kind):
        """-> (run(i, outputs, stream), make_outputs(i)) for victim stream i."""
        if kind in self._victims:
            return self._victims[kind]
        lib, dev, H, W, NB, g = self.lib, self.dev, self.H, self.W, self.NB, self.g
        P, Q, nbins = _fft_sizes(H, W)
        if kind == "fft":
            C, Cout = 225, 128
            tq, tp = _twiddles(Q, dev), _twiddles(P, dev)
            corr = [torch.randn(NB, C, H * W, generator=g).to(dev) for _ in range(NF)]
            inv = [torch.rand(NB, H * W, generator=g).to(dev) + 0.5 for _ in range(NF)]
            Yin = [torch.randn(NB, Cout, nbins, 2, generator=g).to(dev) for _ in range(NF)]
            bp = torch.ones(3 * 128, device=dev)
            shb = lib.os2d_shb_bytes(Cout, H, W)

            def run(i, out, st):
                s = ctypes.c_void_p(st.cuda_stream)
                _lib.check(lib.os2d_fft_forward(_lib.ptr(corr[i]), _lib.ptr(inv[i]), _lib.ptr(out[0]), _lib.ptr(tq), _lib.ptr(tp), NB, C, H, W, s), "fwd")
                _lib.check(lib.os2d_fft_inverse(_lib.ptr(Yin[i]), _lib.ptr(bp), _lib.ptr(out[1]), _lib.ptr(tq), _lib.ptr(tp), NB, Cout, H, W,
                                                _lib.ptr(self.status), s), "inv")

            def outputs(i):
                return [torch.zeros(C, NB, nbins, 2, device=dev), torch.zeros(NB * shb, dtype=torch.uint8, device=dev)]
            keep = (tq, tp, corr, inv, Yin, bp)
        elif kind == "gemm16":
            # the split-half GEMM as the head runs it: spectra in quads of bins x channels on both sides (dft_mfma.hip's layouts)
            w16 = self.net.spectra(H, W, split=True)[0]
            nb2 = _dft_sizes(H, W)[2]
            xs = lib.os2d_dft_xscale(H, W)
            cpad = lib.os2d_dft_channel_stride(225)
            X = [(torch.rand(nb2 // 4, NB, cpad, 4, 2, generator=g) * 40.0 - 20.0).to(dev) for _ in range(NF)]

            def run(i, out, st):
                _lib.check(lib.os2d_spectral_gemm_f16_quads(_lib.ptr(w16), _lib.ptr(X[i]), _lib.ptr(out[0]), NB, 225, 128, nb2, xs,
                                                            ctypes.c_void_p(st.cuda_stream)), "gemm16")

            def outputs(i):
                return [torch.zeros(nb2 // 4, NB, 128, 4, 2, device=dev)]
            keep = (w16, X)
        elif kind == "dft":
            # the matrix-product transforms (half-precision MFMA + VALU splits, hand-made LDS-only barriers)
            C, Cout = 225, 128
            nb2 = _dft_sizes(H, W)[2]
            cpad = lib.os2d_dft_channel_stride(C)
            mats = self.net.spectra(H, W, split=True)[1]
            corr = [torch.randn(NB, C, H * W, generator=g).to(dev) for _ in range(NF)]
            inv = [(torch.rand(NB, H * W, generator=g) * 0.2 + 0.05).to(dev) for _ in range(NF)]
            Yin = [torch.randn(nb2 // 4, NB, Cout, 4, 2, generator=g).to(dev) for _ in range(NF)]
            bp = torch.ones(3 * 128, device=dev)
            shb = lib.os2d_shb_bytes(Cout, H, W)

            def run(i, out, st):
                s = ctypes.c_void_p(st.cuda_stream)
                _lib.check(lib.os2d_dft_forward(_lib.ptr(corr[i]), _lib.ptr(inv[i]), _lib.ptr(out[0]), _lib.ptr(mats), NB, C, H, W, s), "dft fwd")
                _lib.check(lib.os2d_dft_inverse(_lib.ptr(Yin[i]), _lib.ptr(bp), _lib.ptr(out[1]), _lib.ptr(mats), NB, Cout, H, W,
                                                _lib.ptr(self.status), s), "dft inv")

            def outputs(i):
                return [torch.zeros(nb2 // 4, NB, cpad, 4, 2, device=dev), torch.zeros(NB * shb, dtype=torch.uint8, device=dev)]
            keep = (mats, corr, inv, Yin, bp)
        elif kind == "corr":
            Cf = 1024
            fm = [synthetic.make_feature_map(Cf, H, W, seed=10 + i).to(dev) for i in range(NF)]
            qp = torch.rand(NB, Cf, 256, generator=g).to(dev) / 32.0
            qs = torch.zeros(lib.os2d_class_split_bytes(NB, Cf), dtype=torch.uint8, device=dev)
            _lib.check(lib.os2d_class_split(_lib.ptr(qp), _lib.ptr(qs), NB, Cf, _lib.current_stream(dev)), "class_split")
            ws = [torch.zeros(lib.os2d_corr_f16x3_workspace_bytes(1, Cf, H, W), dtype=torch.uint8, device=dev) for _ in range(NF)]
            rshb_bytes = NB * lib.os2d_shb_bytes(225, H, W)

            def run(i, out, st):
                _lib.check(lib.os2d_corr_f16x3(_lib.ptr(fm[i]), _lib.ptr(qs), _lib.ptr(out[0]), _lib.ptr(out[1]), 1, NB, Cf, H, W,
                                               _lib.ptr(ws[i]), ws[i].numel(), ctypes.c_void_p(st.cuda_stream)), "corr")

            def outputs(i):
                return [torch.zeros(NB, 225, H * W, device=dev), torch.zeros(rshb_bytes, dtype=torch.uint8, device=dev)]
            keep = (fm, qp, qs, ws)
        elif kind == "corrp":
            # the correlation with the classes packed along M: fixed-point sums combined by 64-bit atomics + the norms pass
            Cf = 1024
            fm = [synthetic.make_feature_map(Cf, H, W, seed=10 + i).to(dev) for i in range(NF)]
            qp = torch.rand(NB, Cf, 256, generator=g).to(dev) / 32.0
            qp[:, :, 225:] = 0.0                                   # rows 225 .. 255 of the class operand are zero by contract
            qs = torch.zeros(lib.os2d_class_split_bytes(NB, Cf), dtype=torch.uint8, device=dev)
            _lib.check(lib.os2d_class_split(_lib.ptr(qp), _lib.ptr(qs), NB, Cf, _lib.current_stream(dev)), "class_split")
            ws = [torch.zeros(lib.os2d_corr_f16x3_packed_workspace_bytes(1, NB, Cf, H, W), dtype=torch.uint8, device=dev) for _ in range(NF)]

            def run(i, out, st):
                _lib.check(lib.os2d_corr_f16x3_packed(_lib.ptr(fm[i]), _lib.ptr(qs), _lib.ptr(out[0]), _lib.ptr(out[1]), 1, NB, Cf, H, W, 1,
                                                      _lib.ptr(ws[i]), ws[i].numel(), ctypes.c_void_p(st.cuda_stream)), "corr packed")

            def outputs(i):
                return [torch.zeros(NB, 225, H * W, device=dev), torch.zeros(NB, H * W, device=dev)]
            keep = (fm, qp, qs, ws)
        elif kind == "sample":          # the resampler: the other kernel with hundreds of packed-FP32 instructions when they are on
            corr = [torch.randn(NB, 225, H * W, generator=g).to(dev) for _ in range(NF)]
            par = torch.zeros(NB, 6, H * W)
            par[:, 0], par[:, 4] = 1.0, 1.0
            par = (par + 0.1 * torch.randn(NB, 6, H * W, generator=g)).to(dev)

            def run(i, out, st):
                _lib.check(lib.os2d_sample_decode(_lib.ptr(corr[i]), _lib.ptr(par), NB, H, W, 6, 1, 16, 16, _lib.ptr(out[0]), _lib.ptr(out[1]),
                                                  _lib.ptr(out[2]), ctypes.c_void_p(st.cuda_stream)), "sample_decode")

            def outputs(i):
                return [torch.zeros(NB, k, H * W, device=dev) for k in (4, 1, 8)]
            keep = (corr, par)
        else:
            raise ValueError(kind)
        self._victims[kind] = (run, outputs, keep)
        return self._victims[kind]

    def reference(self, kind):
        run, outputs, _ = self.victim(kind)
        main = torch.cuda.current_stream(self.dev)
        ref = [outputs(i) for i in range(NF)]
        for i in range(NF):
            run(i, ref[i], main)
        torch.cuda.synchronize()
        return ref

    def contend(self, kind, ref, aggressor=True, reps=3):
        """One round: `reps` x (aggressors on their streams, the victim on its four streams); -> list of
        (victim stream, output index) whose bytes differ from the reference, and the outputs."""
        run, outputs, _ = self.victim(kind)
        out = [outputs(i) for i in range(NF)]
        torch.cuda.synchronize()
        for _ in range(reps):
            if aggressor:
                for j in range(NA):
                    self.aggress(j)
            for i in range(NF):
                run(i, out[i], self.fstreams[i])
        torch.cuda.synchronize()
        bad = [(i, k) for i in range(NF) for k in range(len(out[i])) if not torch.equal(out[i][k], ref[i][k])]
        return bad, out
