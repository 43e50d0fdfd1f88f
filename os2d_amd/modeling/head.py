"""Host-side mirror of the reference's head API (os2d/modeling/head.py) on top of libos2d_hip.so.

Same class names, constructor arguments, attribute names and state-dict keys as the reference so that a caller of
``Os2dHeadCreator.create_os2d_head`` / ``Os2dHead.forward`` can switch over:

    build_os2d_head_creator  reference head.py:12-15
    Os2dAlignment            reference head.py:43-193   (holds the TransformNet; transform assembly runs in HIP)
    Os2dHeadCreator          reference head.py:204-268
    Os2dHead                 reference head.py:271-435  (forward = ONE call of os2d_head_forward for all classes)
    TransformationNet        reference head.py:604-661

All arithmetic of the path runs in the hand-written gfx950 kernels (os2d_amd/csrc); PyTorch only owns device
memory, the stream and the parameter containers.  There is no CPU / eager fallback: CPU tensors raise.
Training through the head (autograd) is out of scope and raises as well.
"""
import collections
import itertools
import ctypes
import logging
import os

import torch
import torch.nn as nn

from .. import _lib
from ..structures.feature_map import FeatureMapSize
from .box_coder import BoxGridGenerator

TEMPLATE = 15
QROWS = 256
MAX_W_DIRECT7 = 209     # OS2D_MAX_W_DIRECT7: widest map of the direct 7x7 kernels; beyond it the layer runs in the frequency domain
FFT_MIN_PAIRS = 7       # precision "fft" / "fftx3": image x class pairs below which the direct 7x7 kernel is used instead
                        # (measured crossover at 60 x 80, tools/time_small_batches.py: 6 pairs 0.413 vs 0.416 ms, 8 pairs 0.46 vs 0.53)
PRECISIONS = {"f32": 0, "f16x3": 1, "f16x2": 2, "fft": 3, "fftx3": 4, "fft32": 5}     # OS2D_PRECISION_* of include/os2d_hip.h
FFT_MODES = ("fft", "fftx3", "fft32")
FP32_MODES = ("f32", "fft32")           # no fp16 value anywhere: fp32-MFMA correlation and 5x5 layers
DEFAULT_PRECISION = "fftx3"


def resolve_precision(precision=None):
    """Arithmetic of the two large TransformNet convolutions: "f32" (exact fp32 MFMA) or "f16x3" (fp16 hi/lo split on
    the half-precision matrix cores, fp32-equivalent results: same 2.4e-7 agreement with the reference on every
    parity case), or "f16x2" (as f16x3, but the 7x7 layer takes its weights as fp16 roundings only: 2/3 of the
    matrix-core work, box regression within 5e-5 and scores within 1e-6 of fp32, inside the 1e-4 parity bound), or "fft"
    (as f16x3, but the 7x7 layer runs in the frequency domain in fp32: real FFT -> one complex GEMM per bin on the fp32
    matrix cores -> inverse FFT; fp32-equivalent, 16.7x fewer multiply-adds; maps that do not fit the in-LDS transform and
    small class batches fall back to f16x3), or "fftx3" (as fft, with the per-bin GEMM on the half-precision matrix cores:
    spectra split into fp16 hi + lo, the arithmetic of f16x3), or "fft32" (strictly fp32 and fast: the correlation and the 5x5
    layers on the fp32 matrix cores as in "f32", the 7x7 layer in the frequency domain in fp32 as in "fft" - no fp16 value
    anywhere; 2.6x the throughput of "f32").  Default from $OS2D_PRECISION, else DEFAULT_PRECISION."""
    return _resolve(precision)[0]


def _resolve(precision=None):
    """-> (name, pinned).  A trailing "!" on a frequency-domain mode ("fftx3!", "fft!") pins the ROUTE: the 7x7 layer runs in
    the frequency domain whatever the number of (image, class) pairs of the call (``FFT_MIN_PAIRS`` is a speed heuristic;
    the two routes differ in the last bits, so a caller that needs results independent of how classes are batched - a
    class-sharded run with a ragged tail rank - pins it or passes ``route_pairs`` to ``Os2dHead.forward``)."""
    precision = precision or os.environ.get("OS2D_PRECISION", DEFAULT_PRECISION)
    pinned = isinstance(precision, str) and precision.endswith("!")
    name = precision[:-1] if pinned else precision
    if name not in PRECISIONS or (pinned and name not in FFT_MODES):
        raise ValueError("precision must be one of {} (\"fft!\" / \"fftx3!\" pin the frequency-domain route), got {!r}".format(
            sorted(PRECISIONS), precision))
    return name, pinned


def build_os2d_head_creator(do_simple_affine, is_cuda, use_inverse_geom_model, feature_map_stride,
                            feature_map_receptive_field):
    """reference head.py:12-15."""
    aligner = Os2dAlignment(do_simple_affine, is_cuda, use_inverse_geom_model)
    return Os2dHeadCreator(aligner, feature_map_stride, feature_map_receptive_field)


# --------------------------------------------------------------------------------------------- workspace
_WORKSPACES = {}


def workspace_cap_bytes():
    """Upper bound of ONE scratch buffer (classes are processed in chunks that fit).  An MI355X has 288 GB of HBM3E, so
    the default is generous: 16 GiB holds ~1150 classes at 60x80 in one chunk (all 1024 classes of BASELINE.json's
    configs[2] in a single pass of large launches)."""
    return int(float(os.environ.get("OS2D_WORKSPACE_GB", "16")) * (1 << 30))


def workspace_total_cap_bytes():
    """Upper bound of ALL scratch buffers of a device together (one per HIP stream in use: 7 for the pyramid runner)."""
    return int(float(os.environ.get("OS2D_WORKSPACE_TOTAL_GB", "96")) * (1 << 30))


def get_workspace(device, wanted, minimum):
    """Grow-only scratch buffer, one per (device, stream): concurrent heads on different HIP streams (the pyramid
    runner) must not share intermediates.  The pyramid runner draws its streams from a fixed pool
    (os2d_amd/engine/pyramid.py), so the number of buffers is bounded by the number of level slots; their total is kept
    below ``workspace_total_cap_bytes`` by shrinking what a NEW or growing buffer may take (never below ``minimum``, the
    footprint of one class)."""
    index = device.index if device.index is not None else torch.cuda.current_device()
    key = (device.type, index, torch.cuda.current_stream(device).cuda_stream)
    buf = _WORKSPACES.get(key)
    others = sum(b.numel() for k, b in _WORKSPACES.items() if b is not None and k != key and k[1] == index)
    room = max(workspace_total_cap_bytes() - others, 0)
    size = max(min(wanted, workspace_cap_bytes(), room), minimum)
    if buf is None or buf.numel() < size:
        _WORKSPACES[key] = None
        buf = torch.empty(size, dtype=torch.uint8, device=device)
        buf[:4096].zero_()       # the range words of a head call live at the start of its workspace (include/os2d_hip.h)
        _WORKSPACES[key] = buf
    return buf


def release_workspaces():
    _WORKSPACES.clear()


_STATUS_WORDS = {}
STATUS_SLOTS = 1024


class _StatusSlots(object):
    """The range-status words of one device: pinned arrays of ``STATUS_SLOTS`` int32 that are never freed, a free list, and
    one slot per LIVE head."""

    def __init__(self):
        self.arrays = []
        self.free = []

    def take(self):
        if not self.free:
            words = torch.zeros(STATUS_SLOTS, dtype=torch.int32).pin_memory()
            self.arrays.append(words)
            self.free.extend(words[i:i + 1] for i in range(STATUS_SLOTS - 1, -1, -1))
        word = self.free.pop()
        word[0] = 0
        return word

    def give_back(self, word):
        self.free.insert(0, word)          # reused last: kernels of the head that owned it have long drained by then


def device_status_word(device, owner=None):
    """A sticky range-status word of the split-fp16 kernels for ONE head: an int32 in mapped pinned host memory.  The words
    of a device live in arrays that are allocated once and kept for the life of the process: kernels in flight store to a
    word through a raw pointer (only when an activation leaves the fp16 range), so the memory must never go back to the host
    allocator while any kernel of any head may still run - a per-head tensor (round 2) could be garbage-collected with its
    head while that head's kernels were still queued.  Every LIVE head owns one word (round 3 shared ONE word
    between all heads of a device: a flag raised by head A's kernels could be consumed and cleared by head B - ADVICE r3; round 4
    handed slots out round-robin and zeroed a recycled slot under its live owner after 1024 heads - ADVICE r4): a word returns
    to the free list when ``owner`` is garbage-collected, goes to the back of that list, and the arrays grow by another
    ``STATUS_SLOTS`` words when every word is owned."""
    index = device.index if device.index is not None else torch.cuda.current_device()
    slots = _STATUS_WORDS.get(index)
    if slots is None:
        slots = _STATUS_WORDS[index] = _StatusSlots()
    word = slots.take()
    if owner is not None:
        import weakref
        weakref.finalize(owner, slots.give_back, word)
    return word


class _StreamOrdered(object):
    """A cached device operand together with the event that marks the end of the kernels that produced it.  Consumers on
    OTHER streams (the per-level streams of the pyramid runner) wait for that event before their kernels read the
    buffers; on the producing stream plain stream order is enough."""

    def __init__(self, key, value, device):
        self.key, self.value = key, value
        stream = torch.cuda.current_stream(device)
        self._stream = stream.cuda_stream
        self._event = torch.cuda.Event()
        self._event.record(stream)

    def get(self, device):
        stream = torch.cuda.current_stream(device)
        if stream.cuda_stream != self._stream:
            stream.wait_event(self._event)
            for t in self._tensors():     # the cache may drop the operand while this stream's kernels still read it
                t.record_stream(stream)
        return self.value

    def _tensors(self):
        values = self.value if isinstance(self.value, (tuple, list)) else (self.value,)
        return [t for t in values if isinstance(t, torch.Tensor)]

    def nbytes(self):
        return sum(t.numel() * t.element_size() for t in self._tensors())


def spectra_cache_cap_bytes(split=True):
    """Upper bound for the cached weight spectra of one arithmetic FAMILY of the frequency-domain modes on ONE DEVICE - over all
    transform sizes and all TransformationNets of the process (least recently used entries are dropped first; 654 MB for the
    64 x 84 transform of a 60 x 80 map).  $OS2D_FFT_CACHE_BYTES overrides both families.
      split = True  ("fftx3", the default precision): every map is planned on six canonical transform sizes (overlap-save tiles;
                    os2d_amd/csrc/dft_mfma.h): 2.4 GB in total, whatever the dataset - an eighth of the device's memory, at
                    most 8 GiB;
      split = False ("fft" / "fft32"): one set of spectra per FFT-friendly transform size - 52 sizes = 35.6 GB for a dataset fed
                    at its own aspect ratios and 7 pyramid scales (reference os2d/data/dataloader.py:326;
                    tools/bench_size_churn.py) - a quarter of the device's memory, at most 64 GiB (under the 8 GiB
                    cap of round 4 an LRU cache missed on EVERY call of such a cyclic access pattern)."""
    env = os.environ.get("OS2D_FFT_CACHE_BYTES")
    if env:
        return int(env)
    total = 64 << 30
    if torch.cuda.is_available():
        total = torch.cuda.get_device_properties(torch.cuda.current_device()).total_memory
    return int(min(8 << 30, total // 8)) if split else int(min(64 << 30, total // 4))


class _SpectraStore(object):
    """The cached weight spectra of one device, shared by every TransformationNet of the process (a per-net cap let N
    nets pin N times the cap, invisibly to the caching allocator's out-of-memory retry).  Entries are keyed by (net id, slot),
    least recently used first; a net that goes away takes its entries with it."""
    _stores = {}

    @classmethod
    def of(cls, device):
        index = device.index if device.index is not None else torch.cuda.current_device()
        store = cls._stores.get(index)
        if store is None:
            store = cls._stores[index] = cls()
        return store

    def __init__(self):
        self.entries = collections.OrderedDict()

    def get(self, net_id, slot, key):
        entry = self.entries.get((net_id, slot))
        if entry is None or entry.key != key:
            return None
        self.entries.move_to_end((net_id, slot))
        return entry

    def drop_stale(self, net_id, key):
        for k in [k for k, e in self.entries.items() if k[0] == net_id and e.key != key]:
            del self.entries[k]

    def drop_net(self, net_id):
        for k in [k for k in self.entries if k[0] == net_id]:
            del self.entries[k]

    def put(self, net_id, slot, entry):
        family = bool(slot[2])                                 # slot = (P, Q, split): the two families have caps of their own
        cap, used = spectra_cache_cap_bytes(family), entry.nbytes()
        for k in list(self.entries):                          # oldest first, whichever net it belongs to
            if used + sum(e.nbytes() for kk, e in self.entries.items() if bool(kk[1][2]) == family) <= cap:
                break
            if bool(k[1][2]) == family:
                del self.entries[k]
        self.entries[(net_id, slot)] = entry

    def slots_of(self, net_id):
        return [k[1] for k in self.entries if k[0] == net_id]

    def nbytes(self):
        return sum(e.nbytes() for e in self.entries.values())


def split_rows_f16(T):
    """(float64 MODEL of what spectra_pack.hip does on the device; used by tests/test_range_plan.py.)
    fp16 hi + lo split of a float64 tensor whose leading dimension indexes rows that get their own power-of-two scale
    (precision "fftx3": the weight spectra; one row per output channel): row r is multiplied by 2^wexp[r], the largest
    power of two that keeps its largest |entry| <= 32768 (< 65504: no overflow, also not through rounding), then
    hi = rn16(v), lo = rn16(v - hi): hi + lo carries 22 bits of v wherever |v| >= 2^-3 and an absolute 2^-25 below (fp16
    subnormal spacing).  Returns (hi, lo, wexp); works in place on T (which holds the residual afterwards)."""
    dims = tuple(range(1, T.dim()))
    amax = T.abs().amax(dim=dims).clamp_min(1e-300)
    wexp = torch.floor(torch.log2(32768.0 / amax)).clamp(-100, 100)
    T *= torch.exp2(wexp).view(-1, *([1] * (T.dim() - 1)))
    hi = T.to(torch.float16)
    T -= hi.double()
    lo = T.to(torch.float16)
    return hi, lo, wexp


def _require_device_f32(t, name):
    if not isinstance(t, torch.Tensor):
        raise TypeError("{} must be a torch.Tensor".format(name))
    if not t.is_cuda:
        raise RuntimeError("{} is on {}: the OS2D head runs only on a HIP device (libos2d_hip.so); there is no "
                           "CPU fallback".format(name, t.device))
    if t.dtype != torch.float32:
        raise RuntimeError("{} must be float32, got {}".format(name, t.dtype))
    return t.contiguous()


_NET_IDS = itertools.count(1)


# --------------------------------------------------------------------------------------------- TransformNet
class _SpectraView(object):
    """dict-like access to ONE net's entries of a device store (tests / tools: len, in, clear, values)."""

    def __init__(self, store, net_id):
        self.store, self.net_id = store, net_id

    def __len__(self):
        return len(self.store.slots_of(self.net_id))

    def __contains__(self, slot):
        return (self.net_id, slot) in self.store.entries

    def __iter__(self):
        return iter(self.store.slots_of(self.net_id))

    def values(self):
        return [e for k, e in self.store.entries.items() if k[0] == self.net_id]

    def clear(self):
        self.store.drop_net(self.net_id)


def _drop_spectra_of(net_id):
    for store in _SpectraStore._stores.values():
        store.drop_net(net_id)


class TransformationNet(nn.Module):
    """Parameter container with the reference's layout (head.py:604-646): ``conv`` = Sequential(Conv 225->128 k7,
    BatchNorm, ReLU, Conv 128->64 k5, BatchNorm, ReLU) and ``linear`` = Conv 64->output_dim k5, the latter
    initialised to the identity transform.  The compute is the MFMA implicit-GEMM kernels; ``packed()`` folds
    eval-mode BatchNorm and re-lays the filters out for them (cached until a parameter changes)."""

    def __init__(self, output_dim=6, use_cuda=True, normalization="batchnorm", kernel_sizes=[7, 5],
                 channels=[128, 64], input_feature_dim=15 * 15, num_groups=16):
        super(TransformationNet, self).__init__()
        if list(kernel_sizes) != [7, 5] or list(channels) != [128, 64] or input_feature_dim != 225:
            raise RuntimeError("the HIP TransformNet kernels are built for the OS2D architecture "
                               "(225 -> 128 k7 -> 64 k5 -> P k5)")
        if normalization.lower() != "batchnorm":
            raise RuntimeError("only 'batchnorm' TransformNet normalisation is supported (the reference "
                               "hard-codes it, head.py:72)")
        mods = []
        ch_in = input_feature_dim
        for ch_out, k in zip(channels, kernel_sizes):
            mods += [nn.Conv2d(ch_in, ch_out, kernel_size=k, padding=k // 2), nn.BatchNorm2d(ch_out),
                     nn.ReLU(inplace=True)]
            ch_in = ch_out
        self.conv = nn.Sequential(*mods)
        k = kernel_sizes[-1]
        self.linear = nn.Conv2d(ch_in, output_dim, kernel_size=(k, k), padding=k // 2)
        # identity transform at initialisation (head.py:632-642)
        with torch.no_grad():
            self.linear.weight.zero_()
            self.linear.bias.zero_()
            if output_dim == 6:
                self.linear.bias[0] = 1
                self.linear.bias[4] = 1
            elif output_dim == 4:
                self.linear.bias[0] = 1
                self.linear.bias[2] = 1
        self.output_dim = output_dim
        self._packed_cache = {}
        self._new_net_id()
        if use_cuda:
            self.conv.cuda()
            self.linear.cuda()

    def _new_net_id(self):
        """Key of this net's entries in the per-device spectra store: a process-wide counter, never reused (``id()`` values
        are, after garbage collection), with a finalizer that drops the entries when the net goes away."""
        import weakref
        self._net_id = next(_NET_IDS)
        weakref.finalize(self, _drop_spectra_of, self._net_id)

    def __deepcopy__(self, memo):
        """A copy is a net of its own: fresh id (sharing the original's made the two evict each other's spectra on every
        alternating call, and the original's finalizer dropped the clone's entries - ADVICE r4), no cached packings."""
        import copy
        cls = self.__class__
        clone = cls.__new__(cls)
        memo[id(self)] = clone
        for k, v in self.__dict__.items():
            if k not in ("_net_id", "_packed_cache"):
                setattr(clone, k, copy.deepcopy(v, memo))
        clone._packed_cache = {}
        clone._new_net_id()
        return clone

    def __setstate__(self, state):
        super(TransformationNet, self).__setstate__(state)
        self._packed_cache = {}
        self._new_net_id()

    def freeze_bn(self):
        for layer in self.modules():
            if isinstance(layer, nn.BatchNorm2d):
                layer.eval()

    def check_ready(self):
        """The HIP kernels need the parameters on a HIP device and eval-mode BatchNorm; said here, before ``packed`` /
        ``spectra`` run torch operators on them (CPU-resident parameters used to surface as torch errors)."""
        dev = self.linear.weight.device
        if dev.type != "cuda":
            raise RuntimeError("TransformationNet parameters are on {}: move the model to the HIP device "
                               "(no CPU fallback)".format(dev))
        if self.training and any(isinstance(m, nn.BatchNorm2d) and m.training for m in self.modules()):
            raise RuntimeError("TransformationNet is in training mode: the HIP path implements eval-mode "
                               "BatchNorm (running statistics) only; call .eval()")
        return dev

    def _state_key(self):
        ts = list(self.parameters()) + list(self.buffers())
        return tuple((t.data_ptr(), t._version, str(t.device)) for t in ts)

    def _folded(self):
        """Eval-mode BatchNorm folded into the convolutions, in float64: [(w [Cout,Cin,k,k], b [Cout])] x 3
        (head.py:619-629).  Used for the range analysis of the split-fp16 path only; the kernels fold in fp32."""
        out = []
        for conv, bn in ((self.conv[0], self.conv[1]), (self.conv[3], self.conv[4]), (self.linear, None)):
            w, b = conv.weight.detach().double(), conv.bias.detach().double()
            if bn is not None:
                s = bn.weight.detach().double() / torch.sqrt(bn.running_var.detach().double() + bn.eps)
                w = w * s.view(-1, 1, 1, 1)
                b = (b - bn.running_mean.detach().double()) * s + bn.bias.detach().double()
            out.append((w, b))
        return out

    def range_plan(self):
        """Power-of-two scales of the split-fp16 ("f16x3") path, chosen so that NO value can leave the fp16 range for
        finite inputs and every operand keeps its 22 bits.

        Every fp32 value x of the path is stored as fp16 hi + lo of x * 2^e; hi must stay below 65504 and, for the full
        22 bits, above 2^-3 (so that lo is a normal fp16 number).  Exponents are per CHANNEL:
          * activations: the TransformNet input is L2-normalised over its 225 channels (head.py:650), so by
            Cauchy-Schwarz |conv1[o]| <= B1[o] = 7 * ||w1[o]||_2 + |b1[o]| at every location (49 taps, each seeing a vector
            of norm <= 1); then |conv2[o]| <= B2[o] = min(sum_c B1[c] * ||w2[o,c]||_1, 5 * ||B1||_2 * ||w2[o]||_2) + |b2[o]|
            (ReLU only shrinks).  Channel o of a layer's output is stored with the exponent that puts its bound at <= 2^15:
            overflow is impossible, values down to 2^-18 of the bound keep all 22 bits (typical activations sit 2^7 ..
            2^13 below their bound);
          * weights: w[o][c] meets the input channel c stored with exponent e_in[c], so it is packed as
            w[o][c] * 2^(-e_in[c] + e_w[o]) with e_w[o] the largest exponent keeping that row's maximum <= 16384.
        Channel-wise rescalings of the network (BatchNorm scales differing by orders of magnitude, compensated in the
        next layer) therefore cost no precision: the exponents undo them up to a factor of two.
        Returns dict(in_exp=[3 int tensors], out_exp=[2 int tensors, None], weight_exp=[3 int tensors], bounds=(B1, B2))."""
        (w1, b1), (w2, b2), (w3, b3) = self._folded()

        def floor_log2(x):     # float64 tensor -> int tensor, 0 where x is 0 / non-finite
            ok = torch.isfinite(x) & (x > 0)
            e = torch.floor(torch.log2(torch.where(ok, x, torch.ones_like(x))))
            return torch.where(ok, e, torch.zeros_like(e)).clamp(-60, 60).to(torch.int32)

        def out_exp(bound):
            return floor_log2(32768.0 / bound.clamp_min(1e-300)) * (bound > 0).to(torch.int32)

        def weight_exp(w, in_exp):
            eff = w * torch.exp2(-in_exp.double()).view(1, -1, 1, 1)
            amax = eff.abs().amax(dim=(1, 2, 3))
            return floor_log2(16384.0 / amax.clamp_min(1e-300)) * (amax > 0).to(torch.int32)

        lib = _lib.load()
        bound1 = 7.0 * w1.flatten(1).norm(dim=1) + b1.abs()
        l1 = (w2.abs().sum(dim=(2, 3)) * bound1.view(1, -1)).sum(dim=1)
        l2 = 5.0 * bound1.norm() * w2.flatten(1).norm(dim=1)
        bound2 = torch.minimum(l1, l2) + b2.abs()
        e0 = torch.full((225,), lib.os2d_rnorm_exp(), dtype=torch.int32, device=w1.device)
        e1, e2 = out_exp(bound1), out_exp(bound2)
        return dict(in_exp=[e0, e1, e2], out_exp=[e1, e2, None],
                    weight_exp=[weight_exp(w1, e0), weight_exp(w2, e1), weight_exp(w3, e2)], bounds=(bound1, bound2))

    def packed(self, precision=None):
        """Packed TransformNet for the kernels: (w1, b1, w2, b2, w3, b3).
        precision "f32": os2d_pack_conv layouts; "f16x3": the split-half layout of os2d_pack_conv_f16x3 with the scales
        of ``range_plan``.  Cached until a parameter changes; the cache entry carries an event so that other streams never
        read half-written buffers."""
        precision = resolve_precision(precision)
        if precision in ("f16x2", "fft", "fftx3"):
            precision = "f16x3"        # same packed weights and scales (f16x2 skips the lo halves of layer 1; fft replaces
                                       # layer 1 by ``spectra`` and keeps its bias / output scales)
        elif precision == "fft32":
            precision = "f32"          # fp32 packing; layer 1's weights are replaced by ``spectra``, its folded bias is used
        key = (precision,) + self._state_key()
        dev = self.linear.weight.device
        cached = self._packed_cache.get(precision)
        if cached is not None and cached.key == key:
            return cached.get(dev)
        lib = _lib.load()
        self.check_ready()
        with torch.cuda.device(dev):
            stream = _lib.current_stream(dev)
            cur = torch.cuda.current_stream(dev)
            out = []
            P = self.output_dim
            plan = self.range_plan() if precision == "f16x3" else None
            layers = ((1, self.conv[0], self.conv[1]), (2, self.conv[3], self.conv[4]), (3, self.linear, None))
            for layer, conv, bn in layers:
                w = _require_device_f32(conv.weight.detach(), "conv weight")
                b = _require_device_f32(conv.bias.detach(), "conv bias")
                pb = torch.zeros(lib.os2d_packed_bias_floats(layer), dtype=torch.float32, device=dev)
                if bn is not None:
                    bnp = [_require_device_f32(t.detach(), "bn") for t in (bn.weight, bn.bias, bn.running_mean, bn.running_var)]
                    eps = float(bn.eps)
                else:
                    bnp, eps = [None] * 4, 0.0
                if precision == "f16x3":
                    exps = [plan[k][layer - 1] for k in ("weight_exp", "in_exp", "out_exp")]
                    exps = [e.to(dev).contiguous() if e is not None else None for e in exps]
                    pw = torch.empty(lib.os2d_packed_conv_bytes(layer, PRECISIONS[precision]), dtype=torch.uint8, device=dev)
                    _lib.check(lib.os2d_pack_conv_f16x3(layer, P, _lib.ptr(w), _lib.ptr(b), *[_lib.ptr(t) for t in bnp],
                                                        ctypes.c_float(eps), *[_lib.ptr(e) for e in exps],
                                                        _lib.ptr(pw), _lib.ptr(pb), stream), "os2d_pack_conv_f16x3")
                    for e in exps:
                        if e is not None:
                            e.record_stream(cur)
                else:
                    pw = torch.empty(lib.os2d_packed_conv_floats(layer), dtype=torch.float32, device=dev)
                    _lib.check(lib.os2d_pack_conv(layer, P, _lib.ptr(w), _lib.ptr(b), *[_lib.ptr(t) for t in bnp],
                                                  ctypes.c_float(eps), _lib.ptr(pw), _lib.ptr(pb), stream), "os2d_pack_conv")
                out += [pw, pb]
            result = tuple(out)
            self._packed_cache[precision] = _StreamOrdered(key, result, dev)
        return result

    def spectra(self, H, W, split=False):
        """Frequency-domain form of the 7x7 layer for an H x W map: (wspec, tables A, tables B, nbins), cached per TRANSFORM
        size (P, Q) - which the many map sizes of a dataset share - until a parameter changes, least recently used first out
        above ``spectra_cache_cap_bytes()`` (event-tracked like ``packed``).  None only when the library has no transform
        plan for the map.
          ``split`` False (precision "fft" / "fft32"): wspec = complex64 weight spectra for os2d_spectral_gemm, tables = the two
            twiddle tables (twQ, twP) of the in-LDS FFTs (fft.hip; sizes 2^a 3^b or 42 / 84, maps beyond one in-LDS transform
            are cut into overlap-save tiles, os2d_fft_tiles);
          ``split`` True (precision "fftx3"): wspec = the weight spectra pre-split into fp16 hi + lo with per-row power-of-two
            scales (os2d_spectral_gemm_f16) in the bin order of the matrix-product transforms (os2d_dft_sizes: any P % 4 == 0,
            even Q; tiles beyond 64 x 94), tables A = the four operand matrices of os2d_dft_matrices_build, tables B = None.
        The BatchNorm-folded 7x7 filters are centred on the origin of the P x Q grid (tap (t, s) at ((3 - t) mod P, (3 - s) mod
        Q): the circular convolution then IS the zero-padded correlation of head.py:619 for the first H x W samples) and
        transformed once in float64 on the device (a 7-term DFT per axis, spectra_pack.hip) from exact float64 tables."""
        lib = _lib.load()
        self.check_ready()
        cP, cQ, cN = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        if split:
            rc = lib.os2d_dft_sizes(int(H), int(W), ctypes.byref(cP), ctypes.byref(cQ), ctypes.byref(cN), None)
        else:
            rc = lib.os2d_fft_sizes(int(H), int(W), ctypes.byref(cP), ctypes.byref(cQ), ctypes.byref(cN))
        if rc != 0:
            return None
        P, Q, nbins = cP.value, cQ.value, cN.value
        dev = self.linear.weight.device
        key = self._state_key()
        slot = (P, Q, bool(split))
        store = _SpectraStore.of(dev)
        cached = store.get(self._net_id, slot, key)
        if cached is not None:
            return cached.get(dev)
        store.drop_stale(self._net_id, key)                      # a parameter changed: every cached size of this net is stale
        with torch.cuda.device(dev), torch.no_grad():
            (w1, _), _, _ = self._folded()                       # float64 [128,225,7,7]: the BatchNorm fold

            def table64(n):
                m = torch.arange(n, dtype=torch.float64)
                ang = m * (-2.0 * torch.pi / n)
                return torch.stack([torch.cos(ang), torch.sin(ang)], 1).to(dev).contiguous()
            wfold = w1.to(dev).contiguous()
            tp64, tq64 = table64(P), table64(Q)
            scratch = torch.empty(1024, dtype=torch.uint8, device=dev)
            stream = _lib.current_stream(dev)
            if split:
                wspec = torch.empty(lib.os2d_spectral_weight16_bytes(225, nbins), dtype=torch.uint8, device=dev)
                _lib.check(lib.os2d_spectral_weights_build_dft(_lib.ptr(wfold), _lib.ptr(tp64), _lib.ptr(tq64), 225, 128, P, Q, nbins,
                                                               _lib.ptr(wspec), _lib.ptr(scratch), stream),
                           "os2d_spectral_weights_build_dft")
                mats = torch.empty(lib.os2d_dft_matrices_bytes(P, Q), dtype=torch.uint8, device=dev)
                _lib.check(lib.os2d_dft_matrices_build(_lib.ptr(tp64), _lib.ptr(tq64), P, Q, _lib.ptr(mats), stream),
                           "os2d_dft_matrices_build")
                result = (wspec, mats, None, nbins)
            else:
                wspec = torch.empty(lib.os2d_spectral_weight_bytes(225, 128, nbins) // 4, dtype=torch.float32, device=dev)
                _lib.check(lib.os2d_spectral_weights_build(_lib.ptr(wfold), _lib.ptr(tp64), _lib.ptr(tq64), 225, 128, P, Q, nbins,
                                                           0, _lib.ptr(wspec), _lib.ptr(scratch), stream), "os2d_spectral_weights_build")

                def table(n):
                    m = torch.arange(n, dtype=torch.float64)
                    ang = -2.0 * torch.pi * m / n
                    return torch.stack([torch.cos(ang), torch.sin(ang)], 1).float().to(dev).contiguous()
                result = (wspec, table(Q), table(P), nbins)
            cur = torch.cuda.current_stream(dev)
            for t in (wfold, tp64, tq64, scratch):
                t.record_stream(cur)
            store.put(self._net_id, slot, _StreamOrdered(key, result, dev))
        return result

    @property
    def _spectra_cache(self):
        """{slot: entry} view of this net's cached weight spectra (the store itself is per device, shared by all nets)."""
        dev = self.linear.weight.device
        if dev.type != "cuda":
            return {}
        store = _SpectraStore.of(dev)
        return _SpectraView(store, self._net_id)

    def forward(self, corr_maps, precision="f32"):
        """corr_maps [N,225,H,W] -> transform parameters [N,P,H,W] (reference head.py:648-655): input normalisation + the
        three convolution kernels of the selected arithmetic ("f32": os2d_corr_normalize + os2d_transform_conv;
        "f16x3" / "f16x2": os2d_corr_normalize_f16x3 + os2d_transform_conv_f16x3 - the kernels the fused head runs)."""
        corr_maps = _require_device_f32(corr_maps, "corr_maps")
        if corr_maps.dim() != 4 or corr_maps.size(1) != 225:
            raise RuntimeError("corr_maps must be [N,225,H,W], got {}".format(tuple(corr_maps.shape)))
        if torch.is_grad_enabled() and (corr_maps.requires_grad or any(p.requires_grad for p in self.parameters())) and self.training:
            raise RuntimeError("autograd through the HIP TransformNet is not implemented (training is out of scope)")
        precision = resolve_precision(precision)
        lib = _lib.load()
        N, _, H, W = corr_maps.shape
        dev = corr_maps.device
        P = self.output_dim
        w1, b1, w2, b2, w3, b3 = self.packed(precision)
        out = torch.empty(N, P, H, W, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            stream = _lib.current_stream(dev)
            if precision in FP32_MODES:      # (the frequency-domain modes differ from their families only inside the fused head)
                plane = lib.os2d_plane_floats(H, W)
                r = torch.empty(N * 226 * plane, dtype=torch.float32, device=dev)
                h1 = torch.empty(N * 128 * plane, dtype=torch.float32, device=dev)
                h2 = torch.empty(N * 64 * plane, dtype=torch.float32, device=dev)
                _lib.check(lib.os2d_corr_normalize(_lib.ptr(corr_maps), _lib.ptr(r), N, H, W, stream), "os2d_corr_normalize")
                _lib.check(lib.os2d_transform_conv(1, _lib.ptr(r), _lib.ptr(w1), _lib.ptr(b1), _lib.ptr(h1), N, P, H, W, stream), "conv1")
                _lib.check(lib.os2d_transform_conv(2, _lib.ptr(h1), _lib.ptr(w2), _lib.ptr(b2), _lib.ptr(h2), N, P, H, W, stream), "conv2")
                _lib.check(lib.os2d_transform_conv(3, _lib.ptr(h2), _lib.ptr(w3), _lib.ptr(b3), _lib.ptr(out), N, P, H, W, stream), "conv3")
            else:
                terms1 = 2 if precision == "f16x2" else 3
                r = torch.empty(N * lib.os2d_shb_bytes(225, H, W), dtype=torch.uint8, device=dev)
                h1 = torch.empty(N * lib.os2d_shb_bytes(128, H, W), dtype=torch.uint8, device=dev)
                h2 = torch.empty(N * lib.os2d_shb_bytes(64, H, W), dtype=torch.uint8, device=dev)
                status = torch.zeros(1, dtype=torch.int32, device=dev)
                _lib.check(lib.os2d_corr_normalize_f16x3(_lib.ptr(corr_maps), _lib.ptr(r), N, H, W, stream), "os2d_corr_normalize_f16x3")
                for layer, src, wp, bp, dst, terms in ((1, r, w1, b1, h1, terms1), (2, h1, w2, b2, h2, 3), (3, h2, w3, b3, out, 3)):
                    _lib.check(lib.os2d_transform_conv_f16x3(layer, _lib.ptr(src), _lib.ptr(wp), _lib.ptr(bp), _lib.ptr(dst), N, P,
                                                             H, W, terms, _lib.ptr(status), stream), "os2d_transform_conv_f16x3")
                self.last_status = status      # device int32: OS2D_STATUS_F16_RANGE if an activation left the fp16 range
        return out


class Os2dAlignment(nn.Module):
    """reference head.py:43-193.  Owns the TransformNet and the transformation-model flags.  The reference's
    ``forward`` returns a materialised [N,H,W,15,15,2] grid tensor; here the grid never exists: the transform
    assembly (head.py:81-153), F.affine_grid (head.py:184) and everything downstream are fused in
    os2d_sample_decode, driven from ``Os2dHead.forward``."""

    def __init__(self, do_simple_affine, is_cuda, use_inverse_geom_model):
        super(Os2dAlignment, self).__init__()
        self.model_type = "affine" if not do_simple_affine else "simple_affine"
        self.use_inverse_geom_model = use_inverse_geom_model
        transform_net_output_dim = 6 if self.model_type == "affine" else 4
        self.out_grid_size = FeatureMapSize(w=TEMPLATE, h=TEMPLATE)
        self.reference_feature_map_size = FeatureMapSize(w=TEMPLATE, h=TEMPLATE)
        self.network_stride = FeatureMapSize(w=1, h=1)
        self.network_receptive_field = FeatureMapSize(w=TEMPLATE, h=TEMPLATE)
        self.input_feature_dim = TEMPLATE * TEMPLATE
        self.parameter_regressor = TransformationNet(output_dim=transform_net_output_dim, use_cuda=is_cuda,
                                                     normalization="batchnorm", kernel_sizes=[7, 5],
                                                     channels=[128, 64], input_feature_dim=self.input_feature_dim)

    @property
    def num_transform_params(self):
        return self.parameter_regressor.output_dim

    def prepare_transform_parameters_for_grid_sampler(self, transform_parameters):
        """reference head.py:81-153: [N,P,H,W] regressed parameters -> theta [N*H*W,2,3] (full / simplified affine,
        optionally inverted), with os2d_alignment_grids."""
        return self._grids(transform_parameters, want_theta=True, want_grids=False)[0]

    def _grids(self, params, want_theta, want_grids):
        params = _require_device_f32(params, "transform_parameters")
        N, P, H, W = params.shape
        assert P == self.num_transform_params, \
            "Tranformation parameter vector has to be of dimension {0}, have {1} instead".format(self.num_transform_params, P)
        lib = _lib.load()
        dev = params.device
        theta = torch.empty(N * H * W, 2, 3, dtype=torch.float32, device=dev) if want_theta else None
        grids = torch.empty(N, H, W, TEMPLATE, TEMPLATE, 2, dtype=torch.float32, device=dev) if want_grids else None
        with torch.cuda.device(dev):
            _lib.check(lib.os2d_alignment_grids(_lib.ptr(params), N, H, W, P, 1 if self.use_inverse_geom_model else 0,
                                                _lib.ptr(theta), _lib.ptr(grids), _lib.current_stream(dev)),
                       "os2d_alignment_grids")
        return theta, grids

    def forward(self, corr_maps, precision="f32"):
        """reference head.py:155-193: corr_maps [N,225,H,W] -> resampling grids in local coordinates
        [N,H,W,15,15,2] (TransformNet -> theta -> F.affine_grid(align_corners=True)).  ``Os2dHead.forward`` never
        materialises these grids (they are fused into os2d_sample_decode); this entry point serves callers of the class
        API that want them (6.2 MB per class at 60x80)."""
        assert corr_maps.size(1) == self.input_feature_dim, \
            "The dimension 1 of corr_maps={0} should be equal to self.input_feature_dim={1}".format(corr_maps.size(1), self.input_feature_dim)
        transform_parameters = self.parameter_regressor(corr_maps, precision=precision)
        return self._grids(transform_parameters, want_theta=False, want_grids=True)[1]


# --------------------------------------------------------------------------------------------- head creator
class Os2dHeadCreator(nn.Module):
    """reference head.py:204-268: a sub-module of the model (it owns the TransformNet parameters) that creates
    ``Os2dHead`` objects holding class features."""

    def __init__(self, aligner, feature_map_stride, feature_map_receptive_field):
        super(Os2dHeadCreator, self).__init__()
        self.aligner = aligner
        rec_field, stride = self.get_rec_field_and_stride_after_concat_nets(
            feature_map_receptive_field, feature_map_stride,
            self.aligner.network_receptive_field, self.aligner.network_stride)
        self.feature_map_stride = feature_map_stride
        self.feature_map_receptive_field = feature_map_receptive_field
        self.box_grid_generator_image_level = BoxGridGenerator(box_size=rec_field, box_stride=stride)
        self.box_grid_generator_feature_map_level = BoxGridGenerator(box_size=self.aligner.network_receptive_field,
                                                                     box_stride=self.aligner.network_stride)

    @staticmethod
    def get_rec_field_and_stride_after_concat_nets(receptive_field_netA, stride_netA, receptive_field_netB, stride_netB):
        """net(x) = netB(netA(x)): rf = strideA*(rfB-1)+rfA, stride = strideA*strideB (reference head.py:223-238)."""
        if isinstance(receptive_field_netA, FeatureMapSize):
            f = Os2dHeadCreator.get_rec_field_and_stride_after_concat_nets
            rw, sw = f(receptive_field_netA.w, stride_netA.w, receptive_field_netB.w, stride_netB.w)
            rh, sh = f(receptive_field_netA.h, stride_netA.h, receptive_field_netB.h, stride_netB.h)
            return FeatureMapSize(w=rw, h=rh), FeatureMapSize(w=sw, h=sh)
        return stride_netA * (receptive_field_netB - 1) + receptive_field_netA, stride_netA * stride_netB

    @staticmethod
    def resize_feature_maps_to_reference_size(ref_size, feature_maps):
        """reference head.py:241-259, as a HIP kernel: returns the resized (NOT normalised) maps [B,C,15,15]."""
        q15, _ = _prepare_class_maps(feature_maps, normalise=False)
        return q15

    def create_os2d_head(self, class_feature_maps):
        """class_feature_maps: list of [1,C,h_i,w_i] device tensors (reference head.py:261-268)."""
        q15, qp = _prepare_class_maps(class_feature_maps, normalise=True)
        return Os2dHead(q15, self.aligner, self.box_grid_generator_image_level,
                        self.box_grid_generator_feature_map_level, _prepared=qp,
                        _stride=self.feature_map_stride, _rec_field=self.feature_map_receptive_field)


def _prepare_class_maps(class_feature_maps, normalise=True):
    """Run os2d_class_prepare on each [1,C,h,w] (or [C,h,w]) map.  Returns (q15 [B,C,15,15], qp [B,C,256])."""
    lib = _lib.load()
    if isinstance(class_feature_maps, torch.Tensor):
        class_feature_maps = [m.unsqueeze(0) for m in class_feature_maps]
    if len(class_feature_maps) == 0:
        raise RuntimeError("need at least one class feature map")
    maps = []
    for fm in class_feature_maps:
        fm = _require_device_f32(fm, "class feature map")
        if fm.dim() == 4:
            assert fm.size(0) == 1, "Can process only batches of size 1, but have {0}".format(fm.size(0))
            fm = fm[0]
        maps.append(fm)
    C = maps[0].size(0)
    dev = maps[0].device
    B = len(maps)
    for fm in maps:
        if fm.size(0) != C:
            raise RuntimeError("class feature maps disagree on the feature dimension: {} vs {}".format(fm.size(0), C))
        if fm.device != dev:
            raise RuntimeError("class feature maps are on different devices: {} vs {}".format(fm.device, dev))
    q15 = torch.empty(B, C, TEMPLATE, TEMPLATE, dtype=torch.float32, device=dev)
    qp = torch.empty(B, C, QROWS, dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        stream = _lib.current_stream(dev)
        # ONE launch for all classes (the reference loops them, head.py:261-268): pointer + size tables on the device
        for b0 in range(0, B, 65535):
            chunk = maps[b0:b0 + 65535]
            ptrs = torch.tensor([m.data_ptr() for m in chunk], dtype=torch.int64).to(dev, non_blocking=False)
            sizes = torch.tensor([[m.size(1), m.size(2)] for m in chunk], dtype=torch.int32).to(dev, non_blocking=False)
            ws = torch.empty(lib.os2d_class_prepare_workspace_floats(len(chunk), C), dtype=torch.float32, device=dev)
            _lib.check(lib.os2d_class_prepare_batch(_lib.ptr(ptrs), _lib.ptr(sizes), len(chunk), C, 1 if normalise else 0,
                                                    _lib.ptr(q15[b0:b0 + len(chunk)]), _lib.ptr(qp[b0:b0 + len(chunk)]),
                                                    _lib.ptr(ws), stream), "os2d_class_prepare_batch")
            cur = torch.cuda.current_stream(dev)
            ptrs.record_stream(cur), sizes.record_stream(cur), ws.record_stream(cur)
            for m in chunk:
                m.record_stream(cur)
    return q15, qp


# --------------------------------------------------------------------------------------------- head
class Os2dHead(nn.Module):
    """reference head.py:271-435.  Holds the L2-normalised 15x15 class feature maps of B classes and computes, for
    a batch of image feature maps, the localisation / recognition outputs of every (image, class) pair.

    Differences from the reference, by design:
      * one ``forward`` handles all B classes in a single library call (the reference's evaluation loops B=1
        heads, os2d/engine/evaluate.py:323-331) - the per-class results are identical;
      * eval mode only: ``output_recognition_transform_detached`` is the same tensor as ``output_recognition``
        (as in the reference when no gradient is required, head.py:400-402).
    """

    def __init__(self, class_feature_maps, aligner, box_grid_generator_image_level,
                 box_grid_generator_feature_map_level, pool_border_width=2, _prepared=None, _stride=None,
                 _rec_field=None):
        super(Os2dHead, self).__init__()
        if pool_border_width != 2:
            raise RuntimeError("the HIP resampling kernel is built for pool_border_width=2 (the only value the "
                               "reference uses, head.py:280)")
        class_feature_maps = _require_device_f32(class_feature_maps, "class_feature_maps")
        if class_feature_maps.dim() != 4 or class_feature_maps.size(2) != TEMPLATE or class_feature_maps.size(3) != TEMPLATE:
            raise RuntimeError("class_feature_maps must be [B,C,15,15], got {}".format(tuple(class_feature_maps.shape)))
        if _prepared is None:
            # public constructor, as in the reference: maps are resized but not yet normalised (head.py:293)
            class_feature_maps, _prepared = _prepare_class_maps(class_feature_maps, normalise=True)
        self.class_feature_maps = class_feature_maps          # normalised, [B,C,15,15]
        self._qp = _prepared                                  # GEMM operand [B,C,256]
        self._qs = None                                       # its fp16 hi/lo split (f16x3 mode), built on first use
        self.class_batch_size = self.class_feature_maps.size(0)
        self.box_grid_generator_image_level = box_grid_generator_image_level
        self.box_grid_generator_feature_map_level = box_grid_generator_feature_map_level
        # pooling mask (head.py:296-302): informational - the kernel hard-codes the 11x11 inner window
        mask = torch.zeros(self.class_batch_size, 1, TEMPLATE, TEMPLATE, dtype=torch.float32,
                           device=self.class_feature_maps.device)
        mask[:, :, pool_border_width:TEMPLATE - pool_border_width, pool_border_width:TEMPLATE - pool_border_width] = 1
        self.class_pool_mask = mask / mask.sum(dim=(2, 3), keepdim=True)
        self.aligner = aligner
        self.last_precision = None
        self.precision = None      # None: follow $OS2D_PRECISION (default "fftx3"); or one of PRECISIONS
        # sticky status word of the split-fp16 kernels in mapped pinned host memory (one per device, process lifetime:
        # ``device_status_word``): the kernels store to it only when an activation leaves the fp16 range (impossible for
        # finite inputs, see TransformationNet.range_plan), the host reads it without synchronising
        self._status = None
        self.strict_range = os.environ.get("OS2D_STRICT_RANGE", "0") not in ("0", "", "false")
        box = box_grid_generator_image_level
        self._stride = int(box.box_stride.w)
        # image-level box = stride*(15-1) + receptive field (head.py:223-238)
        self._rec_field = int(box.box_size.w - self._stride * (TEMPLATE - 1))
        if box.box_stride.w != box.box_stride.h or box.box_size.w != box.box_size.h:
            raise RuntimeError("anisotropic strides / receptive fields are not supported by the HIP head")

    def _split_class_operand(self):
        """qs [B, C/8, hi|lo, 256] x 8 halves for the f16x3 correlation (os2d_class_split), built on first use; carries
        the event of the kernel that wrote it so that other streams can consume it safely."""
        dev = self._qp.device
        if self._qs is None:
            lib = _lib.load()
            B, C = self._qp.size(0), self._qp.size(1)
            qs = torch.empty(lib.os2d_class_split_bytes(B, C), dtype=torch.uint8, device=dev)
            with torch.cuda.device(dev):
                _lib.check(lib.os2d_class_split(_lib.ptr(self._qp), _lib.ptr(qs), B, C, _lib.current_stream(dev)),
                           "os2d_class_split")
            self._qs = _StreamOrdered(None, qs, dev)
        return self._qs.get(dev)

    def prepare(self, precision=None):
        """Build everything ``forward`` caches on first use (packed TransformNet of the selected arithmetic, the fp16
        split of the class operand) on the CURRENT stream.  Callers that fan out over several streams (the pyramid
        runner) call this first; the caches are event-tracked either way."""
        precision = resolve_precision(precision or self.precision)
        self.aligner.parameter_regressor.packed(precision)
        if precision not in FP32_MODES:
            self._split_class_operand()
        return self

    def _status_word(self):
        if self._status is None:
            self._status = device_status_word(self._qp.device, owner=self)
        return self._status

    def range_status(self, synchronize=False):
        """Sticky status bits raised by the split-fp16 kernels (OS2D_STATUS_F16_RANGE = 1: an activation left the fp16
        range).  Without ``synchronize`` this reads the mapped host word as it is (kernels in flight may still set it)."""
        if synchronize:
            torch.cuda.synchronize(self._qp.device)
        return int(self._status_word()[0])

    def clear_range_status(self):
        """Reset the sticky word (the caller has seen it).  Kernels of calls still in flight may raise it again."""
        self._status_word()[0] = 0

    @classmethod
    def cat(cls, heads):
        """Merge per-class heads (as the reference's evaluation creates them) into one class-batched head."""
        h0 = heads[0]
        q15 = torch.cat([h.class_feature_maps for h in heads], 0)
        qp = torch.cat([h._qp for h in heads], 0)
        return cls(q15, h0.aligner, h0.box_grid_generator_image_level, h0.box_grid_generator_feature_map_level,
                   _prepared=qp)

    def forward(self, feature_maps, out=None, stage_events=None, precision=None, strict_range=None, route_pairs=None):
        """feature_maps [A,C,H,W] -> (loc [A,B,4,H,W], cls [A,B,1,H,W], cls_detached (same), corners [A,B,8,H,W]).

        ``out``: optional preallocated (loc, cls, corners) device tensors of exactly those shapes (contiguous) - used
        by the class-sharded wrapper to let the kernels write straight into the all-gather buffer.
        ``stage_events``: optional ctypes array of 13 event handles for os2d_head_forward_profiled (bench.py).
        Non-finite input (a NaN / Inf feature; nothing else gets past the range plan): the split-fp16 kernels raise a range word
        in the call's workspace and the LAST kernel of the same call writes NaN into every output of the affected image - where
        the reference has NaN around the offending cells (head.py:339, 650: torch.relu / norm propagate it) - and raises the
        sticky host word (``range_status``).  No synchronisation, no effect on later calls.
        ``strict_range`` (default $OS2D_STRICT_RANGE, off): synchronise after a split-fp16 call and, if the word was raised,
        re-run the call in exact fp32 (the reference's own NaN pattern) before returning.
        ``route_pairs``: the number of (image, class) pairs the ROUTE decision of the frequency-domain modes is made for
        (default: this call's own A * B).  A class-sharded caller passes the GLOBAL pair count, so that a ragged tail rank
        holding fewer than ``FFT_MIN_PAIRS`` classes takes the same arithmetic route as the others and the sharded result
        stays bit-equal to the unsharded one (os2d_amd/parallel.py); ``precision="fftx3!"`` / ``"fft!"`` pins the
        frequency-domain route regardless of the pair count."""
        feature_maps = _require_device_f32(feature_maps, "feature_maps")
        if feature_maps.dim() != 4:
            raise RuntimeError("feature_maps must be [A,C,H,W], got {}".format(tuple(feature_maps.shape)))
        if torch.is_grad_enabled() and feature_maps.requires_grad:
            raise RuntimeError("autograd through the HIP head is not implemented (training is out of scope): "
                               "call under torch.no_grad()")
        A, C, H, W = feature_maps.shape
        B = self.class_batch_size
        class_feature_dim = self.class_feature_maps.size(1)
        assert C == class_feature_dim, \
            "Feature dimensionality of input={0} and class={1} feature maps has to equal".format(C, class_feature_dim)
        if self._qp.device != feature_maps.device:
            raise RuntimeError("class features are on {} but feature_maps on {}".format(self._qp.device, feature_maps.device))
        lib = _lib.load()
        dev = feature_maps.device
        regressor = self.aligner.parameter_regressor
        P = regressor.output_dim
        precision, pinned = _resolve(precision or self.precision)
        regressor.check_ready()                  # device / eval-mode errors before any torch op can trip over them
        spectra = None
        if W > MAX_W_DIRECT7:
            # wider than the direct 7x7 kernels take (the reference has no width limit, head.py:619): the layer runs in the
            # frequency domain - tiled to any width - whatever the class batch, in the arithmetic family that was asked for
            precision = {"f32": "fft32", "f16x3": "fftx3", "f16x2": "fftx3"}.get(precision, precision)
            pinned = True
        if precision in FFT_MODES:
            # the frequency-domain 7x7 layer pays off from 7 pairs on (it streams 0.65 GB of weight spectra per call);
            # below that the direct f16x3 kernel does the layer.  The decision is made on ``route_pairs`` when given
            # (class-sharded callers: the global count), and not at all when the route is pinned
            pairs = A * B if route_pairs is None else int(route_pairs)
            spectra = regressor.spectra(H, W, split=precision == "fftx3") if (pinned or pairs >= FFT_MIN_PAIRS) else None
            if spectra is None:
                precision = "f32" if precision == "fft32" else "f16x3"       # the direct kernels of the same arithmetic family
        self.last_precision = precision          # the arithmetic that actually ran (bench.py / tests)
        w1, b1, w2, b2, w3, b3 = regressor.packed(precision)
        if out is None:
            loc = torch.empty(A, B, 4, H, W, dtype=torch.float32, device=dev)
            cls = torch.empty(A, B, 1, H, W, dtype=torch.float32, device=dev)
            corners = torch.empty(A, B, 8, H, W, dtype=torch.float32, device=dev)
        else:
            loc, cls, corners = out
            for t, k in ((loc, 4), (cls, 1), (corners, 8)):
                if tuple(t.shape) != (A, B, k, H, W) or not t.is_contiguous() or t.device != dev or t.dtype != torch.float32:
                    raise RuntimeError("out tensors must be contiguous float32 [A,B,{},H,W] on {}".format(k, dev))
        full = ctypes.c_size_t()
        one = ctypes.c_size_t()
        _lib.check(lib.os2d_head_workspace_bytes_ex(A, B, C, H, W, P, PRECISIONS[precision], ctypes.byref(full)), "os2d_head_workspace_bytes_ex")
        _lib.check(lib.os2d_head_workspace_bytes_ex(A, 1, C, H, W, P, PRECISIONS[precision], ctypes.byref(one)), "os2d_head_workspace_bytes_ex")
        with torch.cuda.device(dev):     # hipFuncSetAttribute / launches act on the CURRENT device
            ws = get_workspace(dev, full.value, one.value)
            status = self._status_word() if precision not in FP32_MODES else None
            _lib.check(lib.os2d_head_forward_ex(
                _lib.ptr(feature_maps), _lib.ptr(self._qp), _lib.ptr(w1), _lib.ptr(b1), _lib.ptr(w2), _lib.ptr(b2),
                _lib.ptr(w3), _lib.ptr(b3), A, B, C, H, W, P, 1 if self.aligner.use_inverse_geom_model else 0,
                self._stride, self._rec_field, _lib.ptr(loc), _lib.ptr(cls), _lib.ptr(corners),
                _lib.ptr(ws), ws.numel(), _lib.current_stream(dev), PRECISIONS[precision],
                _lib.ptr(self._split_class_operand()) if precision not in FP32_MODES else None, stage_events, None,
                _lib.host_ptr(status), *([_lib.ptr(t) for t in spectra[:3]] if spectra is not None else [None, None, None])),
                "os2d_head_forward_ex")
        strict = self.strict_range if strict_range is None else strict_range
        if strict and precision not in FP32_MODES:
            torch.cuda.current_stream(dev).synchronize()
            if int(self._status[0]) != 0:
                self.clear_range_status()
                logging.getLogger("OS2D").warning("OS2D head: non-finite input or an activation outside the fp16 range; strict_range "
                                                  "re-runs this call in precision='f32'")
                return self.forward(feature_maps, out=out, stage_events=stage_events, precision="f32")
        return loc, cls, cls, corners


def normalize_feature_map_L2(feature_maps, epsilon=1e-6):
    """reference head.py:597-601.  Provided for API completeness (plain tensor expression; NOT used by the HIP
    path, which folds both normalisations into its GEMM epilogue)."""
    return feature_maps / (feature_maps.norm(dim=1, keepdim=True) + epsilon)
