"""
Launchers for the feature kernels (aps_amd/csrc/feats.hip): host-side argument marshalling only.
"""
import ctypes as C
import os
from dataclasses import dataclass, field
from typing import List, Optional, Tuple

import numpy as np
import torch as th

from aps_amd import _native as nat
from aps_amd.const import EPSILON


@dataclass
class SpectralPlan:
    """What the magnitude branch computes after |X|: power -> [mel] -> [log] -> [row cmvn]"""
    power: int = 1
    mel: Optional["MelBands"] = None
    apply_log: bool = False
    log_eps: float = EPSILON
    log_lower_bound: float = 0.0
    norm_mean: bool = False
    norm_var: bool = False
    cmvn_eps: float = EPSILON


class MelBands(object):
    """Banded form of a mel matrix [M, F] on the device: per row the first non-zero bin, the run
    length and the packed weights.  Any dense matrix is legal (the run then covers everything)."""

    def __init__(self, filters: th.Tensor):
        w = filters.detach().float().cpu().numpy()
        M, F = w.shape
        start = np.zeros(M, dtype=np.int32)
        length = np.zeros(M, dtype=np.int32)
        offset = np.zeros(M, dtype=np.int32)
        packed = []
        pos = 0
        for m in range(M):
            nz = np.nonzero(w[m])[0]
            if nz.size:
                start[m], length[m] = nz[0], nz[-1] - nz[0] + 1
                packed.append(w[m, nz[0]:nz[-1] + 1])
            offset[m] = pos
            pos += int(length[m])
        dev = filters.device
        self.num_mels, self.num_bins = M, F
        self.dense = filters.detach().float().contiguous()  # the GEMM form (training path)
        self.start = th.from_numpy(start).to(dev)
        self.length = th.from_numpy(length).to(dev)
        self.offset = th.from_numpy(offset).to(dev)
        flat = np.concatenate(packed) if packed else np.zeros(1, dtype=np.float32)
        self.weight = th.from_numpy(np.ascontiguousarray(flat, dtype=np.float32)).to(dev)
        self.key = (filters.data_ptr(), filters._version, str(dev))

    @staticmethod
    def cached(module, filters: th.Tensor) -> "MelBands":
        key = (filters.data_ptr(), filters._version, str(filters.device))
        bands = getattr(module, "_aps_bands", None)
        if bands is None or bands.key != key:
            bands = MelBands(filters)
            module._aps_bands = bands
        return bands


class NanGuard(object):
    """Device side of check_valid (aps/transform/asr.py:33-45): the feature kernels bump a device
    counter when they write a NaN, so detection costs no extra pass over the features.
        policy "sync"     : read the counter right after the launch (reference behaviour: the
                            ValueError is raised by the call that produced the NaN; host stalls)
        policy "deferred" : the counter is copied to pinned memory asynchronously (every
                            `period`-th launch: the copy is a 4 us node on the stream) and
                            inspected at a later call / at flush() -- same detection, no stall
        policy "manual"   : the kernels keep counting, the owner reads `count()` when it wants
                            (e.g. after replaying a captured graph, where host reads are illegal)
        policy "off"      : counter not passed to the kernels at all
    """

    period = 16  # deferred: launches between two counter read-backs

    class _PerDevice(object):
        """one device's counter, its pinned mirror and the read-back in flight.  Never released: the counter's
        address is baked into every graph captured with it (replays keep bumping it)."""

        def __init__(self, device):
            self.flag = th.zeros(1, dtype=th.int32, device=device)
            self.host = th.zeros(1, dtype=th.int32).pin_memory()
            self.event = th.cuda.Event()
            self.pending = None
            self.since = 0

    def __init__(self):
        self._dev = {}   # device index -> _PerDevice (a process that alternates devices keeps every count)

    def _state(self, device, create: bool = True):
        dev = th.device(device)
        key = dev.index if dev.index is not None else th.cuda.current_device()
        st = self._dev.get(key)
        if st is None and create:
            st = self._dev[key] = NanGuard._PerDevice(th.device("cuda", key))
        return st

    def known(self, device) -> bool:
        """has this device's counter been created (a stream capture cannot create one)"""
        return self._state(device, create=False) is not None

    def pointer(self, device) -> th.Tensor:
        return self._state(device).flag

    def _raise(self, st, count, shape):
        st.flag.zero_()
        raise ValueError(f"Detect NANs in feature matrices ({count} wavefront rows), " +
                         f"shape = {shape}...")

    def flush(self):
        """every device's launches since its last read-back: fetch the counters now, raise on a non-zero one"""
        for st in list(self._dev.values()):
            if st.pending is None and st.since:
                with th.cuda.device(st.flag.device):
                    st.host.copy_(st.flag, non_blocking=True)
                    st.event.record()
                st.pending, st.since = ("?",), 0
        for st in list(self._dev.values()):
            if st.pending is not None:
                st.event.synchronize()
                shape, st.pending = st.pending, None
                if int(st.host[0]) != 0:
                    self._raise(st, int(st.host[0]), shape)

    def count(self) -> int:
        """synchronising read of the device counters (and reset), summed over the devices used so far"""
        total = 0
        for st in self._dev.values():
            c = int(st.flag.item())
            if c:
                st.flag.zero_()
            total += c
        return total

    def after_launch(self, policy: str, shape, device=None):
        """`device`: where the launch ran (default: the current device)"""
        if policy in ("off", "manual"):
            return
        st = self._state(device if device is not None else th.device("cuda", th.cuda.current_device()), create=False)
        if st is None:
            return
        if policy == "sync":
            count = int(st.flag.item())
            if count:
                self._raise(st, count, shape)
            return
        # deferred
        if st.pending is not None and st.event.query():
            shape_p, st.pending = st.pending, None
            if int(st.host[0]) != 0:
                self._raise(st, int(st.host[0]), shape_p)
        st.since += 1
        if st.pending is None and st.since >= self.period:
            with th.cuda.device(st.flag.device):
                st.host.copy_(st.flag, non_blocking=True)
                st.event.record()
            st.pending, st.since = tuple(shape), 0


def _feat_params(F, C_, ref, plan: Optional[SpectralPlan], num_pairs, ipd_sin) -> nat.FeatParams:
    if plan is None:
        plan = SpectralPlan()
    return nat.FeatParams(F, C_, ref, plan.power, plan.mel.num_mels if plan.mel else 0,
                          int(plan.apply_log), int(plan.norm_mean), int(plan.norm_var), num_pairs,
                          int(ipd_sin), float(plan.log_eps), float(plan.log_lower_bound),
                          float(plan.cmvn_eps))


def _mel_ptrs(plan: Optional[SpectralPlan]):
    if plan is None or plan.mel is None:
        return None, None, None, None
    m = plan.mel
    return nat.ptr(m.start), nat.ptr(m.length), nat.ptr(m.offset), nat.ptr(m.weight)


_PAIR_CACHE = {}


def reim_axis(inp: th.Tensor, dim: int, op: int, eps: float = 0.0) -> th.Tensor:
    """N x ... x 2 x ... -> N x ...: atan2(imag, real) (op 0) or sqrt(real^2 + imag^2 + eps) (op 1) along
    `dim` -- PhaseTransform / MagnitudeTransform called on their own with any axis (aps_reim_axis)"""
    nat.require_device(inp)
    d = dim % inp.dim()
    if inp.shape[d] != 2:
        raise RuntimeError(f"axis {dim} holds {inp.shape[d]} values, not (real, imag)")
    x = nat.f32c(inp)
    inner = 1
    for k in inp.shape[d + 1:]:
        inner *= k
    out = th.empty(inp.shape[:d] + inp.shape[d + 1:], device=inp.device, dtype=th.float32)
    if out.numel():
        nat.check(nat.load().aps_reim_axis(nat.ptr(x), nat.ptr(out), x.numel() // (2 * inner), inner, op,
                                           float(eps), nat.stream_of(inp)), "aps_reim_axis")
    return out


def _pair_tensors(il: tuple, ir: tuple, device):
    """channel-pair index lists on the device, uploaded once per (pairs, device)"""
    key = (il, ir, str(device))
    hit = _PAIR_CACHE.get(key)
    if hit is None:
        hit = (th.tensor(il, dtype=th.int32, device=device),
               th.tensor(ir, dtype=th.int32, device=device))
        _PAIR_CACHE[key] = hit
    return hit


def store_features(store: th.Tensor,
                   plan: Optional[SpectralPlan],
                   ref_channel: int = 0,
                   pairs: Optional[Tuple[List[int], List[int]]] = None,
                   ipd_sin: bool = False,
                   nan_flag: Optional[th.Tensor] = None) -> th.Tensor:
    """store N x C x T x F x 2 (or N x T x F x 2) -> N x T x D.
    plan=None: no magnitude branch; pairs=None: no IPD branch."""
    nat.require_device(store)
    lib = nat.load()
    if store.dim() == 4:
        store = store.unsqueeze(1)
    N, Cn, T, F, _ = store.shape
    if plan is not None and plan.mel is not None and plan.mel.num_bins != F:
        raise RuntimeError(f"mel matrix expects {plan.mel.num_bins} bins, spectrogram has {F}")
    num_pairs = 0
    pl = pr = None
    if pairs is not None:
        il, ir = pairs
        if Cn < 2 or max(il + ir) >= Cn or min(il + ir) < 0:
            raise RuntimeError(f"IPD pair index out of range for {Cn} channels: {il} / {ir}")
        num_pairs = len(il)
        pl, pr = _pair_tensors(tuple(il), tuple(ir), store.device)
    ref = ref_channel if plan is not None else -1
    if plan is not None and not (0 <= ref < Cn):
        raise RuntimeError(f"ref_channel {ref} out of range for {Cn} channels")
    p = _feat_params(F, Cn, ref, plan, num_pairs, ipd_sin)
    D0 = 0 if plan is None else (plan.mel.num_mels if plan.mel else F)
    D = D0 + num_pairs * (2 if ipd_sin else 1) * F
    out = th.empty(N, T, D, device=store.device, dtype=th.float32)
    ms, ml, mo, mw = _mel_ptrs(plan)
    rc = lib.aps_enh_features(nat.ptr(store), N, T, store.stride(0), store.stride(1),
                              store.stride(2), C.byref(p), ms, ml, mo, mw, nat.ptr(pl),
                              nat.ptr(pr), nat.ptr(out), nat.ptr(nan_flag), nat.stream_of(store))
    nat.check(rc, "aps_enh_features")
    return out


def abs_features(y: th.Tensor, plan: SpectralPlan, abs_eps: float,
                 nan_flag: Optional[th.Tensor] = None) -> th.Tensor:
    """complex rows (..., F, 2) interleaved -> (..., D): |(re+eps) + i im| -> [mel][log][cmvn]"""
    if nat.needs_grad(y):
        # training: magnitude -> mel GEMM -> log + CMVN rows, each with a HIP backward (grad_ops)
        from aps_amd.grad_ops import LogCmvnFn, MagnitudeFn, activation
        from aps_amd.nn_ops import linear
        x = MagnitudeFn.apply(y, float(abs_eps))
        if plan.power == 2:
            x = activation(x, "square")
        if plan.mel is not None:
            x = linear(x, plan.mel.dense)
        if plan.apply_log or plan.norm_mean or plan.norm_var:
            tail = SpectralPlan(1, None, plan.apply_log, plan.log_eps, plan.log_lower_bound,
                                plan.norm_mean, plan.norm_var, plan.cmvn_eps)
            x = LogCmvnFn.apply(x, tail)
        return x
    nat.require_device(y)
    lib = nat.load()
    if y.stride(-1) != 1 or y.stride(-2) != 2:
        y = y.contiguous()
    lead = y.shape[:-2]
    F = y.shape[-2]
    rows = y.reshape(-1, F, 2)
    if rows.stride(-1) != 1 or rows.stride(-2) != 2:
        rows = rows.contiguous()
    p = _feat_params(F, 1, 0, plan, 0, False)
    D = plan.mel.num_mels if plan.mel else F
    out = th.empty(*lead, D, device=y.device, dtype=th.float32)
    ms, ml, mo, mw = _mel_ptrs(plan)
    rc = lib.aps_abs_features(nat.ptr(rows), rows.shape[0], rows.stride(0), float(abs_eps),
                              C.byref(p), ms, ml, mo, mw, nat.ptr(out), nat.ptr(nan_flag),
                              nat.stream_of(y))
    nat.check(rc, "aps_abs_features")
    return out


def row_features(x: th.Tensor, plan: SpectralPlan,
                 nan_flag: Optional[th.Tensor] = None) -> th.Tensor:
    """real rows (..., F) -> (..., D): [power] -> [mel] -> [log] -> [row cmvn]"""
    if nat.needs_grad(x):
        # behind a trainable mel projection (or any differentiable producer): log + CMVN rows with
        # their HIP backward; a mel projection in a differentiable chain is a GEMM (grad_ops.LinearFn)
        from aps_amd.grad_ops import LogCmvnFn, activation
        from aps_amd.nn_ops import linear
        if plan.power == 2:
            x = activation(x, "square")
        if plan.mel is not None:
            x = linear(x, plan.mel.dense)
        if plan.apply_log or plan.norm_mean or plan.norm_var:
            tail = SpectralPlan(1, None, plan.apply_log, plan.log_eps, plan.log_lower_bound,
                                plan.norm_mean, plan.norm_var, plan.cmvn_eps)
            x = LogCmvnFn.apply(x, tail)
        return x
    nat.require_device(x)
    lib = nat.load()
    x = x.float()
    F = x.shape[-1]
    rows = x.reshape(-1, F)
    if rows.stride(-1) != 1:
        rows = rows.contiguous()
    if plan.mel is not None and plan.mel.num_bins != F:
        raise RuntimeError(f"mel matrix expects {plan.mel.num_bins} bins, input has {F}")
    p = _feat_params(F, 1, 0, plan, 0, False)
    D = plan.mel.num_mels if plan.mel else F
    out = th.empty(*x.shape[:-1], D, device=x.device, dtype=th.float32)
    ms, ml, mo, mw = _mel_ptrs(plan)
    rc = lib.aps_row_features(nat.ptr(rows), rows.shape[0], rows.stride(0), C.byref(p), ms, ml, mo,
                              mw, nat.ptr(out), nat.ptr(nan_flag), nat.stream_of(x))
    nat.check(rc, "aps_row_features")
    return out


class _TfMaskFunction(th.autograd.Function):
    """masking with aps_tf_mask_backward (gradients to the mask and to the spectrogram)"""

    @staticmethod
    def forward(ctx, store, mask):
        ctx.save_for_backward(store, mask)
        return tf_mask_store(store, mask)

    @staticmethod
    def backward(ctx, grad_out):
        store, mask = ctx.saved_tensors
        lib = nat.load()
        N, T, F, _ = store.shape
        cplx = mask.dim() == 4
        m = mask.float()
        if cplx and m.stride(-1) != 1:
            m = m.contiguous()
        g = nat.f32c(grad_out)
        need_m, need_x = ctx.needs_input_grad[1], ctx.needs_input_grad[0]
        gm = th.empty(mask.shape, device=g.device, dtype=th.float32) if need_m else None
        gx = th.empty(N, T, F, 2, device=g.device, dtype=th.float32) if need_x else None
        gs = (gm.stride(0), gm.stride(2), gm.stride(1)) if need_m else (0, 0, 0)
        rc = lib.aps_tf_mask_backward(nat.ptr(store), N, T, F, store.stride(0), store.stride(1),
                                      nat.ptr(m), m.stride(0), m.stride(2), m.stride(1), int(cplx),
                                      nat.ptr(g), nat.ptr(gm), gs[0], gs[1], gs[2], nat.ptr(gx),
                                      nat.stream_of(g))
        nat.check(rc, "aps_tf_mask_backward")
        return gx, (gm.to(mask.dtype) if need_m else None)


class SingularGuard(NanGuard):
    """The same device counter for matrices the reference's `ComplexTensor.inverse()` -> th.inverse would
    have raised on (aps/cplx.py:268-278): the MVDR solve kernels and aps_cplx_inverse bump it once per matrix
    whose elimination meets a zero or non-finite pivot.  Same policies as NanGuard; raises
    torch.linalg.LinAlgError (a RuntimeError), what th.inverse raises."""

    def _raise(self, st, count, shape):
        st.flag.zero_()
        raise th.linalg.LinAlgError(f"linalg.inv: {count} of the batch's matrices are singular (a zero or "
                                    f"non-finite pivot), input shape = {shape}")


# the MVDR solves' guard (aps_mvdr_weights / aps_mvdr_weight / aps_mvdr_attention_weight): one per process
MVDR_SINGULAR = SingularGuard()


def mvdr_singular_flag(device) -> Optional[th.Tensor]:
    """the counter handed to the MVDR solve kernels (None inside a stream capture that would have to create it)"""
    g = MVDR_SINGULAR
    if not g.known(device) and th.cuda.is_current_stream_capturing():
        return None
    return g.pointer(device)


def mvdr_singular_check(policy: str, shape, device=None) -> None:
    """after an MVDR solve launch: "sync" raises at once like the reference's Rn.inverse() (a host stall per
    call), "deferred" (the default of MvdrBeamformer) reads the counter back asynchronously every 16th launch
    and raises at a later call / at MVDR_SINGULAR.flush(), "manual" / "off" leave it to MVDR_SINGULAR.count();
    inside a stream capture nothing is read"""
    if th.cuda.is_current_stream_capturing():
        return
    MVDR_SINGULAR.after_launch(policy, shape, device)


def length_map(lens: th.Tensor, add: int, div: int, post: int) -> th.Tensor:
    """trunc((lens + add) / div) + post on integer lengths.  Lengths on the GPU take ONE launch
    (aps_length_map) instead of torch's three or four; host-side lengths are plain arithmetic."""
    if not lens.is_cuda or lens.dtype != th.int64:
        return th.div(lens + add, div, rounding_mode="trunc") + post
    lens = lens.contiguous()
    out = th.empty_like(lens)
    rc = nat.load().aps_length_map(nat.ptr(lens), nat.ptr(out), lens.numel(), int(add), int(div),
                                   int(post), nat.stream_of(lens))
    nat.check(rc, "aps_length_map")
    return out


def tf_mask_store(store: th.Tensor, mask: th.Tensor) -> th.Tensor:
    """store N x T x F x 2, mask reference-shaped N x F x T (real) or N x F x T x 2 (complex)
    -> store N x T x F x 2"""
    if nat.needs_grad(store, mask):
        return _TfMaskFunction.apply(store, mask)
    nat.require_device(store, mask)
    lib = nat.load()
    N, T, F, _ = store.shape
    cplx = mask.dim() == 4
    mask = mask.float()
    if cplx and mask.stride(-1) != 1:
        mask = mask.contiguous()
    out = th.empty(N, T, F, 2, device=store.device, dtype=th.float32)
    rc = lib.aps_tf_mask(nat.ptr(store), N, T, F, store.stride(0), store.stride(1), nat.ptr(mask),
                         mask.stride(0), mask.stride(2), mask.stride(1), int(cplx), nat.ptr(out),
                         nat.stream_of(store))
    nat.check(rc, "aps_tf_mask")
    return out


# ------------------------------------------------------------------------------------------------
# context / utterance-level layers (aps_amd/csrc/context.hip)
# ------------------------------------------------------------------------------------------------
def splice(feats: th.Tensor, lctx: int, rctx: int, subsampling: int = 1) -> th.Tensor:
    """N x ... x T x F -> N x ... x (T // sub) x (lctx + rctx + 1) F with edge frames repeated"""
    nat.require_device(feats)
    lib = nat.load()
    x = nat.f32c(feats)
    T, F = x.shape[-2:]
    U = x.numel() // (T * F)
    To = T // subsampling
    out = th.empty(*x.shape[:-2], To, (lctx + rctx + 1) * F, device=x.device, dtype=th.float32)
    if out.numel():
        rc = lib.aps_splice(nat.ptr(x), nat.ptr(out), U, T, F, T * F, F, lctx, rctx, subsampling,
                            nat.stream_of(x))
        nat.check(rc, "aps_splice")
    return out


def delta(feats: th.Tensor, scale: th.Tensor, ctx: int, order: int,
          as_channel: bool = False) -> th.Tensor:
    """N x (C) x T x F -> N x (C) x T x (1 + order) F, or N x (1 + order) x T x F (as_channel)"""
    nat.require_device(feats, scale)
    lib = nat.load()
    x = nat.f32c(feats)
    T, F = x.shape[-2:]
    U = x.numel() // (T * F)
    sc = nat.f32c(scale)
    K = 1 + order
    if as_channel:
        if x.dim() != 3:
            raise RuntimeError("delta_as_channel expects N x T x F features")
        out = th.empty(x.shape[0], K, T, F, device=x.device, dtype=th.float32)
        out[:, 0].copy_(x)
        utt, row, step = K * T * F, F, T * F  # block k of utterance u starts at u * utt + k * step
    else:
        out = th.empty(*x.shape[:-1], K * F, device=x.device, dtype=th.float32)
        out[..., :F].copy_(x)
        utt, row, step = T * K * F, K * F, F
    base = out.data_ptr()
    for k in range(1, K):
        rc = lib.aps_delta(base + 4 * (k - 1) * step, base + 4 * k * step, nat.ptr(sc), U, T, F, ctx,
                           utt, row, utt, row, nat.stream_of(x))
        nat.check(rc, "aps_delta")
    return out


def cmvn_utterance(feats: th.Tensor, norm_mean: bool, norm_var: bool, eps: float) -> th.Tensor:
    """N x (C) x T x F, statistics over each utterance-channel's T x F matrix"""
    nat.require_device(feats)
    lib = nat.load()
    x = nat.f32c(feats)
    count = x.shape[-1] * x.shape[-2]
    out = th.empty_like(x)
    rc = lib.aps_cmvn_utterance(nat.ptr(x), nat.ptr(out), x.numel() // count, count,
                                int(norm_mean), int(norm_var), float(eps), nat.stream_of(x))
    nat.check(rc, "aps_cmvn_utterance")
    return out


def cmvn_global(feats: th.Tensor, gmean: th.Tensor, gstd: th.Tensor, norm_mean: bool,
                norm_var: bool) -> th.Tensor:
    nat.require_device(feats, gmean, gstd)
    lib = nat.load()
    x = nat.f32c(feats)
    F = x.shape[-1]
    if gmean.numel() != F or gstd.numel() != F:
        raise RuntimeError(f"global cmvn statistics have {gmean.numel()} entries, features {F}")
    out = th.empty_like(x)
    rc = lib.aps_cmvn_global(nat.ptr(x), nat.ptr(nat.f32c(gmean)), nat.ptr(nat.f32c(gstd)),
                             nat.ptr(out), x.numel() // F, F, int(norm_mean), int(norm_var),
                             nat.stream_of(x))
    nat.check(rc, "aps_cmvn_global")
    return out
