"""
Launchers for the encoder kernels (aps_amd/csrc/nn.hip): host-side argument marshalling only.
Activations are batch-major [N, T, D] / [rows, D], fp32, contiguous.
"""
import os
from ctypes import c_void_p as C_void_p
from typing import Optional

import torch as th

from aps_amd import _native as nat

# optional profiling sink: a list that receives (start_event, stop_event, flops, kernel) per GEMM
# launch, kernel = "f32" (gemm_f32_kernel) | "panel" (gemm_panel_kernel) | "split" (gemm_fp16x2_kernel, or
# gemm_split_bd_kernel under SPLIT_LAYOUT 1)
GEMM_TIMELINE = None
# bench.py: when a list, every GEMM launch of `linear` appends (re-issue callable, flops, kind, tensors kept
# alive) -- the launch sequence of a step can then be replayed back to back between ONE pair of events
GEMM_RECORD = None

# LayerNorm folded into the consuming GEMM (aps_linear_layernorm); APS_NO_LN_FUSE=1 keeps the
# stand-alone LayerNorm launches for A/B measurements
LN_FUSION = not os.environ.get("APS_NO_LN_FUSE")

ACTIVATIONS = {None: 0, "none": 0, "relu": 1, "swish": 2, "sigmoid": 3, "tanh": 4, "gelu": 5}


def _rows_view(x: th.Tensor, K: int):
    """x (..., K) as a [rows, K] matrix with a uniform row pitch, without copying when x is a
    column slice of a contiguous buffer (e.g. one half of a paired LSTM output)"""
    if x.dtype == th.float32 and x.stride(-1) == 1 and x.dim() >= 2:
        pitch = x.stride(-2)
        uniform = pitch >= K and pitch % 4 == 0 and x.data_ptr() % 16 == 0
        for d in range(x.dim() - 2):  # leading dims must continue the same pitch
            uniform = uniform and x.stride(d) == x.stride(d + 1) * x.shape[d + 1]
        if uniform:
            rows = x.numel() // K
            return x.as_strided((rows, K), (pitch, 1)), pitch
    a = nat.f32c(x).reshape(-1, K)
    return a, K


def _ln_folded(weight: th.Tensor, bias: Optional[th.Tensor], norm: th.nn.LayerNorm):
    """(W diag(gamma), colsum, b + W beta) for aps_linear_layernorm, refreshed when any source
    tensor changes.  The cache lives on the LayerNorm module (one entry per consuming weight
    storage): norm and projection belong to the same layer and share its lifetime, and views of a
    weight (the 1 x 1 conv weights are passed as `.view(2D, D)`) hit the same entry."""
    parts = [weight, bias, norm.weight, norm.bias]
    key = tuple((t.data_ptr(), t._version, tuple(t.shape)) for t in parts if t is not None)
    table = norm.__dict__.setdefault("_aps_fold", {})
    hit = table.get(weight.data_ptr())
    if hit is not None and hit[0] == key:
        return hit[1:]
    w = weight.detach().double()
    gamma = norm.weight.detach().double() if norm.weight is not None else th.ones_like(w[0])
    wg = w * gamma[None, :]
    b = bias.detach().double() if bias is not None else th.zeros_like(w[:, 0])
    if norm.bias is not None:
        b = b + w @ norm.bias.detach().double()
    out = (wg.float().contiguous(), wg.sum(1).float().contiguous(), b.float().contiguous())
    table[weight.data_ptr()] = (key,) + out
    return out


# fp32 GEMMs on the bf16 matrix pipe (aps_linear_split, csrc/gemm_split.hip): "1" / "0" force it
# on / off for every eligible launch (A/B runs, tests); by default the launches with at least
# SPLIT_MIN_TILES 64 x 128 output tiles take it.  Crossover measured on MI355X
# (scripts/split_gemm_bench.py, split against fp32, us): M = 2016: N = 512 19.0 / 14.6, N = 1024
# 23.9 / 24.0, N = 1536 29.1 / 33.9, N = 2048 32.4 / 44.2; M = 4032, N = 512 (252 tiles) 23.5 / 22.0;
# M = 6048, N = 512 (378 tiles) 28.1 / 30.8 -- below ~1.2 tiles per CU the fp32 kernel's 64 x 64
# tiles fill the chip better.  Round 3 (the fp16 two-plane kernel, joint step at BASELINE's 32
# utterances per GPU, M = 2016, two batches in flight; scripts/gpu_min_tiles.sh): threshold 320
# 10 480 utt/s, 250 (the N >= 1024 projections move over: 256 tiles) 11 430, 190 11 490, 120 (all of
# them) 11 400 with the one-stream step 4.6 instead of 4.1 ms -> 256 (one tile per CU and up).  With the
# 64 x 64 form of the kernel for launches of <= FP16X2_NARROW_TILES tiles (gemm_fp16x2.hip: twice the
# workgroups) the N = 512 projections of that batch pay too: 11 160 -> 11 650 utt/s at 128
# (scripts/gpu_r03_s14.sh) -> 128.
SPLIT_MODE = os.environ.get("APS_GEMM_SPLIT")
SPLIT_MIN_TILES = int(os.environ.get("APS_GEMM_SPLIT_MIN_TILES", "128"))
# launches of at most this many 64 x 128 tiles run as 64 x 64 tiles (mirrors the library's rule; the
# fp32-recomputation counter of nn_ops.fp16x2_wide_tiles counts tiles of the form that ran)
FP16X2_NARROW_TILES = int(os.environ.get("APS_GEMM_NARROW_TILES", "400"))
CONV_SPLIT_MIN_TILES = 256  # (the convolution's two-plane form has its own tile shapes: one tile per CU and up)


def fp16x2_tiles(M: int, N: int) -> int:
    """output tiles of an aps_linear_fp16x2 / aps_linear_panel launch (what `fp16x2_wide_tiles` counts)"""
    if SPLIT_LAYOUT == 3 and (PANEL_FORM or SPLIT_MODE == "1" or _panel_pays(M, N)):
        lib = nat.load()
        rows, cols = lib.aps_linear_panel_rows(M, N, PANEL_FORM), lib.aps_linear_panel_cols(M, N, PANEL_FORM)
        return ((M + rows - 1) // rows) * ((N + cols - 1) // cols)
    wide = ((M + 63) // 64) * ((N + 127) // 128)
    return ((M + 63) // 64) * ((N + 63) // 64) if wide <= FP16X2_NARROW_TILES else wide
# weight image: 1 = fragment image of the bf16 three-plane form (the 64 x 128 kernel whose waves fetch
# their weight operands straight into registers), 2 = the
# fragment image of the two-plane fp16 form (aps_linear_fp16x2: three products per term instead of
# six, operands scaled per row, tiles outside the planes' range recomputed in fp32), 3 = the same image
# and arithmetic in the PANEL form (aps_linear_panel, csrc/gemm_panel.hip: K walked in chunks whose
# planes are formed inside the kernel into one static LDS image -- no planes pass over A, no barrier
# inside a chunk, a power-of-two scale per (row, chunk); the default since round 4)
SPLIT_LAYOUT = int(os.environ.get("APS_GEMM_SPLIT_LAYOUT", "3"))
# the panel kernel's tile: 0 = the default, 1 | 2 | 3 = 32 x 128 at two workgroups per CU, 64 x 128, 32 x 128 at
# four workgroups per CU, 4 | 5 = the K-group forms of 16 | 8 waves, 6 = the LDS-DMA form (tests, A/B runs)
PANEL_FORM = 0
# single stream: launches of at most KGROUP_MAX_TILES tiles of 32 x 128 (one per CU) on the K-group form (form 4);
# APS_GEMM_KGROUP=0: never
KGROUP_SINGLE_STREAM = os.environ.get("APS_GEMM_KGROUP", "1") != "0"
KGROUP_MAX_TILES = 256
# streams the caller keeps busy at once BESIDES what `lstm_share()` says (replicas.PipelinedReplicas: several
# worker streams and one stream on which the persistent LSTM launches run one after the other, each sized for the
# whole chip): > 1 keeps the four-wave GEMM tiles
STREAMS_IN_FLIGHT = 1
# replicas.PipelinedReplicas: called with "lstm_begin" / "lstm_end" around a persistent LSTM-stack launch, so that a
# step can be cut into stages at those points (the stages of different batches then run on different streams)
STAGE_HOOK = None
CONV_SPLIT_MIN_CO = int(os.environ.get("APS_CONV_SPLIT_MIN_CO", "16"))
# the convolutions on the fp16 two-plane arithmetic (aps_conv2d_nhwc_fp16x2) instead of the bf16 form:
# "1" every eligible convolution, "0" none; unset: the call sites that ask for it (`fp16=True`: the
# DCCRN blocks -- 9 430 -> 10 070 mixtures/s, conv 3.82 -> 3.41 ms per 64; the conformer's conv2d
# subsampling keeps the bf16 form it was profiled with)
_CONV16_ENV = os.environ.get("APS_CONV_FP16X2")
CONV_FP16X2 = None if _CONV16_ENV is None else _CONV16_ENV == "1"
def _weight_owner(weight: th.Tensor) -> Optional[th.Tensor]:
    """the long-lived tensor a derived-weight cache may hang on: a Parameter, or the Parameter a
    view was taken from (`conv.weight.view(2D, D)`); None for temporaries, which are never cached
    (a recycled data_ptr would alias a stale entry)"""
    if isinstance(weight, th.nn.Parameter) or getattr(weight, "_aps_persistent", False):
        return weight  # (a module's own cached re-layout of a weight marks itself persistent)
    base = weight._base
    return base if isinstance(base, th.nn.Parameter) else None


def _split_planes(w: th.Tensor, owner, tag: str, layout: Optional[int] = None, with_source: bool = False):
    """planes image of the weight matrix w [N, K] (aps_linear_split_weight / aps_linear_fp16x2_weight),
    cached on `owner` (a Parameter or the LayerNorm-fold cache entry's dict) until the source changes.
    "Changes" is torch's version counter (optimiser steps, load_state_dict, any in-place op under
    no_grad); writes through `.data` bypass it, like they do for every derived-weight cache here.
    with_source: also return the contiguous fp32 matrix the image was made from (the fp16 two-plane
    kernels read it on their fp32 path; the cache entry keeps it alive next to the image)."""
    layout = SPLIT_LAYOUT if layout is None else layout
    layout = 2 if layout == 3 else layout  # (the panel kernel reads the two-plane fp16 image)
    table = owner.__dict__.setdefault("_aps_split", {}) if not isinstance(owner, dict) else owner
    key = (tag, w.data_ptr(), w._version, tuple(w.shape), w.device, layout)
    hit = table.get(tag)
    if hit is None or hit[0] != key:
        if hit is not None:
            _prefetch_forget(hit[1])  # (the replaced image must not stay pinned by the launch-order table)
        lib = nat.load()
        N, K = w.shape
        wc = nat.f32c(w.detach())
        if layout == 2:
            planes = th.empty(lib.aps_linear_fp16x2_size(N, K) // 2, device=w.device, dtype=th.int16)
            nat.check(lib.aps_linear_fp16x2_weight(nat.ptr(wc), nat.ptr(planes), N, K, K, nat.stream_of(w)),
                      "aps_linear_fp16x2_weight")
        else:
            planes = th.empty(lib.aps_linear_split_size(N, K) // 2, device=w.device, dtype=th.int16)
            nat.check(lib.aps_linear_split_weight(nat.ptr(wc), nat.ptr(planes), N, K, K, layout,
                                                  nat.stream_of(w)), "aps_linear_split_weight")
        hit = table[tag] = (key, planes, wc)
    return (hit[1], hit[2]) if with_source else hit[1]


# Tiles the fp16 two-plane kernels recomputed on their fp32 path (operands whose in-row range the two
# planes do not hold, csrc/gemm_fp16x2.hip): one sticky int32 counter per device, handed to every
# launch.  A diagnostic, not an error: the results are right either way, only slower.
_WIDE_COUNT = {}


def _refuse_first_use_in_capture(what: str) -> None:
    """process-wide device buffers (this counter, the LSTM kernels' workspace) are created on first use:
    inside a stream capture they would land in that graph's private pool -- re-zeroed by every replay,
    dangling once the graph is freed -- so the first use must be an eager one (GraphReplicas runs the
    eager step first; a hand-rolled capture has to do the same)"""
    if th.cuda.is_available() and th.cuda.is_current_stream_capturing():
        raise RuntimeError(f"aps_amd: {what} would be allocated inside a stream capture; run the step once "
                           "eagerly before capturing it")


def _wide_counter(device: th.device) -> th.Tensor:
    key = device.index if device.index is not None else th.cuda.current_device()
    t = _WIDE_COUNT.get(key)
    if t is None:
        _refuse_first_use_in_capture("the fp32-path tile counter of the two-plane GEMMs")
        t = _WIDE_COUNT[key] = th.zeros(1, dtype=th.int32, device=th.device("cuda", key))
    return t


def fp16x2_wide_tiles(device=None) -> int:
    """tiles recomputed in fp32 on `device` since the process started (blocking read)"""
    dev = th.device("cuda", th.cuda.current_device()) if device is None else th.device(device)
    return int(_wide_counter(dev).item())


# What follows what: the projections of a step run in the same order step after step, so every
# aps_linear_panel launch is told which weight image the NEXT launch of the stream read last time
# (learnt on the fly: `_PF_NEXT[image]` = the image of the launch that followed it) and requests it on
# its way out (aps_linear_panel's next_image hint: a step's images do not survive in the caches from
# one step to the next, and a launch of the 32-utterance batch that starts on cold weights spends more
# time waiting for HBM than computing).  Only a hint: a wrong guess costs a few idle requests.  The
# table holds the image tensors themselves, so a hinted address stays allocated while its image is the
# current one of its weight (a captured graph replays hints of current images only: one captured before a
# weight update reads stale images everywhere, hint or not); entries of an image are dropped when
# `_split_planes` replaces it, and the table is bounded.  APS_GEMM_PREFETCH=0: off (A/B runs).
PREFETCH_NEXT = os.environ.get("APS_GEMM_PREFETCH", "1") != "0"
_PF_PREV = {}   # device index -> data_ptr of the previous launch's image
_PF_NEXT = {}   # data_ptr of an image -> the image (tensor) the following launch read


def _prefetch_hint(planes: th.Tensor):
    """(pointer, bytes) of what followed `planes` the last time it was launched; records the order"""
    if not PREFETCH_NEXT:
        return None, 0
    dev = planes.device.index
    key = planes.data_ptr()
    prev = _PF_PREV.get(dev)
    if prev is not None and prev != key:
        if len(_PF_NEXT) >= 4096:  # (models come and go in one process: bounded)
            _PF_NEXT.clear()
        _PF_NEXT[prev] = planes
    _PF_PREV[dev] = key
    nxt = _PF_NEXT.get(key)
    if nxt is None or nxt.device != planes.device:
        return None, 0
    return C_void_p(nxt.data_ptr()), nxt.numel() * nxt.element_size()


def _prefetch_forget(planes: th.Tensor) -> None:
    """a weight image is being replaced (its source changed: optimiser step, load_state_dict): drop what the
    launch-order table holds for it -- as a key (its address may be recycled for an unrelated image) and as a
    value (the table's reference would otherwise be the only thing keeping the stale image allocated)"""
    key = planes.data_ptr()
    _PF_NEXT.pop(key, None)
    for k in [k for k, v in _PF_NEXT.items() if v is planes]:
        del _PF_NEXT[k]
    for dev in [d for d, p in _PF_PREV.items() if p == key]:
        del _PF_PREV[dev]


def prefetch_chain_reset() -> None:
    """forget the learnt launch order (tests)"""
    _PF_PREV.clear()
    _PF_NEXT.clear()


def _panel_pays(M: int, N: int) -> bool:
    """the panel kernel against the planes-pass kernel (scripts/panel_gemm_probe.py, round 4): ahead for
    every launch of fewer than 512 tiles of 64 x 128 (BASELINE's 32 utterances per GPU: M = 2016) and for
    the N <= 640 projections of the merged batch; behind where many column tiles re-split the same rows
    (M = 8064: N = 1024 44 - 53 against 43 us, N = 1536 65 - 85 against 56, N = 5000 217 against 168)"""
    return ((M + 63) // 64) * ((N + 127) // 128) < 512 or N <= 640


def _use_split(M: int, N: int, K: int, min_tiles: Optional[int] = None) -> bool:
    if SPLIT_MODE is not None:
        return SPLIT_MODE == "1"
    min_tiles = SPLIT_MIN_TILES if min_tiles is None else min_tiles
    return ((M + 63) // 64) * ((N + 127) // 128) >= min_tiles and K >= 128


def linear(x: th.Tensor, weight: th.Tensor, bias: Optional[th.Tensor] = None,
           residual: Optional[th.Tensor] = None, relu: bool = False, act: Optional[str] = None,
           alpha: float = 1.0, ln: Optional[th.nn.LayerNorm] = None, chain: bool = False) -> th.Tensor:
    """y = act(x W^T + b) * alpha (+ residual), x (..., K), W [N, K] -> (..., N); fp32 MFMA GEMM
    with the epilogue fused (tf.linear + activation + scaling + residual add of the reference).
    act: None | "relu" | "swish" | "sigmoid" | "tanh" | "gelu".  chain: accepted and ignored (round 2's
    row-maxima hand-over between GEMMs; every fp16 two-plane GEMM forms the planes of its input in a
    pass of its own now, which finds the row maxima on the way)."""
    if relu:
        act = "relu"
    if act not in ACTIVATIONS:
        raise ValueError(f"linear: unknown activation {act}")
    ln_params = () if ln is None else (ln.weight, ln.bias)
    if nat.needs_grad(x, weight, bias, residual, *ln_params):
        # training: the un-fused form keeps what the backward needs (grad_ops.LinearFn)
        from aps_amd.grad_ops import LinearFn
        if ln is not None:
            x = layernorm(x, ln.weight, ln.bias, ln.eps)
        return LinearFn.apply(x, weight, bias, residual, ACTIVATIONS[act], float(alpha))
    nat.require_device(x, weight, bias, residual)
    lib = nat.load()
    K = x.shape[-1]
    N = weight.shape[0]
    if weight.shape[1] != K:
        raise RuntimeError(f"linear: weight {tuple(weight.shape)} does not match input dim {K}")
    if ln is not None and (K % 4 or not LN_FUSION):
        # odd K: the K padding below would enter the row statistics (LN_FUSION off: A/B runs)
        x, ln = layernorm(x, ln.weight, ln.bias, ln.eps), None
    a, lda = _rows_view(x, K)
    M = a.shape[0]
    owner = _weight_owner(weight) if K % 4 == 0 and _use_split(M, N, K) else None
    if owner is not None:
        return _linear_split(lib, x, a, lda, weight, owner, bias, residual, act, alpha, ln, chain)
    w = nat.f32c(weight)
    ldw = K
    if K % 4:  # pad K so every row start is 16-byte aligned (rare: odd feature sizes)
        pad = 4 - K % 4
        a = th.nn.functional.pad(a.contiguous(), (0, pad))
        w = th.nn.functional.pad(w, (0, pad))
        lda = ldw = K + pad
    out = th.empty(M, N, device=x.device, dtype=th.float32)
    res = None
    if residual is not None:
        res = nat.f32c(residual).reshape(M, N)
    timeline = GEMM_TIMELINE
    if timeline is not None:
        e0, e1 = th.cuda.Event(enable_timing=True), th.cuda.Event(enable_timing=True)
        e0.record()
    if ln is not None:
        if tuple(ln.normalized_shape) != (K,):
            raise RuntimeError(f"linear: LayerNorm over {ln.normalized_shape}, input has {K}")
        wg, cs, bb = _ln_folded(weight, bias, ln)
        rc = lib.aps_linear_layernorm(nat.ptr(a), nat.ptr(wg), nat.ptr(bb), nat.ptr(cs),
                                      nat.ptr(res), nat.ptr(out), M, N, K, lda, ldw, N,
                                      ACTIVATIONS[act], float(alpha), float(ln.eps),
                                      nat.stream_of(x))
        nat.check(rc, "aps_linear_layernorm")
    else:
        rc = lib.aps_linear(nat.ptr(a), nat.ptr(w),
                            nat.ptr(None if bias is None else nat.f32c(bias)), nat.ptr(res),
                            nat.ptr(out), M, N, K, lda, ldw, N, ACTIVATIONS[act], float(alpha),
                            nat.stream_of(x))
        nat.check(rc, "aps_linear")
    if timeline is not None:
        e1.record()
        timeline.append((e0, e1, 2.0 * M * N * K, "f32"))
    return out.view(*x.shape[:-1], N)


def _linear_split(lib, x, a, lda, weight, owner, bias, residual, act, alpha, ln, chain=False) -> th.Tensor:
    """`linear` on aps_linear_split: the weight's bf16 planes are cached on its Parameter (on the
    LayerNorm fold's cache entry for the folded weight)"""
    M, K = a.shape
    N = weight.shape[0]
    out = th.empty(M, N, device=x.device, dtype=th.float32)
    res = None if residual is None else nat.f32c(residual).reshape(M, N)
    timeline = GEMM_TIMELINE
    if timeline is not None:
        e0, e1 = th.cuda.Event(enable_timing=True), th.cuda.Event(enable_timing=True)
        e0.record()
    if ln is not None:
        if tuple(ln.normalized_shape) != (K,):
            raise RuntimeError(f"linear: LayerNorm over {ln.normalized_shape}, input has {K}")
        wg, cs, bb = _ln_folded(weight, bias, ln)
        # the folded weight lives in the fold cache of `ln`: its planes go next to it
        planes, w32 = _split_planes(wg, ln.__dict__.setdefault("_aps_fold_split", {}),
                                    str(weight.data_ptr()), with_source=True)
        bb_, cs_, eps = bb, cs, float(ln.eps)
    else:
        planes, w32 = _split_planes(weight, owner, "w", with_source=True)
        bb_, cs_, eps = (None if bias is None else nat.f32c(bias)), None, 0.0
    kind = "split"
    if SPLIT_LAYOUT == 3 and (PANEL_FORM or SPLIT_MODE == "1" or _panel_pays(M, N)):
        # The K-group form (16 waves per 32 x 128 tile, csrc/gemm_panel.hip) for the launches of one tile per CU
        # -- N = 512 at M = 2016: 9.7 against 12.0 us -- while ONE stream is launching: a 16-wave workgroup owns
        # its CU, so beside another stream's launches (GraphReplicas(replicas > 1) holds lstm_share() > 1) the
        # four-wave form overlaps better (profiles/r05_rejected_experiments.txt (1)).
        form = PANEL_FORM
        if form == 0 and KGROUP_SINGLE_STREAM and K <= 1024 and lstm_share() == 1 and STREAMS_IN_FLIGHT == 1 and \
                ((M + 31) // 32) * ((N + 127) // 128) <= KGROUP_MAX_TILES:
            form = 4
        kind = "kgroup" if lib.aps_linear_panel_form(M, N, K, form) >= 4 else "panel"
        nxt, nxt_bytes = _prefetch_hint(planes)
        fn, fargs = lib.aps_linear_panel, (
            nat.ptr(a), nat.ptr(planes), nat.ptr(w32), nat.ptr(bb_), nat.ptr(cs_), nat.ptr(res), nat.ptr(out),
            nat.ptr(_wide_counter(x.device)), M, N, K, lda, K, N, ACTIVATIONS[act], float(alpha), eps, nxt,
            nxt_bytes, form, nat.stream_of(x))
        rc = fn(*fargs)
    elif SPLIT_LAYOUT in (2, 3):
        # the call's workspace: the planes image of A (formed by the call's first launch), its row
        # exponents / wide flags / LayerNorm statistics
        ws = th.empty(lib.aps_linear_fp16x2_workspace(M, K), device=x.device, dtype=th.uint8)
        fn, fargs = lib.aps_linear_fp16x2, (
            nat.ptr(a), nat.ptr(planes), nat.ptr(w32), nat.ptr(bb_), nat.ptr(cs_), nat.ptr(res), nat.ptr(out),
            nat.ptr(ws), nat.ptr(_wide_counter(x.device)), M, N, K, lda, K, N, ACTIVATIONS[act], float(alpha),
            eps, nat.stream_of(x))
        rc = fn(*fargs)
    else:
        ws = None
        fn, fargs = lib.aps_linear_split, (
            nat.ptr(a), nat.ptr(planes), nat.ptr(bb_), nat.ptr(cs_), nat.ptr(res), nat.ptr(out), M, N, K, lda,
            N, ACTIVATIONS[act], float(alpha), eps, SPLIT_LAYOUT, nat.stream_of(x))
        rc = fn(*fargs)
    nat.check(rc, "aps_linear_split")
    if GEMM_RECORD is not None:
        keep = (a, planes, w32, bb_, cs_, res, out, ws if kind == "split" else None)
        GEMM_RECORD.append((lambda fn=fn, fargs=fargs: fn(*fargs), 2.0 * M * N * K, kind, keep))
    if timeline is not None:
        e1.record()
        timeline.append((e0, e1, 2.0 * M * N * K, kind))
    return out.view(*x.shape[:-1], N)


def layernorm(x: th.Tensor, weight: th.Tensor, bias: th.Tensor, eps: float = 1e-5,
              residual: Optional[th.Tensor] = None) -> th.Tensor:
    """LayerNorm(x (+ residual)) over the last axis"""
    if nat.needs_grad(x, weight, bias, residual):
        from aps_amd.grad_ops import LayerNormFn
        return LayerNormFn.apply(x, residual, weight, bias, float(eps))
    nat.require_device(x, weight, bias, residual)
    lib = nat.load()
    D = x.shape[-1]
    xc = nat.f32c(x)
    rc_ = None if residual is None else nat.f32c(residual)
    out = th.empty_like(xc)
    rc = lib.aps_layernorm(nat.ptr(xc), nat.ptr(rc_), nat.ptr(nat.f32c(weight)),
                           nat.ptr(nat.f32c(bias)), nat.ptr(out), xc.numel() // D, D, float(eps),
                           nat.stream_of(x))
    nat.check(rc, "aps_layernorm")
    return out


def posenc_add(x: th.Tensor, div_term: th.Tensor, factor: float = 1.0, t0: int = 0) -> th.Tensor:
    """x N x T x D -> x * factor + sinusoid(t0 + t)"""
    if nat.needs_grad(x):
        from aps_amd.grad_ops import PosencFn
        return PosencFn.apply(x, div_term, float(factor), int(t0))
    nat.require_device(x, div_term.detach())
    lib = nat.load()
    N, T, D = x.shape
    xc = nat.f32c(x)
    out = th.empty_like(xc)
    rc = lib.aps_posenc_add(nat.ptr(xc), nat.ptr(nat.f32c(div_term)), nat.ptr(out), N, T, D,
                            float(factor), int(t0), nat.stream_of(x))
    nat.check(rc, "aps_posenc_add")
    return out


def attention_core(qkv: th.Tensor, num_heads: int, lens: Optional[th.Tensor] = None,
                   rel: Optional[th.Tensor] = None, rel_zero: Optional[int] = None,
                   rel_u: Optional[th.Tensor] = None, rel_v: Optional[th.Tensor] = None,
                   query_from_value: bool = False, chunk_size: int = 1, lctx: int = -1,
                   rctx: int = -1, add_mask: Optional[th.Tensor] = None,
                   dropout: Optional[th.nn.Dropout] = None) -> th.Tensor:
    """qkv N x T x 3D (q | k | v, heads contiguous inside each) -> context N x T x D.
    rel [R, dh] (shared) or [H, R, dh] (per head): relative position table, score(i, j) +=
    q_i . rel[j - i + rel_zero] (rel_zero defaults to the middle row, R = 2T - 1);
    rel_u / rel_v [H, dh]: Transformer-XL biases; query_from_value: the XL quirk of the reference;
    chunk_size / lctx / rctx: context window (negative = open); add_mask T x T: any additive
    mask (0 / -inf or a bias)"""
    drop_p = dropout.p if dropout is not None and dropout.training else 0.0
    if nat.needs_grad(qkv, rel, rel_u, rel_v) or drop_p > 0:
        general = rel_u is not None or rel_v is not None or query_from_value or \
            (chunk_size, lctx, rctx) != (1, -1, -1) or (rel is not None and rel.dim() == 3) or \
            (drop_p > 0 and qkv.shape[-1] // 3 // num_heads not in (32, 64))  # (row kernels: 32 / 64)
        if add_mask is not None:
            # an additive mask tensor (src_mask / tgt_mask that is not a context window, impl.py:104-114,
            # decoder.py:150-186): the general form's kernels take it (round 5); it is data, no gradient
            if tuple(add_mask.shape) != (qkv.shape[1], qkv.shape[1]):
                raise RuntimeError(f"attention_core: add_mask {tuple(add_mask.shape)} != "
                                   f"({qkv.shape[1]}, {qkv.shape[1]})")
            if add_mask.device != qkv.device:
                raise RuntimeError(f"attention_core: add_mask on {add_mask.device}, qkv on {qkv.device}")
            general = True
        if general:
            from aps_amd.grad_ops import AttentionXlFn, draw_seed
            return AttentionXlFn.apply(qkv, rel, rel_u, rel_v, lens, num_heads, rel_zero,
                                       bool(query_from_value), int(chunk_size), int(lctx), int(rctx),
                                       float(drop_p), draw_seed() if drop_p > 0 else 0, add_mask)
        from aps_amd.grad_ops import AttentionFn, draw_seed
        return AttentionFn.apply(qkv, rel, lens, num_heads, rel_zero, float(drop_p),
                                 draw_seed() if drop_p > 0 else 0)
    nat.require_device(qkv, lens, rel, rel_u, rel_v, add_mask)
    lib = nat.load()
    N, T, D3 = qkv.shape
    D = D3 // 3
    dh = D // num_heads
    qc = nat.f32c(qkv)
    if lens is not None:
        lens = lens.to(device=qkv.device, dtype=th.int64).contiguous()
    ctx = th.empty(N, T, D, device=qkv.device, dtype=th.float32)
    rel_len, head_stride = 0, 0
    if rel is not None:
        rel = nat.f32c(rel)
        if rel.shape[-1] != dh or rel.dim() not in (2, 3) or \
                (rel.dim() == 3 and rel.shape[0] != num_heads):
            raise RuntimeError(f"attention_core: rel table {tuple(rel.shape)} != [(H,) R, {dh}]")
        rel_len = rel.shape[-2]
        head_stride = rel_len * dh if rel.dim() == 3 else 0
        if rel_zero is None:
            rel_zero = (rel_len - 1) // 2
    for t in (rel_u, rel_v):
        if t is not None and tuple(t.shape) != (num_heads, dh):
            raise RuntimeError(f"attention_core: rel_u / rel_v must be [{num_heads}, {dh}]")
    if add_mask is not None:
        if tuple(add_mask.shape) != (T, T):
            raise RuntimeError(f"attention_core: add_mask {tuple(add_mask.shape)} != ({T}, {T})")
        add_mask = nat.f32c(add_mask)
    rc = lib.aps_attention_core(nat.ptr(qc), nat.ptr(lens), nat.ptr(rel), int(rel_zero or 0),
                                rel_len, head_stride,
                                nat.ptr(None if rel_u is None else nat.f32c(rel_u)),
                                nat.ptr(None if rel_v is None else nat.f32c(rel_v)),
                                2 if query_from_value else 0, int(chunk_size), int(lctx), int(rctx),
                                nat.ptr(add_mask), nat.ptr(ctx), N, T, num_heads, dh,
                                nat.stream_of(qkv))
    nat.check(rc, "aps_attention_core")
    return ctx


def attention_cross(q: th.Tensor, kv: th.Tensor, num_heads: int,
                    key_lens: Optional[th.Tensor] = None,
                    add_mask: Optional[th.Tensor] = None,
                    dropout: Optional[th.nn.Dropout] = None) -> th.Tensor:
    """q N x Tq x D (query projection), kv N x Tk x 2D (key | value projections of the memory,
    heads contiguous inside each) -> context N x Tq x D; keys at j >= key_lens[n] are masked;
    add_mask Tq x Tk: additive (0 / -inf or a bias), the decoder layer's memory_mask;
    dropout: nn.Dropout on the attention weights (active in train() mode)"""
    drop_p = dropout.p if dropout is not None and dropout.training else 0.0
    if nat.needs_grad(q, kv) or drop_p > 0:
        if add_mask is not None and tuple(add_mask.shape) != (q.shape[1], kv.shape[1]):
            raise RuntimeError(f"attention_cross: add_mask {tuple(add_mask.shape)} != ({q.shape[1]}, {kv.shape[1]})")
        if add_mask is not None and add_mask.device != q.device:
            raise RuntimeError(f"attention_cross: add_mask on {add_mask.device}, q on {q.device}")
        from aps_amd.grad_ops import AttentionCrossFn, draw_seed
        return AttentionCrossFn.apply(q, kv, key_lens, num_heads, float(drop_p),
                                      draw_seed() if drop_p > 0 else 0, add_mask)
    nat.require_device(q, kv, key_lens, add_mask)
    lib = nat.load()
    N, Tq, D = q.shape
    Tk = kv.shape[1]
    if kv.shape[0] != N or kv.shape[2] != 2 * D:
        raise RuntimeError(f"attention_cross: kv {tuple(kv.shape)} does not match q {tuple(q.shape)}")
    if key_lens is not None:
        key_lens = key_lens.to(device=q.device, dtype=th.int64).contiguous()
    if add_mask is not None:
        if tuple(add_mask.shape) != (Tq, Tk):
            raise RuntimeError(f"attention_cross: add_mask {tuple(add_mask.shape)} != ({Tq}, {Tk})")
        add_mask = nat.f32c(add_mask)
    ctx = th.empty(N, Tq, D, device=q.device, dtype=th.float32)
    rc = lib.aps_attention_cross(nat.ptr(nat.f32c(q)), nat.ptr(nat.f32c(kv)), nat.ptr(key_lens),
                                 nat.ptr(add_mask), nat.ptr(ctx), N, Tq, Tk, num_heads,
                                 D // num_heads, nat.stream_of(q))
    nat.check(rc, "aps_attention_cross")
    return ctx


def embedding_posenc(table: th.Tensor, ids: th.Tensor, div_term: th.Tensor, factor: float = 1.0,
                     t0: int = 0) -> th.Tensor:
    """table V x D, ids N x T (int64) -> N x T x D = table[ids] * factor + sinusoid(t0 + t)"""
    if nat.needs_grad(table):
        from aps_amd.grad_ops import EmbeddingPosencFn
        return EmbeddingPosencFn.apply(table, ids, div_term, float(factor), int(t0))
    nat.require_device(table.detach(), ids, div_term.detach())
    lib = nat.load()
    N, T = ids.shape
    V, D = table.shape
    out = th.empty(N, T, D, device=table.device, dtype=th.float32)
    rc = lib.aps_embedding_posenc(nat.ptr(nat.f32c(table)), nat.ptr(ids.to(th.int64).contiguous()),
                                  nat.ptr(nat.f32c(div_term)), nat.ptr(out), N, T, D, V,
                                  float(factor), int(t0), None, nat.stream_of(table))
    nat.check(rc, "aps_embedding_posenc")
    return out


def glu_dwconv(x: th.Tensor, weight: th.Tensor, bias: Optional[th.Tensor],
               scale: Optional[th.Tensor], shift: Optional[th.Tensor],
               swish: bool = True, causal: bool = False,
               pad_bias: Optional[th.Tensor] = None, act: Optional[str] = None) -> th.Tensor:
    """x N x T x 2D -> act(scale * (depthwise_conv_T(glu(x)) + bias) + shift) N x T x D;
    act: "swish" | "relu" | "gelu" | "none" (default: swish if `swish` else none);
    weight [D, 1, K] | [D, K] (the depthwise Conv1d weight), zero padding (K - 1) / 2; causal:
    K - 1 frames of left context whose out-of-range frames carry glu(pad_bias) (see aps_amd.h)"""
    if nat.needs_grad(x, weight, bias, scale, shift, pad_bias):
        plain = scale is None and shift is None and \
            (act == "none" or (act is None and not swish))
        if not plain:
            raise NotImplementedError("aps_amd: glu_dwconv backward exists for the plain GLU + "
                                      "depthwise convolution; compose BatchNorm / activation with "
                                      "grad_ops.batchnorm_rows / activation")
        from aps_amd.grad_ops import GluDwconvFn
        return GluDwconvFn.apply(x, weight, bias, causal, pad_bias if causal else None)
    nat.require_device(x, weight, bias, scale, shift, pad_bias)
    lib = nat.load()
    N, T, D2 = x.shape
    D = D2 // 2
    w = nat.f32c(weight).reshape(D, -1)
    K = w.shape[1]
    xc = nat.f32c(x)
    out = th.empty(N, T, D, device=x.device, dtype=th.float32)

    def opt(t):
        return nat.ptr(None if t is None else nat.f32c(t))

    codes = {"none": 0, "swish": 1, "relu": 2, "gelu": 3}
    if act is None:
        act = "swish" if swish else "none"
    if act not in codes:
        raise ValueError(f"glu_dwconv: unknown activation {act}")
    code = codes[act]
    rc = lib.aps_glu_dwconv(nat.ptr(xc), nat.ptr(w), opt(bias), opt(scale), opt(shift),
                            nat.ptr(out), N, T, D, K, code, int(causal), opt(pad_bias),
                            nat.stream_of(x))
    nat.check(rc, "aps_glu_dwconv")
    return out


# ------------------------------------------------------------------------------------------------
# LSTM (aps_lstm_layer): one batched input GEMM + one persistent recurrence launch per layer and
# direction
# ------------------------------------------------------------------------------------------------
LSTM_HIDDEN_SIZES = (64, 128, 256, 320, 384, 512, 640, 768, 1024)
LSTM_MAX_BATCH = int(os.environ.get("APS_LSTM_MAX_BATCH", "128"))
# test switch: read the hand-off timeout counter right after every launch (a blocking copy).  The
# default is the deferred check of _LstmStatus below: never skipped, never a stall.
LSTM_CHECK = False


# layer-pipelined single launch for unidirectional stacks (aps_lstm_stack); APS_NO_LSTM_STACK=1
# keeps one launch per layer (A/B measurements)
LSTM_STACK = not os.environ.get("APS_NO_LSTM_STACK")
LSTM_STACK_SIZES = (64, 128, 256, 512)
LSTM_STACK_MAX_BATCH = 64

# How many memory-synchronised launches (the persistent LSTM grids) may be in flight on the device at
# once.  Every workgroup of every such launch has to be resident together, so each launch sizes its
# grid for 1 / share of the chip (the `share` argument of aps_lstm_layer / aps_lstm_stack).  Library
# state, not an environment variable: GraphReplicas holds it at its replica count for as long as it
# lives, so every launch of the process in that time -- captured or eager -- respects it.
_LSTM_SHARE = []  # one entry per holder (a GraphReplicas object, a concurrent_launches context)


def lstm_share() -> int:
    """the share every persistent launch is sized for right now: the largest one held, 1 if none"""
    return max(_LSTM_SHARE, default=1)


def push_lstm_share(n: int) -> None:
    if n < 1:
        raise ValueError(f"share must be >= 1, got {n}")
    _LSTM_SHARE.append(int(n))


def pop_lstm_share(n: int) -> None:
    """release one hold of `n` (holders may be released in any order)"""
    if int(n) in _LSTM_SHARE:
        _LSTM_SHARE.remove(int(n))


class _LstmStatus:
    """Per-device hand-off status of the persistent LSTM kernels.  `ws` is the workspace every launch
    on the device gets: word 0 is a sticky count of expired hand-off waits (a workgroup that was not
    resident, an input NaN that reproduces the sentinel); a launch it happened in returns NaNs.  The
    count is ALWAYS reported: after every eager launch it is copied to pinned memory without
    stalling the stream and examined at the next launch / `lstm_timeouts()` (the "deferred" policy of
    the NaN guard); launches recorded into a graph cannot copy, so whoever replays the graph reads
    `lstm_timeouts(device)` afterwards (GraphReplicas.synchronize does)."""

    def __init__(self, device: th.device) -> None:
        # status words + the placement tables of the team form (aps_lstm_workspace)
        self.ws = th.zeros(nat.load().aps_lstm_workspace(512) // 4, dtype=th.int32, device=device)
        self.host = th.zeros(4, dtype=th.int32).pin_memory()
        self.event = None
        self.reported = 0

    def _raise_if(self, count: int, what: str) -> None:
        if count > self.reported:
            self.reported = count
            raise RuntimeError(
                f"{what}: an inter-workgroup hand-off of a persistent LSTM launch timed out "
                f"({count} expired waits so far on {self.ws.device}): a workgroup was not resident "
                "(more concurrent launches than the `share` they were sized for?) or an input NaN "
                "reproduced the sentinel; the outputs of that launch are NaN")

    def poll(self, what: str) -> None:
        """non-blocking: look at the last completed copy"""
        if self.event is not None and self.event.query():
            self._raise_if(int(self.host[0]), what)

    def after_launch(self, what: str, block: bool = False) -> None:
        if th.cuda.is_current_stream_capturing():
            return
        self.poll(what)
        with th.cuda.device(self.ws.device):
            self.host.copy_(self.ws[:4], non_blocking=True)
            self.event = th.cuda.Event()
            self.event.record()
        if block:
            self.event.synchronize()
            self._raise_if(int(self.host[0]), what)

    def count(self) -> int:
        """blocking read of the sticky counter"""
        return int(self.ws[0].item())


_LSTM_STATUS = {}


def _lstm_status(device: th.device) -> _LstmStatus:
    key = device.index if device.index is not None else th.cuda.current_device()
    st = _LSTM_STATUS.get(key)
    if st is None:
        _refuse_first_use_in_capture("the persistent LSTM kernels' workspace")
        st = _LSTM_STATUS[key] = _LstmStatus(th.device("cuda", key))
    return st


def lstm_timeouts(device=None, check: bool = True) -> int:
    """expired hand-off waits on `device` since the process started (blocking read); raises on a
    count that has not been reported yet unless check=False"""
    dev = th.device("cuda", th.cuda.current_device()) if device is None else th.device(device)
    st = _lstm_status(dev)
    n = st.count()
    if check:
        st._raise_if(n, "lstm_timeouts")
    return n


def _lstm_stack_forward(lib, layers, x: th.Tensor, lens: Optional[th.Tensor]):
    """all layers of a unidirectional stack in one launch (layers pipelined inside the kernel);
    layers = [(w_ih, w_hh, b_ih | None, b_hh | None)]; returns the list of layer outputs or None"""
    import ctypes as C
    N, T, _ = x.shape
    L = len(layers)
    H = layers[0][1].shape[1]
    pre0 = linear(x, layers[0][0], layers[0][2])
    # Placement matters: with the layers' outputs exactly N T H floats apart (a multiple of 64 KB at
    # the benchmark shape) the hand-off traffic of the layers collides in the memory channels and
    # the joint step measured 10 % slower (6 520 against 7 250-7 290 utt/s for any gap of 256 B ...
    # 1 MB) -> one block, layers 4 KB further apart than their size
    per = N * T * H + int(os.environ.get("APS_LSTM_Y_PAD_BYTES", "4096")) // 4
    flat = th.empty(L * per, device=x.device, dtype=th.float32)
    ys = [flat[l * per:l * per + N * T * H].view(N, T, H) for l in range(L)]
    keep = []  # the tensors whose pointers go into the arrays must outlive the call

    def ptrs(tensors):
        arr = (C.c_void_p * L)()
        for i, t in enumerate(tensors):
            if t is not None:
                t = nat.f32c(t)
                keep.append(t)
                arr[i] = t.data_ptr()
        return arr

    w_ih = ptrs([None] + [lay[0] for lay in layers[1:]])
    w_hh = ptrs([lay[1] for lay in layers])
    b_ih = ptrs([None] + [lay[2] for lay in layers[1:]])
    b_hh = ptrs([lay[3] for lay in layers])
    yp = ptrs(ys)
    status = _lstm_status(x.device)
    hook = STAGE_HOOK
    if hook is not None:
        hook("lstm_begin")   # (the launch below goes to whatever stream is current after this)
    rc = lib.aps_lstm_stack(nat.ptr(pre0), w_ih, w_hh, b_ih, b_hh, nat.ptr(lens), yp, N, T, H, L,
                            lstm_share(), nat.ptr(status.ws), nat.stream_of(x))
    if hook is not None:
        hook("lstm_end")
    if rc == nat.ERR_UNSUPPORTED:  # no resident decomposition for this geometry: layer by layer
        return None
    nat.check(rc, "aps_lstm_stack")
    status.after_launch("aps_lstm_stack", block=LSTM_CHECK)
    return ys


def _lstm_chunks(lib, run, N: int, x: th.Tensor, what: str) -> None:
    """utterances are independent: the batch goes through `run(n0, n1)` in chunks of at most
    LSTM_MAX_BATCH; a chunk the library has no resident decomposition for (APS_ERR_UNSUPPORTED) is
    halved down to 16 utterances"""
    n0, chunk = 0, LSTM_MAX_BATCH
    status = _lstm_status(x.device)
    # every persistent launch is a stage of its own for a staged capture (replicas.PipelinedReplicas moves it to the
    # one stream that runs such launches one after the other): single-layer, bidirectional, paired and
    # stack-declined recurrences as well as the stack launch of _lstm_stack_forward
    hook = STAGE_HOOK
    while n0 < N:
        n1 = min(N, n0 + chunk)
        if hook is not None:
            hook("lstm_begin")
        rc = run(n0, n1, status.ws)
        if hook is not None:
            hook("lstm_end")
        if rc == nat.ERR_UNSUPPORTED and n1 - n0 > 16:
            chunk = max(16, (n1 - n0 + 1) // 2)
            continue
        nat.check(rc, what)
        status.after_launch(what, block=LSTM_CHECK)
        n0 = n1


def _lstm_layer_params(rnn: th.nn.LSTM):
    """[(w_ih, w_hh, b_ih | None, b_hh | None)] of a unidirectional nn.LSTM"""
    out = []
    for l in range(rnn.num_layers):
        out.append((getattr(rnn, f"weight_ih_l{l}"), getattr(rnn, f"weight_hh_l{l}"),
                    getattr(rnn, f"bias_ih_l{l}") if rnn.bias else None,
                    getattr(rnn, f"bias_hh_l{l}") if rnn.bias else None))
    return out


def lstm_layers_forward(layers, x: th.Tensor, lens: Optional[th.Tensor], has_bias: bool):
    """unidirectional stack on raw parameter tensors -> the outputs of EVERY layer (the backward
    recomputes gates and cells from them).  layers = [(w_ih, w_hh[, b_ih, b_hh])]."""
    lib = nat.load()
    layers = [(lay[0], lay[1], lay[2] if has_bias else None, lay[3] if has_bias else None)
              for lay in layers]
    N, T, _ = x.shape
    H = layers[0][1].shape[1]
    if H not in LSTM_HIDDEN_SIZES:
        raise NotImplementedError(f"aps_amd LSTM: hidden size {H} has no recurrence kernel")
    if 2 <= len(layers) <= 4 and N <= LSTM_STACK_MAX_BATCH and H in LSTM_STACK_SIZES and LSTM_STACK:
        ys = _lstm_stack_forward(lib, layers, x, lens)
        if ys is not None:
            return ys
    ys, out = [], x
    for w_ih, w_hh, b_ih, b_hh in layers:
        y = th.empty(N, T, H, device=x.device, dtype=th.float32)
        pre = linear(out, w_ih, b_ih)

        def run(n0, n1, ws, pre=pre, y=y, w_hh=w_hh, b_hh=b_hh):
            return lib.aps_lstm_layer(nat.ptr(pre[n0:n1]), None, nat.ptr(w_hh), None,
                                      nat.ptr(b_hh), None,
                                      nat.ptr(None if lens is None else lens[n0:n1]),
                                      nat.ptr(y[n0:n1]), n1 - n0, T, H, 1, lstm_share(),
                                      nat.ptr(ws), nat.stream_of(x))

        _lstm_chunks(lib, run, N, x, "aps_lstm_layer")
        ys.append(y)
        out = y
    return ys


def lstm_bidir_layers_forward(layers, x: th.Tensor, lens: Optional[th.Tensor], has_bias: bool):
    """bidirectional stack on raw parameter tensors -> the N x T x 2H outputs of EVERY layer.
    layers = [((w_ih, w_hh[, b_ih, b_hh]) forward, (...) backward direction)]"""
    lib = nat.load()
    N, T, _ = x.shape
    H = layers[0][0][1].shape[1]
    if H not in LSTM_HIDDEN_SIZES:
        raise NotImplementedError(f"aps_amd LSTM: hidden size {H} has no recurrence kernel")
    ys, out = [], x
    for fwd, bwd in layers:
        y = th.empty(N, T, 2 * H, device=x.device, dtype=th.float32)
        pre = [linear(out, p[0], p[2] if has_bias else None) for p in (fwd, bwd)]
        b_hh = [p[3] if has_bias else None for p in (fwd, bwd)]

        def run(n0, n1, ws, pre=pre, y=y, fwd=fwd, bwd=bwd, b_hh=b_hh):
            return lib.aps_lstm_layer(nat.ptr(pre[0][n0:n1]), nat.ptr(pre[1][n0:n1]),
                                      nat.ptr(fwd[1]), nat.ptr(bwd[1]), nat.ptr(b_hh[0]),
                                      nat.ptr(b_hh[1]),
                                      nat.ptr(None if lens is None else lens[n0:n1]),
                                      nat.ptr(y[n0:n1]), n1 - n0, T, H, 1, lstm_share(),
                                      nat.ptr(ws), nat.stream_of(x))

        _lstm_chunks(lib, run, N, x, "aps_lstm_layer")
        ys.append(y)
        out = y
    return ys


RNN_STEP_MODES = {"GRU": 0, "RNN_TANH": 1, "RNN_RELU": 2, "LSTM": 3}


def rnn_step_supported(rnn: th.nn.Module, x: th.Tensor) -> bool:
    """can `rnn` run step by step on aps_rnn_step?  (any batch-first nn.GRU / nn.RNN / nn.LSTM on
    the GPU, outside autograd: the recurrences that have no persistent kernel)"""
    return (isinstance(rnn, th.nn.RNNBase) and rnn.mode in RNN_STEP_MODES and rnn.batch_first and
            x.is_cuda and x.dim() == 3 and not nat.needs_grad(x, *rnn.parameters()) and
            not (rnn.training and rnn.dropout > 0 and rnn.num_layers > 1))


def rnn_step_forward(rnn: th.nn.RNNBase, x: th.Tensor, lens: Optional[th.Tensor] = None) -> th.Tensor:
    """nn.GRU / nn.RNN / nn.LSTM (any hidden size, with or without proj_size) forward with zero
    initial state: x N x T x D -> N x T x (dirs H_out), frames past lens[n] zero.  The input
    projection of a layer is ONE GEMM over the whole sequence; the recurrence then costs one
    aps_linear (h W_hh^T + b_hh) and one aps_rnn_step launch per time step (+ one aps_linear for
    the projection of a projected LSTM) -- the sequential form, captured into the step's hipGraph
    like everything else.  The reverse direction runs on time-reversed utterances."""
    from aps_amd.grad_ops import reverse_time
    nat.require_device(x, lens, *rnn.parameters())
    lib = nat.load()
    mode = RNN_STEP_MODES[rnn.mode]
    G = {0: 3, 1: 1, 2: 1, 3: 4}[mode]
    N, T, _ = x.shape
    H = rnn.hidden_size
    P = rnn.proj_size if getattr(rnn, "proj_size", 0) > 0 else 0
    Ho = P if P else H
    if lens is not None:
        lens = lens.to(device=x.device, dtype=th.int64).contiguous()
    st = nat.stream_of(x)
    out = nat.f32c(x)
    for layer in range(rnn.num_layers):
        ys = []
        for d in range(2 if rnn.bidirectional else 1):
            sfx = f"_l{layer}" + ("_reverse" if d else "")
            w_ih, w_hh = getattr(rnn, "weight_ih" + sfx), getattr(rnn, "weight_hh" + sfx)
            b_ih = getattr(rnn, "bias_ih" + sfx) if rnn.bias else None
            b_hh = getattr(rnn, "bias_hh" + sfx) if rnn.bias else None
            w_hr = getattr(rnn, "weight_hr" + sfx) if P else None
            inp = reverse_time(out, lens) if d else out
            gx = linear(inp, w_ih, b_ih)  # N x T x G H
            y = th.empty(N, T, Ho, device=x.device, dtype=th.float32)
            h = th.zeros(N, Ho, device=x.device, dtype=th.float32)
            c = th.zeros(N, H, device=x.device, dtype=th.float32) if mode == 3 else None
            for t in range(T):
                gh = linear(h, w_hh, b_hh)  # N x G H
                h_new = th.empty(N, H, device=x.device, dtype=th.float32)
                c_new = th.empty_like(c) if c is not None else None
                frame = y[:, t]
                if P:  # the cell's h is projected before it is emitted / fed back
                    # (state freeze past the length: the cell keeps h_full = its own previous
                    # output there, which is not the projected state -- so masked rows are fixed up)
                    rc = lib.aps_rnn_step(nat.ptr(gx[:, t]), T * G * H, nat.ptr(gh), nat.ptr(None),
                                          nat.ptr(c), nat.ptr(lens), t, nat.ptr(h_new),
                                          nat.ptr(c_new), nat.ptr(None), 0, N, H, mode, st)
                    nat.check(rc, "aps_rnn_step")
                    hp = linear(h_new, w_hr)
                    if lens is not None:
                        live = (lens > t)[:, None]
                        hp = th.where(live, hp, h)
                        frame.copy_(th.where(live, hp, th.zeros_like(hp)))
                    else:
                        frame.copy_(hp)
                    h, c = hp, c_new
                else:
                    rc = lib.aps_rnn_step(nat.ptr(gx[:, t]), T * G * H, nat.ptr(gh), nat.ptr(h),
                                          nat.ptr(c), nat.ptr(lens), t, nat.ptr(h_new),
                                          nat.ptr(c_new), nat.ptr(frame), T * Ho, N, H, mode, st)
                    nat.check(rc, "aps_rnn_step")
                    h, c = h_new, c_new
            ys.append(reverse_time(y, lens) if d else y)
        out = ys[0] if len(ys) == 1 else th.cat(ys, dim=-1)
    return out


def rnn_step_trainable(rnn: th.nn.Module, x: th.Tensor) -> bool:
    """can `rnn` TRAIN step by step on aps_rnn_step / aps_rnn_step_backward?  (batch-first nn.GRU / nn.RNN /
    nn.LSTM with or without a projection, on the GPU, under autograd or in train() mode with inter-layer dropout)"""
    return (isinstance(rnn, th.nn.RNNBase) and rnn.mode in RNN_STEP_MODES and rnn.batch_first and
            x.is_cuda and x.dim() == 3)


def rnn_step_train(rnn: th.nn.RNNBase, x: th.Tensor, lens: Optional[th.Tensor] = None) -> th.Tensor:
    """`rnn_step_forward` under autograd: every layer and direction a grad_ops.RnnStepFn (BPTT on HIP:
    aps_rnn_step_backward + the GEMMs), nn.RNNBase's dropout between the layers in train() mode as the
    counter-based mask of grad_ops.DropoutFn"""
    from aps_amd.grad_ops import DropoutFn, LstmProjStepFn, RnnStepFn, draw_seed
    mode = RNN_STEP_MODES[rnn.mode]
    proj = getattr(rnn, "proj_size", 0) > 0
    if lens is not None:
        lens = lens.to(device=x.device, dtype=th.int64).contiguous()
    out = x
    for layer in range(rnn.num_layers):
        ys = []
        for d in range(2 if rnn.bidirectional else 1):
            sfx = f"_l{layer}" + ("_reverse" if d else "")
            w_ih, w_hh = getattr(rnn, "weight_ih" + sfx), getattr(rnn, "weight_hh" + sfx)
            b_ih = getattr(rnn, "bias_ih" + sfx) if rnn.bias else None
            b_hh = getattr(rnn, "bias_hh" + sfx) if rnn.bias else None
            if proj:
                ys.append(LstmProjStepFn.apply(out, lens, bool(d), w_ih, w_hh, getattr(rnn, "weight_hr" + sfx),
                                               b_ih, b_hh))
            else:
                ys.append(RnnStepFn.apply(out, lens, mode, bool(d), w_ih, w_hh, b_ih, b_hh))
        out = ys[0] if len(ys) == 1 else th.cat(ys, dim=-1)
        if rnn.training and rnn.dropout > 0 and layer + 1 < rnn.num_layers:
            out = DropoutFn.apply(out, float(rnn.dropout), draw_seed())
    return out


def lstm_supported(rnn: th.nn.Module, x: th.Tensor) -> bool:
    """can `rnn` run on aps_lstm_layer? (otherwise the caller keeps torch's MIOpen path)"""
    return (isinstance(rnn, th.nn.LSTM) and rnn.batch_first and rnn.proj_size == 0 and
            rnn.hidden_size in LSTM_HIDDEN_SIZES and x.is_cuda and x.dim() == 3 and
            x.shape[0] * x.shape[1] * rnn.hidden_size * (2 if rnn.bidirectional else 1) * 4 < 2**31)


def lstm_forward(rnn: th.nn.LSTM, x: th.Tensor, lens: Optional[th.Tensor] = None) -> th.Tensor:
    """nn.LSTM(batch_first=True) forward with zero initial state: x N x T x D -> N x T x (dirs H).
    Frames at t >= lens[n] come out as zeros (pad_packed_sequence semantics); the caller trims the
    time axis to max(lens) if it needs the reference's shape."""
    if lens is not None:
        lens = lens.to(device=x.device, dtype=th.int64).contiguous()
    layer_drop = rnn.training and rnn.dropout > 0 and rnn.num_layers > 1
    if nat.needs_grad(x, *rnn.parameters()) or layer_drop:
        from aps_amd.grad_ops import DropoutFn, LstmFn, draw_seed
        per_layer = []
        for l in range(rnn.num_layers):
            flat = []
            for sfx in ([""] if not rnn.bidirectional else ["", "_reverse"]):
                names = ["weight_ih", "weight_hh"] + (["bias_ih", "bias_hh"] if rnn.bias else [])
                flat += [getattr(rnn, f"{n}_l{l}{sfx}") for n in names]
            per_layer.append(flat)
        bias, bidir = bool(rnn.bias), bool(rnn.bidirectional)
        if not layer_drop:
            return LstmFn.apply(x, lens, rnn.num_layers, bias, bidir, *sum(per_layer, []))
        out = x  # train(): nn.LSTM drops the output of every layer but the last
        for l, flat in enumerate(per_layer):
            out = LstmFn.apply(out, lens, 1, bias, bidir, *flat)
            if l + 1 < rnn.num_layers:
                out = DropoutFn.apply(out, float(rnn.dropout), draw_seed())
        return out
    nat.require_device(x, lens, *rnn.parameters())
    lib = nat.load()
    N, T, _ = x.shape
    H = rnn.hidden_size
    dirs = 2 if rnn.bidirectional else 1
    out = nat.f32c(x)
    if dirs == 1 and 2 <= rnn.num_layers <= 4 and N <= LSTM_STACK_MAX_BATCH and \
            H in LSTM_STACK_SIZES and LSTM_STACK:
        stacked = _lstm_stack_forward(lib, _lstm_layer_params(rnn), out, lens)
        if stacked is not None:
            return stacked[-1]
    for layer in range(rnn.num_layers):
        y = th.empty(N, T, dirs * H, device=x.device, dtype=th.float32)
        pre, w_hh, b_hh = [], [], []
        for d in range(dirs):
            sfx = f"_l{layer}" + ("_reverse" if d else "")
            b_ih = getattr(rnn, "bias_ih" + sfx) if rnn.bias else None
            pre.append(linear(out, getattr(rnn, "weight_ih" + sfx), b_ih))  # N x T x 4H
            w_hh.append(nat.f32c(getattr(rnn, "weight_hh" + sfx)))
            b_hh.append(nat.f32c(getattr(rnn, "bias_hh" + sfx)) if rnn.bias else None)
        if dirs == 1:
            pre.append(None), w_hh.append(None), b_hh.append(None)
        def run(n0, n1, ws):
            return lib.aps_lstm_layer(nat.ptr(pre[0][n0:n1]),
                                      nat.ptr(None if pre[1] is None else pre[1][n0:n1]),
                                      nat.ptr(w_hh[0]), nat.ptr(w_hh[1]), nat.ptr(b_hh[0]),
                                      nat.ptr(b_hh[1]),
                                      nat.ptr(None if lens is None else lens[n0:n1]),
                                      nat.ptr(y[n0:n1]), n1 - n0, T, H, 1, lstm_share(),
                                      nat.ptr(ws), nat.stream_of(x))

        _lstm_chunks(lib, run, N, x, "aps_lstm_layer")
        out = y
    return out


# ------------------------------------------------------------------------------------------------
# channels-last convolution (aps_conv2d_nhwc)
# ------------------------------------------------------------------------------------------------
CONV_ACTS = {None: 0, "none": 0, "relu": 1, "leaky_relu": 5}
# optional profiling sink like GEMM_TIMELINE: (start_event, stop_event, flops) per conv launch
CONV_TIMELINE = None


def conv2d_nhwc(x: th.Tensor, weight: th.Tensor, scale: Optional[th.Tensor] = None,
                shift: Optional[th.Tensor] = None, stride=(1, 1), padding=(0, 0),
                transposed: bool = False, output_padding=(0, 0), act: Optional[str] = None,
                slope: float = 0.01, residual: Optional[th.Tensor] = None,
                crop=(0, 0), fp16: bool = False) -> th.Tensor:
    """x N x H x W x Ci, weight Co x KH x KW x Ci (channels-last form of the nn.Conv2d /
    nn.ConvTranspose2d weight, see include/aps_amd.h) -> N x Ho x Wo x Co with
    act(scale * conv + shift) (+ residual).  `crop` drops that many trailing output rows / columns
    (they are simply not computed): the truncation of the causal blocks, dcunet.py:90-100"""
    if nat.needs_grad(x, weight, scale, shift, residual):
        # training: the convolution (+ bias) with its adjoints, the rest of the epilogue as the
        # stand-alone passes that keep what their backward needs.  A BatchNorm folded into `scale`
        # is an eval-mode construct: training-mode blocks normalise with grad_ops.batchnorm_rows
        if scale is not None:
            raise NotImplementedError("aps_amd: conv2d_nhwc under autograd takes no folded "
                                      "BatchNorm scale (use grad_ops.batchnorm_rows behind it)")
        if act == "leaky_relu" and slope != 0.01:
            raise NotImplementedError("aps_amd: leaky_relu backward exists for slope 0.01")
        from aps_amd.grad_ops import Conv2dNhwcFn, ScaleAddFn, activation
        out = Conv2dNhwcFn.apply(x, weight, shift, tuple(stride), tuple(padding), bool(transposed),
                                 tuple(output_padding), tuple(crop))
        out = activation(out, act)
        return out if residual is None else ScaleAddFn.apply(out, residual, 1.0)
    nat.require_device(x, weight, scale, shift, residual)
    lib = nat.load()
    xc, w = nat.f32c(x), nat.f32c(weight)
    N, H, W, Ci = xc.shape
    Co, KH, KW, Ci2 = w.shape
    if Ci2 != Ci:
        raise RuntimeError(f"conv2d_nhwc: weight has {Ci2} input channels, input {Ci}")
    sh, sw = stride
    ph, pw = padding
    if transposed:
        Ho = (H - 1) * sh - 2 * ph + KH + output_padding[0]
        Wo = (W - 1) * sw - 2 * pw + KW + output_padding[1]
    else:
        Ho = (H + 2 * ph - KH) // sh + 1
        Wo = (W + 2 * pw - KW) // sw + 1
    Ho, Wo = Ho - crop[0], Wo - crop[1]
    if Ho <= 0 or Wo <= 0:
        raise RuntimeError(f"conv2d_nhwc: empty output ({Ho} x {Wo})")
    out = th.empty(N, Ho, Wo, Co, device=x.device, dtype=th.float32)
    res = None if residual is None else nat.f32c(residual)
    if res is not None and tuple(res.shape) != tuple(out.shape):
        raise RuntimeError(f"conv2d_nhwc: residual {tuple(res.shape)} != output {tuple(out.shape)}")

    def opt(t):
        return nat.ptr(None if t is None else nat.f32c(t))

    timeline = CONV_TIMELINE
    if timeline is not None:
        e0, e1 = th.cuda.Event(enable_timing=True), th.cuda.Event(enable_timing=True)
        e0.record()
    # the bf16-split form (csrc/gemm_split.hip:conv_split_kernel) for the layers whose weight has a
    # long-lived owner (the modules' cached channels-last weights mark themselves), Ci a multiple of
    # 32, at least 16 output channels and enough tiles to fill the chip
    owner = _weight_owner(weight) if Ci % 32 == 0 and Co >= CONV_SPLIT_MIN_CO and \
        _use_split(N * Ho * Wo, Co, KH * KW * Ci, CONV_SPLIT_MIN_TILES) else None
    if owner is not None:
        if fp16 if CONV_FP16X2 is None else CONV_FP16X2:
            planes, w32 = _split_planes(w.view(Co, KH * KW * Ci), owner, "conv16", layout=2,
                                        with_source=True)
            pixexp = th.empty(N * H * W, device=x.device, dtype=th.int32)  # exponent of every input pixel
            rc = lib.aps_conv2d_nhwc_fp16x2(nat.ptr(xc), nat.ptr(planes), nat.ptr(w32), opt(scale),
                                            opt(shift), nat.ptr(res), nat.ptr(out), nat.ptr(pixexp),
                                            nat.ptr(_wide_counter(x.device)), N, H, W, Ci,
                                            Co, KH, KW, sh, sw, ph, pw, Ho, Wo, int(transposed),
                                            CONV_ACTS[act], float(slope), nat.stream_of(x))
        else:
            planes = _split_planes(w.view(Co, KH * KW * Ci), owner, "conv", layout=1)
            rc = lib.aps_conv2d_nhwc_split(nat.ptr(xc), nat.ptr(planes), opt(scale), opt(shift),
                                           nat.ptr(res), nat.ptr(out), N, H, W, Ci, Co, KH, KW, sh, sw,
                                           ph, pw, Ho, Wo, int(transposed), CONV_ACTS[act], float(slope),
                                           nat.stream_of(x))
        nat.check(rc, "aps_conv2d_nhwc_split")
    else:
        rc = lib.aps_conv2d_nhwc(nat.ptr(xc), nat.ptr(w), opt(scale), opt(shift), nat.ptr(res),
                                 nat.ptr(out), N, H, W, Ci, Co, KH, KW, sh, sw, ph, pw, Ho, Wo,
                                 int(transposed), CONV_ACTS[act], float(slope), nat.stream_of(x))
        nat.check(rc, "aps_conv2d_nhwc")
    if timeline is not None:
        e1.record()
        # algorithmic flops of the dense form (every tap of every output pixel; the zero taps of
        # padding / the stride holes of the transposed form are counted like the reference's
        # flop counter counts them for conv2d; for conv_transpose2d the useful taps are 1/(sh sw))
        useful = 1.0 / (sh * sw) if transposed else 1.0
        timeline.append((e0, e1, 2.0 * N * Ho * Wo * Co * KH * KW * Ci * useful,
                         f"{'deconv' if transposed else 'conv'} {N}x{H}x{W}x{Ci}->{Ho}x{Wo}x{Co} "
                         f"k{KH}x{KW} s{sh}x{sw}"))
    return out


def lstm_pair_forward(rnn_a: th.nn.LSTM, rnn_b: th.nn.LSTM, x: th.Tensor):
    """Two independent, identically shaped unidirectional nn.LSTM stacks over the SAME input (the
    real / imaginary LSTMs of a complex LSTM) in one launch per layer: N x T x D -> (ya, yb), each
    N x T x H (column halves of one N x T x 2H buffer)."""
    for r in (rnn_a, rnn_b):
        if r.bidirectional or r.proj_size != 0 or not r.batch_first or \
                (r.training and r.dropout > 0 and r.num_layers > 1):
            raise NotImplementedError("lstm_pair_forward: unidirectional eval-mode LSTMs only")
    if (rnn_a.hidden_size, rnn_a.num_layers, rnn_a.input_size, rnn_a.bias) != \
            (rnn_b.hidden_size, rnn_b.num_layers, rnn_b.input_size, rnn_b.bias):
        raise RuntimeError("lstm_pair_forward: the two LSTMs must have the same geometry")
    nat.require_device(x, *rnn_a.parameters(), *rnn_b.parameters())
    lib = nat.load()
    N, T, _ = x.shape
    H = rnn_a.hidden_size
    ins = [nat.f32c(x), nat.f32c(x)]
    for layer in range(rnn_a.num_layers):
        y = th.empty(N, T, 2 * H, device=x.device, dtype=th.float32)
        pre, w_hh, b_hh = [], [], []
        for r, inp in zip((rnn_a, rnn_b), ins):
            sfx = f"_l{layer}"
            pre.append(linear(inp, getattr(r, "weight_ih" + sfx),
                              getattr(r, "bias_ih" + sfx) if r.bias else None))
            w_hh.append(nat.f32c(getattr(r, "weight_hh" + sfx)))
            b_hh.append(nat.f32c(getattr(r, "bias_hh" + sfx)) if r.bias else None)
        def run(n0, n1, ws):
            return lib.aps_lstm_layer(nat.ptr(pre[0][n0:n1]), nat.ptr(pre[1][n0:n1]),
                                      nat.ptr(w_hh[0]), nat.ptr(w_hh[1]), nat.ptr(b_hh[0]),
                                      nat.ptr(b_hh[1]), None, nat.ptr(y[n0:n1]), n1 - n0, T, H, 0,
                                      lstm_share(), nat.ptr(ws), nat.stream_of(x))

        _lstm_chunks(lib, run, N, x, "aps_lstm_layer (pair)")
        ins = [y[..., :H], y[..., H:]]
    return ins[0], ins[1]
