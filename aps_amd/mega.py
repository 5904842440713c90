"""
Host side of aps_conformer_stack (csrc/conformer_mega.hip, round 6): the conformer encoder stack of
aps/asr/transformer/impl.py:432-541, 718-756 as ONE launch per batch, a workgroup per utterance.

`conformer_stack(encoder, x, lens, rel)` is what `ApsTransformerEncoder.run` calls for a stack the kernel is built
for (pre-norm macaron conformer layers with learnt relative positions, D = 512, FF = 1024, 64-wide heads, 15 taps, eval
mode, T <= 64 encoder frames -- BASELINE configs[4]'s encoder at 4 s utterances); everything else, training and
autograd stay on the per-launch path, which is also this kernel's oracle-checked twin
(tests/test_gpu_mega.py compares the two and both against the CPU oracle).

The per-layer tables (ApsMegaLayer, include/aps_amd.h) point at the SAME derived weights the per-launch path uses: the
LayerNorm-folded matrices of nn_ops._ln_folded and the two-plane fragment images of nn_ops._split_planes, cached where
that path caches them and rebuilt when a parameter's version counter moves.
"""
import ctypes as C
import os
from typing import Optional

import torch as th

from . import _native as nat
from . import nn_ops

# "auto": the stacks the kernel takes run on it while FOUR or more streams are launching (nn_ops.STREAMS_IN_FLIGHT:
# replicas.PipelinedReplicas with three or more workers).  A workgroup per utterance leaves a 32-utterance batch on 32
# of the 256 CUs: per batch the launch costs 4.0 / 2.1 / 1.5 / 1.2 / 0.93 ms with 1 / 2 / 3 / 4 / 6 of them in flight
# against 2.0 ms for the per-launch path on one stream and ~2.0 with three in flight -- a lone batch and two whole steps
# in flight (GraphReplicas(replicas=2): 12.2 k utt/s per launch against 9.7 k on this kernel) keep one launch per
# projection.  True / False force it on / off (tests, A/B runs; APS_MEGA=1 | 0 in the environment)
ENABLED = {"1": True, "0": False}.get(os.environ.get("APS_MEGA", ""), "auto")
# launches of aps_conformer_stack since import (tests assert the path they mean to exercise ran)
CALLS = 0
# bench.py: set to a list to collect (encoder, x, lens, rel) of every call (the measurement legs re-issue them)
RECORD = None


class MegaGemm(C.Structure):
    _fields_ = [("image", C.c_void_p), ("w32", C.c_void_p), ("bias", C.c_void_p), ("colsum", C.c_void_p),
                ("N", C.c_int32), ("ksteps_total", C.c_int32), ("kstep0", C.c_int32), ("ldw", C.c_int32),
                ("act", C.c_int32), ("pad0", C.c_int32), ("alpha", C.c_float), ("ln_eps", C.c_float)]


class MegaLayer(C.Structure):
    _fields_ = [(n, MegaGemm) for n in ("ff1_up", "ff1_dn0", "ff1_dn1", "qkv", "out", "pw1", "pw2", "ff2_up",
                                        "ff2_dn0", "ff2_dn1")] + \
               [("dw_w", C.c_void_p), ("dw_b", C.c_void_p), ("bn_scale", C.c_void_p), ("bn_shift", C.c_void_p),
                ("conv_act", C.c_int32), ("pad1", C.c_int32)]


_CONV_ACT = {"none": 0, "swish": 1, "relu": 2, "gelu": 3}


def _layer_ok(mod, D: int) -> bool:
    from aps_amd.asr.transformer.impl import ApsConformerEncoderLayer, RelMultiheadAttention
    if not isinstance(mod, ApsConformerEncoderLayer) or not mod.pre_norm or mod.feedforward1 is None:
        return False
    att = mod.self_attn
    if type(att) is not RelMultiheadAttention or att.head_dim != 64 or att.embed_dim != D or att.in_proj_bias is None:
        return False
    c = mod.convolution
    if mod.padding > 0 or c[2].kernel_size[0] != 15 or c[2].weight.shape[0] != D or c[3].training:
        return False
    if mod.activation not in ("swish", "relu", "gelu") or mod.activation not in nn_ops.ACTIVATIONS:
        return False
    for ffn in (mod.feedforward1, mod.feedforward2):
        if ffn[0].out_features != 2 * D or ffn[0].in_features != D or ffn[0].bias is None or ffn[3].bias is None:
            return False
    return True


def supported(encoder, x: th.Tensor, rel: Optional[th.Tensor], window) -> bool:
    """can `encoder` (an ApsTransformerEncoder) run x [N, T, D] on aps_conformer_stack?"""
    if rel is None or window is not None or encoder.training or not x.is_cuda or x.dim() != 3:
        return False
    N, T, D = x.shape
    if T > 64 or D != 512 or N > 65535 or rel.dim() != 2 or rel.shape[-1] != 64:
        return False
    ok = encoder.__dict__.get("_aps_mega_ok")
    if ok is None:
        ok = encoder.__dict__["_aps_mega_ok"] = all(_layer_ok(m, D) for m in encoder.layers)
    return ok


def _gemm(weight, bias, ln, act, alpha, khalf=None, with_bias=True):
    """one projection phase; returns (MegaGemm, tensors to keep alive)"""
    if ln is not None:
        wg, cs, bb = nn_ops._ln_folded(weight, bias, ln)
        planes, w32 = nn_ops._split_planes(wg, ln.__dict__.setdefault("_aps_fold_split", {}), str(weight.data_ptr()),
                                           with_source=True)
        bias_t, colsum, eps = bb, cs, float(ln.eps)
    else:
        owner = nn_ops._weight_owner(weight)
        if owner is None:
            raise RuntimeError("aps_amd.mega: a projection weight that is neither a Parameter nor a view of one")
        planes, w32 = nn_ops._split_planes(weight, owner, "w", with_source=True)
        bias_t, colsum, eps = (None if bias is None else nat.f32c(bias.detach())), None, 0.0
    N, K = w32.shape
    if K % 512 or N % 32:
        raise RuntimeError(f"aps_amd.mega: projection {N} x {K} is not made of 512-wide phases / 32-column blocks")
    if not with_bias:
        bias_t = None
    g = MegaGemm(planes.data_ptr(), w32.data_ptr(), 0 if bias_t is None else bias_t.data_ptr(),
                 0 if colsum is None else colsum.data_ptr(), N, K // 32, 16 * (khalf or 0), K,
                 nn_ops.ACTIVATIONS[act], 0, float(alpha), eps)
    return g, (planes, w32, bias_t, colsum)


def _build(encoder, device):
    keep = []
    layers = (MegaLayer * len(encoder.layers))()
    for i, mod in enumerate(encoder.layers):
        L = layers[i]
        D = mod.norm_attn.normalized_shape[0]

        def put(name, *a, **k):
            g, refs = _gemm(*a, **k)
            setattr(L, name, g)
            keep.append(refs)

        for tag, ffn, ln in (("ff1", mod.feedforward1, mod.norm_ffn1), ("ff2", mod.feedforward2, mod.norm_ffn2)):
            put(tag + "_up", ffn[0].weight, ffn[0].bias, ln, mod.activation, 1.0)
            put(tag + "_dn0", ffn[3].weight, ffn[3].bias, None, None, mod.macaron_factor, khalf=0)
            put(tag + "_dn1", ffn[3].weight, ffn[3].bias, None, None, mod.macaron_factor, khalf=1, with_bias=False)
        att = mod.self_attn
        put("qkv", att.in_proj_weight, att.in_proj_bias, mod.norm_attn, None, 1.0)
        put("out", att.out_proj.weight, att.out_proj.bias, None, None, 1.0)
        c = mod.convolution
        put("pw1", c[0].weight.view(2 * D, D), c[0].bias, mod.norm_conv, None, 1.0)
        put("pw2", c[5].weight.view(D, D), c[5].bias, None, None, 1.0)
        scale, shift = mod._bn_affine()
        dw_w = nat.f32c(c[2].weight.detach().reshape(D, 15))
        dw_b = None if c[2].bias is None else nat.f32c(c[2].bias.detach())
        L.dw_w, L.dw_b = dw_w.data_ptr(), (0 if dw_b is None else dw_b.data_ptr())
        L.bn_scale, L.bn_shift = scale.data_ptr(), shift.data_ptr()
        L.conv_act = _CONV_ACT[mod.activation]
        keep.append((dw_w, dw_b, scale, shift))
    raw = bytes(layers)
    table = th.frombuffer(bytearray(raw), dtype=th.uint8).to(device)
    return table, keep


def _version_key(encoder):
    # (one pass over ~20 tensors per layer, per call: how many there are, the sum of their version counters -- any
    # in-place update, optimiser step or load_state_dict moves it -- and the sum of their addresses -- a parameter that
    # was REPLACED (module.weight = ..., .to(), pruning hooks) moves that)
    ts = list(encoder.parameters()) + list(encoder.buffers())
    return (len(ts), sum(t._version for t in ts), sum(t.data_ptr() >> 4 for t in ts))


def layer_table(encoder, device) -> th.Tensor:
    """the DEVICE array of ApsMegaLayer of `encoder`, rebuilt when a parameter or buffer changes"""
    key = (_version_key(encoder), str(device))
    hit = encoder.__dict__.get("_aps_mega")
    if hit is None or hit[0] != key:
        nn_ops._refuse_first_use_in_capture("the layer table of aps_conformer_stack")
        table, keep = _build(encoder, device)
        hit = encoder.__dict__["_aps_mega"] = (key, table, keep)
    return hit[1]


def conformer_stack(encoder, x: th.Tensor, lens: Optional[th.Tensor], rel: th.Tensor) -> Optional[th.Tensor]:
    """x N x T x D through every layer of `encoder` (NOT its final norm) in one launch; None if the library refuses
    the shape (the caller keeps the per-launch path)"""
    nat.require_device(x, rel)
    lib = nat.load()
    N, T, D = x.shape
    FF = encoder.layers[0].feedforward1[0].out_features
    H = encoder.layers[0].self_attn.num_heads
    table = layer_table(encoder, x.device)
    out = nat.f32c(x).clone()
    relc = nat.f32c(rel)
    if lens is not None:
        lens = lens.to(device=x.device, dtype=th.int64).contiguous()
    scratch = th.empty(N * lib.aps_conformer_stack_scratch(D, FF), device=x.device, dtype=th.float32)
    rc = lib.aps_conformer_stack(nat.ptr(out), nat.ptr(lens), nat.ptr(table), len(encoder.layers), nat.ptr(relc),
                                 (relc.shape[0] - 1) // 2, relc.shape[0], N, T, D, FF, H, nat.ptr(scratch),
                                 nat.ptr(nn_ops._wide_counter(x.device)), nat.stream_of(x))
    if rc == nat.ERR_UNSUPPORTED:
        return None
    nat.check(rc, "aps_conformer_stack")
    global CALLS
    CALLS += 1
    if RECORD is not None:
        RECORD.append((encoder, x, lens, rel))
    return out


def projection_flops(encoder, N: int, T: int) -> float:
    """ALGORITHMIC flops of the projections of one aps_conformer_stack launch: 2 T sum(N K) per utterance and layer
    (attention and the depthwise convolution, < 3 % of it, run on the fp32 pipes and are not counted)"""
    total = 0.0
    for mod in encoder.layers:
        ws = [mod.feedforward1[0].weight, mod.feedforward1[3].weight, mod.feedforward2[0].weight,
              mod.feedforward2[3].weight, mod.self_attn.in_proj_weight, mod.self_attn.out_proj.weight,
              mod.convolution[0].weight, mod.convolution[5].weight]
        total += sum(float(w.numel()) for w in ws)
    return 2.0 * N * T * total


def wanted() -> bool:
    """`ENABLED` = "auto": on while four or more streams are launching (replicas.PipelinedReplicas holds
    nn_ops.STREAMS_IN_FLIGHT = workers + 1)"""
    if ENABLED == "auto":
        return nn_ops.STREAMS_IN_FLIGHT >= 4
    return bool(ENABLED)
