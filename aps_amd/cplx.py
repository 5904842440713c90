"""
(real, imag) pair container with the interface the reference's networks exchange
(aps/cplx.py:18-185).  Here it is only a CARRIER between the HIP kernels: the arithmetic the
reference does with it on the MVDR path (A @ B^H, inverse, trace, division) lives in
aps_amd/csrc/mvdr.hip.  The light accessors below are views / single torch ops (plumbing).

Where a call leaves the HIP kernels (documented, not hidden): `@` and `inverse()` run on
aps_cplx_matmul / aps_cplx_inverse for GPU float32 operands of covariance size outside autograd
(K <= 64, C <= 8: every use on the MVDR path and the reference's own tests, aps/cplx.py:301-364); other
shapes, other dtypes and calls under autograd are torch.matmul / torch.linalg.inv on the same device --
a general-purpose container has no single hot shape to write a kernel for, and the trainable MVDR
path does not go through it (MvdrBeamformer.forward_trainable uses the grad_ops functions).
"""
from typing import Optional

import torch as th

from aps_amd import _native


class ComplexTensor(object):

    def __init__(self, real: th.Tensor, imag: Optional[th.Tensor] = None, polar: bool = False):
        imag = th.zeros_like(real) if imag is None else imag
        if polar:
            self.real = th.cos(imag) * real
            self.imag = th.sin(imag) * real
        else:
            self.real = real
            self.imag = imag

    # ---- structure ------------------------------------------------------------------------
    @property
    def shape(self) -> th.Size:
        return self.real.shape

    @property
    def device(self) -> th.device:
        return self.real.device

    @property
    def dtype(self) -> th.dtype:
        return self.real.dtype

    def size(self) -> th.Size:
        return self.real.size()

    def dim(self) -> int:
        return self.real.dim()

    def _map(self, fn) -> "ComplexTensor":
        return ComplexTensor(fn(self.real), fn(self.imag))

    def transpose(self, dim0, dim1) -> "ComplexTensor":
        return self._map(lambda x: x.transpose(dim0, dim1))

    def view(self, *shape) -> "ComplexTensor":
        return self._map(lambda x: x.view(*shape))

    def contiguous(self) -> "ComplexTensor":
        return self._map(lambda x: x.contiguous())

    def to(self, *args, **kwargs) -> "ComplexTensor":
        return self._map(lambda x: x.to(*args, **kwargs))

    def cpu(self) -> "ComplexTensor":
        return self._map(lambda x: x.cpu())

    def cuda(self) -> "ComplexTensor":
        return self._map(lambda x: x.cuda())

    def __getitem__(self, item) -> "ComplexTensor":
        return self._map(lambda x: x[item])

    def masked_fill(self, mask, value) -> "ComplexTensor":
        return self._map(lambda x: x.masked_fill(mask, value))

    def masked_select(self, mask) -> "ComplexTensor":
        return self._map(lambda x: x.masked_select(mask))

    def sum(self, dim=None, keepdim=False) -> "ComplexTensor":
        return self._map(lambda x: x.sum(dim=dim, keepdim=keepdim))

    # ---- value accessors -------------------------------------------------------------------
    def conj(self) -> "ComplexTensor":
        return ComplexTensor(self.real, -1.0 * self.imag)

    def conj_transpose(self, dim0, dim1) -> "ComplexTensor":
        return self.transpose(dim0, dim1).conj()

    def abs(self) -> th.Tensor:
        return (self.real**2 + self.imag**2).sqrt()

    def angle(self) -> th.Tensor:
        return th.atan2(self.imag, self.real)

    def as_real(self) -> th.Tensor:
        return th.stack([self.real, self.imag], dim=-1)

    def __add__(self, other):
        if isinstance(other, (ComplexTensor, complex)):
            return ComplexTensor(self.real + other.real, self.imag + other.imag)
        return ComplexTensor(self.real + other, self.imag)

    __radd__ = __add__

    def __mul__(self, other):
        if isinstance(other, (ComplexTensor, complex)):
            return ComplexTensor(self.real * other.real - self.imag * other.imag,
                                 self.imag * other.real + self.real * other.imag)
        return ComplexTensor(self.real * other, self.imag * other)

    __rmul__ = __mul__

    # ---- remaining algebra of the reference class (aps/cplx.py:47-110, 221-278).  Not on the
    # kernels' path (the MVDR chain consumes covariances in aps_amd/csrc/mvdr.hip); provided so that
    # code written against the reference's ComplexTensor keeps working.  Plain torch ops on
    # whatever device the halves live on.
    def __neg__(self) -> "ComplexTensor":
        return ComplexTensor(-self.real, -self.imag)

    def __sub__(self, other):
        if isinstance(other, (ComplexTensor, complex)):
            return ComplexTensor(self.real - other.real, self.imag - other.imag)
        return ComplexTensor(self.real - other, self.imag)

    def __rsub__(self, other):
        return (-self).__add__(other)

    def __truediv__(self, other):
        if isinstance(other, (ComplexTensor, complex)):
            den = other.real**2 + other.imag**2
            return ComplexTensor((self.real * other.real + self.imag * other.imag) / den,
                                 (self.imag * other.real - self.real * other.imag) / den)
        return ComplexTensor(self.real / other, self.imag / other)

    def __rtruediv__(self, other):
        den = self.real**2 + self.imag**2
        if isinstance(other, (ComplexTensor, complex)):
            return ComplexTensor((other.real * self.real + other.imag * self.imag) / den,
                                 (other.imag * self.real - other.real * self.imag) / den)
        return ComplexTensor(other * self.real / den, -other * self.imag / den)

    def __matmul__(self, other):
        """(...) x M x K @ (...) x K x N (aps/cplx.py:242-266).  GPU float32 operands of covariance size
        (K <= 64, equal or absent batch dims on `other`) run on aps_cplx_matmul -- one launch for the
        four real products and two sums of the reference; anything else on torch.matmul"""
        o_re, o_im = (other.real, other.imag) if isinstance(other, ComplexTensor) else (other, None)
        out = _hip_matmul(self.real, self.imag, o_re, o_im)
        if out is not None:
            return out
        if o_im is not None:
            return ComplexTensor(th.matmul(self.real, o_re) - th.matmul(self.imag, o_im),
                                 th.matmul(self.real, o_im) + th.matmul(self.imag, o_re))
        return ComplexTensor(th.matmul(self.real, o_re), th.matmul(self.imag, o_re))

    def __rmatmul__(self, other):
        if isinstance(other, ComplexTensor):
            return other.__matmul__(self)
        out = _hip_matmul(other, None, self.real, self.imag)
        if out is not None:
            return out
        return ComplexTensor(th.matmul(other, self.real), th.matmul(other, self.imag))

    def inverse(self) -> "ComplexTensor":
        """inverse of (...) x C x C complex matrices (cplx.py:268-278: the reference inverts the real
        2C x 2C embedding [[R, -I], [I, R]] by LU).  GPU float32, C <= 8: complex Gauss-Jordan with
        partial pivoting in registers, one matrix per lane (aps_cplx_inverse), in eval AND under autograd
        (the adjoint G_A = -Y^H G Y^H on aps_cplx_matmul: both modes share the arithmetic); a singular
        matrix raises torch.linalg.LinAlgError like th.inverse does (checked right after the call -- th.inverse
        synchronises for the same check; inside a stream capture, or with `cplx.INVERSE_SINGULAR_POLICY = "deferred"`,
        the check is left to `singular_matrices`).
        Otherwise the embedding on torch.linalg.inv."""
        C_ = self.real.shape[-1]
        if self.real.is_cuda and self.real.dtype == th.float32 and 1 <= C_ <= 8 and \
                self.real.shape[-2] == C_ and self.imag.is_cuda and self.imag.dtype == th.float32:
            if _native.needs_grad(self.real, self.imag):
                o_re, o_im = _InverseFn.apply(self.real, self.imag)
            else:
                o_re, o_im = _hip_inverse(self.real, self.imag)
            return ComplexTensor(o_re, o_im)
        top = th.cat([self.real, -self.imag], -1)
        bot = th.cat([self.imag, self.real], -1)
        inv = th.linalg.inv(th.cat([top, bot], -2))
        return ComplexTensor(inv[..., :C_, :C_], inv[..., C_:, :C_])

    def __repr__(self) -> str:
        return f"ComplexTensor(shape={tuple(self.shape)}, device={self.device})"


_SINGULAR = {}  # device index -> sticky int32 counter of singular matrices (aps_cplx_inverse)
# When `ComplexTensor.inverse()` looks at that counter: "sync" (default) right behind the call, the reference's
# behaviour -- th.inverse raises from the call and synchronises to do so (aps/cplx.py:268-278); "deferred": never inside
# the call -- no host stall per inverse; the caller reads `singular_matrices()` where it has to stop anyway, as inside a
# stream capture.  (The MVDR solves have the same choice as MvdrBeamformer.singular_policy.)
INVERSE_SINGULAR_POLICY = "sync"


def _singular_counter(device: th.device):
    key = device.index if device.index is not None else th.cuda.current_device()
    t = _SINGULAR.get(key)
    if t is None and not th.cuda.is_current_stream_capturing():
        t = _SINGULAR[key] = th.zeros(1, dtype=th.int32, device=th.device("cuda", key))
    return t


def singular_matrices(device=None) -> int:
    """matrices `ComplexTensor.inverse()` met with a zero / non-finite pivot on `device` since the last read
    (blocking; for callers that invert inside a captured graph, where the call itself cannot look)"""
    dev = th.device("cuda", th.cuda.current_device()) if device is None else th.device(device)
    t = _singular_counter(dev)
    if t is None:
        return 0
    c = int(t.item())
    if c:
        t.zero_()
    return c


def _hip_inverse(real: th.Tensor, imag: th.Tensor):
    lib = _native.load()
    C_ = real.shape[-1]
    re, im = _native.f32c(real.detach()), _native.f32c(imag.detach())
    o_re, o_im = th.empty_like(re), th.empty_like(im)
    B = re.numel() // (C_ * C_)
    if B > 0:
        flag = _singular_counter(re.device)
        _native.check(lib.aps_cplx_inverse(_native.ptr(re), _native.ptr(im), _native.ptr(o_re), _native.ptr(o_im),
                                           B, C_, _native.ptr(flag), _native.stream_of(re)), "aps_cplx_inverse")
        if INVERSE_SINGULAR_POLICY not in ("sync", "deferred"):
            raise ValueError(f"cplx.INVERSE_SINGULAR_POLICY must be sync | deferred, got {INVERSE_SINGULAR_POLICY!r}")
        if flag is not None and INVERSE_SINGULAR_POLICY == "sync" and not th.cuda.is_current_stream_capturing():
            bad = int(flag.item())
            if bad:
                flag.zero_()
                raise th.linalg.LinAlgError(f"linalg.inv: {bad} of {B} matrices are singular (a zero or "
                                            f"non-finite pivot), input shape = {tuple(real.shape)}")
    return o_re, o_im


class _InverseFn(th.autograd.Function):
    """Y = A^-1 on aps_cplx_inverse with its adjoint: for a real loss with G = dL/dRe Y + i dL/dIm Y,
    dL/dA = -Y^H G Y^H (from dY = -Y dA Y), two aps_cplx_matmul launches"""

    @staticmethod
    def forward(ctx, real, imag):
        o_re, o_im = _hip_inverse(real, imag)
        ctx.save_for_backward(o_re, o_im)
        return o_re, o_im

    @staticmethod
    def backward(ctx, g_re, g_im):
        o_re, o_im = ctx.saved_tensors
        yh = ComplexTensor(o_re.transpose(-1, -2).contiguous(), -o_im.transpose(-1, -2).contiguous())
        g = ComplexTensor(_native.f32c(g_re), _native.f32c(g_im))
        ga = yh @ g @ yh
        return -ga.real, -ga.imag


def _hip_matmul(a_re, a_im, b_re, b_im):
    """A @ B on aps_cplx_matmul when both are GPU float32, at least 2-D, K <= 64, and B's batch dims
    either equal A's or are absent (one matrix for every A); None = not this kernel's case"""
    if not (isinstance(a_re, th.Tensor) and isinstance(b_re, th.Tensor)):
        return None
    if not (a_re.is_cuda and b_re.is_cuda and a_re.dtype == b_re.dtype == th.float32):
        return None
    if a_re.dim() < 2 or b_re.dim() < 2 or a_re.shape[-1] != b_re.shape[-2] or a_re.shape[-1] > 64:
        return None
    if _native.needs_grad(a_re, a_im, b_re, b_im):
        return None
    batch = tuple(a_re.shape[:-2])
    if tuple(b_re.shape[:-2]) not in (batch, ()):
        return None
    M, K, N = a_re.shape[-2], a_re.shape[-1], b_re.shape[-1]
    B = 1
    for d in batch:
        B *= d
    if B == 0 or M == 0 or N == 0:
        return None
    lib = _native.load()
    ar, br = _native.f32c(a_re), _native.f32c(b_re)
    ai = None if a_im is None else _native.f32c(a_im)
    bi = None if b_im is None else _native.f32c(b_im)
    c_re = th.empty(*batch, M, N, device=a_re.device, dtype=th.float32)
    c_im = th.empty_like(c_re)
    _native.check(lib.aps_cplx_matmul(_native.ptr(ar), _native.ptr(ai), _native.ptr(br), _native.ptr(bi),
                                      _native.ptr(c_re), _native.ptr(c_im), B, M, K, N, M * K,
                                      K * N if b_re.dim() > 2 else 0, _native.stream_of(ar)),
                  "aps_cplx_matmul")
    return ComplexTensor(c_re, c_im)
