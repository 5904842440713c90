"""
(real, imag) pair container with the interface the reference's networks exchange
(aps/cplx.py:18-185).  Here it is only a CARRIER between the HIP kernels: the arithmetic the
reference does with it on the MVDR path (A @ B^H, inverse, trace, division) lives in
aps_amd/csrc/mvdr.hip.  The light accessors below are views / single torch ops (plumbing).
"""
from typing import Optional

import torch as th


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
        if isinstance(other, ComplexTensor):
            return ComplexTensor(th.matmul(self.real, other.real) - th.matmul(self.imag, other.imag),
                                 th.matmul(self.real, other.imag) + th.matmul(self.imag, other.real))
        return ComplexTensor(th.matmul(self.real, other), th.matmul(self.imag, other))

    def __rmatmul__(self, other):
        if isinstance(other, ComplexTensor):
            return other.__matmul__(self)
        return ComplexTensor(th.matmul(other, self.real), th.matmul(other, self.imag))

    def inverse(self) -> "ComplexTensor":
        """inverse of (...) x C x C complex matrices through the real 2C x 2C embedding
        [[R, -I], [I, R]] (cplx.py:268-278)"""
        C_ = self.real.shape[-1]
        top = th.cat([self.real, -self.imag], -1)
        bot = th.cat([self.imag, self.real], -1)
        inv = th.linalg.inv(th.cat([top, bot], -2))
        return ComplexTensor(inv[..., :C_, :C_], inv[..., C_:, :C_])

    def __repr__(self) -> str:
        return f"ComplexTensor(shape={tuple(self.shape)}, device={self.device})"
