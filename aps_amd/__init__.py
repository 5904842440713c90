"""
aps_amd: MI355X (gfx950) native implementation of the aps front-end hot path
(waveform -> framed STFT -> spectral / spatial features -> mask based MVDR -> features),
behind the reference's own plugin surface (aps.transform.AsrTransform / EnhTransform,
aps.libs registries).  Arithmetic runs in hand written HIP kernels reached through the C-ABI
declared in include/aps_amd.h; there is no CPU fallback.
"""
__version__ = "0.1.0"
