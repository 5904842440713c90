"""
aps_amd: MI355X (gfx950) native implementation of the aps front-end hot path
(waveform -> framed STFT -> spectral / spatial features -> mask based MVDR -> features),
behind the reference's own plugin surface (aps.transform.AsrTransform / EnhTransform,
aps.libs registries).  Arithmetic runs in hand written HIP kernels reached through the C-ABI
declared in include/aps_amd.h; there is no CPU fallback.
"""
__version__ = "0.1.0"

import os as _os
import sys as _sys

# Several batches in flight (aps_amd.replicas.PipelinedReplicas: the head stream + 3 workers, `EnhASRBase.serve`) need
# every stream on a hardware queue of its own.  HIP multiplexes its streams onto GPU_MAX_HW_QUEUES queues (4 when
# unset) and reads the variable when the runtime INITIALISES -- the first HIP call of the process, not `import
# torch` -- so the package asks for 8 here unless the caller chose a number or the runtime is already up
# (tests/test_gpu_replicas.py::test_import_sets_the_hardware_queues measures that this takes effect).
HW_QUEUES_WANTED = 8


def _default_hardware_queues() -> None:
    if "GPU_MAX_HW_QUEUES" in _os.environ:
        return
    torch = _sys.modules.get("torch")
    if torch is not None and torch.cuda.is_initialized():
        import warnings
        warnings.warn("aps_amd was imported after the HIP runtime started with the default 4 hardware queues: more than "
                      "three streams in flight (replicas.PipelinedReplicas, EnhASRBase.serve) will share queues, i.e. "
                      "serialise (measured 11.3 k against 14.7 k utt/s).  Import aps_amd -- or set GPU_MAX_HW_QUEUES=8 "
                      "-- before the first CUDA / HIP call of the process.")
        return
    _os.environ["GPU_MAX_HW_QUEUES"] = str(HW_QUEUES_WANTED)


_default_hardware_queues()
