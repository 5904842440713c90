"""
ctypes binding of the C-ABI in include/aps_amd.h.

The library is built ahead of time (python -m aps_amd.build).  It is NOT built or faked at import
time: if it is missing, every call raises, loudly -- the product path has no CPU fallback.
torch must be imported before the library is opened so that both bind the same libamdhip64.
"""
import ctypes as C
import os

import torch as th

_LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "libaps_amd.so")
ABI_VERSION = 58


class StftParams(C.Structure):
    _fields_ = [("fft_size", C.c_int32), ("frame_len", C.c_int32), ("frame_hop", C.c_int32),
                ("num_bins", C.c_int32), ("center", C.c_int32), ("polar", C.c_int32),
                ("pre_emphasis", C.c_float), ("eps", C.c_float), ("scale", C.c_float)]


class FeatParams(C.Structure):
    _fields_ = [("num_bins", C.c_int32), ("num_channels", C.c_int32), ("ref_channel", C.c_int32),
                ("power", C.c_int32), ("num_mels", C.c_int32), ("apply_log", C.c_int32),
                ("norm_mean", C.c_int32), ("norm_var", C.c_int32), ("num_pairs", C.c_int32),
                ("ipd_sin", C.c_int32), ("log_eps", C.c_float), ("log_lower_bound", C.c_float),
                ("cmvn_eps", C.c_float)]


_P = C.c_void_p
_I64 = C.c_int64
_I32 = C.c_int32
_F = C.c_float


# name -> (restype, argtypes); must list every symbol include/aps_amd.h declares
SIGNATURES = {
    "aps_status_string": (C.c_char_p, [C.c_int]),
    "aps_abi_version": (C.c_int, []),
    "aps_stft_num_frames": (_I64, [_I64, C.POINTER(StftParams)]),
    "aps_stft_forward": (C.c_int, [_P, _I64, _I64, _P, C.POINTER(StftParams), _P, _I64, _I64, _I64,
                                   _P]),
    "aps_stft_forward_pcm16": (C.c_int, [_P, _I64, _I64, _P, C.POINTER(StftParams), _P, _I64, _I64, _I64,
                                         _P]),
    "aps_stft_inverse": (C.c_int, [_P, _I64, _I64, _I64, _I64, _P, C.POINTER(StftParams), _P, _I64,
                                   _P, _P]),
    "aps_stft_backward": (C.c_int, [_P, _I64, _I64, _I64, _I64, _P, C.POINTER(StftParams), _P, _I64,
                                    _P, _P]),
    "aps_stft_inverse_backward": (C.c_int, [_P, _I64, _I64, _P, C.POINTER(StftParams), _P, _I64,
                                            _I64, _I64, _P, _P]),
    "aps_enh_features": (C.c_int, [_P, _I64, _I64, _I64, _I64, _I64, C.POINTER(FeatParams), _P, _P,
                                   _P, _P, _P, _P, _P, _P, _P]),
    "aps_stft_features": (C.c_int, [_P, _I64, _I64, _I64, _P, C.POINTER(StftParams),
                                    C.POINTER(FeatParams), _P, _P, _P, _P, _P, _P, _P, _I64, _I64,
                                    _I64, _P, _P, _P]),
    "aps_stft_features_pcm16": (C.c_int, [_P, _I64, _I64, _I64, _P, C.POINTER(StftParams),
                                          C.POINTER(FeatParams), _P, _P, _P, _P, _P, _P, _P, _I64, _I64,
                                          _I64, _P, _P, _P]),
    "aps_abs_features": (C.c_int, [_P, _I64, _I64, _F, C.POINTER(FeatParams), _P, _P, _P, _P, _P,
                                   _P, _P]),
    "aps_row_features": (C.c_int, [_P, _I64, _I64, C.POINTER(FeatParams), _P, _P, _P, _P, _P, _P,
                                   _P]),
    "aps_mvdr_beamform_features": (C.c_int, [_P, _P, _I64, _I64, _I64, _I64, _I64, _I64, _I64, _F,
                                             C.POINTER(FeatParams), _P, _P, _P, _P, _P, _P, _P, _P]),
    "aps_mvdr_process_mask": (C.c_int, [_P, _P, _I64, _I64, _I64, _I32, _I32, _P, _P]),
    "aps_mvdr_covariance_workspace": (_I64, [_I64, _I64, _I64, _I64]),
    "aps_mvdr_covariance": (C.c_int, [_P, _I64, _I64, _I64, _I64, _I64, _I64, _I64, _P, _P, _I64, _P,
                                      _I32, _P, _P, _P, _P, _P, _P, _P]),
    "aps_mvdr_weights_workspace": (_I64, [_I64, _I64, _I64, _I64, _I64]),
    "aps_mvdr_weights": (C.c_int, [_P, _I64, _I64, _I64, _I64, _I64, _I64, _I64, _P, _P, _I64, _P, _I32, _I64,
                                   _P, _P, _P, _P, _F, _P, _P, _P, _P, _P, _P, _P]),
    "aps_mvdr_attention_scratch": (_I64, [_I64, _I64, _I64]),
    "aps_mvdr_channel_attention": (C.c_int, [_P, _I64, _I64, _I64, _I64, _P, _P, _P, _P, _P, _P,
                                             _P]),
    "aps_mvdr_attention_weight": (C.c_int, [_P, _P, _P, _I64, _I64, _I64, _I64, _P, _P, _P, _P, _F,
                                            _P, _P, _P, _P, _P]),
    "aps_mvdr_weight": (C.c_int, [_P, _P, _P, _I64, _I64, _I64, _F, _P, _P, _P]),
    "aps_mvdr_beamform": (C.c_int, [_P, _P, _I64, _I64, _I64, _I64, _I64, _I64, _I64, _P, _P]),
    "aps_linear": (C.c_int, [_P, _P, _P, _P, _P, _I64, _I64, _I64, _I64, _I64, _I64, _I32, _F,
                             _P]),
    "aps_linear_layernorm": (C.c_int, [_P, _P, _P, _P, _P, _P, _I64, _I64, _I64, _I64, _I64, _I64,
                                       _I32, _F, _F, _P]),
    "aps_linear_split_size": (C.c_int64, [_I64, _I64]),
    "aps_linear_split_weight": (C.c_int, [_P, _P, _I64, _I64, _I64, _I32, _P]),
    "aps_linear_split": (C.c_int, [_P, _P, _P, _P, _P, _P, _I64, _I64, _I64, _I64, _I64, _I32, _F, _F,
                                   _I32, _P]),
    "aps_linear_fp16x2_size": (C.c_int64, [_I64, _I64]),
    "aps_linear_fp16x2_weight": (C.c_int, [_P, _P, _I64, _I64, _I64, _P]),
    "aps_linear_fp16x2_workspace": (C.c_int64, [_I64, _I64]),
    "aps_linear_fp16x2": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I64, _I64, _I64, _I64, _I64, _I64,
                                    _I32, _F, _F, _P]),
    "aps_linear_panel_rows": (_I32, [_I64, _I64, _I32]),
    "aps_linear_panel_cols": (_I32, [_I64, _I64, _I32]),
    "aps_linear_panel_form": (_I32, [_I64, _I64, _I64, _I32]),
    "aps_linear_panel": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _I64, _I64, _I64, _I64, _I64, _I64,
                                   _I32, _F, _F, _P, _I64, _I32, _P]),
    "aps_layernorm": (C.c_int, [_P, _P, _P, _P, _P, _I64, _I64, _F, _P]),
    "aps_posenc_add": (C.c_int, [_P, _P, _P, _I64, _I64, _I64, _F, _I32, _P]),
    "aps_attention_core": (C.c_int, [_P, _P, _P, _I64, _I64, _I64, _P, _P, _I32, _I32, _I32, _I32, _P,
                                     _P, _I64, _I64, _I64, _I64, _P]),
    "aps_splice": (C.c_int, [_P, _P, _I64, _I64, _I64, _I64, _I64, _I32, _I32, _I32, _P]),
    "aps_delta": (C.c_int, [_P, _P, _P, _I64, _I64, _I64, _I32, _I64, _I64, _I64, _I64, _P]),
    "aps_cmvn_utterance": (C.c_int, [_P, _P, _I64, _I64, _I32, _I32, _F, _P]),
    "aps_cmvn_global": (C.c_int, [_P, _P, _P, _P, _I64, _I64, _I32, _I32, _P]),
    "aps_conv2d_nhwc": (C.c_int, [_P, _P, _P, _P, _P, _P] + [_I64] * 13 + [_I32, _I32, _F, _P]),
    "aps_conv2d_nhwc_split": (C.c_int, [_P, _P, _P, _P, _P, _P] + [_I64] * 13 + [_I32, _I32, _F, _P]),
    "aps_conv2d_nhwc_fp16x2": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P] + [_I64] * 13 + [_I32, _I32, _F, _P]),
    "aps_dccrn_mask": (C.c_int, [_P, _P, _P, _I64, _I64, _I32, _I32, _I32, _F, _P]),
    "aps_store_magnitude": (C.c_int, [_P, _P, _I64, _F, _P]),
    "aps_reim_axis": (C.c_int, [_P, _P, _I64, _I64, _I32, _F, _P]),
    "aps_ipd_from_phase": (C.c_int, [_P, _P, _P, _I64, _I64, _I64, _I64, _I32, _I32, _P, _P]),
    "aps_lstm_workspace": (_I64, [_I64]),
    "aps_lstm_layer": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _I64, _I64, _I64, _I32, _I32, _P,
                                 _P]),
    "aps_lstm_timed_out": (C.c_int, [_P, _P]),
    "aps_lstm_stack": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _I64, _I64, _I64, _I64, _I32, _P, _P]),
    "aps_conformer_stack_scratch": (_I64, [_I64, _I64]),
    "aps_conformer_stack": (C.c_int, [_P, _P, _P, _I32, _P, _I64, _I64, _I64, _I64, _I64, _I64, _I64, _P, _P, _P]),
    "aps_glu_dwconv": (C.c_int, [_P, _P, _P, _P, _P, _P, _I64, _I64, _I64, _I64, _I32, _I32, _P,
                                 _P]),
    "aps_embedding_posenc": (C.c_int, [_P, _P, _P, _P, _I64, _I64, _I64, _I64, _F, _I32, _P, _P]),
    "aps_attention_cross": (C.c_int, [_P, _P, _P, _P, _P, _I64, _I64, _I64, _I64, _I64, _P]),
    "aps_lstm_cell": (C.c_int, [_P, _P, _P, _P, _I64, _I64, _P]),
    "aps_rnn_step": (C.c_int, [_P, _I64, _P, _P, _P, _P, _I64, _P, _P, _P, _I64, _I64, _I64, _I32, _P]),
    "aps_att_step": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I64, _I64, _I64, _I64,
                               _I64, _I64, _I32, _F, _P]),
    "aps_att_step_heads": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I64, _I64, _I64,
                                     _I64, _I64, _I64, _I64, _I32, _F, _P]),
    "aps_length_map": (C.c_int, [_P, _P, _I64, _I64, _I64, _I64, _P]),
    "aps_fixed_beamform": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _I64, _I64, _I64, _I64, _I64, _P]),
    "aps_directional_feature": (C.c_int, [_P, _P, _I64, _P, _P, _P, _I32, _P, _I64, _I64, _I64, _I64,
                                          _I64, _I64, _I64, _F, _F, _P]),
    "aps_speed_perturb": (C.c_int, [_P, _P, _P, _P, _P, _P, _I32, _P, _I64, _I64, _I64, _P]),
    "aps_spec_augment": (C.c_int, [_P, _P, _P, _I64, _I64, _I64, _I64, _I32, _I32, _I32, _P, _P]),
    "aps_mask_nonlinear": (C.c_int, [_P, _P, _I64, _I64, _I32, _F, _F, _F, _P]),
    "aps_mask_nonlinear_backward": (C.c_int, [_P, _P, _P, _I64, _I64, _I32, _F, _F, _F, _P]),
    "aps_tf_mask": (C.c_int, [_P, _I64, _I64, _I64, _I64, _I64, _P, _I64, _I64, _I64, _I32, _P,
                              _P]),
    "aps_tf_mask_backward": (C.c_int, [_P, _I64, _I64, _I64, _I64, _I64, _P, _I64, _I64, _I64, _I32,
                                       _P, _P, _I64, _I64, _I64, _P, _P]),
    # ---- backward (grad.hip / grad_core.h)
    "aps_act_forward": (C.c_int, [_P, _P, _P, _I64, _I32, _F, _P]),
    "aps_act_backward": (C.c_int, [_P, _P, _P, _I64, _I32, _F, _P]),
    "aps_row_bias_add": (C.c_int, [_P, _P, _P, _I64, _I64, _P]),
    "aps_gather_rows_backward": (C.c_int, [_P, _P, _P, _I64, _I64, _I64, _P]),
    "aps_transpose": (C.c_int, [_P, _P, _I64, _I64, _I64, _I64, _P]),
    "aps_fixed_beamform_backward": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I64, _I64, _I64, _I64,
                                              _I64, _P]),
    "aps_rnn_step_backward": (C.c_int, [_P, _I64, _P, _P, _P, _P, _I64, _P, _I64, _P, _P, _P, _P, _I64, _P, _P,
                                        _I64, _I64, _I32, _P]),
    "aps_gemm_tn_workspace": (_I64, [_I64, _I64, _I64]),
    "aps_gemm_tn": (C.c_int, [_P, _P, _P, _P, _P, _I64, _I64, _I64, _I64, _I64, _I64, _P]),
    "aps_colreduce_workspace": (_I64, [_I64, _I64]),
    "aps_colreduce": (C.c_int, [_I32, _P, _P, _P, _P, _I64, _I64, _I64, _I64, _F, _I32, _P, _P, _P]),
    "aps_layernorm_backward": (C.c_int, [_P, _P, _P, _P, _P, _P, _I64, _I64, _F, _P]),
    "aps_batchnorm_workspace": (_I64, [_I64, _I64]),
    "aps_batchnorm_stats": (C.c_int, [_P, _I64, _I64, _F, _F, _P, _P, _P, _P, _P, _P]),
    "aps_batchnorm_apply": (C.c_int, [_P, _P, _P, _P, _P, _P, _I64, _I64, _P]),
    "aps_batchnorm_backward": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _I64, _I64, _P]),
    "aps_softmax_rows": (C.c_int, [_P, _P, _I64, _I64, _P]),
    "aps_softmax_rows_backward": (C.c_int, [_P, _P, _P, _I64, _I64, _P]),
    "aps_magnitude_forward": (C.c_int, [_P, _P, _I64, _F, _P]),
    "aps_magnitude_backward": (C.c_int, [_P, _P, _P, _I64, _F, _P]),
    "aps_log_cmvn_backward": (C.c_int, [_P, _P, _P, _I64, _I64, _I32, _I32, _I32, _F, _F, _F, _P]),
    "aps_glu_dwconv_backward_workspace": (_I64, [_I64, _I64, _I64, _I64]),
    "aps_glu_dwconv_backward": (C.c_int, [_P, _P, _P, _P, _P, _I64, _I64, _I64, _I64, _P, _P]),
    "aps_glu_dwconv_backward_causal": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _I64, _I64, _I64, _I64,
                                                 _P, _P]),
    "aps_im2col_nhwc": (C.c_int, [_P, _P] + [_I64] * 13 + [_P]),
    "aps_attention_backward_workspace": (_I64, [_I64, _I64, _I64]),
    "aps_attention_backward": (C.c_int, [_P, _P, _P, _I64, _I64, _P, _P, _P, _I64, _I64, _I64, _I64,
                                         _F, _I64, _P, _P]),
    "aps_attention_backward_xl": (C.c_int, [_P, _P, _P, _I64, _I64, _I64, _P, _P, _I32, _I32, _I32, _I32, _P, _P,
                                            _P, _P, _P, _I64, _I64, _I64, _I64, _F, _I64, _P, _P, _P]),
    "aps_attention_forward_xl_dropout": (C.c_int, [_P, _P, _P, _I64, _I64, _I64, _P, _P, _I32, _I32, _I32,
                                                   _I32, _P, _I64, _I64, _I64, _I64, _F, _I64, _P, _P]),
    "aps_attention_cross_forward_dropout": (C.c_int, [_P, _P, _P, _P, _I64, _I64, _I64, _I64, _I64, _F,
                                                      _I64, _P, _P]),
    "aps_attention_cross_backward_workspace": (_I64, [_I64, _I64, _I64]),
    "aps_attention_cross_backward": (C.c_int, [_P, _P, _P, _P, _P, _P, _I64, _I64, _I64, _I64, _I64, _F,
                                               _I64, _P, _P, _P]),
    "aps_embedding_backward": (C.c_int, [_P, _P, _P, _P, _I64, _I64, _I64, _F, _P]),
    "aps_dropout": (C.c_int, [_P, _P, _I64, _F, _I64, _P]),
    "aps_attention_forward_dropout": (C.c_int, [_P, _P, _P, _I64, _I64, _P, _I64, _I64, _I64, _I64,
                                                _F, _I64, _P, _P]),
    "aps_time_shift": (C.c_int, [_P, _P, _I64, _I64, _I64, _P]),
    "aps_reverse_time": (C.c_int, [_P, _P, _P, _I64, _I64, _I64, _P]),
    "aps_lstm_gate_scan": (C.c_int, [_P, _P, _P, _P, _P, _P, _I64, _I64, _I64, _P]),
    "aps_lstm_backward_step": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _I64, _I64, _I64, _I64, _P]),
    "aps_lstm_backward_sweep": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _I64, _I64, _I64, _P]),
    "aps_mvdr_offdiag_abs": (C.c_int, [_P, _P, _I64, _I64, _I64, _P]),
    "aps_mvdr_offdiag_abs_backward": (C.c_int, [_P, _P, _P, _I64, _I64, _I64, _P]),
    "aps_mvdr_weight_backward": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _I64, _I64, _I64, _F, _P]),
    "aps_mvdr_beamform_backward": (C.c_int, [_P, _P, _P, _I64, _I64, _I64, _I64, _I64, _I64, _I64,
                                             _P]),
    "aps_cplx_matmul": (C.c_int, [_P, _P, _P, _P, _P, _P, _I64, _I64, _I64, _I64, _I64, _I64, _P]),
    "aps_cplx_inverse": (C.c_int, [_P, _P, _P, _P, _I64, _I64, _P, _P]),
    "aps_dccrn_mask_backward": (C.c_int, [_P, _P, _P, _P, _P, _I64, _I64, _I32, _I32, _I32, _F, _P]),
    "aps_cacgmm_log_pdf": (C.c_int, [_P, _P, _P, _I64, _I64, _I64, _I64, _I64, _I64, _I64, _F, _P]),
    "aps_cacgmm_log_pdf_backward": (C.c_int, [_P, _P, _P, _P, _I64, _I64, _I64, _I64, _I64, _I64,
                                              _I64, _F, _P]),
    "aps_mvdr_covariance_backward": (C.c_int, [_P, _P, _P, _P, _P, _P, _I64, _I64, _I64, _I64, _I64,
                                               _I64, _I64, _I32, _P, _P]),
}

_lib = None


class NativeLibraryError(RuntimeError):
    pass


def library_path() -> str:
    return _LIB_PATH


def load():
    """Open libaps_amd.so and type every entry point.  Raises if the extension is absent."""
    global _lib, _LIB_PATH
    if _lib is not None:
        return _lib
    # A/B measurements: another build of the SAME ABI (e.g. the previous commit's kernels)
    _LIB_PATH = os.environ.get("APS_AMD_LIB", _LIB_PATH)
    if not os.path.exists(_LIB_PATH):
        raise NativeLibraryError(
            f"aps_amd HIP extension not built: {_LIB_PATH} is missing. Run "
            "`python -m aps_amd.build` (or __graft_entry__.build()). There is no CPU fallback.")
    lib = C.CDLL(_LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if a declared symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if lib.aps_abi_version() != ABI_VERSION:
        raise NativeLibraryError(
            f"ABI mismatch: library {lib.aps_abi_version()} vs binding {ABI_VERSION}; rebuild")
    _lib = lib
    return lib


ERR_UNSUPPORTED = -2  # APS_ERR_UNSUPPORTED (include/aps_amd.h)


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = load().aps_status_string(rc).decode()
        raise RuntimeError(f"{what} failed: {msg} (status {rc})")


def ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def stream_of(t: th.Tensor):
    return C.c_void_p(th.cuda.current_stream(t.device).cuda_stream)


def needs_grad(*tensors) -> bool:
    """does autograd have to record this call? (the differentiable ops route through their
    autograd.Function then; inside Function.forward grad mode is off and this is False)"""
    return th.is_grad_enabled() and any(t is not None and t.requires_grad for t in tensors)


def require_device(*tensors) -> th.device:
    """All tensors must be fp32 CUDA(HIP) tensors on one device; no autograd through the kernels
    (ops that have a backward kernel check `needs_grad` first and never get here with grad on)."""
    dev = None
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError(
                "aps_amd kernels run on the GPU only (got a CPU tensor); there is no CPU fallback")
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise RuntimeError(f"tensors on different devices: {dev} vs {t.device}")
        if th.is_grad_enabled() and t.requires_grad:
            raise NotImplementedError(
                "aps_amd: backward through the HIP kernels is not implemented yet "
                "(forward path only; wrap the call in torch.no_grad())")
    return dev


def f32c(t: th.Tensor) -> th.Tensor:
    """fp32 + contiguous (plumbing copy only when the caller hands something else)"""
    if t.dtype != th.float32:
        t = t.float()
    return t if t.is_contiguous() else t.contiguous()
