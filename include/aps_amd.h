/*
 * aps_amd.h  --  C-ABI of the MI355X (gfx950) front-end hot path of funcwj/aps.
 *
 * The reference has NO FFI on this path: its boundary is a Python class surface
 * (aps.transform.AsrTransform / EnhTransform, STFT / iSTFT modules, MvdrBeamformer).  This header
 * is the C-ABI this build adds underneath that surface.  Each entry point names the reference
 * function (file:line under the reference tree) whose arithmetic it replaces.
 *
 * Conventions
 *   - plain device pointers + sizes; the caller owns every buffer (inputs and outputs);
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); all work is enqueued
 *     on it, nothing synchronises;
 *   - return 0 on success, negative aps_status otherwise; aps_status_string() explains;
 *   - all tensors fp32.  Complex data are interleaved (re, im) float pairs.
 *   - SPECTROGRAM STORE layout: [seq, frame, bin, 2] with the bin axis fastest
 *     (seq = n * C + c).  Strides are passed in floats so padded pitches are legal; the
 *     (re, im) pair is always contiguous and the bin stride is always 2 floats.
 *     The reference layout N x C x F x T x 2 is the permuted VIEW of this store.
 */
#ifndef APS_AMD_H_
#define APS_AMD_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
  APS_OK = 0,
  APS_ERR_INVALID = -1,     /* bad pointer / size / stride */
  APS_ERR_UNSUPPORTED = -2, /* legal in the reference, not implemented by this build */
  APS_ERR_LAUNCH = -3       /* hipLaunch / hipGetLastError failure */
} aps_status;

const char* aps_status_string(int status);
/* ABI version, bumped on any signature change */
int aps_abi_version(void);

/* ---------------------------------------------------------------------------------------------
 * Framed STFT.   Replaces _forward_stft (aps/transform/utils.py:227-290) incl. reflect padding,
 * per-frame pre-emphasis, windowing, dense DFT, onesided selection and the polar option; the
 * native counterpart in the reference is StreamingSTFT::Compute (csrc/utils/stft.cc:17-23) over
 * FFTComputer::RealFFT (csrc/utils/fft.cc:59-90).
 * ------------------------------------------------------------------------------------------- */
typedef struct {
  int32_t fft_size;    /* W: DFT size (any >= 2; powers of two take the FFT path)            */
  int32_t frame_len;   /* L: samples taken per frame = kernel width K.shape[2] (L <= W)      */
  int32_t frame_hop;   /* H                                                                   */
  int32_t num_bins;    /* F written per frame: W/2+1 (onesided) or W                          */
  int32_t center;      /* 1: reflect-pad L/2 samples on both sides (utils.py:257-260)         */
  int32_t polar;       /* 1: write (sqrt(re^2+im^2+eps), atan2(im,re)) (utils.py:285-288)     */
  float pre_emphasis;  /* > 0: Kaldi per-frame pre-emphasis (utils.py:263-270)                */
  float eps;           /* magnitude floor for polar                                           */
  float scale;         /* folded DFT-matrix scale: 1, or 1/sqrt(W) for normalized             */
} aps_stft_params;

/* number of frames for `num_samples` (utils.py:653-662); <= 0 when the signal is too short */
int64_t aps_stft_num_frames(int64_t num_samples, const aps_stft_params* p);

/* wav [num_seq, num_samples] contiguous;  window [frame_len];
 * out: store, element (s, t, f) at out + s*stride_seq + t*stride_frame + 2*f */
int aps_stft_forward(const float* wav, int64_t num_seq, int64_t num_samples, const float* window,
                     const aps_stft_params* p, float* out, int64_t stride_seq,
                     int64_t stride_frame, int64_t num_frames, void* stream);

/* The same transform of int16 PCM: a sample is wav[i] / 32768 (exact in fp32), i.e. what the reference's reader
 * hands to its modules (read_audio(..., norm=True): aps/io/audio.py:41-44) formed inside the kernel instead of on
 * the host -- half the bytes over PCIe and half the sample traffic.  wav 4-byte aligned. */
int aps_stft_forward_pcm16(const int16_t* wav, int64_t num_seq, int64_t num_samples, const float* window,
                           const aps_stft_params* p, float* out, int64_t stride_seq,
                           int64_t stride_frame, int64_t num_frames, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Inverse STFT.  Replaces _inverse_stft (aps/transform/utils.py:293-360): Hermitian extension,
 * inverse DFT, synthesis window, overlap-add, window^2 normaliser, centre crop.
 * Native counterpart: StreamingiSTFT (csrc/utils/stft.cc:25-51).
 * spec: store [num_seq, num_frames, F, 2];  wav_out [num_seq, num_samples_out] contiguous with
 * num_samples_out = (T-1)*H + L - (center ? 2*(L/2) : 0).
 * `p->scale` carries 1/W (or 1/sqrt(W) for normalized); p->polar means the input is polar.
 * workspace: caller-owned float [num_seq * num_frames * frame_len] (windowed frames before OLA).
 * ------------------------------------------------------------------------------------------- */
int aps_stft_inverse(const float* spec, int64_t num_seq, int64_t num_frames, int64_t stride_seq,
                     int64_t stride_frame, const float* window, const aps_stft_params* p,
                     float* wav_out, int64_t num_samples_out, float* workspace, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Backward of the two transforms (both are linear maps for polar = 0, so each
 * one's adjoint runs on the other's machinery).  They make STFT / iSTFT differentiable the way the
 * reference's conv1d / conv_transpose1d forms are under autograd (utils.py:227-360), e.g. for the
 * time-domain objectives of aps/task/sse.py that back-propagate through iSTFT.
 *   aps_stft_backward:  grad_store (layout as aps_stft_forward's out) -> grad_wav [num_seq,
 *     num_samples]; pre_emphasis = 0 and polar = 0 only (else APS_ERR_UNSUPPORTED);
 *     workspace float [num_seq * num_frames * frame_len].
 *   aps_stft_inverse_backward:  grad_wav [num_seq, num_samples_out] -> grad_store (layout as
 *     aps_stft_inverse's spec); polar = 0 only; workspace float [num_seq * ((T-1)*H + L)].
 * ------------------------------------------------------------------------------------------- */
int aps_stft_backward(const float* grad_store, int64_t num_seq, int64_t num_frames,
                      int64_t stride_seq, int64_t stride_frame, const float* window,
                      const aps_stft_params* p, float* grad_wav, int64_t num_samples,
                      float* workspace, void* stream);
int aps_stft_inverse_backward(const float* grad_wav, int64_t num_seq, int64_t num_samples_out,
                              const float* window, const aps_stft_params* p, float* grad_store,
                              int64_t stride_seq, int64_t stride_frame, int64_t num_frames,
                              float* workspace, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Spectral + spatial features from a spectrogram store.  Replaces the chain
 *   RefChannelTransform -> MagnitudeTransform -> TFTransposeTransform -> PowerTransform
 *   [-> MelTransform] [-> LogTransform] [-> CmvnTransform(per row)]   (aps/transform/asr.py:280-618,
 *   aps/transform/enh.py:21-49) and PhaseTransform -> IpdTransform (aps/transform/enh.py:52-143),
 * concatenated as EnhTransform.forward does (enh.py:595-613).
 * ------------------------------------------------------------------------------------------- */
typedef struct {
  int32_t num_bins;       /* F                                                                */
  int32_t num_channels;   /* C in the store (1 for single channel input)                      */
  int32_t ref_channel;    /* channel the magnitude branch reads; < 0: no magnitude branch     */
  int32_t power;          /* 1 or 2  (PowerTransform)                                         */
  int32_t num_mels;       /* 0: no mel projection                                             */
  int32_t apply_log;      /* LogTransform present                                             */
  int32_t norm_mean;      /* CmvnTransform flags; both 0 = no cmvn                            */
  int32_t norm_var;
  int32_t num_pairs;      /* IPD channel pairs (0: no spatial branch)                         */
  int32_t ipd_sin;        /* append sin(IPD) after the cos block                              */
  float log_eps;          /* clamp floor of LogTransform                                      */
  float log_lower_bound;  /* > 0: log(lower_bound + x) instead of the clamp                   */
  float cmvn_eps;
} aps_feat_params;

/* store: element (n, c, t, f) at store + n*stride_n + c*stride_c + t*stride_t + 2*f.
 * mel_* describe the mel matrix [num_mels, F] in banded form: row m has nonzeros in
 * [mel_start[m], mel_start[m] + mel_len[m]) whose values sit at mel_w[mel_off[m] ...].
 * pair_l / pair_r: int32 [num_pairs] channel indices.
 * out [N, T, D] contiguous, D = (F or num_mels, if ref_channel >= 0) + num_pairs*(1+ipd_sin)*F
 * nan_count: optional device int32; incremented when a NaN is written (the device side of
 * check_valid, aps/transform/asr.py:33-45, so the host need not re-read the features). */
int aps_enh_features(const float* store, int64_t N, int64_t T, int64_t stride_n, int64_t stride_c,
                     int64_t stride_t, const aps_feat_params* p, const int32_t* mel_start,
                     const int32_t* mel_len, const int32_t* mel_off, const float* mel_w,
                     const int32_t* pair_l, const int32_t* pair_r, float* out, int32_t* nan_count,
                     void* stream);

/* Framed STFT fused with the feature chain (aps_stft_forward + aps_enh_features in one launch;
 * the spectrogram is written but never re-read).  EnhTransform.encode + forward
 * (aps/transform/enh.py:571-613), and with C = 1 / store_out = NULL the waveform rooted
 * AsrTransform chains `spectrogram|fbank [-log] [-cmvn]` (aps/transform/asr.py:902-971) without
 * materialising the spectrogram.  wav [N, C, S]; store_out (optional) element (n*C+c, t, f) at
 * store_out + (n*C+c)*stride_seq + t*stride_frame + 2*f; feats_out [N, T, D] as aps_enh_features.
 * Returns APS_ERR_UNSUPPORTED outside the 512-point fast path (callers then use the two kernels). */
int aps_stft_features(const float* wav, int64_t N, int64_t C, int64_t num_samples,
                      const float* window, const aps_stft_params* p, const aps_feat_params* q,
                      const int32_t* mel_start, const int32_t* mel_len, const int32_t* mel_off,
                      const float* mel_w, const int32_t* pair_l, const int32_t* pair_r,
                      float* store_out, int64_t stride_seq, int64_t stride_frame,
                      int64_t num_frames, float* feats_out, int32_t* nan_count, void* stream);

/* aps_stft_features on int16 PCM (samples wav[i] / 32768, see aps_stft_forward_pcm16; aps/io/audio.py:41-44) */
int aps_stft_features_pcm16(const int16_t* wav, int64_t N, int64_t C, int64_t num_samples,
                            const float* window, const aps_stft_params* p, const aps_feat_params* q,
                            const int32_t* mel_start, const int32_t* mel_len, const int32_t* mel_off,
                            const float* mel_w, const int32_t* pair_l, const int32_t* pair_r,
                            float* store_out, int64_t stride_seq, int64_t stride_frame,
                            int64_t num_frames, float* feats_out, int32_t* nan_count, void* stream);

/* AbsTransform on a complex input followed by [mel] [log] [cmvn]: the "abs-mel-log-cmvn" chain
 * EnhASRBase applies to the beamformer output (aps/transform/asr.py:306-332, enh_att.py:92-93).
 * y: interleaved complex rows [R, F, 2] (row stride in floats); eps is added to the REAL part. */
int aps_abs_features(const float* y, int64_t num_rows, int64_t stride_row, float abs_eps,
                     const aps_feat_params* p, const int32_t* mel_start, const int32_t* mel_len,
                     const int32_t* mel_off, const float* mel_w, float* out, int32_t* nan_count,
                     void* stream);

/* The same tail on REAL rows x [R, F] (row stride in floats): stand-alone PowerTransform /
 * MelTransform / LogTransform / CmvnTransform(per row) layers (aps/transform/asr.py:335-618). */
int aps_row_features(const float* x, int64_t num_rows, int64_t stride_row,
                     const aps_feat_params* p, const int32_t* mel_start, const int32_t* mel_len,
                     const int32_t* mel_off, const float* mel_w, float* out, int32_t* nan_count,
                     void* stream);

/* ---------------------------------------------------------------------------------------------
 * Mask based MVDR  (aps/asr/filter/mvdr.py).
 * ------------------------------------------------------------------------------------------- */
/* MvdrBeamformer._process_mask called on its own (aps/asr/filter/mvdr.py:103-116; aps_mvdr_covariance /
 * aps_mvdr_weights fold it into their pass): mask [N, T, F] -> out [N, F, T]: frames t >= x_len[n] zeroed
 * (x_len NULL: none), divided by max_t |mask| + EPSILON per (n, f) when mask_norm.
 * complement = 1: out [N, T, F] (the mask's own layout) = 1 - the processed mask: the implicit noise mask
 * of MvdrBeamformer.forward (mvdr.py:135), as an operand for aps_mvdr_covariance_backward */
int aps_mvdr_process_mask(const float* mask, const int64_t* x_len, int64_t N, int64_t T, int64_t F,
                          int32_t mask_norm, int32_t complement, float* out, void* stream);

/* _process_mask (mvdr.py:103-116) + estimate_covar (mvdr.py:42-61) for the speech and the noise
 * mask in one pass over the spectrogram.
 *   mask_s, mask_n: [N, T, F] contiguous (as the mask net emits them); mask_n may be NULL, the
 *   noise mask is then 1 - processed speech mask (mvdr.py:136);
 *   x_len: int64 [N] valid frame counts or NULL;  mask_norm: divide by max_t|mask| + EPSILON.
 *   mask_ld (round 5; here and in aps_mvdr_weights): floats between two frames of a mask, utterances T mask_ld
 *   apart; 0 = F (dense [N, T, F]).  2 F reads the halves of the mask net's [N, T, 2 F] output in place
 *   (mvdr.py:132-135 chunks it; the copies into dense halves were two launches and 2 N T F floats per step).
 *   cov_s, cov_n: [N, F, C, C, 2] contiguous.
 *   offdiag: optional [N, C, F] output, |mean_{j != c} Rs[n, f, c, j]| -- the only quantity
 *   ChannelAttention reads from Rs (mvdr.py:165-170); pass it to aps_mvdr_attention_weight.
 *   pmask_s / pmask_n: optional [N, F, T] outputs of the processed masks (NULL to skip).
 *   workspace: caller-owned, aps_mvdr_covariance_workspace(N, C, T, F) bytes (frame-segment
 *   partial sums; the result is deterministic: fixed reduction order, no atomics). */
int64_t aps_mvdr_covariance_workspace(int64_t N, int64_t C, int64_t T, int64_t F);
int aps_mvdr_covariance(const float* store, int64_t N, int64_t C, int64_t T, int64_t F,
                        int64_t stride_n, int64_t stride_c, int64_t stride_t, const float* mask_s,
                        const float* mask_n, int64_t mask_ld, const int64_t* x_len, int32_t mask_norm, float* cov_s,
                        float* cov_n, float* offdiag, float* pmask_s, float* pmask_n,
                        float* workspace, void* stream);

/* Masks -> beamformer weights (mvdr.py:132-140) as one call: covariance partials, segment fold
 * (into coalesced packed triangles + the off-diagonal magnitudes ChannelAttention needs),
 * attention scores, softmax + per-bin solve.  This is what MvdrBeamformer.forward uses; the
 * stage-by-stage entry points expose the same arithmetic piecewise.  cov_s / cov_n (both or
 * neither) are optional full Hermitian outputs.  workspace: aps_mvdr_weights_workspace bytes.
 * singular_count (round 5; here and in the two entry points below; or NULL): device int32, bumped
 * once per (n, f) whose Rn + eps I has a zero or non-finite pivot -- where the reference's Rn.inverse()
 * raises (aps/cplx.py:268-278 -> th.inverse); sticky, read by the caller when it wants (no host sync). */
int64_t aps_mvdr_weights_workspace(int64_t N, int64_t C, int64_t T, int64_t F, int64_t A);
int aps_mvdr_weights(const float* store, int64_t N, int64_t C, int64_t T, int64_t F,
                     int64_t stride_n, int64_t stride_c, int64_t stride_t, const float* mask_s,
                     const float* mask_n, int64_t mask_ld, const int64_t* x_len, int32_t mask_norm, int64_t A,
                     const float* proj_w, const float* proj_b, const float* gvec_w,
                     const float* gvec_b, float eps, float* workspace, float* cov_s, float* cov_n,
                     float* u_out, float* weight_out, int32_t* singular_count, void* stream);

/* ChannelAttention (mvdr.py:148-174): u = softmax_c(gvec . tanh(proj |offdiag-mean Rs| + b)).
 * proj_w [A, F], proj_b [A], gvec_w [A], gvec_b [1];
 * scratch: caller-owned, aps_mvdr_attention_scratch(N, C, A) bytes;  u_out [N, C]. */
int64_t aps_mvdr_attention_scratch(int64_t N, int64_t C, int64_t A);
int aps_mvdr_channel_attention(const float* cov_s, int64_t N, int64_t C, int64_t F, int64_t A,
                               const float* proj_w, const float* proj_b, const float* gvec_w,
                               const float* gvec_b, float* scratch, float* u_out, void* stream);

/* _derive_weight (mvdr.py:75-101): w = (Rn+eps I)^-1 Rs u / (tr((Rn+eps I)^-1 Rs) + eps).
 * weight_out [N, F, C, 2].  2 <= C <= 8. */
int aps_mvdr_weight(const float* cov_s, const float* cov_n, const float* u, int64_t N, int64_t C,
                    int64_t F, float eps, float* weight_out, int32_t* singular_count, void* stream);

/* ChannelAttention + _derive_weight in two launches (the softmax over channels is folded into the
 * weight kernel): what MvdrBeamformer.forward needs between covariance and beamform.
 * offdiag: the [N, C, F] by-product of aps_mvdr_covariance, or NULL (derived from cov_s).
 * scratch as for aps_mvdr_channel_attention; u_out [N, C]; weight_out [N, F, C, 2]. */
int aps_mvdr_attention_weight(const float* cov_s, const float* cov_n, const float* offdiag,
                              int64_t N, int64_t C, int64_t F, int64_t A, const float* proj_w, const float* proj_b,
                              const float* gvec_w, const float* gvec_b, float eps, float* scratch,
                              float* u_out, float* weight_out, int32_t* singular_count, void* stream);

/* beamform (mvdr.py:29-39, 142-145): y[n,t,f] = sum_c conj(w[n,f,c]) x[n,c,t,f].
 * y_out [N, T, F, 2] contiguous. */
int aps_mvdr_beamform(const float* store, const float* weight, int64_t N, int64_t C, int64_t T,
                      int64_t F, int64_t stride_n, int64_t stride_c, int64_t stride_t, float* y_out,
                      void* stream);
/* beamform + AbsTransform + [mel] [log] [row CMVN] in one pass -- SURVEY 8(d) P3: what EnhASRBase does with
 * the beamformer's output (aps/asr/enh_att.py:86-93; aps/asr/filter/mvdr.py:29-39 followed by
 * aps/transform/asr.py:306-332 and its tail), whose complex beam output is never returned.  p / mel_* as in
 * aps_abs_features; y_out [N, T, F, 2] or NULL (NULL: the beam output is not written at all); feats_out
 * [N, T, D], D = num_mels or F.  APS_ERR_UNSUPPORTED: the caller runs aps_mvdr_beamform + aps_abs_features. */
int aps_mvdr_beamform_features(const float* store, const float* weight, int64_t N, int64_t C, int64_t T,
                               int64_t F, int64_t stride_n, int64_t stride_c, int64_t stride_t, float abs_eps,
                               const aps_feat_params* p, const int32_t* mel_start, const int32_t* mel_len,
                               const int32_t* mel_off, const float* mel_w, float* y_out, float* feats_out,
                               int32_t* nan_count, void* stream);

/* length arithmetic on device-resident int64 lengths, one launch: out[i] = trunc((in[i] + add) /
 * div) + post -- the frame-count / output-length formulas of aps/transform/utils.py:653-662,
 * aps/asr/base/component.py:187-190, 290-297 and aps/transform/asr.py:1017-1019 */
int aps_length_map(const int64_t* in, int64_t* out, int64_t n, int64_t add, int64_t div,
                   int64_t post, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Context / utterance-level feature layers of AsrTransform.  Feature matrices are [U, T, F] with
 * U = utterances x channels; *_utt / *_row are the pitches (in floats) between utterances / frames.
 * ------------------------------------------------------------------------------------------- */
/* SpliceTransform (aps/transform/asr.py:687-728; splice_feature, aps/transform/utils.py:193-224):
 * out[u, to, c*F + f] = in[u, clamp(to*subsampling + c - lctx, 0, T-1), f], c = 0..lctx+rctx,
 * to = 0..T/subsampling - 1; out contiguous [U, T/subsampling, (lctx+rctx+1) F] */
int aps_splice(const float* in, float* out, int64_t U, int64_t T, int64_t F, int64_t in_utt,
               int64_t in_row, int32_t lctx, int32_t rctx, int32_t subsampling, void* stream);
/* one order of DeltaTransform (asr.py:731-782):
 * out[u,t,f] = sum_c scale[c] * in[u, clamp(t + c - ctx, 0, T-1), f], c = 0..2ctx (summed left to
 * right like the reference); scale [2ctx+1] = the module's frozen `scale` parameter */
int aps_delta(const float* in, float* out, const float* scale, int64_t U, int64_t T, int64_t F,
              int32_t ctx, int64_t in_utt, int64_t in_row, int64_t out_utt, int64_t out_row,
              void* stream);
/* CmvnTransform with utterance ("all band", per_band=False) statistics over the `count` = T*F
 * values of each of the U contiguous utterance-channels (asr.py:587-596) */
int aps_cmvn_utterance(const float* x, float* out, int64_t U, int64_t count, int32_t norm_mean,
                       int32_t norm_var, float eps, void* stream);
/* CmvnTransform with global statistics: (x - gmean[f]) / gstd[f] (asr.py:605-609) */
int aps_cmvn_global(const float* x, const float* gmean, const float* gstd, float* out, int64_t rows,
                    int64_t F, int32_t norm_mean, int32_t norm_var, void* stream);

/* DCCRN complex ratio masks (aps/sse/bss/dccrn.py:217-242): dec [rows, 2S] = decoder output, channels
 * s / S + s the real / imaginary mask of speaker s; m' = nl(|m|) m / |m| with |m| = sqrt(mr^2 +
 * mi^2 + eps); out [S, rows, 2] = m' (apply = 0) or m' * X with X = store [rows, 2] (apply = 1).
 * non_linear: 0 none, 1 relu, 2 tanh, 3 softplus, 4 sigmoid (MaskNonLinear, sse/base.py:112-156)
 * cplx = 0 is the real-valued network (dccrn.py:234-241): dec [rows, S], m = nl(dec), out [S, rows]
 * = m (apply = 0) or [S, rows, 2] = (re X, im X) m (apply = 1). */
int aps_dccrn_mask(const float* dec, const float* store, float* out, int64_t rows, int64_t S,
                   int32_t non_linear, int32_t apply, int32_t cplx, float eps, void* stream);
/* PhaseTransform / MagnitudeTransform / IpdTransform called as modules with any `dim` / `eps`
 * (aps/transform/enh.py:52-143, aps/transform/asr.py:280-303; inside EnhTransform.forward the phase is
 * never formed and the magnitude is part of the feature launch): aps_reim_axis: x [outer, 2, inner] with
 * the (re, im) axis in the middle -> [outer, inner], op 0 atan2(im, re), op 1 sqrt(re^2 + im^2 + eps);
 * aps_ipd_from_phase: phase [N, C, T, F] -> [N, T, M, F], cos(p_l - p_r) of the num_pairs pairs,
 * followed by their sines when with_sin (M = 2 num_pairs) */
int aps_reim_axis(const float* x, float* out, int64_t outer, int64_t inner, int32_t op, float eps,
                  void* stream);
int aps_ipd_from_phase(const float* phase, const int32_t* pair_l, const int32_t* pair_r, int64_t N,
                       int64_t C, int64_t T, int64_t F, int32_t num_pairs, int32_t with_sin, float* out,
                       void* stream);
/* magnitude input of the real-valued DCCRN (dccrn.py:259): out[r] = sqrt(re^2 + im^2 + eps) of the
 * interleaved rows store [rows, 2] */
int aps_store_magnitude(const float* store, float* out, int64_t rows, float eps, void* stream);

/* ---------------------------------------------------------------------------------------------
 * FixedBeamformer.forward (aps/transform/enh.py:303-384): real / imag [N,C,F,T] (two tensors, as the
 * reference passes them), weights w_real / w_imag [B,C,F] (the reference's [B,C,F,1] parameters),
 *   out[n,b,f,t] = sum_c conj(w[b,c,f]) x[n,c,f,t]
 * beam == NULL: all beams, out_* [N,B,F,T]; beam [N] (int64, 0 <= beam[n] < B): the selected beam
 * per utterance, out_* [N,F,T].  squeeze / trans of the reference are views on the host side.
 * ------------------------------------------------------------------------------------------- */
int aps_fixed_beamform(const float* real, const float* imag, const float* w_real,
                       const float* w_imag, const int64_t* beam, float* out_real, float* out_imag,
                       int64_t N, int64_t C, int64_t F, int64_t T, int64_t B, void* stream);

/* ---------------------------------------------------------------------------------------------
 * DfTransform (aps/transform/enh.py:146-300), geometry "7@" (the only one the reference has,
 * :211-229): phase [N,C,T,F]; doa[n * doa_stride + d] radians (doa_stride 0: the same D sampled
 * directions for every utterance, :204-209); neg_omega[f] = -pi sr f / (F - 1) (:190-193);
 * index_l / index_r: num_pairs <= 16 microphone pairs;
 *   out[((n D + d) T + t) ld_out + out_offset + f] = mean_p cos(phase[l_p] - phase[r_p] - dif_p)
 * (ld_out / out_offset place several speakers' features side by side, the cat of :293-296).
 * D <= 64, else APS_ERR_UNSUPPORTED.
 * ------------------------------------------------------------------------------------------- */
int aps_directional_feature(const float* phase, const float* doa, int64_t doa_stride,
                            const float* neg_omega, const int32_t* index_l, const int32_t* index_r,
                            int32_t num_pairs, float* out, int64_t ld_out, int64_t out_offset,
                            int64_t N, int64_t C, int64_t T, int64_t F, int64_t D, float radius,
                            float velocity, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Training-time augmentation of the ASR feature transform.  The random draws are made on the host
 * in the reference's order; these entry points apply them.
 *   aps_speed_perturb: SpeedPerturbTransform.forward in training mode (aps/transform/asr.py:166-195,
 *     perturb_speed aps/transform/augment.py:86-109).  wav [N,S]; choice[n] in [0, num_filters]
 *     picks filters[choice] = a polyphase bank [dst, src, taps] (taps odd) that maps every block of
 *     src samples to dst samples; num_filters = keep the signal.  out [N,S_out], S_out = the
 *     longest new length of the batch, zero past each utterance's own.
 *   aps_spec_augment: SpecAugTransform.forward once the coin flip said "augment" (asr.py:660-684,
 *     tf_mask / random_mask augment.py:13-83).  x [N,C,T,F]; bands int32 [N, num_freq + num_time, 2]
 *     = (begin, length) of utterance n's frequency bands then time bands (length 0: none); values
 *     inside a band become x * 0 (mask_zero) or the mean of the whole input (workspace: 8 bytes).
 * ------------------------------------------------------------------------------------------- */
int aps_speed_perturb(const float* wav, const int64_t* choice, const float* const* filters,
                      const int32_t* src, const int32_t* dst, const int32_t* taps,
                      int32_t num_filters, float* out, int64_t N, int64_t S, int64_t S_out,
                      void* stream);
int aps_spec_augment(const float* x, const int32_t* bands, float* out, int64_t N, int64_t C,
                     int64_t T, int64_t F, int32_t num_freq, int32_t num_time, int32_t mask_zero,
                     void* workspace, void* stream);

/* ---------------------------------------------------------------------------------------------
 * TF masking (aps/sse/base.py:23-47): out[n,t,f] = x[n,ch,t,f] * mask[n,t,f]
 * mask: real [N,T,F] (mask_complex = 0) or complex [N,T,F,2]; mask strides in floats.
 * ------------------------------------------------------------------------------------------- */
int aps_tf_mask(const float* store, int64_t N, int64_t T, int64_t F, int64_t stride_n,
                int64_t stride_t, const float* mask, int64_t mask_stride_n, int64_t mask_stride_t,
                int64_t mask_stride_f, int32_t mask_complex, float* out, void* stream);
/* MaskNonLinear.forward (aps/sse/base.py:112-156): out = clamp(f(x) * scale, vmin, vmax); code 0
 * identity, 1 relu, 2 tanh, 3 softplus, 4 sigmoid (elementwise over sources * inner values), 5
 * softmax over the leading axis of x [sources, inner].  No clamp: vmin = -inf / vmax = +inf. */
int aps_mask_nonlinear(const float* x, float* out, int64_t sources, int64_t inner, int32_t code,
                       float scale, float vmin, float vmax, void* stream);
/* its adjoint (training an SSE network through its mask activation, aps/sse/base.py:141-156): g_x =
 * scale f'(x) g_out where clamp(f(x) scale, vmin, vmax) let the value through (bounds included, as
 * torch's clamp_min / clamp_max do); code 5: the softmax's Jacobian over the sources */
int aps_mask_nonlinear_backward(const float* x, const float* g_out, float* g_x, int64_t sources,
                                int64_t inner, int32_t code, float scale, float vmin, float vmax,
                                void* stream);
/* aps_tf_mask's backward: grad_out [N,T,F,2] -> grad_mask (real: g.re x.re + g.im x.im; complex: conj(x) g;
 * strides in floats like the mask's; may be NULL) and grad_store [N,T,F,2] contiguous (g m or
 * g conj(M); may be NULL) */
int aps_tf_mask_backward(const float* store, int64_t N, int64_t T, int64_t F, int64_t stride_n,
                         int64_t stride_t, const float* mask, int64_t mask_stride_n,
                         int64_t mask_stride_t, int64_t mask_stride_f, int32_t mask_complex,
                         const float* grad_out, float* grad_mask, int64_t gm_stride_n,
                         int64_t gm_stride_t, int64_t gm_stride_f, float* grad_store, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Transformer encoder pieces (aps/asr/transformer/impl.py, pose.py; aps/asr/base/encoder.py).
 * ------------------------------------------------------------------------------------------- */
/* nn.Linear with fused epilogue on fp32 MFMA:
 *   C[M,N] = act(A[M,K] W[N,K]^T + bias) * alpha + residual.
 * bias [N] / residual [M,N] (leading dim ldc) may be NULL; act: 0 none, 1 relu, 2 swish, 3 sigmoid, 4 tanh,
 * 5 gelu (erf form, nn.GELU; transformer/utils.py:113-123).  lda,
 * ldw multiples of 4, A and W 16-byte aligned.  (tf.linear + activation + residual add,
 * impl.py:147-185, 389-429; alpha = 0.5 is the conformer's macaron half step, impl.py:519-540;
 * the 1 x 1 Conv1d layers of the conformer convolution, impl.py:478-489, are the same GEMM) */
int aps_linear(const float* A, const float* W, const float* bias, const float* residual, float* C,
               int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldw, int64_t ldc,
               int32_t act, float alpha, void* stream);

/* LayerNorm over the K axis of A folded into the projection that consumes it (the pre-norm
 * sub-layers norm -> Linear of impl.py:404-429, 519-540):
 *   C = act(LN(A) W^T + b) * alpha + residual,  LN(a) = (a - mean) / sqrt(var + eps) * gamma + beta
 * evaluated as rstd_r (A W'^T - mean_r colsum) + b' on the RAW rows with the host-prepared
 *   W_gamma = W diag(gamma) [N,K],  colsum[n] = sum_k W_gamma[n,k],  bias_beta = b + W beta [N];
 * mean_r / rstd_r are accumulated from the A tiles as they pass through the staging registers. */
int aps_linear_layernorm(const float* A, const float* W_gamma, const float* bias_beta,
                         const float* colsum, const float* residual, float* C, int64_t M, int64_t N,
                         int64_t K, int64_t lda, int64_t ldw, int64_t ldc, int32_t act, float alpha,
                         float eps, void* stream);

/* The same two projections on the bf16 matrix pipe, fp32 in / fp32 out and as accurate as the fp32
 * MFMA (every operand is split exactly into three bf16 planes, six products per term; error
 * against float64 measured equal to aps_linear's, csrc/gemm_split.hip).  The weight is split once:
 *   aps_linear_split_size(N, K)   bytes of the planes image of a weight [N, K]
 *   aps_linear_split_weight       W [N, K] (row pitch ldw) -> planes (16-byte aligned); layout 0 =
 *                                 row image (planes staged through LDS), 1 = fragment image (the
 *                                 64 x 128 kernel whose waves fetch their weight operands straight
 *                                 into registers); aps_linear_split takes the same `layout`
 *   aps_linear_split              C = act(A W^T + bias) * alpha + residual on the planes of W; with
 *                                 colsum != NULL the LayerNorm fold of aps_linear_layernorm (planes
 *                                 of W_gamma, bias = bias_beta, eps)
 * (tf.linear / nn.Linear of the encoder layers, impl.py:147-185, 389-429, 478-540, at the batch
 * sizes where the fp32 matrix rate is the bound) */
int64_t aps_linear_split_size(int64_t N, int64_t K);
int aps_linear_split_weight(const float* W, void* planes, int64_t N, int64_t K, int64_t ldw,
                            int32_t layout, void* stream);
int aps_linear_split(const float* A, const void* planes, const float* bias, const float* colsum,
                     const float* residual, float* C, int64_t M, int64_t N, int64_t K, int64_t lda,
                     int64_t ldc, int32_t act, float alpha, float eps, int32_t layout, void* stream);

/* The same GEMM with HALF the matrix work: two fp16 planes and three products per term
 * (csrc/gemm_fp16x2.hip).  Both operands are scaled per row by a power of two that brings the row
 * maximum into [2^14, 2^15); the planes are h = rn_f16(x') and l = rn_f16((x' - h) 2^11), the cross
 * terms h l + l h accumulate apart from h h and are folded in with 2^-11 in the epilogue.  An element
 * keeps 22 bits while it lies within 2^-28 of its row maximum; a tile that meets a non-zero element
 * more than 2^30 below its row maximum is detected while its planes are formed and recomputed on the fp32 MFMA from the fp32 operands inside the same
 * launch.  For every finite input the representation loses at most 2^-19 sum_k |a_k| |w_k| (2^-20.5
 * measured on operands without such elements), next to the rounding of the fp32 accumulation every
 * fp32 evaluation carries (K / 16 sequential additions here, K / 2 in aps_linear and on the fp32 path;
 * scripts/split_fp16_emulation.py, tests/test_fp16x2_arithmetic.py,
 * tests/test_gpu_encoder.py::test_fp16x2_wide_range_*).
 *   aps_linear_fp16x2_size(N, K)  bytes of the image of a weight [N, K]
 *   aps_linear_fp16x2_weight      W [N, K] (row pitch ldw, 16-byte aligned rows) -> image: the
 *                                 fragment-ordered planes, the int32 row exponents, the int32 "wide"
 *                                 flags of the rows
 *   aps_linear_fp16x2_workspace(M, K)  bytes of device workspace one call needs (16-byte aligned)
 *   aps_linear_fp16x2             as aps_linear_split; W32 (row pitch ldw) is the fp32 weight the
 *                                 image was made from (read only by tiles on the fp32 path).  The
 *                                 call first forms the planes of A in `workspace` -- one pass over A
 *                                 that also yields every row's exponent, its "wide" flag and the row
 *                                 statistics of the LayerNorm fold: [K step][plane][row, padded to
 *                                 64][32 k] f16, so that the GEMM's staging lanes move whole 16-byte
 *                                 runs into LDS with no arithmetic in between -- then runs the GEMM.
 *                                 wide_count (or NULL): device int32 that counts the tiles the call
 *                                 recomputed in fp32 (sticky, for diagnostics)
 * (same reference call sites as aps_linear_split) */
int64_t aps_linear_fp16x2_size(int64_t N, int64_t K);
int aps_linear_fp16x2_weight(const float* W, void* image, int64_t N, int64_t K, int64_t ldw,
                             void* stream);
int64_t aps_linear_fp16x2_workspace(int64_t M, int64_t K);
int aps_linear_fp16x2(const float* A, const void* image, const float* W32, const float* bias,
                      const float* colsum, const float* residual, float* C, void* workspace,
                      int32_t* wide_count, int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldw,
                      int64_t ldc, int32_t act, float alpha, float eps, void* stream);

/* aps_linear_fp16x2's arithmetic and weight image in the PANEL form (csrc/gemm_panel.hip, round 4): a
 * workgroup owns 32 or 64 rows x 128 columns and walks K in chunks of 256 / 128 -- the chunk's fp32
 * rows go global -> registers -> (row maxima, scale, split) -> one static LDS image of both planes,
 * the chunk's K steps then run without a barrier or any A traffic, and its accumulators are folded
 * into a running fp32 sum with ONE power of two per (row, chunk) (exact).  No planes pass over A, no
 * workspace: the LayerNorm fold's row statistics are gathered by the staging lanes.  An element now
 * only has to lie within 2^-30 of the largest magnitude of its own chunk of its row (a finer granule
 * than aps_linear_fp16x2's whole row; same detection, same in-launch fp32 recomputation of the tiles
 * that do not fit, same bound).  K must be a multiple of 4.
 *   form                          0 = the default; 1 | 2 | 3 = the caller's choice: 32 rows x 128 columns at
 *                                 two workgroups per CU (chunks of 256), 64 x 128 (chunks of 128), 32 x 128 at
 *                                 four workgroups per CU (chunks of 128, 128 VGPRs: the default); 4 | 5 = the
 *                                 K-GROUP forms (round 5, K <= 1024): the same 32 x 128 tile owned by 16 (8)
 *                                 waves, K cut into 4 (2) groups that stage, split and multiply their own
 *                                 columns side by side, the partial tiles summed in LDS in group order
 *                                 (bit-reproducible), the epilogue spread over every lane -- faster for a
 *                                 launch of one tile per CU alone on the chip (N = 512 at M = 2016: 9.7
 *                                 against 12.0 us), slower for larger launches and beside another stream's:
 *                                 the Python host asks for form 4 only for such launches of a single stream
 *   aps_linear_panel_rows(M, N, form)   32 | 64: the panel height the call will use
 *   aps_linear_panel_cols(M, N, form)   128: its column-tile width (the "tile" of wide_count is rows x cols)
 *   aps_linear_panel_form(M, N, K, form)  the form the call will run: 1 ... 5 as above (tests / bench labels)
 *   next_image / next_bytes       a HINT (or NULL / 0): device memory the next launch of the stream
 *                                 will read first -- normally the weight image of the next projection.
 *                                 Every workgroup requests its share of it on its way out, so the
 *                                 next launch finds it in L2 instead of HBM (a step's weight images
 *                                 do not survive in the caches from one step to the next); it must stay
 *                                 allocated while the launch runs, is never written and never changes
 *                                 a result
 * (same reference call sites as aps_linear_split: every nn.Linear / 1 x 1 convolution of the encoder
 * path, aps/asr/transformer/impl.py:377-541, aps/asr/base/encoder.py:87-184) */
int32_t aps_linear_panel_rows(int64_t M, int64_t N, int32_t form);
int32_t aps_linear_panel_cols(int64_t M, int64_t N, int32_t form);
int32_t aps_linear_panel_form(int64_t M, int64_t N, int64_t K, int32_t form);
int aps_linear_panel(const float* A, const void* image, const float* W32, const float* bias,
                     const float* colsum, const float* residual, float* C, int32_t* wide_count,
                     int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldw, int64_t ldc,
                     int32_t act, float alpha, float eps, const void* next_image, int64_t next_bytes,
                     int32_t form, void* stream);

/* out = LayerNorm(x (+ residual)) * gamma + beta over rows of D  (nn.LayerNorm, impl.py:396-428) */
int aps_layernorm(const float* x, const float* residual, const float* gamma, const float* beta,
                  float* out, int64_t rows, int64_t D, float eps, void* stream);

/* out[n,t,:] = x[n,t,:] * factor + sinusoid(t0 + t)  (InputSinPosEncoding, pose.py:93-118);
 * div_term [D/2] as the reference's frozen parameter */
int aps_posenc_add(const float* x, const float* div_term, float* out, int64_t N, int64_t T,
                   int64_t D, float factor, int32_t t0, void* stream);

/* token embedding + position encoding of the transformer decoder (aps/asr/transformer/decoder.py:
 * 150-153): out[n,t,:] = table[ids[n,t],:] * factor + sinusoid(t0 + t); table [V, D], ids int64
 * [N, T]; bad_count (optional device int32) counts ids outside [0, V) (they embed as zeros) */
int aps_embedding_posenc(const float* table, const int64_t* ids, const float* div_term, float* out,
                         int64_t N, int64_t T, int64_t D, int64_t V, float factor, int32_t t0,
                         int32_t* bad_count, void* stream);

/* softmax((q k^T [+ rel term]) / sqrt(dh) + masks) v for every (utterance, head)
 * (impl.py:90-114).  qkv [N, T, 3, H, dh] = the in-projection output; lens int64 [N] valid key
 * counts or NULL; ctx [N, T, H, dh].  head_dim in {32, 64, 128}.
 * rel (or NULL): relative position table [rel_len, dh] shared by the heads (rel_head_stride = 0) or
 * per head [H, rel_len, dh] (rel_head_stride = rel_len * dh); the score of (query i, key j) gains
 * q_i . rel[j - i + rel_zero] (rows outside the table count as zero) -- RelMultiheadAttention's
 * digit_shift(q E^T) term, impl.py:258-292 + utils.py:14-39, with E = RelPosEncoding(arange(-T+1,
 * T)) (pose.py:65-88, encoder.py:91-95): rel_len = 2T - 1, rel_zero = T - 1.
 * rel_u / rel_v [H, dh] (or NULL): Transformer-XL biases, score = (q + u_h) k_j + (q + v_h)
 * rel_h[j - i + rel_zero] (XlMultiheadAttention.dot_att, impl.py:322-343, rel_h = rel_proj(sinusoid)
 * per head); query_slot selects which projection plays "q": 0 = query, 2 = value (the reference's
 * XL forward passes `value`, impl.py:366 -- reproduced).
 * chunk / lctx / rctx: context window of prep_context_mask (transformer/utils.py:60-98): key j is
 * visible to query i iff max((i/chunk - lctx) chunk, 0) <= j < (i/chunk + rctx + 1) chunk; a
 * negative lctx / rctx leaves that side open (chunk = 1, lctx = rctx = -1: no window).
 * add_mask (or NULL): any additive [T, T] mask (0 / -inf or a bias), the `src_mask` argument of the
 * encoder layers (impl.py:404, 507) when it is not a context window. */
int aps_attention_core(const float* qkv, const int64_t* lens, const float* rel, int64_t rel_zero,
                       int64_t rel_len, int64_t rel_head_stride, const float* rel_u,
                       const float* rel_v, int32_t query_slot, int32_t chunk, int32_t lctx,
                       int32_t rctx, const float* add_mask, float* ctx, int64_t N, int64_t T,
                       int64_t H, int64_t head_dim, void* stream);
/* cross attention of the transformer decoder (nn.MultiheadAttention(tgt, memory, memory),
 * aps/asr/transformer/decoder.py:78-86): q [N, Tq, H, dh] (the query projection of the target),
 * kv [N, Tk, 2, H, dh] (key | value projections of the memory), key_lens int64 [N] valid memory
 * frames or NULL (memory_key_padding_mask), add_mask [Tq, Tk] additive (0 / -inf or a bias: the
 * layer's memory_mask, decoder.py:51, 85) or NULL, ctx [N, Tq, H, dh]; head_dim in {32, 64, 128} */
int aps_attention_cross(const float* q, const float* kv, const int64_t* key_lens,
                        const float* add_mask, float* ctx, int64_t N, int64_t Tq, int64_t Tk,
                        int64_t H, int64_t head_dim, void* stream);

/* Conformer convolution module between its two pointwise layers (impl.py:478-489):
 *   out[n,t,d] = act(scale[d] * (sum_k weight[d,k] * glu(x)[n, t + k - (K-1)/2, d] + bias[d])
 *                    + shift[d]),   glu(x)[n,t,d] = x[n,t,d] * sigmoid(x[n,t,D+d]), 0 outside [0,T)
 * = GLU(dim=channels) -> depthwise Conv1d(K, padding (K-1)/2, groups D) -> BatchNorm1d (eval
 * affine folded into scale/shift, NULL = identity) -> activation (`swish`: 0 none, 1 Swish, 2 ReLU,
 * 3 GELU erf form; transformer/utils.py:113-123).
 * x [N, T, 2D], weight [D, K] (Conv1d weight [D,1,K]), bias/scale/shift [D], out [N, T, D]; K odd,
 * K <= 63.  causal = 1 (casual_conv1d, impl.py:468-505): taps t - (K-1) .. t, and since the
 * reference pads the module INPUT with K-1 zero frames, a frame left of 0 carries
 * glu(pad_bias) = pad_bias[d] * sigmoid(pad_bias[D+d]) with pad_bias [2D] the bias of the
 * preceding pointwise layer (NULL: zeros). */
int aps_glu_dwconv(const float* x, const float* weight, const float* bias, const float* scale,
                   const float* shift, float* out, int64_t N, int64_t T, int64_t D, int64_t K,
                   int32_t swish, int32_t causal, const float* pad_bias, void* stream);

/* ---------------------------------------------------------------------------------------------
 * 2-D convolution / transposed convolution, channels-last, implicit GEMM on fp32 MFMA with the
 * per-channel affine (bias, folded eval-mode BatchNorm2d), activation and residual fused:
 *   y[n,ho,wo,co] = act(scale[co] * sum_{kh,kw,ci} x[n,hi,wi,ci] w[co,kh,kw,ci] + shift[co])
 *                   (+ residual[n,ho,wo,co])
 *   forward:    hi = ho*sh + kh - ph,        wi = wo*sw + kw - pw          (zero padding)
 *   transposed: hi = (ho + ph - kh) / sh,    wi = (wo + pw - kw) / sw      (where divisible)
 * x [N,H,W,Ci], w [Co,KH,KW,Ci] (nn.Conv2d weight permuted (0,2,3,1); nn.ConvTranspose2d weight
 * permuted (1,2,3,0): the gather form needs no kernel flip), y [N,Ho,Wo,Co]; scale / shift /
 * residual may be NULL; act: 0 none, 1 relu, 5 leaky relu with `slope`.
 * Replaces Conv2d + BatchNorm2d + ReLU of the encoder's conv2d subsampling
 * (aps/asr/base/component.py:251-307) and the Conv2d / ConvTranspose2d + BatchNorm2d + LeakyReLU
 * blocks of DCCRN (aps/sse/enh/dcunet.py:24-170; a complex layer = one real layer on real|imag
 * stacked channels).  Ci % 32 == 0 runs on MFMA, any other Ci (first layers) on a direct kernel.
 * ------------------------------------------------------------------------------------------- */
int aps_conv2d_nhwc(const float* x, const float* w, const float* scale, const float* shift,
                    const float* residual, float* y, int64_t N, int64_t H, int64_t W, int64_t Ci,
                    int64_t Co, int64_t KH, int64_t KW, int64_t sh, int64_t sw, int64_t ph,
                    int64_t pw, int64_t Ho, int64_t Wo, int32_t transposed, int32_t act, float slope,
                    void* stream);

/* aps_conv2d_nhwc on the bf16 matrix pipe (the arithmetic of aps_linear_split: exact three-way
 * splits, six products, fp32-accurate): `planes` = aps_linear_split_weight(w viewed as
 * [Co, KH KW Ci], layout 1).  Ci must be a multiple of 32 (a K step = 32 channels of one tap).
 * (the complex convolution blocks of DCCRN, dcunet.py:24-274; Conv2d subsampling, component.py:251-307) */
int aps_conv2d_nhwc_split(const float* x, const void* planes, const float* scale, const float* shift,
                          const float* residual, float* y, int64_t N, int64_t H, int64_t W,
                          int64_t Ci, int64_t Co, int64_t KH, int64_t KW, int64_t sh, int64_t sw,
                          int64_t ph, int64_t pw, int64_t Ho, int64_t Wo, int32_t transposed,
                          int32_t act, float slope, void* stream);

/* aps_conv2d_nhwc with the arithmetic of aps_linear_fp16x2 (two fp16 planes, three products, rows
 * and weight rows scaled by powers of two, tiles with operands outside the planes' range recomputed
 * in fp32): `image` = aps_linear_fp16x2_weight(w viewed as [Co, KH KW Ci]), w32 = that fp32 weight;
 * pixexp = int32 [N H W] device workspace the call fills with the exponent of every input pixel (a
 * row of the implicit GEMM takes the smallest exponent among the pixels its taps read); wide_count
 * as in aps_linear_fp16x2.  Ci must be a multiple of 32.  Same call sites as aps_conv2d_nhwc_split */
int aps_conv2d_nhwc_fp16x2(const float* x, const void* image, const float* w32, const float* scale,
                           const float* shift, const float* residual, float* y, int32_t* pixexp,
                           int32_t* wide_count, int64_t N, int64_t H, int64_t W, int64_t Ci,
                           int64_t Co, int64_t KH, int64_t KW, int64_t sh, int64_t sw, int64_t ph,
                           int64_t pw, int64_t Ho, int64_t Wo, int32_t transposed, int32_t act,
                           float slope, void* stream);

/* ---------------------------------------------------------------------------------------------
 * LSTM recurrence of the RNN mask estimator (PyTorchRNNEncoder -> nn.LSTM batch_first,
 * aps/asr/base/encoder.py:87-184, aps/asr/base/component.py:26-55, 145-190): one persistent launch
 * per layer, both directions of a bidirectional layer inside it.
 *   pre_*  [N, T, 4H] = x W_ih^T + b_ih (aps_linear), torch gate order i | f | g | o;
 *          pre_bwd / w_hh_bwd / b_hh_bwd NULL for a unidirectional layer
 *   w_hh_* [4H, H], b_hh_* [4H] or NULL, zero initial state
 *   lens   int64 [N] valid frames or NULL; packed-sequence semantics: y[n, t >= len] = 0 and the
 *          backward direction runs each utterance from its own last frame
 *   y      [N, T, dirs * H] (forward | backward columns), 16-byte aligned, fully overwritten
 *   second_reverse: 1 = the second parameter set is the backward direction (nn.LSTM
 *          bidirectional); 0 = it is an independent second forward LSTM over the same time axis
 *          (DCCRN's real / imaginary LSTM pair, aps/sse/bss/dccrn.py:54-94) run in the same launch
 *   share:  >= 1, how many launches of this kind may be in flight on the device at once (graph
 *          replicas on separate streams): the grid is sized for 1 / share of the resident slots,
 *          because every workgroup of every such launch has to be resident together (an explicit
 *          argument: no process-global state steers the launch geometry)
 *   workspace: aps_lstm_workspace(H) bytes of device memory the CALLER zeroes once: word 0 counts
 *          expired hand-off waits and is STICKY (never reset by a launch), so one persistent word
 *          per device serves eager launches and captured graphs alike and can be read at any
 *          later point (aps_lstm_timed_out).  After an expired wait the layer output holds NaNs
 *          (the un-arrived operand words are the NaN-patterned sentinel), so downstream NaN
 *          guards fire as well.  Behind the 16 status bytes the library keeps the placement tables
 *          of the team form (below), one per launch, cycled.
 * Arithmetic: the recurrent product h W_hh^T runs on the f16 matrix pipe as three products of
 * two-plane operands (h = h_hi + h_lo exactly representable parts, |h| < 1; W_hh rows scaled by a
 * power of two), fp32 accumulation: the gate pre-activations are within 2^-20 sum_k |w_rk| of the
 * fp32 product (hidden sizes that are not a multiple of 128 keep the exact-fp32 MFMA form); the
 * tests hold the layer output to 1e-5 of torch's float64 LSTM after 249 steps either way.
 * Team form (H = 128 / 256 / 512 and at most 8 (direction, 16- or 32-utterance block) pairs): the 32
 * workgroups that exchange h are the blocks with equal id % 8, which the dispatcher is observed
 * to place on one XCD; every workgroup publishes the XCC id it runs on, and only a team whose 32
 * ids agree hands h over through that XCD's L2 (plain stores + L1-bypassing loads); any other
 * placement runs the placement-independent protocol (write-through stores).
 * H in {64, 128, 256, 320, 384, 512, 640, 768, 1024}, N <= 128, N*T*dirs*H*4 < 2^31; otherwise
 * APS_ERR_UNSUPPORTED (callers keep the MIOpen path for those).  The launch is decomposed into
 * (unit block, utterance block) workgroups that must all be resident; when no decomposition of
 * this (H, N, dirs) fits the device the call returns APS_ERR_UNSUPPORTED before touching y
 * beyond the sentinel fill (callers retry with fewer utterances: N <= 16 always fits).
 * aps_lstm_timed_out: 1 if the workspace counter is non-zero (blocking read on `stream`).
 * ------------------------------------------------------------------------------------------- */
int64_t aps_lstm_workspace(int64_t H);
int aps_lstm_layer(const float* pre_fwd, const float* pre_bwd, const float* w_hh_fwd,
                   const float* w_hh_bwd, const float* b_hh_fwd, const float* b_hh_bwd,
                   const int64_t* lens, float* y, int64_t N, int64_t T, int64_t H,
                   int32_t second_reverse, int32_t share, void* workspace, void* stream);
int aps_lstm_timed_out(const void* workspace, void* stream);
/* Unidirectional nn.LSTM stack (2 <= L <= 4 layers) in ONE launch, layers pipelined: layer l >= 1
 * consumes y[l-1] live (x_t gathered with the same write-once sentinel protocol as h_{t-1}), so it
 * trails the layer below by about one step and needs no input GEMM.  pre0 [N,T,4H] = layer 0's
 * x W_ih^T + b_ih; w_ih / w_hh / b_ih / b_hh / y: arrays of L device pointers ([4H,H], [4H] or
 * NULL, y[l] [N,T,H] fully overwritten; w_ih[0] / b_ih[0] unused).  H in {64,128,256,512}, N <= 64
 * and a resident decomposition of L layers; otherwise APS_ERR_UNSUPPORTED (run the layers with
 * aps_lstm_layer).  share / workspace as above. */
int aps_lstm_stack(const float* pre0, const float* const* w_ih, const float* const* w_hh,
                   const float* const* b_ih, const float* const* b_hh, const int64_t* lens,
                   float* const* y, int64_t N, int64_t T, int64_t H, int64_t L, int32_t share,
                   void* workspace, void* stream);

/* ---------------------------------------------------------------------------------------------
 * RNN attention decoder, one target position at a time (aps/asr/base/decoder.py:69-218).  The
 * projections of a step run on aps_linear; these two entry points do the rest.
 *   aps_lstm_cell: one nn.LSTM time step with carried state (decoder.py:128-135): pre [N, 4H] =
 *     x W_ih^T + b_ih + h W_hh^T + b_hh (gate order i | f | g | o), c_prev [N, H] or NULL (zeros)
 *     -> h_out, c_out [N, H].
 *   aps_att_step: attention over the encoder frames for every utterance (attention.py:76-259):
 *     enc_part [N, T, A] = enc_proj(enc_pad), enc_pad [N, T, D], dec_part [N, A] = dec_proj(dec),
 *     mode 0 "ctx": score = w . tanh(enc_part + dec_part); 1 "dot": score = scale enc_part .
 *     dec_part; 2 "loc": score = w . tanh(enc_part + dec_part + att(F(ali_prev))) with the location
 *     filter F [C, 2L+1] (+ bias [C]) and the 1 x 1 conv att [A, C]; ali_prev [N, T] or NULL = the
 *     uniform initial alignment.  ali [N, T] = softmax over the frames t < enc_len[n] (enc_len
 *     NULL: all), ctx [N, D] = sum_t ali[t] enc_pad[t].
 * ------------------------------------------------------------------------------------------- */
int aps_lstm_cell(const float* pre, const float* c_prev, float* h_out, float* c_out, int64_t N,
                  int64_t H, void* stream);

/* One time step of an nn.GRU / nn.RNN (tanh, relu) / nn.LSTM cell on the gate pre-activations: the
 * step-by-step form of the recurrences without a persistent kernel (PyTorchRNNEncoder with
 * rnn = "gru" / "rnn_tanh" / "rnn_relu", LSTMs of other hidden sizes or with proj_size;
 * aps/asr/base/encoder.py:87-184, var_len_rnn_forward).  gx [N, G H] (row pitch ldx) = x_t W_ih^T +
 * b_ih from the whole-sequence GEMM, gh [N, G H] = h_{t-1} W_hh^T + b_hh; gate orders as torch (GRU
 * r | z | n, LSTM i | f | g | o); mode 0 GRU, 1 tanh, 2 relu, 3 LSTM (c_prev / c_out).  Rows with
 * t >= lens[n] keep their state and write zeros to y (packed-sequence semantics); y (row pitch ldy)
 * may be NULL. */
int aps_rnn_step(const float* gx, int64_t ldx, const float* gh, const float* h_prev,
                 const float* c_prev, const int64_t* lens, int64_t t, float* h_out, float* c_out,
                 float* y, int64_t ldy, int64_t N, int64_t H, int32_t mode, void* stream);
int aps_att_step(const float* enc_part, const float* enc_pad, const float* dec_part, const float* w,
                 const int64_t* enc_len, const float* ali_prev, const float* loc_filter,
                 const float* loc_filter_bias, const float* loc_att, float* ali, float* ctx,
                 int64_t N, int64_t T, int64_t A, int64_t D, int64_t C, int64_t L, int32_t mode,
                 float scale, void* stream);
/* the multi-head forms MHCtx / MHDot / MHLocAttention (attention.py:266-531): H independent heads
 * in one launch.  key [N, T, H A] = key_proj(enc_pad), value [N, T, H Dv] = enc_proj(enc_pad) (the
 * reference uses Dv = A), dec_part [N, H A], w [H A] (the grouped 1 x 1 conv `w`), ali_prev
 * [N, H, T] or NULL, loc_filter [H C, 2L+1] (+ bias [H C]) and loc_att [H A, C] (the grouped convs
 * `F` and `att`); ali [N, H, T], ctx [N, H Dv] (the input of ctx_proj).  Modes as above. */
int aps_att_step_heads(const float* key, const float* value, const float* dec_part, const float* w,
                       const int64_t* enc_len, const float* ali_prev, const float* loc_filter,
                       const float* loc_filter_bias, const float* loc_att, float* ali, float* ctx,
                       int64_t N, int64_t T, int64_t H, int64_t A, int64_t Dv, int64_t C, int64_t L,
                       int32_t mode, float scale, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Backward (section 8(f) row 1): what `loss.backward()` of the reference's trainer
 * (aps/trainer/ddp.py:124-200) needs for the MVDR front end (aps/asr/filter/mvdr.py:19-174), its RNN
 * mask estimator (aps/asr/base/encoder.py:87-184, component.py:26-55) and the conformer encoder
 * (aps/asr/transformer/impl.py:225-541, base/component.py:251-307).  The Python layer wraps these in
 * torch.autograd.Function objects (aps_amd/grad_ops.py); gradients equal torch autograd through the
 * reference's own CPU arithmetic (tests/test_grad_host.py on the host build of the same functors,
 * tests/test_gpu_backward.py on the GPU).
 * Gradient of a real loss w.r.t. a complex value: (dL/d re, dL/d im) in the value's own layout.
 * The contractions of the backward pass: g_x = g_pre W = aps_linear(g_pre, W^T) (aps_transpose makes
 * W^T), and g_W = g_pre^T x together with g_b = column sums of g_pre in one call of aps_gemm_tn.
 * ------------------------------------------------------------------------------------------- */
/* out = act(pre) * alpha (+ residual) / g_pre = g_out * alpha * act'(pre): the epilogue of
 * aps_linear as its own pass (training keeps `pre`); act codes of aps_linear, plus 6 = nn.LeakyReLU()
 * (slope 0.01: the activation of the DCCRN blocks, aps/sse/enh/dcunet.py:131, 180) and 7 = x^2
 * (PowerTransform(2) inside a differentiable feature chain, aps/transform/asr.py:302-327) */
int aps_act_forward(const float* pre, const float* residual, float* out, int64_t n, int32_t act,
                    float alpha, void* stream);
int aps_act_backward(const float* g_out, const float* pre, float* g_pre, int64_t n, int32_t act,
                     float alpha, void* stream);
/* Adjoint of aps_fixed_beamform (FixedBeamformer with requires_grad, aps/transform/enh.py:303-384):
 * g_b real / imag [N, (B), F, T] -> the input's gradient g_x real / imag [N, C, F, T] (or NULL pair) and the
 * coefficients' g_w real / imag [B, C, F] (or NULL pair; needs x).  beam [N] or NULL as in the forward. */
int aps_fixed_beamform_backward(const float* g_br, const float* g_bi, const float* xr, const float* xi,
                                const float* wr, const float* wi, const int64_t* beam, float* g_xr,
                                float* g_xi, float* g_wr, float* g_wi, int64_t N, int64_t C, int64_t F,
                                int64_t T, int64_t B, void* stream);
/* One time step of nn.GRU / nn.RNN / nn.LSTM backwards (mode and gate order of aps_rnn_step): the
 * recurrences of var_len_rnn_forward (aps/asr/base/component.py:26-55) that run step by step.  gx: this
 * step's rows of x W_ih^T + b_ih (pitch ldx); gh = h_prev W_hh^T + b_hh [N, G H] (recomputed by the caller);
 * g_y: this step's rows of the output gradient (pitch ldgy) or NULL; g_h / g_c [N, H]: the state gradients
 * carried back from step t + 1 or NULL.  Writes this step's rows (pitch ldg) of g_gx and g_gh -- operands of
 * the batched weight-gradient products -- and g_hp [N, H], the previous state's gradient WITHOUT the
 * recurrent term g_gh W_hh (the caller's GEMM adds it), g_cp (LSTM).  Rows with t >= lens[n] pass their
 * state gradient through (packed-sequence semantics). */
int aps_rnn_step_backward(const float* gx, int64_t ldx, const float* gh, const float* h_prev,
                          const float* c_prev, const int64_t* lens, int64_t t, const float* g_y,
                          int64_t ldgy, const float* g_h, const float* g_c, float* g_gx, float* g_gh,
                          int64_t ldg, float* g_hp, float* g_cp, int64_t N, int64_t H, int32_t mode,
                          void* stream);
/* out[r, :] = x[r, :] + b (the conv bias in front of a training-mode BatchNorm; its gradient is a
 * column reduction) */
int aps_row_bias_add(const float* x, const float* b, float* out, int64_t rows, int64_t D,
                     void* stream);
/* adjoint of gathering R rows of an embedding table [V, D] by int64 index (RelPosEncoding,
 * pose.py:65-88: table[r] = embed[clamp(offset_r)]): g_weight[v] = sum_{index[r] = v} g_table[r] */
int aps_gather_rows_backward(const int64_t* index, const float* g_table, float* g_weight, int64_t R,
                             int64_t V, int64_t D, void* stream);
/* out[c, r] = in[r, c] for a [rows, cols] matrix with row pitch ld_in (ld_out >= rows) */
int aps_transpose(const float* in, float* out, int64_t rows, int64_t cols, int64_t ld_in,
                  int64_t ld_out, void* stream);
/* C [I, J] = A^T B for row-major A [M, I] (pitch lda), B [M, J] (pitch ldb), C pitch ldc; colsum [I]
 * (or NULL) = the column sums of A.  What loss.backward() (aps/trainer/ddp.py:161-165) derives for the
 * weight and bias of every nn.Linear (g_W = g_pre^T x, g_b = sum_rows g_pre; aps/asr/transformer/impl.py:
 * 147-185, 389-429), for nn.LSTM's w_ih / w_hh (aps/asr/base/component.py:26-55) and for the im2col form
 * of the convolutions.  fp32 MFMA straight from the row-major operands (no transposed copies), M cut
 * into slabs whose partial products are summed in slab order (deterministic).
 * workspace: aps_gemm_tn_workspace(M, I, J) bytes (may be 0; then NULL is accepted). */
int64_t aps_gemm_tn_workspace(int64_t M, int64_t I, int64_t J);
int aps_gemm_tn(const float* A, const float* B, float* C, float* colsum, void* workspace, int64_t M,
                int64_t I, int64_t J, int64_t lda, int64_t ldb, int64_t ldc, void* stream);
/* out[c] (+)= scale * sum_r f(r, c) over the rows of [rows, cols] matrices (pitches lda / ldb),
 * deterministic two-stage reduction.  mode 0: A; 1: A * B; 2: (A - v1[c])^2;
 * 3: A * (B - v1[c]) * v2[c];  4: two plain sums in one call -- `cols` = 2 D, out[0 .. D) = column sums of
 * A [rows, D], out[D .. 2 D) = those of B [rows, D] (the LayerNorm's g_gamma | g_beta).
 * workspace: aps_colreduce_workspace(rows, cols) bytes. */
int64_t aps_colreduce_workspace(int64_t rows, int64_t cols);
int aps_colreduce(int32_t mode, const float* A, const float* B, const float* v1, const float* v2,
                  int64_t rows, int64_t cols, int64_t lda, int64_t ldb, float scale,
                  int32_t accumulate, float* out, float* workspace, void* stream);
/* nn.LayerNorm backward on rows of D: y = LN(x (+ residual)) * gamma + beta (aps_layernorm);
 * g_x = gradient of x and of the residual; t (or NULL) [rows, D] = g_y * xhat, whose column sums
 * are g_gamma (g_beta = column sums of g_y) */
int aps_layernorm_backward(const float* x, const float* residual, const float* gamma,
                           const float* g_y, float* g_x, float* t, int64_t rows, int64_t D, float eps,
                           void* stream);
/* Training-mode BatchNorm over the rows of [rows, D] (BatchNorm1d of the conformer convolution,
 * impl.py:478-489; BatchNorm2d of the conv2d subsampling on channels-last pixels,
 * component.py:251-307): batch statistics (biased variance for the normalisation, torch's
 * momentum update of running_mean / running_var with the unbiased variance; NULL = not tracked),
 * y = (x - mean) rstd gamma + beta, and the backward g_x = gamma rstd (g_y - sum_gy / rows -
 * xhat sum_gy_xhat / rows) with the two column sums from aps_colreduce (mode 0 and mode 3; NULL
 * sums = statistics were constants, i.e. eval mode).  workspace: aps_batchnorm_workspace bytes. */
int64_t aps_batchnorm_workspace(int64_t rows, int64_t D);
int aps_batchnorm_stats(const float* x, int64_t rows, int64_t D, float eps, float momentum,
                        float* mean, float* rstd, float* running_mean, float* running_var,
                        float* workspace, void* stream);
int aps_batchnorm_apply(const float* x, const float* mean, const float* rstd, const float* gamma,
                        const float* beta, float* y, int64_t rows, int64_t D, void* stream);
int aps_batchnorm_backward(const float* x, const float* mean, const float* rstd, const float* gamma,
                           const float* g_y, const float* sum_gy, const float* sum_gy_xhat,
                           float* g_x, int64_t rows, int64_t D, void* stream);
/* softmax over short rows and its backward (the channel softmax of ChannelAttention, mvdr.py:174) */
int aps_softmax_rows(const float* x, float* y, int64_t rows, int64_t D, void* stream);
int aps_softmax_rows_backward(const float* y, const float* g_y, float* g_x, int64_t rows, int64_t D,
                              void* stream);
/* |z + eps| on n interleaved complex values (AbsTransform on a ComplexTensor, asr.py:306-332: eps
 * joins the REAL part) and its adjoint */
int aps_magnitude_forward(const float* z, float* mag, int64_t n, float eps, void* stream);
int aps_magnitude_backward(const float* z, const float* g_mag, float* g_z, int64_t n, float eps,
                           void* stream);
/* adjoint of [log] -> per-row CMVN (aps_row_features on rows of D; asr.py:431-464, 576-618):
 * m = the input of the log, g_z = gradient of the normalised rows */
int aps_log_cmvn_backward(const float* m, const float* g_z, float* g_m, int64_t rows, int64_t D,
                          int32_t apply_log, int32_t norm_mean, int32_t norm_var, float log_eps,
                          float lower_bound, float cmvn_eps, void* stream);
/* adjoint of GLU -> depthwise Conv1d (aps_glu_dwconv with scale = shift = NULL, swish = 0, causal
 * = 0): g_c [N, T, D] -> g_x [N, T, 2D], g_w [D, K] (or NULL; g_bias = column sums of g_c).
 * workspace: aps_glu_dwconv_backward_workspace bytes. */
int64_t aps_glu_dwconv_backward_workspace(int64_t N, int64_t T, int64_t D, int64_t K);
int aps_glu_dwconv_backward(const float* x, const float* w, const float* g_c, float* g_x, float* g_w,
                            int64_t N, int64_t T, int64_t D, int64_t K, float* workspace,
                            void* stream);
/* the same for causal = 1 (`casual_conv1d`, asr/transformer/impl.py:446, 491-505): K - 1 frames of
 * left context that carry glu(pad_bias) (zeros when pad_bias is NULL); g_pad [2D] (or NULL) = the
 * gradient that reaches pad_bias through those frames (the caller adds it to the projection bias's) */
int aps_glu_dwconv_backward_causal(const float* x, const float* w, const float* g_c,
                                   const float* pad_bias, float* g_x, float* g_w, float* g_pad,
                                   int64_t N, int64_t T, int64_t D, int64_t K, float* workspace,
                                   void* stream);
/* patches[(n, ho, wo), (kh, kw, ci)] of a channels-last image (row pitch ld >= KH KW Ci, zero
 * filled): g_W of aps_conv2d_nhwc = g_y^T patches (one aps_linear); g_x is aps_conv2d_nhwc's
 * transposed form on g_y */
int aps_im2col_nhwc(const float* x, float* out, int64_t N, int64_t H, int64_t W, int64_t Ci,
                    int64_t KH, int64_t KW, int64_t sh, int64_t sw, int64_t ph, int64_t pw,
                    int64_t Ho, int64_t Wo, int64_t ld, void* stream);
/* adjoint of aps_attention_core for absolute / learnt relative positions with length masks (no
 * XL biases, no context window): g_ctx [N, T, H, dh] -> g_qkv [N, T, 3, H, dh] and, with rel,
 * g_rel_partial [N H, rel_len, dh] (its column sums over N H are g_rel).  P is recomputed from q, k
 * in three passes (rows / columns / table), nothing T x T is stored.
 * workspace: aps_attention_backward_workspace bytes. */
int64_t aps_attention_backward_workspace(int64_t N, int64_t T, int64_t H);
int aps_attention_backward(const float* qkv, const int64_t* lens, const float* rel, int64_t rel_zero,
                           int64_t rel_len, const float* g_ctx, float* g_qkv, float* g_rel_partial,
                           int64_t N, int64_t T, int64_t H, int64_t head_dim, float drop_p,
                           int64_t drop_seed, float* workspace, void* stream);

/* the general form of aps_attention_backward (generic kernels: any head size):
 * context windows (chunk / lctx / rctx as in aps_attention_core), per-head relative tables
 * (rel_head_stride = rel_len * dh), the Transformer-XL biases rel_u / rel_v [H, dh] and the query read
 * from the value projection (query_slot 2: XlMultiheadAttention, aps/asr/transformer/impl.py:322-374).
 * g_qkv's q slot receives the gradient of the scores' query row whatever slot it was read from;
 * g_row_k / g_row_e [N, T, H, dh] (or NULL) = sum_j dS k_j / sum_j dS E_ij per row: their column sums
 * over (n, t) are the gradients of rel_u / rel_v; g_rel_partial [N H, rel_len, dh] (summed over n, and
 * over h for a shared table, by the caller).  drop_p / drop_seed: the weight dropout of the forward
 * aps_attention_forward_xl_dropout (0: none).  add_mask (round 5; or NULL): any additive [T, T] mask on
 * the scaled logits (0 / -inf or a bias: `src_mask` / `tgt_mask`, impl.py:104-114, decoder.py:150-186) --
 * data: no gradient flows into it, a -inf pair gets weight and gradient 0 -- in this call, in the forward
 * below and in the two cross-attention calls (memory_mask [Tq, Tk]).
 * workspace: aps_attention_backward_workspace bytes. */
int aps_attention_backward_xl(const float* qkv, const int64_t* lens, const float* rel, int64_t rel_zero,
                              int64_t rel_len, int64_t rel_head_stride, const float* rel_u,
                              const float* rel_v, int32_t query_slot, int32_t chunk, int32_t lctx,
                              int32_t rctx, const float* g_ctx, float* g_qkv, float* g_rel_partial,
                              float* g_row_k, float* g_row_e, int64_t N, int64_t T, int64_t H,
                              int64_t head_dim, float drop_p, int64_t drop_seed, const float* add_mask,
                              float* workspace, void* stream);
/* training forward of the same general form with dropout on the attention weights (impl.py:104 inside
 * XlMultiheadAttention / windowed encoders): ctx [N, T, H dh], rows without a visible key are 0 */
int aps_attention_forward_xl_dropout(const float* qkv, const int64_t* lens, const float* rel,
                                     int64_t rel_zero, int64_t rel_len, int64_t rel_head_stride,
                                     const float* rel_u, const float* rel_v, int32_t query_slot,
                                     int32_t chunk, int32_t lctx, int32_t rctx, float* ctx, int64_t N,
                                     int64_t T, int64_t H, int64_t head_dim, float drop_p,
                                     int64_t drop_seed, const float* add_mask, void* stream);
/* cross attention of the transformer decoder under autograd (aps_attention_cross;
 * aps/asr/transformer/decoder.py:78-86): the train()-mode forward with dropout on the attention weights
 * (nn.MultiheadAttention's `dropout`) and the backward, which recomputes the mask from (drop_p,
 * drop_seed) -- g_q [N, Tq, H, dh], g_kv [N, Tk, 2, H, dh]; generic kernels, any head size.
 * workspace: aps_attention_cross_backward_workspace bytes. */
int aps_attention_cross_forward_dropout(const float* q, const float* kv, const int64_t* key_lens,
                                        float* ctx, int64_t N, int64_t Tq, int64_t Tk, int64_t H,
                                        int64_t head_dim, float drop_p, int64_t drop_seed,
                                        const float* add_mask, void* stream);
int64_t aps_attention_cross_backward_workspace(int64_t N, int64_t Tq, int64_t H);
int aps_attention_cross_backward(const float* q, const float* kv, const int64_t* key_lens,
                                 const float* g_ctx, float* g_q, float* g_kv, int64_t N, int64_t Tq,
                                 int64_t Tk, int64_t H, int64_t head_dim, float drop_p,
                                 int64_t drop_seed, const float* add_mask, float* workspace, void* stream);
/* adjoint of aps_embedding_posenc w.r.t. the table (the decoder's token embedding, decoder.py:150):
 * the R lookups sorted by token id -- sorted_ids [R], order [R] (the row of g behind each sorted
 * position) -- g [R, D]; g_weight [V, D] zero-filled by the caller; g_weight[v] = scale * sum of the
 * rows that looked v up, summed in `order` (deterministic) */
int aps_embedding_backward(const int64_t* sorted_ids, const int64_t* order, const float* g,
                           float* g_weight, int64_t R, int64_t D, int64_t V, float scale, void* stream);
/* nn.Dropout in train() mode, counter based: out[i] = x[i] * keep(seed, i) with keep = 0 or
 * 1 / (1 - p) a hash of (seed, i) -- the backward is the same call on the gradient (no stored mask).
 * aps_attention_forward_dropout: the training forward of aps_attention_core (absolute / learnt
 * relative positions, length masks) with dropout p on the attention WEIGHTS (impl.py:104,
 * `self.dropout(th.softmax(logit))`); aps_attention_backward takes the same (drop_p, drop_seed) and
 * recomputes the mask (0 = no dropout).  workspace: aps_attention_backward_workspace bytes;
 * head_dim 32 / 64. */
int aps_dropout(const float* x, float* out, int64_t n, float p, int64_t seed, void* stream);
int aps_attention_forward_dropout(const float* qkv, const int64_t* lens, const float* rel,
                                  int64_t rel_zero, int64_t rel_len, float* ctx, int64_t N, int64_t T,
                                  int64_t H, int64_t head_dim, float drop_p, int64_t drop_seed,
                                  float* workspace, void* stream);
/* nn.LSTM backward through time for one unidirectional layer (component.py:26-55), given the layer
 * output y of the forward (aps_lstm_layer / aps_lstm_stack):
 *   aps_time_shift:      hprev[n, t] = y[n, t - 1] (0 at t = 0)
 *   (hh = aps_linear(hprev, W_hh): h_{t-1} W_hh^T of every step in one GEMM)
 *   aps_lstm_gate_scan:  gates [N, T, 4H] (activated i | f | g | o) and cells c [N, T, H] from
 *                        pre = x W_ih^T + b_ih, hh and b_hh; nothing happens at t >= lens[n]
 *   aps_lstm_backward_sweep: g_pre [N, T, 4H] for t = T-1 .. 0, each step = aps_linear(g_pre[t+1],
 *                        W_hh^T as w_hh_t [H, 4H]) + aps_lstm_backward_step; g_h_rec, g_c [N, H]
 *                        scratch
 *   (then g_x = aps_linear(g_pre, W_ih^T), g_W_ih = g_pre^T x, g_W_hh = g_pre^T hprev,
 *    g_b = column sums of g_pre) */
int aps_time_shift(const float* y, float* out, int64_t N, int64_t T, int64_t H, void* stream);
/* out[n, t] = x[n, lens[n] - 1 - t] for t < lens[n], zeros past it (lens NULL: T): the backward
 * direction of a bidirectional nn.LSTM is the forward recurrence on time-reversed utterances, so its
 * BPTT runs the same sweep between two of these (the map is its own inverse) */
int aps_reverse_time(const float* x, const int64_t* lens, float* out, int64_t N, int64_t T, int64_t D,
                     void* stream);
int aps_lstm_gate_scan(const float* pre, const float* hh, const float* b_hh, const int64_t* lens,
                       float* gates, float* c, int64_t N, int64_t T, int64_t H, void* stream);
int aps_lstm_backward_step(const float* gates, const float* c, const float* g_y,
                           const float* g_h_rec, const int64_t* lens, float* g_c, float* g_pre,
                           int64_t N, int64_t T, int64_t H, int64_t t, void* stream);
int aps_lstm_backward_sweep(const float* gates, const float* c, const float* g_y,
                            const float* w_hh_t, const int64_t* lens, float* g_pre, float* g_h_rec,
                            float* g_c, int64_t N, int64_t T, int64_t H, void* stream);
/* mask-based MVDR adjoints (mvdr.py:29-174); the spectrogram store is data (no gradient):
 *   aps_mvdr_offdiag_abs(+_backward): v[n,c,f] = |mean_{j != c} Rs[n,f,c,j]| (mvdr.py:165-170);
 *       the backward ACCUMULATES into g_cov
 *   aps_mvdr_weight_backward: adjoint of aps_mvdr_weight: g_w [N,F,C,2] -> g_cov_s, g_cov_n
 *       [N,F,C,C,2] and g_u_part [N,F,C] (its sums over F are g_u)
 *   aps_mvdr_beamform_backward: g_y [N,T,F,2] -> g_w [N,F,C,2]
 *   aps_mvdr_covariance_backward: adjoint of aps_mvdr_covariance for ONE mask (raw mask [N,T,F],
 *       lens = valid frames or NULL, mask_norm as in the forward): g_cov [N,F,C,C,2] -> g_mask.
 *       g_sub [N,T,F] or NULL is subtracted from the gradient of the PROCESSED mask before
 *       _process_mask's own adjoint: the implicit noise mask Rn = estimate_covar(1 - m', X)
 *       (aps/asr/filter/mvdr.py:135) -- its gradient w.r.t. the complement (a call of this function on
 *       aps_mvdr_process_mask(..., complement = 1), lens NULL, mask_norm 0) enters the speech mask here */
int aps_mvdr_offdiag_abs(const float* cov, float* v, int64_t N, int64_t C, int64_t F, void* stream);
int aps_mvdr_offdiag_abs_backward(const float* cov, const float* g_v, float* g_cov, int64_t N,
                                  int64_t C, int64_t F, void* stream);
int aps_mvdr_weight_backward(const float* cov_s, const float* cov_n, const float* u,
                             const float* g_w, float* g_cov_s, float* g_cov_n, float* g_u_part,
                             int64_t N, int64_t C, int64_t F, float eps, void* stream);
int aps_mvdr_beamform_backward(const float* store, const float* g_y, float* g_w, int64_t N,
                               int64_t C, int64_t T, int64_t F, int64_t stride_n, int64_t stride_c,
                               int64_t stride_t, void* stream);
int aps_mvdr_covariance_backward(const float* store, const float* mask, const int64_t* lens,
                                 const float* cov, const float* g_cov, float* g_mask, int64_t N,
                                 int64_t C, int64_t T, int64_t F, int64_t stride_n, int64_t stride_c,
                                 int64_t stride_t, int32_t mask_norm, const float* g_sub, void* stream);

/* MlEnhTask's objective (aps/task/ml.py:38-101), the second covariance consumer of section 8(f)
 * row 4: with R = aps_mvdr_covariance(store, mask, mask_norm = 0) the log-pdf of the complex angular
 * central Gaussian, log_pdf[n,t,f] = -C log max(Re x^H B^-1 x, eps) - log max(det B, eps),
 * B = C R + eps I, and its adjoint g_log_pdf -> g_cov (whose adjoint to the mask is
 * aps_mvdr_covariance_backward).  store [N,C,T,F,2] with element strides, cov [N,F,C,C,2],
 * log_pdf [N,T,F] (the reference's N x F x T is its transposed view). */
int aps_cacgmm_log_pdf(const float* store, const float* cov, float* log_pdf, int64_t N, int64_t C,
                       int64_t T, int64_t F, int64_t stride_n, int64_t stride_c, int64_t stride_t,
                       float eps, void* stream);
int aps_cacgmm_log_pdf_backward(const float* store, const float* cov, const float* g_log_pdf,
                                float* g_cov, int64_t N, int64_t C, int64_t T, int64_t F,
                                int64_t stride_n, int64_t stride_c, int64_t stride_t, float eps,
                                void* stream);

/* ComplexTensor.__matmul__ / .inverse() of the reference (aps/cplx.py:242-278) on small matrices:
 *   aps_cplx_matmul   C[b] = A[b] B[b]: A [B, M, K], B [B, K, N] (batch stride b_batch = 0 broadcasts one
 *                     matrix), halves given apart (a_im / b_im NULL = a real operand), one output
 *                     element per thread (the operands are covariance sized: K <= 64)
 *   aps_cplx_inverse  B matrices [C, C], C = 1 .. 8, complex Gauss-Jordan with partial pivoting in
 *                     registers (the reference inverts the real 2C x 2C embedding: the same matrix);
 *                     singular_count (or NULL): device int32 bumped once per matrix with a zero or
 *                     non-finite pivot -- the matrices th.inverse raises on; the Python wrapper reads it
 *                     after the call and raises torch.linalg.LinAlgError like the reference */
int aps_cplx_matmul(const float* a_re, const float* a_im, const float* b_re, const float* b_im,
                    float* c_re, float* c_im, int64_t B, int64_t M, int64_t K, int64_t N,
                    int64_t a_batch, int64_t b_batch, void* stream);
int aps_cplx_inverse(const float* a_re, const float* a_im, float* o_re, float* o_im, int64_t B,
                     int64_t C, int32_t* singular_count, void* stream);

/* backward of aps_dccrn_mask (aps/sse/bss/dccrn.py:217-242 under autograd): g_out like that call's
 * `out`, g_dec like `dec`; g_store [rows, 2] (the gradient of the masked spectrogram w.r.t. the
 * mixture's STFT, summed over speakers) or NULL (must be NULL when apply = 0) */
int aps_dccrn_mask_backward(const float* dec, const float* store, const float* g_out, float* g_dec,
                            float* g_store, int64_t rows, int64_t S, int32_t non_linear,
                            int32_t apply, int32_t cplx, float eps, void* stream);

/* ---------------------------------------------------------------------------------------------
 * The conformer encoder STACK in one launch per batch (csrc/conformer_mega.hip, round 6): a workgroup owns an
 * utterance (T <= 64 encoder frames) for all layers -- the dependency chain of ConformerEncoderLayer
 * (aps/asr/transformer/impl.py:432-541, pre-norm: x += 1/2 FFN(LN x); x += out_proj(rel-attention(QKV(LN x)));
 * x += pw2(act(BN(dwconv15(GLU(pw1(LN x)))))); x += 1/2 FFN(LN x)) costs barriers inside a workgroup instead of ~11
 * dependent launches per layer.  Same arithmetic as aps_linear_panel (two f16 planes, three products, a power of two
 * per (row, 128-chunk), the fp32 recomputation), aps_attention_core (T <= 64 relative form, exact fp32 MFMA) and
 * aps_glu_dwconv.
 *   ApsMegaGemm   one projection PHASE of contraction 512: `image` = aps_linear_fp16x2_weight of the weight [N,
 *                 K_total] (LayerNorm-folded when colsum is given, as aps_linear_panel takes it), w32 the fp32
 *                 weight it was made from (row pitch ldw), K steps kstep0 .. kstep0 + 15 of ksteps_total = K_total
 *                 / 32 (K_total = 1024: two phases chained through the residual operand); out = alpha * act(LN-fold(A
 *                 W^T) + bias) + residual; act as aps_linear
 *   ApsMegaLayer  the ten phases of a layer, the depthwise weights dw_w [D, 15] / dw_b [D], the eval-mode BatchNorm as
 *                 bn_scale / bn_shift [D] (NULL = identity), conv_act as aps_glu_dwconv's `swish`
 *   aps_conformer_stack_scratch(D, FF)  floats of workspace PER UTTERANCE
 *   aps_conformer_stack   x [N, T, D] IN PLACE (the residual stream), lens int64 [N] valid frames or NULL, `layers`
 *                 a DEVICE array of num_layers ApsMegaLayer, rel [rel_len, 64] the relative position table the layers
 *                 share (score(i, j) += q_i . rel[j - i + rel_zero]), scratch N x aps_conformer_stack_scratch floats,
 *                 wide_count as in aps_linear_panel.  Built for T <= 64, D = 512, FF = 1024, heads = D / 64, 15 taps;
 *                 anything else returns APS_ERR_UNSUPPORTED (the caller keeps the per-launch path). */
typedef struct ApsMegaGemm {
  const void* image;
  const float* w32;
  const float* bias;
  const float* colsum;
  int32_t N, ksteps_total, kstep0, ldw;
  int32_t act, pad0;
  float alpha, ln_eps;
} ApsMegaGemm;
typedef struct ApsMegaLayer {
  ApsMegaGemm ff1_up, ff1_dn0, ff1_dn1, qkv, out, pw1, pw2, ff2_up, ff2_dn0, ff2_dn1;
  const float* dw_w;
  const float* dw_b;
  const float* bn_scale;
  const float* bn_shift;
  int32_t conv_act, pad1;
} ApsMegaLayer;
int64_t aps_conformer_stack_scratch(int64_t D, int64_t FF);
int aps_conformer_stack(float* x, const int64_t* lens, const ApsMegaLayer* layers, int32_t num_layers,
                        const float* rel, int64_t rel_zero, int64_t rel_len, int64_t N, int64_t T, int64_t D,
                        int64_t FF, int64_t heads, float* scratch, int32_t* wide_count, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* APS_AMD_H_ */
