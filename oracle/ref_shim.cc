// Thin C-ABI shim over the REFERENCE's own native FFT / window / streaming (i)STFT classes
// (csrc/utils/{fft,window,stft}.cc).  Only this shim is ours; the reference sources are compiled
// where they lie under /root/reference by oracle/Makefile and never copied into the repo.
// Output: oracle/_ref/libaps_ref.so (git-ignored; travels to the GPU box as a built artefact).
// Test infrastructure only (tests/test_ref_native.py): cross-checks the oracle's windows, framing
// and spectra against the reference's radix-2 RealFFT path.
#include <cstring>
#include <string>
#include <vector>

#include "utils/fft.h"
#include "utils/stft.h"
#include "utils/window.h"

extern "C" {

// window coefficients as csrc/utils/window.cc generates them
void aps_ref_window(const char* name, int32_t len, int32_t periodic, float* out) {
  aps::WindowFunction::Generate(name, out, len, periodic != 0);
}

// in-place packed real FFT of csrc/utils/fft.cc: [R0, R(N/2), R1, I1, ...]
void aps_ref_real_fft(float* buf, int32_t n, int32_t invert) {
  aps::FFTComputer fft(n);
  fft.RealFFT(buf, n, invert != 0);
}

// in-place complex FFT of csrc/utils/fft.cc:24-57: [R0, I0, R1, I1, ...], n_floats = 2 * points
void aps_ref_complex_fft(float* buf, int32_t n_floats, int32_t invert) {
  aps::FFTComputer fft(n_floats);
  fft.ComplexFFT(buf, n_floats, invert != 0);
}

int32_t aps_ref_fft_size(int32_t frame_len, int32_t frame_hop, const char* window,
                         const char* mode) {
  aps::StreamingSTFT stft(frame_len, frame_hop, window, mode);
  return stft.FFTSize();
}

int32_t aps_ref_frame_length(int32_t frame_len, int32_t frame_hop, const char* window,
                             const char* mode) {
  aps::StreamingSTFT stft(frame_len, frame_hop, window, mode);
  return stft.FrameLength();
}

// frame-by-frame STFT (csrc/utils/stft.cc:17-23); out: [T, fft_size] packed spectra.
// returns the number of frames written.
int32_t aps_ref_stft(const float* wav, int32_t num_samples, int32_t frame_len, int32_t frame_hop,
                     const char* window, const char* mode, float* out, int32_t max_frames) {
  aps::StreamingSTFT stft(frame_len, frame_hop, window, mode);
  const int32_t L = stft.FrameLength(), W = stft.FFTSize();
  std::vector<float> frame(W);
  int32_t t = 0;
  for (int32_t beg = 0; beg + L <= num_samples && t < max_frames; beg += frame_hop, ++t) {
    std::memcpy(frame.data(), wav + beg, sizeof(float) * L);
    stft.Compute(frame.data(), L, out + (size_t)t * W);
  }
  return t;
}

// frame-by-frame iSTFT with overlap-add normalisation and flush (csrc/utils/stft.cc:25-51);
// spec: [T, fft_size] packed; out: (T - 1) * hop + frame_length samples
void aps_ref_istft(const float* spec, int32_t num_frames, int32_t frame_len, int32_t frame_hop,
                   const char* window, const char* mode, float* out) {
  aps::StreamingiSTFT istft(frame_len, frame_hop, window, mode);
  const int32_t L = istft.FrameLength(), W = istft.FFTSize();
  std::vector<float> in(W), frame(W);
  for (int32_t t = 0; t < num_frames; ++t) {
    std::memcpy(in.data(), spec + (size_t)t * W, sizeof(float) * W);
    istft.Compute(in.data(), L, frame.data());
    std::memcpy(out + (size_t)t * frame_hop, frame.data(), sizeof(float) * frame_hop);
  }
  std::vector<float> tail(L);
  istft.Flush(tail.data());
  std::memcpy(out + (size_t)num_frames * frame_hop, tail.data(), sizeof(float) * (L - frame_hop));
}

}  // extern "C"
