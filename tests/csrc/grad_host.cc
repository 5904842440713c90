// TEST INFRASTRUCTURE: the index functors of aps_amd/csrc/grad_core.h and the C-ABI marshalling
// of aps_amd/csrc/grad_api.inc compiled for the HOST (g++), every functor run in a plain loop.
// tests/test_grad_host.py drives these `host_*` entry points through ctypes and compares them with
// torch autograd through the CPU oracle, so the adjoint arithmetic and index math of the HIP
// backward kernels are checked without a GPU.
#include <math.h>
#include <stdint.h>
#include <string.h>

#include <vector>

#include "../../include/aps_amd.h"
#include "../../aps_amd/csrc/grad_core.h"

#define APS_CHECK_ARG(cond) \
  do {                      \
    if (!(cond)) return APS_ERR_INVALID; \
  } while (0)

template <class Op>
static int host_each(const Op& op, int64_t n, void*) {
  for (int64_t i = 0; i < n; ++i) op(i);
  return APS_OK;
}

#define APS_GRAD_API(name) host_##name
#define APS_GRAD_EACH(op, n, stream) host_each(op, n, stream)
#include "../../aps_amd/csrc/grad_api.inc"

// the reverse-time sweep of aps_lstm_backward_sweep with a naive matrix product in place of the
// MFMA GEMM launch
extern "C" int host_lstm_backward_sweep(const float* gates, const float* c, const float* g_y,
                                        const float* w_hh_t, const int64_t* lens, float* g_pre,
                                        float* g_h_rec, float* g_c, int64_t N, int64_t T, int64_t H,
                                        void* stream) {
  memset(g_c, 0, sizeof(float) * N * H);
  for (int64_t t = T - 1; t >= 0; --t) {
    const float* rec = nullptr;
    if (t + 1 < T) {
      for (int64_t n = 0; n < N; ++n)
        for (int64_t j = 0; j < H; ++j) {
          double acc = 0;
          const float* a = g_pre + (n * T + t + 1) * 4 * H;
          const float* w = w_hh_t + j * 4 * H;
          for (int64_t k = 0; k < 4 * H; ++k) acc += (double)a[k] * w[k];
          g_h_rec[n * H + j] = (float)acc;
        }
      rec = g_h_rec;
    }
    int rc = host_lstm_backward_step(gates, c, g_y, rec, lens, g_c, g_pre, N, T, H, t, stream);
    if (rc != APS_OK) return rc;
  }
  return APS_OK;
}
