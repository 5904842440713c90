// C [I, J] = A^T B for two tall row-major matrices A [M, I], B [M, J] -- the weight gradient of every
// projection (g_W = g_pre^T x, aps/asr/transformer/impl.py's Linear layers under loss.backward(),
// aps/trainer/ddp.py:161-165), of the LSTM's two weight matrices and of the convolutions' im2col form
// -- with the column sums of A (the bias gradient) riding along.  Round 4: until now these products
// ran as the forward GEMM on TRANSPOSED copies of both operands (two transpose launches per product,
// 318 per training step) plus a two-launch column reduction per bias (247 pairs per step).
//
// The contraction runs over the ROWS of both operands, and the fp32 MFMA 32 x 32 x 2 takes exactly
// that shape straight from memory: lane l supplies A[m + l / 32][i0 + l % 32] and
// B[m + l / 32][j0 + l % 32] -- one coalesced dword per lane and operand (two 128-byte row segments per
// request), no transposition, no LDS, no barrier.  A wave owns a 32 x 32 tile of C; the four waves of a
// workgroup (2 x 2) share their operand segments through the L1.  Two dwords per 64-cycle MFMA and wave
// = 32 B/clk per CU, half of what the vector memory path delivers: the loop is bound by the fp32 matrix
// pipe, as it should be.
//
// M is cut into S slabs (split-K: 512 x 512 outputs are 64 tiles, far too few for 256 CUs) whose
// partial products go to the call's workspace and are summed in slab order by a second launch --
// deterministic, no atomics.  The slab index is the fastest-varying part of the block index, so with
// S = 8 every XCD walks one slab of both operands out of its own L2.
#include <stdint.h>

#include "common.h"

namespace aps {
namespace tn {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr uint32_t kOutside = 0x80000000u;  // a voffset no descriptor of < 2 GB covers: reads give 0
constexpr int kBlockRows = 32;               // rows of M per unrolled block: 16 MFMAs per wave

struct TnArgs {
  const float* A;
  const float* B;
  float* C;        // [I, ldc]           (S == 1) or the partials [S][I][J]
  float* colsum;   // [I] or null        (S == 1) or the partials [S][I]
  int64_t M, I, J, lda, ldb, ldc;
  int32_t tiles_i, tiles_j, slabs;
  int64_t slab_rows;  // a multiple of kBlockRows
};

__global__ __launch_bounds__(256, 2) void gemm_tn_kernel(TnArgs g) {
  const int wv = threadIdx.x >> 6, ln = threadIdx.x & 63;
  const int li = ln & 31, lk = ln >> 5;
  const int64_t lin = blockIdx.x;
  const int s = (int)(lin % g.slabs);
  const int64_t tile = lin / g.slabs;
  const int64_t i0 = (tile / g.tiles_j) * 64 + (wv >> 1) * 32, j0 = (tile % g.tiles_j) * 64 + (wv & 1) * 32;
  if (i0 >= g.I || j0 >= g.J) return;  // (no barrier in this kernel: a wave may leave)
  const int64_t m0 = (int64_t)s * g.slab_rows, m1 = min(g.M, m0 + g.slab_rows);
  // the descriptors end with the slab: rows past it read zeros (the range check covers voffset + soffset,
  // scripts/micro/soffset_range.hip), so the row position is the instruction's SCALAR offset and the loop
  // carries no vector address arithmetic and no edge branch
  auto rsrc_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.A), 0, (uint32_t)(m1 * g.lda * 4),
                                                  0x00020000);
  auto rsrc_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.B), 0, (uint32_t)(m1 * g.ldb * 4),
                                                  0x00020000);
  // this lane's (row parity, column) in either operand; lanes past the edge read zeros (kOutside)
  const bool in_a = i0 + li < g.I, in_b = j0 + li < g.J;
  const uint32_t pa = (uint32_t)g.lda * 4u, pb = (uint32_t)g.ldb * 4u;
  const uint32_t va = in_a ? (uint32_t)lk * pa + (uint32_t)(i0 + li) * 4u : kOutside;
  const uint32_t vb = in_b ? (uint32_t)lk * pb + (uint32_t)(j0 + li) * 4u : kOutside;
  auto fetch = [&](float (&a)[16], float (&b)[16], int64_t m) {
    // (blocks wholly past the slab: every request out of range -- clamp the scalar so it cannot wrap)
    const uint32_t row = (uint32_t)min(m, m1);
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      a[q] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc_a, va, (row + 2 * q) * pa, 0));
      b[q] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc_b, vb, (row + 2 * q) * pb, 0));
    }
  };
  f32x16 acc0, acc1;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc0[e] = acc1[e] = 0.f;
  float cs = 0.f;
  const bool want_cs = g.colsum != nullptr && (tile % g.tiles_j) == 0 && (wv & 1) == 0;
  auto multiply = [&](const float (&a)[16], const float (&b)[16]) {
#pragma unroll
    for (int q = 0; q < 16; q += 2) {  // two accumulators: consecutive MFMAs never depend on each other
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q], b[q], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q + 1], b[q + 1], acc1, 0, 0, 0);
    }
    if (want_cs) {
#pragma unroll
      for (int q = 0; q < 16; ++q) cs += a[q];
    }
  };
  // two register sets: the requests of block t + 1 are in flight under the MFMAs of block t
  float a0[16], b0[16], a1[16], b1[16];
  int64_t m = m0;
  if (m < m1) fetch(a0, b0, m);
  for (; m < m1; m += 2 * kBlockRows) {
    fetch(a1, b1, m + kBlockRows);  // (rows >= m1 read zeros: the last block needs no branch)
    multiply(a0, b0);
    fetch(a0, b0, m + 2 * kBlockRows);
    multiply(a1, b1);
  }
  // C[i0 + (e & 3) + 8 (e >> 2) + 4 lk][j0 + li]
  float* C = g.C + (g.slabs > 1 ? (int64_t)s * g.I * g.J : 0);
  const int64_t ldc = g.slabs > 1 ? g.J : g.ldc;
  if (in_b) {
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int64_t i = i0 + (e & 3) + 8 * (e >> 2) + 4 * lk;
      if (i < g.I) C[i * ldc + j0 + li] = acc0[e] + acc1[e];
    }
  }
  if (want_cs) {
    cs += __shfl_xor(cs, 32, 64);
    if (lk == 0 && in_a) g.colsum[(g.slabs > 1 ? (int64_t)s * g.I : 0) + i0 + li] = cs;
  }
}

// C[i][j] = sum_s P[s][i][j] (slab order), colsum[i] = sum_s pcs[s][i]
__global__ __launch_bounds__(256) void gemm_tn_reduce_kernel(const float* __restrict__ P,
                                                            const float* __restrict__ pcs,
                                                            float* __restrict__ C, float* __restrict__ colsum,
                                                            int64_t I, int64_t J, int64_t ldc, int slabs) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x, IJ = I * J;
  if (idx < IJ) {
    float v = 0.f;
    for (int s = 0; s < slabs; ++s) v += P[(int64_t)s * IJ + idx];
    C[(idx / J) * ldc + idx % J] = v;
  } else if (colsum != nullptr && idx - IJ < I) {
    const int64_t i = idx - IJ;
    float v = 0.f;
    for (int s = 0; s < slabs; ++s) v += pcs[(int64_t)s * I + i];
    colsum[i] = v;
  }
}

static int pick_slabs(int64_t M, int64_t I, int64_t J) {
  const int64_t tiles = ((I + 63) / 64) * ((J + 63) / 64);
  int64_t s = (512 + tiles - 1) / tiles;
  if (s > M / 64) s = M / 64;  // at least 64 rows per slab
  if (s > 64) s = 64;
  return (int)(s < 1 ? 1 : s);
}

}  // namespace tn
}  // namespace aps

extern "C" int64_t aps_gemm_tn_workspace(int64_t M, int64_t I, int64_t J) {
  if (M <= 0 || I <= 0 || J <= 0) return 0;
  const int s = aps::tn::pick_slabs(M, I, J);
  return s > 1 ? (int64_t)s * (I * J + I) * 4 : 0;
}

extern "C" int aps_gemm_tn(const float* A, const float* B, float* C, float* colsum, void* workspace,
                           int64_t M, int64_t I, int64_t J, int64_t lda, int64_t ldb, int64_t ldc,
                           void* stream) {
  using namespace aps::tn;
  APS_CHECK_ARG(A && B && C && M > 0 && I > 0 && J > 0 && lda >= I && ldb >= J && ldc >= J);
  // (32-bit byte offsets inside the buffer descriptors)
  if (M * lda * 4 >= ((int64_t)1 << 31) || M * ldb * 4 >= ((int64_t)1 << 31)) return APS_ERR_UNSUPPORTED;
  hipStream_t st = static_cast<hipStream_t>(stream);
  TnArgs g;
  g.A = A, g.B = B, g.M = M, g.I = I, g.J = J, g.lda = lda, g.ldb = ldb, g.ldc = ldc;
  g.tiles_i = (int32_t)((I + 63) / 64), g.tiles_j = (int32_t)((J + 63) / 64);
  g.slabs = pick_slabs(M, I, J);
  const int64_t per = (M + g.slabs - 1) / g.slabs;
  g.slab_rows = (per + kBlockRows - 1) / kBlockRows * kBlockRows;
  g.slabs = (int32_t)((M + g.slab_rows - 1) / g.slab_rows);  // (rounding may leave the last slab empty)
  float* part = static_cast<float*>(workspace);
  if (g.slabs > 1) {
    APS_CHECK_ARG(workspace != nullptr);
    g.C = part;
    g.colsum = colsum ? part + (int64_t)g.slabs * I * J : nullptr;
  } else {
    g.C = C;
    g.colsum = colsum;
  }
  const int64_t blocks = (int64_t)g.tiles_i * g.tiles_j * g.slabs;
  if (blocks >= ((int64_t)1 << 31)) return APS_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(gemm_tn_kernel, dim3((unsigned)blocks), dim3(256), 0, st, g);
  int rc = aps_launch_status();
  if (rc != APS_OK || g.slabs == 1) return rc;
  const int64_t items = I * J + (colsum ? I : 0);
  hipLaunchKernelGGL(gemm_tn_reduce_kernel, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, st, part,
                     g.colsum, C, colsum, I, J, ldc, (int)g.slabs);
  return aps_launch_status();
}
