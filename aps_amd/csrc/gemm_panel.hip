// fp32 GEMMs on the fp16 matrix pipe, PANEL form (aps_linear_panel; round 4): the arithmetic of
// gemm_fp16x2.hip -- two fp16 planes per operand, three MFMA products per term, main and cross sums in
// their own fp32 accumulators, tiles whose operands the planes cannot hold recomputed on the fp32 MFMA
// inside the launch -- with a different division of labour.
//
// Why.  gemm_fp16x2_kernel streams A through a double-buffered LDS one 32-wide K step at a time: every
// K step of a workgroup is a chain  request -> LDS write -> barrier -> fragment read -> 12 MFMAs  of
// ~1 900 - 2 600 cycles for 384 cycles of matrix work (profiles/r03_gemm_trace_*), hidden only by four
// co-resident workgroups -- which the launches of BASELINE's 32 utterances per GPU (M = 2016: 128 ...
// 384 tiles on 256 CUs) do not have -- and the planes of A come from a pass of their own (a launch and
// a 2 x M x K x 4 byte round trip per GEMM: 1.0 of the 5.3 ms the projections took per 128-utterance
// step in round 3).  Here a workgroup owns RT rows x 128 columns and walks K in CHUNKS of 256 / 128:
//   * the fp32 rows of the chunk go global -> registers (requested late in the previous chunk), the staging lanes
//     find the chunk's row maxima, scale, split and write both planes of the whole chunk to LDS once;
//   * the 8 K steps of the chunk then run with NO barrier and no A traffic at all: fragment reads from
//     the static LDS image, weight fragments from the image of W straight into a register ring
//     requested WRING - 1 steps ahead (and across chunk boundaries), 6 SM MFMAs per step and wave;
//   * a power-of-two scale per (row, chunk) instead of per row: the chunk's accumulators are folded
//     into a running fp32 sum with one ldexp per element (exact), so no pass over the whole row has to
//     precede the first product, and an element only has to lie within 2^-30 of the largest magnitude
//     among the 256 it shares a chunk with (a finer granule than gemm_fp16x2's whole row: the bound of
//     gemm_fp16x2.hip's header holds a fortiori);
//   * the LayerNorm fold's row statistics (sum, sum of squares of the RAW row) ride along in the
//     staging lanes; no side channel, no workspace.
// Same weight image (aps_linear_fp16x2_weight), same epilogue (bias, activation, alpha, residual,
// LayerNorm fold), same detection rule (fit_key) and fp32 recomputation as gemm_fp16x2.hip.
//
// LDS image of a chunk: [plane h | l][row RT][256 f16 + 16 bytes]; the 528-byte row pitch shifts
// consecutive rows by one 16-byte slot, which makes the 16-lane groups of ds_read_b128 (MI355X_MICROARCH
// "LDS": {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} of each half wave) conflict free.
#include <stdint.h>
#include <string.h>

#include <type_traits>

#include "common.h"

namespace aps {
namespace panel {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// (the constants of gemm_fp16x2.hip: the low plane holds (x' - h) 2^11, "fits" = 0 or 2^-16 <= |x'| < 2^15)
constexpr float kLowUp = 2048.f, kLowDown = 1.0f / 2048.f;
constexpr int kFitBias = 15;
constexpr uint32_t kFitMax = 30u;
constexpr uint32_t kOutside = 0x80000000u;  // a voffset no descriptor of < 2 GB covers: reads give 0

__device__ __forceinline__ int32_t scale_exponent(float mx) {
  int be = (int)((__float_as_uint(mx) >> 23) & 0xffu);
  be = be < 1 ? 1 : (be > 254 ? 254 : be);
  return 141 - be;
}

__device__ __forceinline__ f32x16 mfma_f16(u32x4 a, u32x4 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b),
                                                c, 0, 0, 0);
}

// lane exchanges inside a row of 16 lanes on the DPP path (no LDS crossbar round trip, unlike ds_bpermute / __shfl_xor):
// 0xB1 = quad_perm [1, 0, 3, 2], 0x4E = quad_perm [2, 3, 0, 1], 0x141 = row_half_mirror (lane i <-> 7 - i of its 8)
template <int CTRL>
__device__ __forceinline__ float dpp_move(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
// maximum over an aligned group of G = 4 | 8 lanes, in every lane of the group
template <int G>
__device__ __forceinline__ float group_max(float v) {
  static_assert(G == 4 || G == 8, "aligned groups of 4 or 8 lanes");
  v = fmaxf(v, dpp_move<0xB1>(v));
  v = fmaxf(v, dpp_move<0x4E>(v));
  if constexpr (G == 8) v = fmaxf(v, dpp_move<0x141>(v));
  return v;
}

// f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>): a loop whose index is a constant
// expression in every iteration (register-ring positions, LDS offsets)
template <int N, int I = 0, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<N, I + 1>(f);
  }
}

struct PanelArgs {
  const float* A;         // [M, K] fp32, row pitch lda
  const void* Wp;         // image of W (aps_linear_fp16x2_weight)
  const float* W32;       // the fp32 weight the image was made from (row pitch ldw): the fp32 path
  const float* bias;      // [N] or null
  const float* residual;  // [M, N] (ldc) or null
  float* C;
  int32_t* wide_count;    // device counter of tiles recomputed in fp32 (or null)
  const float* ln_cs;     // LayerNorm fold: column sums of W diag(gamma) (null: plain)
  int64_t M, N, K, lda, ldw, ldc;
  float alpha, ln_eps;
  int32_t act, tiles_n, per_xcd, total, ksteps;
  const void* pf;         // what the NEXT launch of the stream will read first (its weight image), or null
  int64_t pf_bytes;
#ifdef APS_PANEL_TRACE
  int32_t trace_slot0;    // first trace slot of this launch, or -1
#endif
};

// 8 fp32 values (already scaled into the planes' range) -> 8 h halves, 8 l halves
__device__ __forceinline__ void split8(const float s[8], u32x4& h, u32x4& l) {
  _Float16 hh[8], ll[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    hh[e] = (_Float16)s[e];
    ll[e] = (_Float16)((s[e] - (float)hh[e]) * kLowUp);
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    h[e] = __builtin_bit_cast(uint32_t, f16x2{hh[2 * e], hh[2 * e + 1]});
    l[e] = __builtin_bit_cast(uint32_t, f16x2{ll[2 * e], ll[2 * e + 1]});
  }
}

#ifdef APS_PANEL_TRACE
// experiments only (scripts/panel_trace_under_load.py with a library built with -DAPS_PANEL_TRACE): s_memtime stamps of
// lane 0 of WAVE 0 of every workgroup of the launches whose (N, K) pass the host-side filter -- a slot of 32 stamps; the
// HOST hands every traced launch its own slot range when it is issued (or captured: a replayed graph node rewrites its
// own slots, the trace holds each node's latest replay), so the kernel pays no atomic for it:
//   0 entry | 1 first chunk's rows arrived | per chunk c < 8: 2+3c planes written (at barrier A) | 3+3c through
//   barrier A | 4+3c MFMA loop + fold done, through barrier B | 26 epilogue begins | 27 end |
//   28 N << 32 | K | 29 C pointer (which batch) | 30 XCC id << 32 | block id | 31 wave | LN << 8 | linear tile << 16
constexpr unsigned kPtSlots = 1u << 19;
__device__ unsigned long long g_panel_trace[(size_t)kPtSlots * 32];
__device__ unsigned int g_pt_ctl[4];  // [3] 1: recording (the other words are host side now)
static unsigned h_pt_next = 0, h_pt_n = 0, h_pt_k = 0;   // next free slot, the (N, K) filter (0: any)
#define PT_STAMP(k) \
  if (ptrace && (k) < 28) ptrace[(k)] = __builtin_amdgcn_s_memtime();
#define PT_CHUNK(k) \
  if (ptrace && (k) < 26) ptrace[(k)] = __builtin_amdgcn_s_memtime();
#else
#define PT_STAMP(k)
#define PT_CHUNK(k)
#endif

// RT rows x TN = 128 columns per workgroup, four waves side by side along N (each RT x 32).
// KC: chunk width (the chunk in flight lives in the staging lanes' registers).
// WRING: register stages of the weight-fragment ring (a K step's four 16-byte fragments per stage).
// MINW: waves per SIMD the kernel is compiled for (the register bound: 2 -> 256 VGPRs, 4 -> 128).
// DMA (round 6, form 'f'): the fp32 rows of the NEXT chunk travel global -> LDS by LDS-DMA (buffer_load ... lds: no
// registers in flight), requested a whole chunk ahead -- right behind barrier A, in front of this chunk's weight
// fragments -- into a staging area in which every wave owns the RT / NW rows its own lanes split (no barrier between
// the DMA and the split, only the wave's own vmcnt).  profiles/r06_panel_trace_under_load.txt: with the rows requested
// late in the previous chunk through registers, "maxima + split" was 2.3 k cycles per chunk alone and 3.8 k beside
// two other streams, most of it the wait for the rows; the freed registers hold a four-stage weight ring instead.
template <int RT, int TN, int KC, bool LN, int WRING, int MINW = 2, bool DMA = false>
__global__ __launch_bounds__(TN * 2, MINW) void gemm_panel_kernel(PanelArgs g) {
  constexpr int NW = TN / 32, NT = NW * 64;    // waves, threads
  constexpr int KS = KC / 32;                  // K steps per chunk of KC
  constexpr int PB = KC * 2 + 16;              // row pitch of a plane in LDS (bytes)
  constexpr int PLANE = RT * PB, SM = RT / 32;
  constexpr int TPR = NT / RT;                 // staging lanes per row
  constexpr int GPT = KC / 8 / TPR;            // 8-element groups per staging lane and chunk
  static_assert(KS % WRING == 0, "the ring position of a chunk's first K step must be 0");
  static_assert(GPT >= 1 && GPT * TPR * 8 == KC, "the staging lanes cover a chunk exactly");
  constexpr int kHand = NW * 32 * 36 * 4;      // the epilogue's hand-over blocks (4.5 KB per wave)
  constexpr int IMG = 2 * PLANE;               // one chunk's image: both planes
  __shared__ __attribute__((aligned(16))) unsigned char s_a[IMG > kHand ? IMG : kHand];
  __shared__ __attribute__((aligned(16))) int32_t s_exp[1][RT];
  __shared__ float2 s_stat[RT];
  __shared__ int32_t s_wide;
  __shared__ __attribute__((aligned(16))) unsigned char s_pf[1024];  // where prefetched lines are dropped
  // DMA: the chunk's fp32 rows as they arrive, [row RT][KC floats]; wave w owns rows w RT / NW ...
  constexpr int STG_ROW = KC * 4, STG_WAVE = (RT / NW) * STG_ROW, STG_INSTR = STG_WAVE / 1024;
  static_assert(!DMA || (STG_WAVE % 1024 == 0 && 1024 % STG_ROW == 0 && 64 / TPR == RT / NW),
                "DMA: whole rows per 1 KB request, a wave splits the rows it requested");
  __shared__ __attribute__((aligned(16))) unsigned char s_stage[DMA ? RT * STG_ROW : 16];
  const int tid = threadIdx.x, ln = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);  // = the wave's 32-column group

  // tile order: workgroup b runs on XCD b % 8 (observed, for speed only): every XCD gets a contiguous
  // range of tiles, so the column tiles of a row panel share one L2
  const int32_t b = (int32_t)blockIdx.x;
  const int32_t lin = (b & 7) * g.per_xcd + (b >> 3);
  if (lin >= g.total || (b >> 3) >= g.per_xcd) return;
  const int32_t pnl = lin / g.tiles_n;
  const int32_t m0 = pnl * RT, n0 = (lin - pnl * g.tiles_n) * TN;
#ifdef APS_PANEL_TRACE
  unsigned long long* ptrace = nullptr;
  if (ln == 0 && wv == 0 && g.trace_slot0 >= 0 && g_pt_ctl[3] != 0) {   // (wave 0 of every workgroup)
    const unsigned slot = (unsigned)g.trace_slot0 + (unsigned)lin;
    if (slot < kPtSlots) {
      ptrace = g_panel_trace + (size_t)slot * 32;
      ptrace[28] = ((unsigned long long)g.N << 32) | (unsigned long long)g.K;
      ptrace[29] = (unsigned long long)reinterpret_cast<uintptr_t>(g.C);
      ptrace[30] = ((unsigned long long)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)) << 32) | (unsigned)b;
      ptrace[31] = (unsigned long long)wv | (LN ? 0x100ull : 0ull) | ((unsigned long long)lin << 16);
    }
  }
#endif
  PT_STAMP(0)

  // ---- operands ----
  const int64_t groups = ((g.N + 127) / 128) * 4;  // 32-column groups of the image
  const int32_t wstep_bytes = (int32_t)(groups * 4096);
  auto rsrc_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(g.Wp), 0,
                                                  (uint32_t)(wstep_bytes * g.ksteps), 0x00020000);
  const int32_t* ew_tab = reinterpret_cast<const int32_t*>(static_cast<const unsigned char*>(g.Wp) +
                                                           (int64_t)wstep_bytes * g.ksteps);
  auto rsrc_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.A), 0,
                                                  (uint32_t)(g.M * g.lda * 4), 0x00020000);
  const int32_t vw = (n0 / 32 + wv) * 4096 + ln * 16;
  // staging lane: row sr of the panel, groups q, q + TPR, ... of the chunk (8 consecutive k each)
  const int sr = tid / TPR, q = tid % TPR;
  const bool row_ok = m0 + sr < g.M;
  const uint32_t va = (uint32_t)((int64_t)(m0 + sr) * g.lda * 4) + (uint32_t)q * 32u;
  const int32_t Ki = (int32_t)g.K;

  u32x4 ra[GPT][2];  // the chunk in flight: [group][first | second four elements]
  auto gload_a = [&](int c) {
    if constexpr (DMA) {
      // request i of this wave: rows (wv RT / NW + i 1024 / STG_ROW ...), lane ln brings 16 bytes of one of them
      constexpr int RPI = 1024 / STG_ROW;       // rows per request
      constexpr int LPR = STG_ROW / 16;         // lanes per row
      const int dr = ln / LPR, dk = (ln % LPR) * 4;
      asm volatile("" ::: "memory");
#pragma unroll
      for (int i = 0; i < STG_INSTR; ++i) {
        const int r = wv * (RT / NW) + i * RPI + dr;
        const int32_t k = c * KC + dk;
        const uint32_t off = (uint32_t)((int64_t)(m0 + r) * g.lda * 4) + (uint32_t)k * 4u;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(
            rsrc_a, (__attribute__((address_space(3))) void*)(s_stage + wv * STG_WAVE + i * 1024), 16,
            (m0 + r < g.M && k < Ki) ? off : kOutside, 0, 0, 0);
      }
      asm volatile("" ::: "memory");
      return;
    }
    const int32_t k0 = c * KC + q * 8;
#pragma unroll
    for (int j = 0; j < GPT; ++j) {
      const int32_t k = k0 + j * TPR * 8;
      const uint32_t off = va + (uint32_t)(c * KC * 4 + j * TPR * 32);
      ra[j][0] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_a, (row_ok && k < Ki) ? off : kOutside, 0, 0);
      ra[j][1] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_a, (row_ok && k + 4 < Ki) ? off + 16u : kOutside, 0, 0);
    }
  };
  u32x4 wb[WRING][2][2];  // [ring stage][MFMA K step][plane]
  const int32_t last_step = g.ksteps - 1;
  auto gload_w = [&](auto stage, int32_t gs) {  // global K step gs (clamped: never out of the image)
    constexpr int P = decltype(stage)::value;
    const int32_t soff = (gs < last_step ? gs : last_step) * wstep_bytes;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int p = 0; p < 2; ++p)
        wb[P][kk][p] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_w, vw, soff + (kk * 2 + p) * 1024, 0);
  };

  f32x16 sum[SM];
#pragma unroll
  for (int i = 0; i < SM; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) sum[i][e] = 0.f;

  const int nchunks = (g.ksteps + KS - 1) / KS;
  gload_a(0);
  // the first WRING - 1 K steps' weight fragments
  static_for<WRING - 1>([&](auto sc) { gload_w(sc, decltype(sc)::value); });
  const int li = ln & 31, lk = ln >> 5;
  const int32_t col = n0 + wv * 32 + li;
  // (the image's tables are padded to whole 128-column groups)
  const int32_t ew = ew_tab[col];
  const int32_t ew_flag = ew_tab[groups * 32 + col];
  // the epilogue's per-column operands, requested now: at their first use they stood 1 000 cycles of
  // L2 round trip in front of the epilogue (scripts/panel_trace.py)
  const bool col_ok = col < g.N;
  const float bv = (g.bias && col_ok) ? g.bias[col] : 0.f;
  const float cs = (LN && col_ok) ? g.ln_cs[col] : 0.f;
  if (tid == 0) s_wide = 0;
  const uint32_t c_bytes = (uint32_t)(g.M * g.ldc * 4);
  auto rsrc_c = __builtin_amdgcn_make_buffer_rsrc(g.C, 0, c_bytes, 0x00020000);
  auto rsrc_r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.residual), 0,
                                                  g.residual ? c_bytes : 0u, 0x00020000);
  // C leaves (and the residual arrives) as 16-byte runs of a row when the layout allows it
  const bool vec = ((g.N | g.ldc) & 3) == 0 && ((reinterpret_cast<uintptr_t>(g.C) |
                   reinterpret_cast<uintptr_t>(g.residual)) & 15) == 0;

  float s1 = 0.f, s2 = 0.f;  // LayerNorm fold: this lane's share of its raw row
  int32_t fitmin = 0;        // smallest frexp exponent of a scaled element (0 for zeros): < -15 = does not fit
  // fragment base of this lane: row ln % 32 of a row block, k half ln / 32
  const unsigned char* const frag = s_a + (ln & 31) * PB + (ln >> 5) * 16;
  unsigned char* const sdst = s_a + sr * PB + q * 16;

  // the staging lanes' share of a chunk: row maximum (and the raw row's sums), then group by group
  auto rowmax = [&]() -> int32_t {
    float mx = 0.f;
#pragma unroll
    for (int j = 0; j < GPT; ++j)
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float v = __uint_as_float(ra[j][h][e]);
          mx = fmaxf(mx, fabsf(v));
          if (LN) {
            s1 += v;
            s2 = fmaf(v, v, s2);
          }
        }
    if constexpr (TPR == 4 || TPR == 8) {
      mx = group_max<TPR>(mx);
    } else {
#pragma unroll
      for (int o = 1; o < TPR; o <<= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    }
    return scale_exponent(mx);
  };
  auto split_group = [&](auto jc, int32_t ex, int buf) {
    constexpr int j = decltype(jc)::value;
    _Float16 hh[8], ll[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float x = __uint_as_float(ra[j][e >> 2][e & 3]);
      const float sc = ldexpf(x, ex);
      fitmin = min(fitmin, __builtin_amdgcn_frexp_expf(sc));
      hh[e] = (_Float16)sc;
      // (x' - h) 2^11 = x 2^(ex + 11) - 2048 h, exact before the one rounding to f16
      ll[e] = (_Float16)fmaf((float)hh[e], -kLowUp, ldexpf(x, ex + 11));
    }
    u32x4 h, l;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      h[e] = __builtin_bit_cast(uint32_t, f16x2{hh[2 * e], hh[2 * e + 1]});
      l[e] = __builtin_bit_cast(uint32_t, f16x2{ll[2 * e], ll[2 * e + 1]});
    }
    *reinterpret_cast<u32x4*>(sdst + buf * IMG + j * TPR * 16) = h;
    *reinterpret_cast<u32x4*>(sdst + buf * IMG + PLANE + j * TPR * 16) = l;
  };

  // (Measured and not kept -- profiles/r04_panel_trace.txt: a second LDS image, the next chunk's rows
  // scaled / split / written to it group by group between this chunk's MFMAs.  The chunk's loop is bound
  // by its weight-fragment fetches (16 KB per K step through a 64 B/clk vector-memory path against 192
  // MFMA cycles), the staging's VALU work delays those requests in the in-order wave one for one: same
  // workgroup life, twice the LDS, and the second batch in flight lost its co-residency.)
  uint32_t vq[SM][4];
  u32x4 rq[SM][4];
  // The launch that follows this one in the stream starts cold: its weight image was last touched a
  // whole step ago (196 MB of images + the step's activations do not stay in the 256 MB Infinity Cache
  // across steps), and at 32 utterances per launch a workgroup's 256 KB of fragments then arrive at
  // the latency of HBM, 12 requests per wave at a time.  So every workgroup asks for its share of the
  // NEXT launch's image -- LDS-DMA requests into a dummy kilobyte: no registers, nothing
  // ever reads them, the line stays in this XCD's L2.
  // Round 6: the requests (and the residual rows') are issued from INSIDE the last chunk, right behind the last weight
  // fragment the launch will ever request, instead of on the way out: the rest of the last chunk's MFMAs, the fold and
  // the epilogue's arithmetic then run while they are in flight (the trace of round 5's form: "epilogue + prefetch,
  // stores drained" was 9.5 k cycles of a 29 k-cycle workgroup life, most of it the wait for these lines).  A FIXED
  // number of requests, the surplus ones out of range (they return at once): the compiler then knows how many
  // younger requests sit behind a fragment and keeps its partial vmcnt waits (behind a loop of unknown length every
  // later wait became vmcnt(0), i.e. waited for the prefetch).
  constexpr int PF_ROUNDS = 16;
  static_assert(!DMA || KC == TN, "DMA: the residual tile reuses the rows' staging area (RT x TN floats)");
  auto tail_requests = [&]() {
    if constexpr (DMA) {
      // the residual TILE by LDS-DMA into the staging area (dead: the last chunk's rows were split before barrier A):
      // no registers in flight through the last chunk's loop; the epilogue reads it from LDS behind a barrier
      constexpr int RPI = 1024 / STG_ROW, LPR = STG_ROW / 16;
      const int dr = ln / LPR, dc = (ln % LPR) * 4;
      asm volatile("" ::: "memory");
#pragma unroll
      for (int i = 0; i < STG_INSTR; ++i) {
        const int64_t row = m0 + wv * (RT / NW) + i * RPI + dr;
        const int32_t colq = n0 + dc;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(
            rsrc_r, (__attribute__((address_space(3))) void*)(s_stage + wv * STG_WAVE + i * 1024), 16,
            (vec && row < g.M && colq < g.N) ? (uint32_t)((row * g.ldc + colq) * 4) : kOutside, 0, 0, 0);
      }
    } else {
      const int rr = ln >> 3, c4 = (ln & 7) * 4;
      const int32_t qcol = n0 + wv * 32 + c4;
#pragma unroll
      for (int i = 0; i < SM; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int64_t row = m0 + i * 32 + rr + 8 * j;
          vq[i][j] = (qcol < g.N && row < g.M) ? (uint32_t)((row * g.ldc + qcol) * 4) : kOutside;
          rq[i][j] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_r, vec ? vq[i][j] : kOutside, 0, 0);
        }
    }
    {
      // (always PF_ROUNDS requests, out of range without a next image: the count behind the residual tile is fixed)
      const int64_t pfb = g.pf != nullptr ? g.pf_bytes : 0;
      const int64_t share = (((pfb + g.per_xcd - 1) / g.per_xcd) + 4095) & ~(int64_t)4095;
      const int rounds = (int)(share > 4096 * PF_ROUNDS ? PF_ROUNDS : share / 4096);
      auto rsrc_p = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(g.pf), 0, (uint32_t)pfb, 0x00020000);
      const uint32_t base = (uint32_t)((int64_t)(b >> 3) * share) + (uint32_t)tid * 16u;
#pragma unroll
      for (int r = 0; r < PF_ROUNDS; ++r)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_p, (__attribute__((address_space(3))) void*)s_pf, 16,
                                                 r < rounds ? base + r * 4096u : kOutside, 0, 0, 0);
      asm volatile("" ::: "memory");
    }
  };
  auto chunk = [&](auto lastc, int c) {
    constexpr bool last = decltype(lastc)::value;  // (its own instantiation: the residual registers are born here)
    // ---- the chunk's planes: maxima, scale, split, one LDS image ----
#ifdef APS_PANEL_TRACE
    if (c == 0) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      PT_STAMP(1)
    }
#endif
    if constexpr (DMA) {
      // this wave's rows of the chunk have landed: they were requested in front of every weight fragment still in
      // flight (loads return in order); at most (WRING - 2) stages may still be out -- all of them for chunk 0
      if (c == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      } else if constexpr (WRING >= 4) {
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      } else if constexpr (WRING == 3) {
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      const unsigned char* const src = s_stage + sr * STG_ROW + q * 32;
#pragma unroll
      for (int j = 0; j < GPT; ++j) {
        ra[j][0] = *reinterpret_cast<const u32x4*>(src + j * TPR * 32);
        ra[j][1] = *reinterpret_cast<const u32x4*>(src + j * TPR * 32 + 16);
      }
    }
    const int32_t ex = rowmax();
    static_for<GPT>([&](auto jc) { split_group(jc, ex, 0); });
    if (q == 0) s_exp[0][sr] = ex;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    PT_CHUNK(2 + 3 * c)
    __builtin_amdgcn_s_barrier();
    PT_CHUNK(3 + 3 * c)
    if constexpr (DMA && !last) gload_a(c + 1);  // (this wave's staging rows were read before the barrier)

    // ---- the chunk's K steps on the static image: no barrier, no A traffic ----
    f32x16 acc[SM], accx[SM];
#pragma unroll
    for (int i = 0; i < SM; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][e] = accx[i][e] = 0.f;
    const int32_t gs0 = c * KS;
    const int kcount = last ? ((g.ksteps - gs0) < KS ? (g.ksteps - gs0) : KS) : KS;  // (only the last can be short)
    // `full`: every K step of the chunk exists, nothing in the loop is conditional.  (Round 6: with the steps of the
    // LAST chunk under a run-time `ks < kcount`, the compiler sank every weight-fragment request out of the step that
    // issues it into the step that consumes it -- a load has no side effect, its only use sits in a later block --
    // so each K step of the last chunk, a quarter of K = 512, waited a whole round trip; the common case, a whole
    // last chunk, now takes the branch-free instantiation, and it requests nothing beyond the last step.)
    auto kstep = [&](auto ksc, auto fullc) {
      constexpr int ks = decltype(ksc)::value;
      constexpr bool full = decltype(fullc)::value;
      constexpr int P = ks % WRING, PN = (ks + WRING - 1) % WRING;
      if (full || ks < kcount) {
        // The next chunk's rows are requested BEHIND the last weight fragment this chunk still needs
        // (loads return in order: requested ahead of them, every fragment wait of the chunk would also
        // wait for the rows); the fragments requested after this point belong to the next chunk, whose
        // split has waited for the rows by then.
        if (!DMA && ks == KS - (WRING - 1) && !last) gload_a(c + 1);
        if constexpr (!last || !full) {
          gload_w(std::integral_constant<int, PN>{}, gs0 + ks + WRING - 1);
        } else if constexpr (ks + WRING - 1 < KS) {
          gload_w(std::integral_constant<int, PN>{}, gs0 + ks + WRING - 1);
        }
        if constexpr (DMA && last && full && ks == (KS >= WRING ? KS - WRING : 0)) tail_requests();
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
          for (int i = 0; i < SM; ++i) {
            const unsigned char* fa = frag + i * 32 * PB + ks * 64 + kk * 32;
            const u32x4 ah = *reinterpret_cast<const u32x4*>(fa);
            const u32x4 al = *reinterpret_cast<const u32x4*>(fa + PLANE);
            accx[i] = mfma_f16(ah, wb[P][kk][1], accx[i]);  // h l
            acc[i] = mfma_f16(ah, wb[P][kk][0], acc[i]);    // h h
            accx[i] = mfma_f16(al, wb[P][kk][0], accx[i]);  // l h
          }
      }
    };
    if constexpr (!last) {
      static_for<KS>([&](auto ksc) { kstep(ksc, std::true_type{}); });
    } else if constexpr (DMA) {
      // (the DMA form is only launched for K % KC == 0: no run-time branch, whose join would cost the partial vmcnt
      // waits behind it -- the compiler falls back to vmcnt(0), i.e. to waiting for the prefetched lines)
      static_for<KS>([&](auto ksc) { kstep(ksc, std::true_type{}); });
    } else {
      if (kcount == KS)
        static_for<KS>([&](auto ksc) { kstep(ksc, std::true_type{}); });
      else
        static_for<KS>([&](auto ksc) { kstep(ksc, std::false_type{}); });
    }
    if constexpr (last && !DMA) tail_requests();   // (round 5's forms: behind the loop)

    // ---- fold: sum += 2^-(ea[row, chunk] + ew[col]) (main + 2^-11 cross) ----
#pragma unroll
    for (int i = 0; i < SM; ++i)
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        const i32x4 ea = *reinterpret_cast<const i32x4*>(&s_exp[0][i * 32 + 8 * r4 + 4 * lk]);
#pragma unroll
        for (int e = 0; e < 4; ++e)
          sum[i][r4 * 4 + e] += ldexpf(fmaf(accx[i][r4 * 4 + e], kLowDown, acc[i][r4 * 4 + e]), -(ea[e] + ew));
      }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // every read of the image and of s_exp is behind us
    PT_CHUNK(4 + 3 * c)
  };
  for (int c = 0; c + 1 < nchunks; ++c) chunk(std::false_type{}, c);
  chunk(std::true_type{}, nchunks - 1);

  // ---- does any operand of this tile fail to fit its scale?  the rows' statistics ----
  {
    bool wide = fitmin < -kFitBias;
    wide = __any(wide || ew_flag != 0);
    if (wide && ln == 0) s_wide = 1;
    if (LN) {
#pragma unroll
      for (int o = 1; o < TPR; o <<= 1) {
        s1 += __shfl_xor(s1, o, 64);
        s2 += __shfl_xor(s2, o, 64);
      }
      if (q == 0) {
        const float mean = s1 / (float)g.K;
        const float var = fmaxf(s2 / (float)g.K - mean * mean, 0.f);
        s_stat[sr] = make_float2(mean, 1.0f / sqrtf(var + g.ln_eps));
      }
    }
  }
  if constexpr (DMA) {
    // this wave's part of the residual tile has landed (PF_ROUNDS younger requests may still be out: a plain
    // __syncthreads() carries a workgroup release, which waits for every LDS-DMA request -- the prefetched lines too)
    asm volatile("s_waitcnt vmcnt(16) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  } else {
    __syncthreads();
  }
  const bool tile_wide = s_wide != 0;
  if (tile_wide) {
    // The fp32 path (gemm_fp16x2_kernel's): the tile once more on v_mfma_f32_32x32x2_f32 from the fp32
    // operands -- exact products, fp32 accumulation; rare, so plain: every lane fetches its own operand
    // rows, 4 k per request (lanes 0-31 k0 .. k0 + 3, lanes 32-63 k0 + 4 .. k0 + 7).
    if (tid == 0 && g.wide_count) atomicAdd(g.wide_count, 1);
#pragma unroll
    for (int i = 0; i < SM; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) sum[i][e] = 0.f;
    auto rsrc_w32 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.W32), 0,
                                                      (uint32_t)(g.N * g.ldw * 4), 0x00020000);
    const int64_t wrow = col < g.N ? col : g.N - 1;
    const uint32_t wo = (uint32_t)(wrow * g.ldw * 4) + lk * 16;
    uint32_t ao[SM];
#pragma unroll
    for (int i = 0; i < SM; ++i) {
      const int64_t arow = m0 + i * 32 + li < g.M ? m0 + i * 32 + li : g.M - 1;
      ao[i] = (uint32_t)(arow * g.lda * 4) + lk * 16;
    }
#pragma unroll 2
    for (int32_t k0 = 0; k0 < Ki; k0 += 8) {
      const int32_t kq = k0 + 4 * lk;
      const bool kin = kq < Ki;  // (K is a multiple of 4: a quad is inside or outside)
      const u32x4 wq = __builtin_amdgcn_raw_buffer_load_b128(rsrc_w32, kin ? wo + k0 * 4 : kOutside, 0, 0);
      u32x4 aq4[SM];
#pragma unroll
      for (int i = 0; i < SM; ++i)
        aq4[i] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_a, kin ? ao[i] + k0 * 4 : kOutside, 0, 0);
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < SM; ++i)
          sum[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(aq4[i][j]), __uint_as_float(wq[j]),
                                                        sum[i], 0, 0, 0);
    }
  }

  PT_STAMP(26)
  // ---- epilogue: LayerNorm fold, bias, activation, alpha, residual; C leaves as 16-byte row runs ----
  auto value = [&](int i, int e) {
    const int trow = i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lk;
    float v = sum[i][e];
    if (LN) v = s_stat[trow].y * (v - s_stat[trow].x * cs);
    v += bv;
    if (g.act == 1) v = fmaxf(v, 0.f);
    if (g.act == 2) v = v / (1.0f + __expf(-v));
    if (g.act == 3) v = 1.0f / (1.0f + __expf(-v));
    if (g.act == 4) v = tanhf(v);
    if (g.act == 5) v = 0.5f * v * (1.0f + erff(v * 0.70710678118654752f));
    return v * g.alpha;
  };
  // (the next launch's weight image was requested from inside the last chunk: `tail_requests`)
  if (vec) {
    // a wave's 32 x 32 block through its own 4.5 KB of LDS (the images are dead): rows leave as 16-byte
    // runs; the residual rows were requested behind the last chunk's fragments
    constexpr int TP = 36;
    float* tb = reinterpret_cast<float*>(s_a) + wv * (32 * TP);
    const int rr = ln >> 3, c4 = (ln & 7) * 4;
#pragma unroll
    for (int i = 0; i < SM; ++i) {
#pragma unroll
      for (int e = 0; e < 16; ++e) tb[((e & 3) + 8 * (e >> 2) + 4 * lk) * TP + li] = value(i, e);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (one wave: LDS serves it in order)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(tb + (rr + 8 * j) * TP + c4);
        u32x4 o;
        if constexpr (DMA) {
          // the residual tile sits in the staging area: [row][TN floats]
          const int trow = i * 32 + rr + 8 * j;
          const f32x4 rs = *reinterpret_cast<const f32x4*>(s_stage + trow * STG_ROW + (wv * 32 + c4) * 4);
          const int64_t row = m0 + trow;
          const int32_t qcol = n0 + wv * 32 + c4;
#pragma unroll
          for (int k = 0; k < 4; ++k) o[k] = __float_as_uint(t[k] + rs[k]);
          __builtin_amdgcn_raw_buffer_store_b128(
              o, rsrc_c, (qcol < g.N && row < g.M) ? (uint32_t)((row * g.ldc + qcol) * 4) : kOutside, 0, 0);
        } else {
#pragma unroll
          for (int k = 0; k < 4; ++k) o[k] = __float_as_uint(t[k] + __uint_as_float(rq[i][j][k]));
          __builtin_amdgcn_raw_buffer_store_b128(o, rsrc_c, vq[i][j], 0, 0);
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the block is read before it is rewritten
    }
  } else {
#pragma unroll
    for (int i = 0; i < SM; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int64_t row = m0 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lk;
        const uint32_t vo = (col_ok && row < g.M) ? (uint32_t)((row * g.ldc + col) * 4) : kOutside;
        const float res = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsrc_r, vo, 0, 0));
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(value(i, e) + res), rsrc_c, vo, 0, 0);
      }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the LDS-DMA requests land before the LDS is given back)
#ifdef APS_PANEL_TRACE
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  PT_STAMP(27)
#endif
}

// ---------------------------------------------------------------------------------------------------
// K-GROUP form (round 5; opt-in, see panel_form below for where it pays): the launches of BASELINE's 32
// utterances per GPU (M = 2016) are 252 ... 756
// tiles of 32 x 128 on 256 CUs -- ONE four-wave workgroup per CU whose life (profiles/r04_panel_trace_v2
// _staged.txt: 19.6 k cycles at N = K = 512) is a CHAIN: first rows 2.4 k, then per chunk split 2.0 k ->
// MFMA loop 4.3 k, epilogue 3.3 k; nothing overlaps anything because nothing else is resident.  Here a
// workgroup is KG x 4 waves: the same 32 x 128 tile, but K is cut into KG groups of KC columns and K group
// g (waves 4 g .. 4 g + 3) stages, splits and multiplies ITS chunk only:
//   * every row of the whole K extent is requested at entry (KG x 256 lanes x 2-4 16-byte loads in flight),
//     one exposed round trip instead of one per chunk;
//   * the split's VALU work and the MFMA loops are spread over 4 waves per SIMD, so one wave's fragment
//     waits are another's MFMAs (what four co-resident workgroups did where a launch had them);
//   * the KG partial tiles meet in LDS (the images are dead by then) and are summed in group order -- a
//     fixed order: results are bit-reproducible -- by ALL KG x 256 lanes, one 16-byte run of a row each:
//     LayerNorm fold, bias, activation, alpha, residual, one 16-byte store (the 4-wave form's epilogue
//     was 3.3 k cycles of one wave per 32 x 32 block going through its hand-over block row by row).
// Scale granule: one power of two per (row, K group) = per 128 / 256 elements of a row, the granule of
// the forms above; same detection rule, same fp32 recomputation (by K group 0) of a tile that does not fit.
// K <= KG x KC; rows of LDS: images KG x 2 x 32 x (2 KC + 16) bytes, then reused as KG x 32 x 136 floats.
template <int KG, int KC, bool LN, int WRING>
__global__ __launch_bounds__(KG * 256) void gemm_kgroup_kernel(PanelArgs g) {
  constexpr int RT = 32, TN = 128;
  constexpr int NT = KG * 256;
  constexpr int KS = KC / 32;                  // K steps per group
  constexpr int PB = KC * 2 + 16;              // row pitch of a plane in LDS (bytes)
  constexpr int PLANE = RT * PB, IMG = 2 * PLANE;
  constexpr int TPR = 8;                       // staging lanes per row (of a K group)
  constexpr int GPT = KC / 8 / TPR;            // 8-element groups per staging lane
  constexpr int PP = 136;                      // row pitch of a partial tile (floats)
  static_assert(KS % WRING == 0 && GPT >= 1, "ring / staging geometry");
  // All LDS is dynamic (the > 64 KB opt-in is sized to the byte: hipFuncSetAttribute refuses a dynamic
  // maximum that does not leave room for a kernel's static LDS): [images | partial tiles][exponents][row
  // sums][row statistics][wide flag][1 KB the prefetched lines are dropped into]
  constexpr int MAIN = KG * IMG > KG * RT * PP * 4 ? KG * IMG : KG * RT * PP * 4;
  extern __shared__ __attribute__((aligned(16))) unsigned char s_dyn[];
  int32_t (*const s_exp)[RT] = reinterpret_cast<int32_t (*)[RT]>(s_dyn + MAIN);
  float2 (*const s_part)[RT] = reinterpret_cast<float2 (*)[RT]>(s_dyn + MAIN + KG * RT * 4);
  float2* const s_stat = reinterpret_cast<float2*>(s_dyn + MAIN + KG * RT * 12);
  int32_t& s_wide = *reinterpret_cast<int32_t*>(s_dyn + MAIN + KG * RT * 12 + RT * 8);
  unsigned char* const s_pf = s_dyn + MAIN + KG * RT * 12 + RT * 8 + 16;
  const int tid = threadIdx.x, ln = tid & 63;
  const int kg = __builtin_amdgcn_readfirstlane(tid >> 8);        // K group of this wave
  const int wv = __builtin_amdgcn_readfirstlane((tid >> 6) & 3);  // its 32-column group
  const int t8 = tid & 255;

  const int32_t b = (int32_t)blockIdx.x;
  const int32_t lin = (b & 7) * g.per_xcd + (b >> 3);
  if (lin >= g.total || (b >> 3) >= g.per_xcd) return;
  const int32_t pnl = lin / g.tiles_n;
  const int32_t m0 = pnl * RT, n0 = (lin - pnl * g.tiles_n) * TN;

  const int64_t groups = ((g.N + 127) / 128) * 4;
  const int32_t wstep_bytes = (int32_t)(groups * 4096);
  auto rsrc_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(g.Wp), 0,
                                                  (uint32_t)(wstep_bytes * g.ksteps), 0x00020000);
  const int32_t* ew_tab = reinterpret_cast<const int32_t*>(static_cast<const unsigned char*>(g.Wp) +
                                                           (int64_t)wstep_bytes * g.ksteps);
  auto rsrc_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.A), 0,
                                                  (uint32_t)(g.M * g.lda * 4), 0x00020000);
  const int32_t vw = (n0 / 32 + wv) * 4096 + ln * 16;
  const int sr = t8 / TPR, q = t8 % TPR;
  const bool row_ok = m0 + sr < g.M;
  const int32_t Ki = (int32_t)g.K;
  const int32_t kbase = kg * KC + q * 8;  // first k of this lane
  const uint32_t va = (uint32_t)((int64_t)(m0 + sr) * g.lda * 4) + (uint32_t)kbase * 4u;

  // ---- every request of the launch's first round trip: the rows, then the first ring stages ----
  u32x4 ra[GPT][2];
#pragma unroll
  for (int j = 0; j < GPT; ++j) {
    const int32_t k = kbase + j * TPR * 8;
    const uint32_t off = va + (uint32_t)(j * TPR * 32);
    ra[j][0] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_a, (row_ok && k < Ki) ? off : kOutside, 0, 0);
    ra[j][1] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_a, (row_ok && k + 4 < Ki) ? off + 16u : kOutside, 0, 0);
  }
  u32x4 wb[WRING][2][2];
  const int32_t last_step = g.ksteps - 1;
  const int32_t gs0 = kg * KS;
  auto gload_w = [&](auto stage, int32_t gs) {
    constexpr int P = decltype(stage)::value;
    const int32_t soff = (gs < last_step ? gs : last_step) * wstep_bytes;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int p = 0; p < 2; ++p)
        wb[P][kk][p] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_w, vw, soff + (kk * 2 + p) * 1024, 0);
  };
  static_for<WRING - 1>([&](auto sc) { gload_w(sc, gs0 + decltype(sc)::value); });
  const int li = ln & 31, lk = ln >> 5;
  const int32_t col = n0 + wv * 32 + li;
  const int32_t ew = ew_tab[col];
  const int32_t ew_flag = ew_tab[groups * 32 + col];
  if (tid == 0) s_wide = 0;
  // the epilogue's operands of this lane: row er (+ 16 for KG = 2: two runs per lane), columns ec .. ec + 3
  constexpr int RUNS = 1024 / NT;
  const int er = tid >> 5, ec = n0 + (tid & 31) * 4;
  const uint32_t c_bytes = (uint32_t)(g.M * g.ldc * 4);
  auto rsrc_c = __builtin_amdgcn_make_buffer_rsrc(g.C, 0, c_bytes, 0x00020000);
  auto rsrc_r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.residual), 0,
                                                  g.residual ? c_bytes : 0u, 0x00020000);
  const bool vec = ((g.N | g.ldc) & 3) == 0 && ((reinterpret_cast<uintptr_t>(g.C) |
                   reinterpret_cast<uintptr_t>(g.residual)) & 15) == 0;
  uint32_t vq[RUNS];
  u32x4 rq[RUNS];
#pragma unroll
  for (int r = 0; r < RUNS; ++r) {
    const int64_t row = m0 + er + r * (NT / 32);
    vq[r] = (ec < g.N && row < g.M) ? (uint32_t)((row * g.ldc + ec) * 4) : kOutside;
    rq[r] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_r, vec ? vq[r] : kOutside, 0, 0);
  }
  float bv[4], cs[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const bool ok = ec + k < g.N;
    bv[k] = (g.bias && ok) ? g.bias[ec + k] : 0.f;
    cs[k] = (LN && ok) ? g.ln_cs[ec + k] : 0.f;
  }

  // ---- this K group's chunk: row maximum, scale, split, its LDS image ----
  unsigned char* const img = s_dyn + kg * IMG;
  float s1 = 0.f, s2 = 0.f;
  int32_t fitmin = 0;
  {
    float mx = 0.f;
#pragma unroll
    for (int j = 0; j < GPT; ++j)
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float v = __uint_as_float(ra[j][h][e]);
          mx = fmaxf(mx, fabsf(v));
          if (LN) {
            s1 += v;
            s2 = fmaf(v, v, s2);
          }
        }
    mx = group_max<TPR>(mx);
    const int32_t ex = scale_exponent(mx);
    unsigned char* const sdst = img + sr * PB + q * 16;
#pragma unroll
    for (int j = 0; j < GPT; ++j) {
      _Float16 hh[8], ll[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float x = __uint_as_float(ra[j][e >> 2][e & 3]);
        const float sc = ldexpf(x, ex);
        fitmin = min(fitmin, __builtin_amdgcn_frexp_expf(sc));
        hh[e] = (_Float16)sc;
        ll[e] = (_Float16)fmaf((float)hh[e], -kLowUp, ldexpf(x, ex + 11));
      }
      u32x4 h, l;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        h[e] = __builtin_bit_cast(uint32_t, f16x2{hh[2 * e], hh[2 * e + 1]});
        l[e] = __builtin_bit_cast(uint32_t, f16x2{ll[2 * e], ll[2 * e + 1]});
      }
      *reinterpret_cast<u32x4*>(sdst + j * TPR * 16) = h;
      *reinterpret_cast<u32x4*>(sdst + PLANE + j * TPR * 16) = l;
    }
    if (q == 0) s_exp[kg][sr] = ex;
    if (LN) {
#pragma unroll
      for (int o = 1; o < TPR; o <<= 1) {
        s1 += __shfl_xor(s1, o, 64);
        s2 += __shfl_xor(s2, o, 64);
      }
      if (q == 0) s_part[kg][sr] = make_float2(s1, s2);
    }
  }
  __syncthreads();
  if (LN && tid < RT) {
    float a1 = 0.f, a2 = 0.f;
#pragma unroll
    for (int k = 0; k < KG; ++k) {  // (group order: the same sums whatever the timing)
      a1 += s_part[k][tid].x;
      a2 += s_part[k][tid].y;
    }
    const float mean = a1 / (float)g.K;
    const float var = fmaxf(a2 / (float)g.K - mean * mean, 0.f);
    s_stat[tid] = make_float2(mean, 1.0f / sqrtf(var + g.ln_eps));
  }

  // ---- the group's K steps on its static image ----
  f32x16 acc, accx;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = accx[e] = 0.f;
  {
    const unsigned char* const frag = img + (ln & 31) * PB + (ln >> 5) * 16;
    const int32_t left = g.ksteps - gs0;
    const int kcount = left < KS ? (left < 0 ? 0 : left) : KS;
    // (`full`: the branch-free instantiation for a whole group -- under a run-time `ks < kcount` the compiler sinks
    // the fragment requests into the steps that consume them, see gemm_panel_kernel)
    auto kstep = [&](auto ksc, auto fullc) {
      constexpr int ks = decltype(ksc)::value;
      constexpr bool full = decltype(fullc)::value;
      constexpr int P = ks % WRING, PN = (ks + WRING - 1) % WRING;
      if (full || ks < kcount) {
        if constexpr (ks + WRING - 1 < KS) gload_w(std::integral_constant<int, PN>{}, gs0 + ks + WRING - 1);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
          const unsigned char* fa = frag + ks * 64 + kk * 32;
          const u32x4 ah = *reinterpret_cast<const u32x4*>(fa);
          const u32x4 al = *reinterpret_cast<const u32x4*>(fa + PLANE);
          accx = mfma_f16(ah, wb[P][kk][1], accx);  // h l
          acc = mfma_f16(ah, wb[P][kk][0], acc);    // h h
          accx = mfma_f16(al, wb[P][kk][0], accx);  // l h
        }
      }
    };
    if (kcount == KS)
      static_for<KS>([&](auto ksc) { kstep(ksc, std::true_type{}); });
    else
      static_for<KS>([&](auto ksc) { kstep(ksc, std::false_type{}); });
  }
  // fold: p = 2^-(ea[row, group] + ew[col]) (main + 2^-11 cross)
  float pv[16];
#pragma unroll
  for (int r4 = 0; r4 < 4; ++r4) {
    const i32x4 ea = *reinterpret_cast<const i32x4*>(&s_exp[kg][8 * r4 + 4 * lk]);
#pragma unroll
    for (int e = 0; e < 4; ++e)
      pv[r4 * 4 + e] = ldexpf(fmaf(accx[r4 * 4 + e], kLowDown, acc[r4 * 4 + e]), -(ea[e] + ew));
  }
  {
    bool wide = fitmin < -kFitBias;
    wide = __any(wide || ew_flag != 0);
    if (wide && ln == 0) s_wide = 1;
  }
  __syncthreads();  // every read of the images is behind us; s_wide is final
  if (s_wide != 0) {
    // the fp32 path: K group 0 recomputes the tile on v_mfma_f32_32x32x2_f32 (exact products), the
    // other groups contribute zeros
    if (tid == 0 && g.wide_count) atomicAdd(g.wide_count, 1);
#pragma unroll
    for (int e = 0; e < 16; ++e) pv[e] = 0.f;
    if (kg == 0) {
      f32x16 sum;
#pragma unroll
      for (int e = 0; e < 16; ++e) sum[e] = 0.f;
      auto rsrc_w32 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.W32), 0,
                                                        (uint32_t)(g.N * g.ldw * 4), 0x00020000);
      const int64_t wrow = col < g.N ? col : g.N - 1;
      const uint32_t wo = (uint32_t)(wrow * g.ldw * 4) + lk * 16;
      const int64_t arow = m0 + li < g.M ? m0 + li : g.M - 1;
      const uint32_t ao = (uint32_t)(arow * g.lda * 4) + lk * 16;
#pragma unroll 2
      for (int32_t k0 = 0; k0 < Ki; k0 += 8) {
        const bool kin = k0 + 4 * lk < Ki;
        const u32x4 wq = __builtin_amdgcn_raw_buffer_load_b128(rsrc_w32, kin ? wo + k0 * 4 : kOutside, 0, 0);
        const u32x4 aq = __builtin_amdgcn_raw_buffer_load_b128(rsrc_a, kin ? ao + k0 * 4 : kOutside, 0, 0);
#pragma unroll
        for (int j = 0; j < 4; ++j)
          sum = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(aq[j]), __uint_as_float(wq[j]), sum, 0, 0, 0);
      }
#pragma unroll
      for (int e = 0; e < 16; ++e) pv[e] = sum[e];
    }
  }
  // ---- the KG partial tiles meet in LDS: [group][row][PP] ----
  {
    float* const pt = reinterpret_cast<float*>(s_dyn) + kg * (RT * PP) + wv * 32 + li;
#pragma unroll
    for (int e = 0; e < 16; ++e) pt[((e & 3) + 8 * (e >> 2) + 4 * lk) * PP] = pv[e];
  }
  __syncthreads();

  // ---- epilogue: a 16-byte run of a row per lane ----
  auto finish = [&](float v, int row, int k) {
    if (LN) v = s_stat[row].y * (v - s_stat[row].x * cs[k]);
    v += bv[k];
    if (g.act == 1) v = fmaxf(v, 0.f);
    if (g.act == 2) v = v / (1.0f + __expf(-v));
    if (g.act == 3) v = 1.0f / (1.0f + __expf(-v));
    if (g.act == 4) v = tanhf(v);
    if (g.act == 5) v = 0.5f * v * (1.0f + erff(v * 0.70710678118654752f));
    return v * g.alpha;
  };
  // (the next launch's weight image: requested behind this lane's last load, see gemm_panel_kernel)
  if (g.pf != nullptr) {
    const int64_t share = (((g.pf_bytes + g.per_xcd - 1) / g.per_xcd) + (NT * 16 - 1)) & ~(int64_t)(NT * 16 - 1);
    const int rounds = (int)(share > 65536 ? 65536 / (NT * 16) : share / (NT * 16));
    auto rsrc_p = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(g.pf), 0, (uint32_t)g.pf_bytes, 0x00020000);
    const uint32_t base = (uint32_t)((int64_t)(b >> 3) * share) + (uint32_t)tid * 16u;
    for (int r = 0; r < rounds; ++r)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_p, (__attribute__((address_space(3))) void*)s_pf, 16,
                                               base + r * (NT * 16), 0, 0, 0);
  }
#pragma unroll
  for (int r = 0; r < RUNS; ++r) {
    const int row = er + r * (NT / 32);
    const float* const pr = reinterpret_cast<const float*>(s_dyn) + row * PP + (tid & 31) * 4;
    f32x4 t = *reinterpret_cast<const f32x4*>(pr);
#pragma unroll
    for (int k = 1; k < KG; ++k) {  // (group order)
      const f32x4 u = *reinterpret_cast<const f32x4*>(pr + k * (RT * PP));
#pragma unroll
      for (int c = 0; c < 4; ++c) t[c] += u[c];
    }
    if (vec) {
      u32x4 o;
#pragma unroll
      for (int c = 0; c < 4; ++c) o[c] = __float_as_uint(finish(t[c], row, c) + __uint_as_float(rq[r][c]));
      __builtin_amdgcn_raw_buffer_store_b128(o, rsrc_c, vq[r], 0, 0);
    } else {
      const int64_t grow = m0 + row;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const uint32_t vo = (ec + c < g.N && grow < g.M) ? (uint32_t)((grow * g.ldc + ec + c) * 4) : kOutside;
        const float res = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsrc_r, vo, 0, 0));
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(finish(t[c], row, c) + res), rsrc_c, vo, 0, 0);
      }
    }
  }
  if (g.pf != nullptr) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <int KG, int KC, bool LN, int WRING>
static int launch_kgroup(PanelArgs g, hipStream_t st) {
  constexpr int IMGS = KG * 2 * 32 * (KC * 2 + 16), PARTS = KG * 32 * 136 * 4;
  constexpr int LDS = (IMGS > PARTS ? IMGS : PARTS) + KG * 32 * 12 + 32 * 8 + 16 + 1024;
  const int64_t panels = (g.M + 31) / 32, tiles_n = (g.N + 127) / 128;
  const int64_t total = panels * tiles_n;
  if (total > 0x7fffff00) return APS_ERR_INVALID;
  g.tiles_n = (int32_t)tiles_n;
  g.total = (int32_t)total;
  g.per_xcd = (int32_t)((total + 7) / 8);
  static ApsPerDevice attr_set;
  if (LDS > 64 * 1024 &&
      !aps_lds_opt_in(attr_set, reinterpret_cast<const void*>(&gemm_kgroup_kernel<KG, KC, LN, WRING>), LDS))
    return APS_ERR_LAUNCH;
  hipLaunchKernelGGL((gemm_kgroup_kernel<KG, KC, LN, WRING>), dim3((unsigned)(g.per_xcd * 8)), dim3(KG * 256),
                     LDS, st, g);
  return aps_launch_status();
}

template <int RT, int TN, int KC, int WRING, bool LN, int MINW = 2, bool DMA = false>
static int launch_panel(PanelArgs g, hipStream_t st) {
  const int64_t panels = (g.M + RT - 1) / RT, tiles_n = (g.N + TN - 1) / TN;
  const int64_t total = panels * tiles_n;
  if (total > 0x7fffff00) return APS_ERR_INVALID;
  g.tiles_n = (int32_t)tiles_n;
  g.total = (int32_t)total;
  g.per_xcd = (int32_t)((total + 7) / 8);
#ifdef APS_PANEL_TRACE
  g.trace_slot0 = -1;
  if ((h_pt_n == 0 || h_pt_n == (unsigned)g.N) && (h_pt_k == 0 || h_pt_k == (unsigned)g.K) &&
      h_pt_next + (unsigned)total <= kPtSlots) {
    g.trace_slot0 = (int32_t)h_pt_next;
    h_pt_next += (unsigned)total;
  }
#endif
  hipLaunchKernelGGL((gemm_panel_kernel<RT, TN, KC, LN, WRING, MINW, DMA>), dim3((unsigned)(g.per_xcd * 8)),
                     dim3(TN * 2), 0, st, g);
  return aps_launch_status();
}

// The forms (all 128 columns wide, four waves side by side):
//   'a' 32 rows, chunks of 256, a four-stage fragment ring: 170 - 200 VGPRs, two workgroups per CU;
//   'c' 64 rows, chunks of 128, two stages: 218 - 250 VGPRs, two per CU;
//   'e' 32 rows, chunks of 128, two stages: 124 - 128 VGPRs, FOUR workgroups per CU (17 KB of LDS each) --
//       a launch whose workgroups spend most of their life waiting (first rows, fragments, stores) fills
//       the chip by occupancy or not at all (scripts/stream_overlap_probe.py: two streams of 'a' launches
//       at N >= 1024 take exactly twice as long as one).  The default.
// Measured (scripts/panel_gemm_probe.py, profiles/r04_panel_probe*.txt; us per launch alone on the chip):
//   M = 2016: N = 512 a 11.2 / c 16.0 / e 12.1 (K = 1024: 16.6 / 25.1 / 18.8), N = 1024 16.8 / 19.6 / 17.0,
//             N = 1536 25.8 / 24.3 / 21.2, N = 5000 85.3 / 67.0 / 62.6
//   M = 8064: N = 512 27.7 / 26.0 / 24.8, N = 1024 52.1 / 48.1 / 46.5, N = 1536 85.7 / 67.4 / 64.0
//   'e' is also ahead wherever a second stream's launches run beside it (two streams, N = 1024: 11.5 against
//   12.6 us per GEMM for 'a', four: 10.4 against 14.3); the 32-utterance joint step 12 570 against 11 870
//   utt/s with two batches in flight, 3.60 against 3.68 ms on one stream.  256-column tiles with eight waves
//   (one split of the rows feeding twice the matrix work) were built and measured no better at any shape
//   (N = 1024: 15.5 us; profiles/r04_panel_probe.txt, columns L32 / L34) and are not in the tree.
// (above ~512 tiles of 64 x 128 and N > 640 the planes-pass kernel aps_linear_fp16x2 is faster than any
// panel form: nn_ops.linear sends those launches there)
// APS_PANEL_FORM=a|c|e forces one (A/B runs); the `form` argument 1 | 2 | 3 likewise.
//   'f' (round 6) 'e' with the rows arriving by LDS-DMA a chunk ahead and a four-stage weight ring: <= 168 VGPRs,
//       three workgroups per CU (35 KB of LDS each); form 6, APS_PANEL_FORM=f.
//   'g' (round 6) 'f' on 64-row tiles (two row blocks per wave: half the weight bytes through the CU's vector-memory
//       path per product): <= 256 VGPRs, two workgroups per CU; form 7, APS_PANEL_FORM=g.
//   'k' (round 5) the K-GROUP form, 16 waves: 32 x 128 tiles, K cut into 4 groups of 128 (K <= 512) or 256
//       (K <= 1024) columns inside the workgroup; 'j' the same with 2 groups of 256 (K <= 512), 8 waves.
//       Measured (profiles/r05_rejected_experiments.txt (1); us per launch alone on the chip, e / k / j):
//       M = 2016: N = 512, K = 512 12.0 / 9.7 / 10.2, N = 512, K = 1024 18.5 / 14.1 / 13.9 -- the launches of 252
//       tiles, one per CU -- but N = 1024 17.1 / 19.5 / 19.7, N = 1536 21.0 / 26.7 / 28.7, M = 8064 worse
//       throughout: a 16-wave workgroup owns its CU, so a launch of more tiles than CUs runs them one after the
//       other, and another stream's launches cannot share the CU.  In the 32-utterance joint step 'k' for the
//       252-tile launches moves one stream from 3.70 to 3.59 ms and two batches in flight from 2.58 to 2.65 ms:
//       the caller (nn_ops) asks for it only while one stream is launching; forms 4 | 5, APS_PANEL_FORM=k|j.
static int panel_form(int64_t M, int64_t N, int64_t K, int32_t form) {
  int f = 0;
  if (form >= 1 && form <= 8) f = "acekjfgh"[form - 1];
  static const int forced = [] {
    const char* e = getenv("APS_PANEL_FORM");
    return (e && e[0] && strchr("acekjfgh", e[0])) ? (int)e[0] : 0;
  }();
  if (!f) f = forced;
  (void)M;
  (void)N;
  if (!f) f = 'e';
  if ((f == 'k' || f == 'j') && K > 1024) f = 'e';
  if (f == 'h') f = N >= 1024 ? 'g' : 'f';   // 64-row tiles where a launch has >= 252 of them
  if ((f == 'f' || f == 'g') && K % 128 != 0) f = (f == 'g' ? 'c' : 'e');   // (the DMA forms walk whole chunks)
  if (f == 'j' && K > 512) f = 'k';
  return f;
}
static int form_rows(int form) { return (form == 'c' || form == 'g') ? 64 : 32; }
static int form_cols(int) { return 128; }

}  // namespace panel
}  // namespace aps

using namespace aps;

#ifdef APS_PANEL_TRACE
// (trace build only; not in include/aps_amd.h) slots handed out so far; copies min(slots, capacity) x 32 stamps
extern "C" int64_t aps_debug_panel_trace(void* host, int64_t bytes) {
  if (hipDeviceSynchronize() != hipSuccess) return -1;
  if (bytes > (int64_t)sizeof(panel::g_panel_trace)) bytes = sizeof(panel::g_panel_trace);
  if (hipMemcpyFromSymbol(host, HIP_SYMBOL(panel::g_panel_trace), (size_t)bytes) != hipSuccess) return -1;
  return (int64_t)panel::h_pt_next;
}
// which launches get slots FROM NOW ON (issued or captured after this call): (N, K) filter, 0 = any; reset = hand the
// slots out from 0 again and zero the trace
extern "C" int aps_debug_panel_trace_filter(int32_t n, int32_t k, int32_t reset) {
  panel::h_pt_n = (unsigned)n, panel::h_pt_k = (unsigned)k;
  if (reset) {
    panel::h_pt_next = 0;
    void* p = nullptr;
    if (hipDeviceSynchronize() != hipSuccess || hipGetSymbolAddress(&p, HIP_SYMBOL(panel::g_panel_trace)) != hipSuccess ||
        hipMemset(p, 0, sizeof(panel::g_panel_trace)) != hipSuccess)
      return APS_ERR_LAUNCH;
  }
  return APS_OK;
}
extern "C" int aps_debug_panel_trace_zero() {
  void* p = nullptr;
  if (hipDeviceSynchronize() != hipSuccess || hipGetSymbolAddress(&p, HIP_SYMBOL(panel::g_panel_trace)) != hipSuccess ||
      hipMemset(p, 0, sizeof(panel::g_panel_trace)) != hipSuccess)
    return APS_ERR_LAUNCH;
  return APS_OK;
}
// recording on / off -- queued on `stream` (a non-blocking stream of the caller's, so that launches in flight on other
// streams keep running) and waited for
extern "C" int aps_debug_panel_trace_ctl(int32_t record, void* stream) {
  hipStream_t st = static_cast<hipStream_t>(stream);
  static unsigned ctl[4];
  ctl[3] = (unsigned)record;
  if (hipMemcpyToSymbolAsync(HIP_SYMBOL(panel::g_pt_ctl), ctl, sizeof(ctl), 0, hipMemcpyHostToDevice, st) != hipSuccess ||
      hipStreamSynchronize(st) != hipSuccess)
    return APS_ERR_LAUNCH;
  return APS_OK;
}
#endif

extern "C" int32_t aps_linear_panel_rows(int64_t M, int64_t N, int32_t form) {
  return panel::form_rows(panel::panel_form(M, N, 512, form));
}
extern "C" int32_t aps_linear_panel_cols(int64_t M, int64_t N, int32_t form) {
  return panel::form_cols(panel::panel_form(M, N, 512, form));
}

extern "C" int32_t aps_linear_panel_form(int64_t M, int64_t N, int64_t K, int32_t form) {
  const char* p = strchr("acekjfgh", panel::panel_form(M, N, K, form));
  return p ? (int32_t)(p - "acekjfgh") + 1 : 0;
}

extern "C" int aps_linear_panel(const float* A, const void* image, const float* W32, const float* bias,
                                const float* colsum, const float* residual, float* C,
                                int32_t* wide_count, int64_t M, int64_t N, int64_t K, int64_t lda,
                                int64_t ldw, int64_t ldc, int32_t act, float alpha, float eps,
                                const void* next_image, int64_t next_bytes, int32_t form, void* stream) {
  APS_CHECK_ARG(A && image && W32 && C && M > 0 && N > 0 && K > 0);
  APS_CHECK_ARG(K % 4 == 0 && lda >= K && ldc >= N && lda % 4 == 0 && ((uintptr_t)A & 15) == 0 &&
                ((uintptr_t)image & 15) == 0);
  APS_CHECK_ARG(ldw >= K && ldw % 4 == 0 && ((uintptr_t)W32 & 15) == 0);
  APS_CHECK_ARG(act >= 0 && act <= 5);
  if (M * lda * 4 >= ((int64_t)1 << 31) || N * ldw * 4 >= ((int64_t)1 << 31) ||
      M * ldc * 4 >= ((int64_t)1 << 31) || aps_linear_fp16x2_size(N, K) >= ((int64_t)1 << 31))
    return APS_ERR_UNSUPPORTED;
  hipStream_t st = static_cast<hipStream_t>(stream);
  panel::PanelArgs g{A, image, W32, bias, residual, C, wide_count, colsum, M, N, K, lda, ldw, ldc,
                     alpha, eps, act, 0, 0, 0, (int32_t)((K + 31) / 32),
                     (next_image && next_bytes > 0 && next_bytes < ((int64_t)1 << 31)) ? next_image : nullptr,
                     next_bytes};
#ifdef APS_PANEL_TRACE
  g.trace_slot0 = -1;
#endif
  switch (panel::panel_form(M, N, K, form)) {
    case 'k':
      if (K <= 512) {
        // (the whole group's weight fragments requested at entry: 124 - 128 VGPRs; APS_KGROUP_RING=2: a two-stage ring)
        static const bool ring2 = [] { const char* e = getenv("APS_KGROUP_RING"); return e && e[0] == '2'; }();
        if (ring2)
          return colsum ? panel::launch_kgroup<4, 128, true, 2>(g, st) : panel::launch_kgroup<4, 128, false, 2>(g, st);
        return colsum ? panel::launch_kgroup<4, 128, true, 4>(g, st) : panel::launch_kgroup<4, 128, false, 4>(g, st);
      }
      return colsum ? panel::launch_kgroup<4, 256, true, 2>(g, st) : panel::launch_kgroup<4, 256, false, 2>(g, st);
    case 'j': return colsum ? panel::launch_kgroup<2, 256, true, 2>(g, st) : panel::launch_kgroup<2, 256, false, 2>(g, st);
    case 'g': return colsum ? panel::launch_panel<64, 128, 128, 4, true, 2, true>(g, st) : panel::launch_panel<64, 128, 128, 4, false, 2, true>(g, st);
    case 'f': return colsum ? panel::launch_panel<32, 128, 128, 4, true, 3, true>(g, st) : panel::launch_panel<32, 128, 128, 4, false, 3, true>(g, st);
    case 'a': return colsum ? panel::launch_panel<32, 128, 256, 4, true>(g, st) : panel::launch_panel<32, 128, 256, 4, false>(g, st);
    case 'c': return colsum ? panel::launch_panel<64, 128, 128, 2, true>(g, st) : panel::launch_panel<64, 128, 128, 2, false>(g, st);
    default: return colsum ? panel::launch_panel<32, 128, 128, 2, true, 4>(g, st) : panel::launch_panel<32, 128, 128, 2, false, 4>(g, st);
  }
}
