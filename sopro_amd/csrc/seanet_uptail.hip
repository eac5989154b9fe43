// Last level of the SEANet decoder in ONE kernel: the last transposed convolution and everything behind it
// (HF:modeling_mimi.py:931-961 - the fourth `MimiConvTranspose1d`, the last `MimiResnetBlock` 408-447, the last layer):
//     h[4t + r, co] = bu[co] + sum_k A[t, k] * Wu[r*64 + co, k],   A[t, :] = [ x[t-1, 0..127] | x[t, 0..127] ]      (seanet_up.hip)
//     h'  = h + Conv1d(32->64, k=1)(ELU(Conv1d(64->32, k=3)(ELU(h)))),   wav = Conv1d(64->1, k=3)(ELU(h'))            (seanet_tail.hip)
// As two kernels the 64-channel activation h makes a round trip through memory: 3.1 GB written + 3.1 GB read per 32 x 200 frames
// (half of that on bf16 rows).  Here it never leaves the CU: a workgroup walks `tiles` consecutive tiles of 32 input rows = 128
// output samples of one utterance.
//
// Eight waves, one workgroup per CU.  The transposed convolution is WEIGHT-STATIONARY as in seanet_up.hip: wave w owns output
// columns 32w .. 32w+31 (sample phase r = w / 2, channels 32 (w % 2) .. + 31) over all of K = 256 - 128 registers of split-bf16
// B fragments for the life of the workgroup.  Its accumulation only reads a staged x tile and registers, so it runs AHEAD of the
// rest and its MFMAs are dealt out one at a time between the vector instructions of the other stages: the matrix cores and the
// vector ALU overlap across the two waves of a SIMD, not within a wave (profiles/r04_mfma_valu_overlap.txt), so both waves should
// have both kinds of work to offer at any time (the first form of this kernel ran stage by stage: profiles/r04_experiments.md).
// Three phases per tile, one workgroup barrier behind each:
//     I1  conv k=3, 64 -> 32 on the split ELU(h) tile of tile k (LDS), wave w = samples 16w .. 16w+15 (v_mfma 16x16x32) -> split
//         ELU(y) tile, rows grouped by sample phase                                        | substeps 10-11 of tile k+1
//     I2  conv k=1, 32 -> 64 in the PRODUCER's layout (wave w: its 32 samples of phase r x its 32 channels), so the skip operand
//         is the raw h the wave still holds in registers; ELU(h') -> fp32 tile             | substeps 12-15 of tile k+1
//     I3  h(k+1) = accumulators + bias: kept raw in registers, ELU + split -> the h tile; last conv k=3, 64 -> 1 of tile k (lane =
//         channel, 16 outputs per wave, transpose-reduction)                               | substeps 0-9 of tile k+2
// x tiles are double-buffered by tile parity and arrive one 16-byte piece per thread and phase.  Both convolutions of the block
// are causal (left padding 2), so a tile needs two rows of ELU(h) and two rows of ELU(h') from its predecessor: they are carried
// through two 544-byte side buffers; a workgroup that does not start at the head of an utterance runs the tile in front of its
// range as a warm-up (stores masked), which leaves exactly these rows.  Arithmetic as in the two kernels: operands x = hi + lo in
// bf16, lo*hi + hi*lo + hi*hi with fp32 accumulation (PASSES 3), or hi*hi only (PASSES 1, the engine's bf16 mode; XH: x arrives
// as bf16 rows).  (What each stage costs was measured on a copy with stages left out: profiles/r04_uptail_ablation.txt.)
#include <type_traits>
#include <utility>

#include "common.h"

namespace {

constexpr int FC = 128;                 // input channels
constexpr int FTI = 32;                 // input rows per tile
constexpr int FTS = 4 * FTI;            // samples per tile
constexpr int FXR = FTI + 1;            // staged x rows
constexpr int FXROW = 2 * FC * 2 + 16;  // 528 B: [128 hi | 128 lo] bf16 + pad
constexpr int FHROW = 272;              // h tile row: [64 hi | 64 lo] bf16 + pad; h' tile row: 64 fp32 + pad
constexpr int FHLD = FHROW / 4;         // 68 floats
constexpr int FHR = FTS + 2;            // h / h' tile rows: two carried rows in front
constexpr int FYROW = 144;              // y tile row: [32 hi | 32 lo] bf16 + pad
constexpr int FW1B = 2 * 6 * 2;         // first convolution: [column tile][k-step][hi | lo] blocks of 64 lanes x 16 B (16x16x32 B operands)
constexpr int FW2B = 2 * 2 * 2;         // second: [channel half][substep][hi | lo] (32x32x16 B operands)
constexpr int FCARRY = 2 * FHROW;       // 544 B: two rows
constexpr int FXBUF = FXR * FXROW;      // one staged x tile; two of them (tile parity)
constexpr int FUSE_LDS = 2 * FXBUF + 2 * FHR * FHROW + FTS * FYROW + (FW1B + FW2B) * 1024 + 2 * FCARRY;
static_assert(FUSE_LDS <= 160 * 1024, "LDS");
static_assert(FXBUF % 16 == 0 && (FHR * FHROW) % 16 == 0 && (FTS * FYROW) % 16 == 0, "16-byte aligned LDS regions");

typedef __bf16 fbf16x8 __attribute__((ext_vector_type(8)));
typedef float ff32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ fbf16x8 ffrag(const uint4& v) { return *reinterpret_cast<const fbf16x8*>(&v); }

__device__ __forceinline__ void fsplit8(const float* __restrict__ p, uint4& hi, uint4& lo) {
  const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
  split2_bf16(a.x, a.y, hi.x, lo.x);
  split2_bf16(a.z, a.w, hi.y, lo.y);
  split2_bf16(b.x, b.y, hi.z, lo.z);
  split2_bf16(b.z, b.w, hi.w, lo.w);
}

// Two lanes that hold neighbouring columns (channels c, c + 1) each hold a packed pair of bf16 - row A in the low half, row B in the
// high half - of their column.  Afterwards the even lane holds row A of columns (c, c + 1) and the odd lane row B of the same two
// columns: one 4-byte LDS store per lane instead of two 2-byte ones.  A DPP move and a byte permute: no LDS traffic.
__device__ __forceinline__ unsigned pair_rows(unsigned mine, unsigned sel) {
  const unsigned got = (unsigned)__builtin_amdgcn_update_dpp(0, (int)mine, 0xB1 /* quad_perm [1, 0, 3, 2]: lane ^ 1 */, 0xF, 0xF, true);
  return __builtin_amdgcn_perm(got, mine, sel);  // bytes 4-7 = got, 0-3 = mine
}

// compile-time loop: f(std::integral_constant<int, 0>()) ... f(std::integral_constant<int, N - 1>())
template <int... I, class F>
__device__ __forceinline__ void sfor_impl(std::integer_sequence<int, I...>, F&& f) {
  (f(std::integral_constant<int, I>()), ...);
}
template <int N, class F>
__device__ __forceinline__ void sfor(F&& f) {
  sfor_impl(std::make_integer_sequence<int, N>(), f);
}
template <int V>
using IC = std::integral_constant<int, V>;

// x: row p of utterance b at x + b * x_seg_stride + p * 128 (fp32, or bf16 with XH); the caller points x at the row BEFORE the
// first input row (a zero row), as for sopro_seanet_up128_*.  wav: sample s of utterance b at wav + b * wav_seg_stride + s.
template <int PASSES, bool XH>
__global__ __launch_bounds__(512, 1) void seanet_uptail_kernel(const void* __restrict__ x_, int64_t x_seg_stride, const float* __restrict__ wu,
                                                               const float* __restrict__ bu, const float* __restrict__ w1,
                                                               const float* __restrict__ b1, const float* __restrict__ w2,
                                                               const float* __restrict__ b2, const float* __restrict__ wf, float bf,
                                                               float* __restrict__ wav, int64_t wav_seg_stride, int T, int tiles) {
  static_assert(PASSES == 1 || PASSES == 3, "passes");
  static_assert(!XH || PASSES == 1, "bf16 rows are a one-pass (bf16 mode) input form");
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  unsigned char* xs = lds;
  unsigned char* hs = xs + 2 * FXBUF;       // split ELU(h) of the tile whose tail runs (rows 0, 1: the two samples in front)
  unsigned char* hp = hs + FHR * FHROW;     // ELU(h') of that tile as fp32 (rows 0, 1 likewise)
  unsigned char* ys = hp + FHR * FHROW;
  uint4* w1s = reinterpret_cast<uint4*>(ys + FTS * FYROW);
  uint4* w2s = w1s + FW1B * 64;
  uint4* chs = w2s + FW2B * 64;    // carried rows of the split ELU(h) tile (samples s0-2, s0-1 of the next tile)
  uint4* cps = chs + FCARRY / 16;  // carried rows of the ELU(h') tile

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.y;
  const int frow = lane & 31, fg = lane >> 5;   // 32x32x16 operands: row / column, k half
  const int col = lane & 15, kq = lane >> 4;    // 16x16x32 operands: row / column, k quarter
  const int ph = wave >> 1, cbase = 32 * (wave & 1);  // this wave's columns of the transposed convolution: sample phase, first channel
  const bool odd = (lane & 1) != 0;
  const unsigned pair_sel = odd ? 0x03020706u : 0x05040100u;  // pair_rows: (got.hi16, mine.hi16) for the odd lane, (mine.lo16, got.lo16) for the even one
  const int ntile = (T + FTI - 1) / FTI;
  const int tile0 = (int)blockIdx.x * tiles;
  if (tile0 >= ntile) return;
  const int kfirst = tile0 > 0 ? tile0 - 1 : 0;  // warm-up tile: leaves the carried rows of the tile in front of the range
  const int klast = min(tile0 + tiles, ntile) - 1;
  const int S = 4 * T;
  const float* xb = XH ? nullptr : reinterpret_cast<const float*>(x_) + (int64_t)b * x_seg_stride;
  const unsigned short* xb16 = XH ? reinterpret_cast<const unsigned short*>(x_) + (int64_t)b * x_seg_stride : nullptr;
  float* wb = wav + (int64_t)b * wav_seg_stride;

  // ---- x tile request / staging in 16-byte pieces per thread (rows t0 .. t0 + 32; rows past the end are redirected to row 0:
  // their results are never stored)
  constexpr int NV = XH ? 2 : 3;  // pieces per thread: 33 rows x 16 (bf16) or x 32 (fp32)
  auto request = [&](int t0, int q) -> uint4 {
    const int idx = tid + q * 512;
    const int rr = XH ? idx >> 4 : idx >> 5, cc = XH ? idx & 15 : idx & 31;
    const int p = t0 + rr;
    const int pc = (rr < FXR && p <= T) ? p : 0;
    // (this kernel runs on long inputs only: x is read once, hundreds of MB - streamed with the non-temporal hint, common.h)
    if constexpr (XH) return bulk_load16(xb16 + (int64_t)pc * FC + cc * 8, true);
    else return bulk_load16(xb + (int64_t)pc * FC + cc * 4, true);
  };
  auto stage = [&](unsigned char* xbuf, int q, const uint4& val) {
    const int idx = tid + q * 512;
    const int rr = XH ? idx >> 4 : idx >> 5, cc = XH ? idx & 15 : idx & 31;
    if (rr < FXR) {
      if constexpr (XH) {
        *reinterpret_cast<uint4*>(xbuf + rr * FXROW + cc * 16) = val;
      } else {
        uint2 hi, lo;
        split2_bf16(__uint_as_float(val.x), __uint_as_float(val.y), hi.x, lo.x);
        split2_bf16(__uint_as_float(val.z), __uint_as_float(val.w), hi.y, lo.y);
        *reinterpret_cast<uint2*>(xbuf + rr * FXROW + cc * 8) = hi;
        if (PASSES == 3) *reinterpret_cast<uint2*>(xbuf + rr * FXROW + 2 * FC + cc * 8) = lo;
      }
    }
  };
  {  // the first two tiles of this workgroup (the second may lie past the range: harmless rows)
    uint4 v0[NV], v1[NV];
#pragma unroll
    for (int q = 0; q < NV; ++q) v0[q] = request(kfirst * FTI, q);
#pragma unroll
    for (int q = 0; q < NV; ++q) v1[q] = request((kfirst + 1) * FTI, q);
#pragma unroll
    for (int q = 0; q < NV; ++q) stage(xs + (kfirst & 1) * FXBUF, q, v0[q]);
#pragma unroll
    for (int q = 0; q < NV; ++q) stage(xs + ((kfirst + 1) & 1) * FXBUF, q, v1[q]);
  }

  // ---- weights of the transposed convolution as (hi, lo) B fragments in registers: n = 32 wave + (lane & 31), k = 16 s + 8 (lane >> 5) .. + 7
  uint4 wh[16], wl[PASSES == 3 ? 16 : 1];
#pragma unroll
  for (int s = 0; s < 16; ++s) {
    uint4 lo;
    fsplit8(wu + (int64_t)(wave * 32 + frow) * (2 * FC) + s * 16 + fg * 8, wh[s], lo);
    if (PASSES == 3) wl[s] = lo;
    if ((s & 3) == 3) asm volatile("" ::: "memory");  // four substeps per memory round
  }
  const float bv = bu[wave * 32 + frow];
  // ---- weight fragments of the residual block -> LDS, once per workgroup
  for (int c = wave; c < 12; c += 8) {  // first convolution, 16x16x32 B operand: n = 16 nt + col, k = 32 s + 8 kq .. + 7 (k = tap * 64 + channel)
    const int nt = c / 6, s = c % 6;
    uint4 hi, lo;
    fsplit8(w1 + (16 * nt + col) * 192 + s * 32 + kq * 8, hi, lo);
    w1s[((nt * 6 + s) * 2 + 0) * 64 + lane] = hi;
    w1s[((nt * 6 + s) * 2 + 1) * 64 + lane] = lo;
  }
  if (wave < 4) {  // second convolution, 32x32x16 B operand: n = 32 half + (lane & 31), k = 16 s + 8 (lane >> 5) .. + 7
    const int half = wave >> 1, s = wave & 1;
    uint4 hi, lo;
    fsplit8(w2 + (half * 32 + frow) * 32 + s * 16 + fg * 8, hi, lo);
    w2s[((half * 2 + s) * 2 + 0) * 64 + lane] = hi;
    w2s[((half * 2 + s) * 2 + 1) * 64 + lane] = lo;
  }
  // one-pass forms have the registers for the first convolution's B fragments too (48): no LDS traffic for them
  uint4 w1r[PASSES == 1 ? 2 : 1][PASSES == 1 ? 6 : 1];
  if (PASSES == 1) {
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int s = 0; s < 6; ++s) {
        uint4 lo;
        fsplit8(w1 + (16 * nt + col) * 192 + s * 32 + kq * 8, w1r[PASSES == 1 ? nt : 0][PASSES == 1 ? s : 0], lo);
      }
  }
  if (tid < 2 * FCARRY / 16) chs[tid] = make_uint4(0u, 0u, 0u, 0u);  // both carries: the zero padding at the head of an utterance
  const float b1v[2] = {b1[col], b1[16 + col]};
  const float b2v = b2[cbase + frow];
  const float wl0 = wf[lane], wl1 = wf[64 + lane], wl2 = wf[128 + lane];  // last layer: tap j of this lane's channel

  f32x16 accP;     // the transposed convolution's accumulators of the tile in flight (32 input rows x this wave's 32 columns)
  float hraw[16];  // raw h of the tile whose tail is running, in the accumulator layout
#pragma unroll
  for (int q = 0; q < 16; ++q) accP[q] = 0.f;

  // ---- the transposed convolution as a stream of MFMAs: number J of a tile is substep J / PASSES (K index = tap * 128 + channel:
  // tap s / 8 = LDS row + tap, channels 16 (s % 8) .. + 15), product J % PASSES of it (lo*hi, hi*lo, hi*hi as in seanet_up.hip).
  // The fragments of a substep are requested when the first MFMA of the substep before it is issued.
  constexpr int NP = 16 * PASSES;
  uint4 pah, pal, nah, nal;
  pal = nal = make_uint4(0u, 0u, 0u, 0u);
  auto p_read = [&](const unsigned char* xt, int s) {
    const unsigned char* p = xt + frow * FXROW + fg * 16 + (s >> 3) * FXROW + (s & 7) * 32;
    nah = *reinterpret_cast<const uint4*>(p);
    if (PASSES == 3) nal = *reinterpret_cast<const uint4*>(p + 2 * FC);
  };
  auto p_mfma = [&](const unsigned char* xt, auto j_) {
    constexpr int J = decltype(j_)::value, s = J / PASSES, piece = J % PASSES;
    if constexpr (piece == 0) {
      pah = nah;
      if (PASSES == 3) pal = nal;
      if constexpr (s + 1 < 16) p_read(xt, s + 1);
    }
    if constexpr (PASSES == 3) {
      if constexpr (piece == 0) accP = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ffrag(pal), ffrag(wh[s]), accP, 0, 0, 0);
      else if constexpr (piece == 1) accP = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ffrag(pah), ffrag(wl[s]), accP, 0, 0, 0);
      else accP = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ffrag(pah), ffrag(wh[s]), accP, 0, 0, 0);
    } else {
      accP = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ffrag(pah), ffrag(wh[s]), accP, 0, 0, 0);
    }
  };
  // MFMAs [J0, J1) dealt evenly over NS slots of vector work: slot I issues its share, then the caller's instructions follow
  auto p_slot = [&](const unsigned char* xt, auto j0_, auto j1_, auto i_, auto ns_) {
    constexpr int J0 = decltype(j0_)::value, J1 = decltype(j1_)::value, I = decltype(i_)::value, NS = decltype(ns_)::value;
    constexpr int lo = J0 + (J1 - J0) * I / NS, hi = J0 + (J1 - J0) * (I + 1) / NS;
    sfor<hi - lo>([&](auto d) { p_mfma(xt, IC<lo + decltype(d)::value>()); });
  };
  constexpr int JA = 10 * PASSES, JB = 12 * PASSES;  // I3: [0, JA) of the tile after next; I1: [JA, JB); I2: [JB, NP)

  // ---- h(k+1): the finished accumulators, raw (+ bias) into registers for the skip operand; ELU + split into the h tile in
  // NSH chunks (h_elu).  Register q of a lane is input row t = 8 (q / 4) + 4 (lane >> 5) + q % 4 = sample 4 t + phase, channel
  // cbase + (lane & 31); (q, q + 1) are tile rows j, j + 4 of this lane's channel: after pair_rows the even lane stores row j,
  // the odd lane row j + 4, channels (c, c + 1) each.
  auto h_raw = [&]() {
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      hraw[q] = accP[q] + bv;
      accP[q] = 0.f;
    }
  };
  unsigned char* const hp0 = hs + (2 + ph + (odd ? 4 : 0)) * FHROW + (cbase + (frow & ~1)) * 2;
  auto h_elu = [&](auto i_) {  // chunk I of 8: registers 2 I, 2 I + 1
    constexpr int q = 2 * decltype(i_)::value;
    const int t = 8 * (q >> 2) + 4 * fg + (q & 3);
    unsigned hi, lo;
    split2_bf16(eluf_(hraw[q]), eluf_(hraw[q + 1]), hi, lo);
    unsigned char* hq = hp0 + 4 * t * FHROW;
    *reinterpret_cast<unsigned*>(hq) = pair_rows(hi, pair_sel);
    if (PASSES == 3) *reinterpret_cast<unsigned*>(hq + 128) = pair_rows(lo, pair_sel);
  };

  // ---- I1: conv k=3, 64 -> 32.  y of tile sample m reads tile rows m, m+1, m+2 (row j = sample m - 2 + j); this wave: m = 16 wave + (0 .. 15).
  // K index = tap * 64 + channel; k-step s covers tap s / 2, channels 32 (s % 2) .. + 31.  Output rows go to y tile row (m % 4) * 32 + m / 4.
  // 10 slots for the transposed convolution's MFMAs [JA, JB): one per k-step, one per epilogue chunk.
  auto conv1 = [&](const unsigned char* xt) {
    ff32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    const unsigned char* a = hs + (16 * wave + col) * FHROW + kq * 16;
    sfor<6>([&](auto s_) {
      constexpr int s = decltype(s_)::value;
      {
        const unsigned char* p = a + (s >> 1) * FHROW + (s & 1) * 64;
        const uint4 ah = *reinterpret_cast<const uint4*>(p);
        if constexpr (PASSES == 3) {
          const uint4 al = *reinterpret_cast<const uint4*>(p + 128);
          const uint4 bh0 = w1s[((0 * 6 + s) * 2 + 0) * 64 + lane], bl0 = w1s[((0 * 6 + s) * 2 + 1) * 64 + lane];
          const uint4 bh1 = w1s[((1 * 6 + s) * 2 + 0) * 64 + lane], bl1 = w1s[((1 * 6 + s) * 2 + 1) * 64 + lane];
          p_slot(xt, IC<JA>(), IC<JB>(), s_, IC<10>());  // (under the fragment reads' latency)
          // the two column tiles' chains alternate: a dependent MFMA waits for its predecessor's result
          acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ffrag(al), ffrag(bh0), acc[0], 0, 0, 0);
          acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ffrag(al), ffrag(bh1), acc[1], 0, 0, 0);
          acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ffrag(ah), ffrag(bl0), acc[0], 0, 0, 0);
          acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ffrag(ah), ffrag(bl1), acc[1], 0, 0, 0);
          acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ffrag(ah), ffrag(bh0), acc[0], 0, 0, 0);
          acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ffrag(ah), ffrag(bh1), acc[1], 0, 0, 0);
        } else {
          p_slot(xt, IC<JA>(), IC<JB>(), s_, IC<10>());
          acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ffrag(ah), ffrag(w1r[0][PASSES == 1 ? s : 0]), acc[0], 0, 0, 0);
          acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ffrag(ah), ffrag(w1r[PASSES == 1 ? 1 : 0][PASSES == 1 ? s : 0]), acc[1], 0, 0, 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    });
    // C: column 16 nt + col, rows m = 16 wave + 4 kq + i: phase i, row-in-phase 4 wave + kq.  (i, i + 1) are y tile rows 32 apart:
    // after pair_rows the even lane stores phase i, the odd lane phase i + 1, columns (n, n + 1) each
    unsigned char* yp0 = ys + ((odd ? 32 : 0) + 4 * wave + kq) * FYROW + (col & ~1) * 2;
    sfor<4>([&](auto c_) {
      constexpr int c = decltype(c_)::value, nt = c >> 1, i = (c & 1) * 2;
      p_slot(xt, IC<JA>(), IC<JB>(), IC<6 + c>(), IC<10>());
      {
        unsigned hi, lo;
        split2_bf16(eluf_(acc[nt][i] + b1v[nt]), eluf_(acc[nt][i + 1] + b1v[nt]), hi, lo);
        unsigned char* yp = yp0 + i * 32 * FYROW + 32 * nt;
        *reinterpret_cast<unsigned*>(yp) = pair_rows(hi, pair_sel);
        if (PASSES == 3) *reinterpret_cast<unsigned*>(yp + 64) = pair_rows(lo, pair_sel);
      }
      __builtin_amdgcn_sched_barrier(0);
    });
  };

  // ---- I2: conv k=1, 32 -> 64 in the producer's layout + skip operand (raw h in registers) -> ELU(h') as fp32.
  // Slots for the transposed convolution's MFMAs [JB, NP): one per MFMA of this convolution, one per pair of output registers.
  constexpr int NC2 = 2 * PASSES;  // MFMAs of this convolution
  auto conv2 = [&](const unsigned char* xt) {
    f32x16 acc2;
#pragma unroll
    for (int q = 0; q < 16; ++q) acc2[q] = 0.f;
    const unsigned char* a = ys + (ph * 32 + frow) * FYROW + fg * 16;
    uint4 ah[2], al[2], bh[2], bl[2];
    {
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        ah[s] = *reinterpret_cast<const uint4*>(a + s * 32);
        bh[s] = w2s[(((wave & 1) * 2 + s) * 2 + 0) * 64 + lane];
        if (PASSES == 3) {
          al[s] = *reinterpret_cast<const uint4*>(a + 64 + s * 32);
          bl[s] = w2s[(((wave & 1) * 2 + s) * 2 + 1) * 64 + lane];
        }
      }
    }
    sfor<NC2>([&](auto m_) {
      constexpr int m = decltype(m_)::value, s = m / PASSES, piece = m % PASSES;
      p_slot(xt, IC<JB>(), IC<NP>(), m_, IC<NC2 + 8>());
      {
        if constexpr (PASSES == 3) {
          if constexpr (piece == 0) acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ffrag(al[s]), ffrag(bh[s]), acc2, 0, 0, 0);
          else if constexpr (piece == 1) acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ffrag(ah[s]), ffrag(bl[s]), acc2, 0, 0, 0);
          else acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ffrag(ah[s]), ffrag(bh[s]), acc2, 0, 0, 0);
        } else {
          acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ffrag(ah[s]), ffrag(bh[s]), acc2, 0, 0, 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    });
    float* hf = reinterpret_cast<float*>(hp);
    sfor<8>([&](auto c_) {
      constexpr int q0 = 2 * decltype(c_)::value;
      p_slot(xt, IC<JB>(), IC<NP>(), IC<NC2 + decltype(c_)::value>(), IC<NC2 + 8>());
      {
#pragma unroll
        for (int q = q0; q < q0 + 2; ++q) {
          const int t = 8 * (q >> 2) + 4 * fg + (q & 3);
          hf[(2 + 4 * t + ph) * FHLD + cbase + frow] = eluf_(hraw[q] + acc2[q] + b2v);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    });
  };

  // ---- I3: h(k+1) out of the accumulators (ELU + split -> h tile, 8 chunks) and the last conv k=3, 64 -> 1 of tile k on the stored
  // ELU(h'): output m reads tile rows m, m+1, m+2; lane = channel, this wave's 16 outputs, then a transpose-reduction over the 64
  // lanes (seanet_tail16_kernel).  WITH_NEXT: the MFMAs [0, JA) of the tile after next run between the chunks.
  auto conv3 = [&](const unsigned char* xt, int s0, bool store, auto with_next_) {
    constexpr bool NEXT = decltype(with_next_)::value != 0;
    constexpr int NS3 = 8 + 8 + 8 + 4 + 2 + 1;  // slots: h chunks, FMA pairs of outputs, exchange steps
    const float* hf = reinterpret_cast<const float*>(hp);
    float xr[18], p16[16];
#pragma unroll
    for (int rr = 0; rr < 18; ++rr) xr[rr] = hf[(16 * wave + rr) * FHLD + lane];
    if constexpr (NEXT) {
      p_read(xt, 0);
      sfor<8>([&](auto c_) {
        p_slot(xt, IC<0>(), IC<JA>(), c_, IC<NS3>());
        h_elu(c_);
        __builtin_amdgcn_sched_barrier(0);
      });
    }
    sfor<8>([&](auto c_) {
      constexpr int i0 = 2 * decltype(c_)::value;
      if constexpr (NEXT) p_slot(xt, IC<0>(), IC<JA>(), IC<8 + decltype(c_)::value>(), IC<NS3>());
#pragma unroll
      for (int i = i0; i < i0 + 2; ++i) p16[i] = fmaf(wl2, xr[i + 2], fmaf(wl1, xr[i + 1], wl0 * xr[i]));
      __builtin_amdgcn_sched_barrier(0);
    });
    // Transpose-reduction over the 64 lanes without LDS traffic.  An exchange step on lane bit B: lanes with the bit set keep the
    // upper half of their values, the others the lower half, and add what the partner lane (lane ^ B) gives away:
    //   B = 32, 16: v_permlane32_swap / v_permlane16_swap (gfx950) swap the upper lanes (odd 16-lane rows) of the register holding
    //               value k with the lower lanes (even rows) of the register holding value k + n: afterwards the two registers hold,
    //               in every lane, its own kept value and the partner's - their sum is the step;
    //   B = 8: a rotation by 8 within the 16-lane row (DPP row_ror:8);  B = 4: row_shr:4 for the lanes of banks 1, 3, row_shl:4 for banks 0, 2;
    //   the last two plain sums over lane bits 0, 1: DPP quad permutes.
    auto fu = [](float v) { return __float_as_uint(v); };
    sfor<8>([&](auto k_) {
      constexpr int k = decltype(k_)::value;
      if constexpr (NEXT) p_slot(xt, IC<0>(), IC<JA>(), IC<16 + k>(), IC<NS3>());
      const auto r = __builtin_amdgcn_permlane32_swap(fu(p16[k]), fu(p16[k + 8]), false, false);
      p16[k] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
      __builtin_amdgcn_sched_barrier(0);
    });
    sfor<4>([&](auto k_) {
      constexpr int k = decltype(k_)::value;
      if constexpr (NEXT) p_slot(xt, IC<0>(), IC<JA>(), IC<24 + k>(), IC<NS3>());
      const auto r = __builtin_amdgcn_permlane16_swap(fu(p16[k]), fu(p16[k + 4]), false, false);
      p16[k] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
      __builtin_amdgcn_sched_barrier(0);
    });
    {
      const bool up8 = (lane & 8) != 0, up4 = (lane & 4) != 0;
      sfor<2>([&](auto k_) {
        constexpr int k = decltype(k_)::value;
        if constexpr (NEXT) p_slot(xt, IC<0>(), IC<JA>(), IC<28 + k>(), IC<NS3>());
        const float keep = up8 ? p16[k + 2] : p16[k], send = up8 ? p16[k] : p16[k + 2];
        p16[k] = keep + __uint_as_float((unsigned)__builtin_amdgcn_update_dpp(0, (int)fu(send), 0x128 /* row_ror:8 */, 0xF, 0xF, true));
        __builtin_amdgcn_sched_barrier(0);
      });
      if constexpr (NEXT) p_slot(xt, IC<0>(), IC<JA>(), IC<30>(), IC<NS3>());
      const float keep = up4 ? p16[1] : p16[0], send = up4 ? p16[0] : p16[1];
      int got = __builtin_amdgcn_update_dpp(0, (int)fu(send), 0x114 /* row_shr:4: from lane - 4 */, 0xF, 0xA, false);
      got = __builtin_amdgcn_update_dpp(got, (int)fu(send), 0x104 /* row_shl:4: from lane + 4 */, 0xF, 0x5, false);
      p16[0] = keep + __uint_as_float((unsigned)got);
    }
    float sum = p16[0];
    sum += __uint_as_float((unsigned)__builtin_amdgcn_update_dpp(0, (int)fu(sum), 0xB1 /* quad_perm [1, 0, 3, 2] */, 0xF, 0xF, true));
    sum += __uint_as_float((unsigned)__builtin_amdgcn_update_dpp(0, (int)fu(sum), 0x4E /* quad_perm [2, 3, 0, 1] */, 0xF, 0xF, true));
    const int sidx = s0 + 16 * wave + (lane >> 2);
    if ((lane & 3) == 0 && store && sidx < S) wb[sidx] = sum + bf;
  };

  // ---- prologue: h of the first tile, the first ten substeps of the second
  __syncthreads();  // x tiles, weight fragments, zeroed carries
  {
    const unsigned char* x0 = xs + (kfirst & 1) * FXBUF;
    p_read(x0, 0);
    sfor<NP>([&](auto j_) { p_mfma(x0, j_); });
    h_raw();
    sfor<8>([&](auto c_) { h_elu(c_); });
    if (wave == 0 && lane < FCARRY / 16) reinterpret_cast<uint4*>(hs)[lane] = chs[lane];
    const unsigned char* x1 = xs + ((kfirst + 1) & 1) * FXBUF;
    p_read(x1, 0);
    sfor<JA>([&](auto j_) { p_mfma(x1, j_); });
  }
  uint4 vq = request((kfirst + 2) * FTI, 0);  // tile k + 2 arrives while tile k's tail runs: piece 0 requested a phase ahead
  __syncthreads();

  for (int k = kfirst; k <= klast; ++k) {
    const bool nxt = k < klast;
    const bool store = k >= tile0;  // (the warm-up tile only leaves its carried rows)
    const unsigned char* x1 = xs + ((k + 1) & 1) * FXBUF;  // x of tile k + 1: substeps 10-15 in I1, I2
    unsigned char* x2 = xs + (k & 1) * FXBUF;              // x of tile k (dead since the last barrier but one) -> tile k + 2 (past the range: harmless rows)
    const int t2 = (k + 2) * FTI;
    // I1
    stage(x2, 0, vq);
    vq = request(t2, 1);
    uint4 vq2 = make_uint4(0u, 0u, 0u, 0u);
    if (NV == 3) vq2 = request(t2, 2);
    conv1(x1);
    if (wave == 7 && lane < FCARRY / 16) chs[lane] = reinterpret_cast<const uint4*>(hs + FTS * FHROW)[lane];  // this tile's last two rows of ELU(h)
    __syncthreads();
    // I2
    stage(x2, 1, vq);
    if (NV == 3) stage(x2, 2, vq2);
    conv2(x1);
    if (wave == 5 && lane < FCARRY / 16) reinterpret_cast<uint4*>(hp)[lane] = cps[lane];  // rows 0, 1: ELU(h') of the two samples in front
    __syncthreads();
    // I3
    if (nxt) {
      vq = request((k + 3) * FTI, 0);
      h_raw();
      conv3(x2, k * FTS, store, IC<1>());
      if (wave == 0 && lane < FCARRY / 16) reinterpret_cast<uint4*>(hs)[lane] = chs[lane];  // rows 0, 1 of the next h tile
    } else {
      conv3(x2, k * FTS, store, IC<0>());
    }
    if (wave == 6 && lane < FCARRY / 16) cps[lane] = reinterpret_cast<const uint4*>(hp + FTS * FHROW)[lane];  // this tile's last two rows of ELU(h')
    if (!nxt) break;
    __syncthreads();
  }
}

int g_uptail_tiles = 0;

template <int PASSES, bool XH>
int launch_uptail(const void* x, int64_t x_seg_stride, const float* wu, const float* bu, const float* w1, const float* b1, const float* w2,
                  const float* b2, const float* wf, float bf, float* wav, int64_t wav_seg_stride, int32_t B, int32_t T, hipStream_t s) {
  const int ntile = (T + FTI - 1) / FTI;
  // ~8-10 workgroups per CU (a short last round of workgroups: 1024 of them on the 192-CU partition measured 5.65 ms, 2048 5.25 ms);
  // a workgroup pays the weight fragments and one warm-up tile, so it should walk a few dozen tiles at least
  int tiles = g_uptail_tiles ? g_uptail_tiles : (int)(((int64_t)ntile * B + 2047) / 2048);
  if (!g_uptail_tiles && tiles < 24) tiles = ntile < 24 ? ntile : 24;
  if (tiles < 1) tiles = 1;
  auto kern = seanet_uptail_kernel<PASSES, XH>;
  SOPRO_SET_MAX_LDS_ONCE(kern, FUSE_LDS);
  hipLaunchKernelGGL(kern, dim3((unsigned)((ntile + tiles - 1) / tiles), (unsigned)B), dim3(512), FUSE_LDS, s, x, x_seg_stride, wu, bu, w1, b1, w2, b2, wf,
                     bf, wav, wav_seg_stride, T, tiles);
  SOPRO_LAUNCH_CHECK();
}

}  // namespace

extern "C" int sopro_seanet_uptail_set_tiles(int tiles) {
  g_uptail_tiles = tiles > 0 ? tiles : 0;
  return 0;
}

extern "C" int sopro_seanet_uptail_f32(const float* x, int64_t x_seg_stride, const float* wu, const float* bu, const float* w1, const float* b1,
                                       const float* w2, const float* b2, const float* wf, float bf, float* wav, int64_t wav_seg_stride, int32_t B,
                                       int32_t T, int32_t passes, void* stream) {
  SOPRO_CHECK_ARG(x && wu && bu && w1 && b1 && w2 && b2 && wf && wav && B > 0 && T > 0, "bad pointers or sizes");
  SOPRO_CHECK_ARG(passes == 1 || passes == 3, "passes must be 3 (three-pass split-bf16) or 1 (bf16 mode)");
  SOPRO_CHECK_ARG(aligned16(x) && aligned16(wu) && aligned16(w1) && aligned16(w2) && (x_seg_stride & 3) == 0, "x, wu, w1, w2 16-byte aligned, x segment stride % 4 == 0");
  SOPRO_CHECK_ARG(B == 1 || (x_seg_stride >= (int64_t)(T + 1) * FC && wav_seg_stride >= (int64_t)4 * T),
                  "segment strides: x holds T + 1 rows of 128 per utterance (one zero row in front), wav 4 T samples");
  if (passes == 3) return launch_uptail<3, false>(x, x_seg_stride, wu, bu, w1, b1, w2, b2, wf, bf, wav, wav_seg_stride, B, T, (hipStream_t)stream);
  return launch_uptail<1, false>(x, x_seg_stride, wu, bu, w1, b1, w2, b2, wf, bf, wav, wav_seg_stride, B, T, (hipStream_t)stream);
}

extern "C" int sopro_seanet_uptail_bf16(const void* x, int64_t x_seg_stride, const float* wu, const float* bu, const float* w1, const float* b1,
                                        const float* w2, const float* b2, const float* wf, float bf, float* wav, int64_t wav_seg_stride, int32_t B,
                                        int32_t T, void* stream) {
  SOPRO_CHECK_ARG(x && wu && bu && w1 && b1 && w2 && b2 && wf && wav && B > 0 && T > 0, "bad pointers or sizes");
  SOPRO_CHECK_ARG(aligned16(x) && aligned16(wu) && aligned16(w1) && aligned16(w2) && (x_seg_stride & 7) == 0, "x, wu, w1, w2 16-byte aligned, x segment stride % 8 == 0");
  SOPRO_CHECK_ARG(B == 1 || (x_seg_stride >= (int64_t)(T + 1) * FC && wav_seg_stride >= (int64_t)4 * T),
                  "segment strides: x holds T + 1 rows of 128 per utterance (one zero row in front), wav 4 T samples");
  return launch_uptail<1, true>(x, x_seg_stride, wu, bu, w1, b1, w2, b2, wf, bf, wav, wav_seg_stride, B, T, (hipStream_t)stream);
}
