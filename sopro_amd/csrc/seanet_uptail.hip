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
// B fragments for the life of the workgroup - and accumulates tile k+1 on the matrix cores WHILE the tail of tile k runs: the
// accumulation only reads the staged x tile and registers, so its 16 K-substeps are dealt over the three phases of the tail
// (6 / 5 / 5) and the two waves that share a SIMD run the two halves of a phase in opposite order (one on the matrix cores
// while the other is in its vector work).  Per tile:
//     I1  conv k=3, 64 -> 32 on the split ELU(h) tile (LDS), wave w = samples 16w .. 16w+15 (v_mfma 16x16x32, weight fragments in
//         LDS as in seanet_tail16) -> split ELU(y) tile, rows grouped by sample phase            | substeps 0-5 of tile k+1
//     I2  conv k=1, 32 -> 64 in the PRODUCER's layout (wave w: its 32 samples of phase r x its 32 channels), so the skip operand
//         is the raw h the wave still holds in registers; ELU(h') -> fp32 tile over the dead h tile | substeps 6-10
//     I3  last conv k=3, 64 -> 1: lane = channel, 16 outputs per wave, transpose-reduction         | substeps 11-15
//     I4  h(k+1) = accumulators + bias: kept raw in registers, ELU + split -> the h tile; x(k+2) staged
// with one workgroup barrier behind each.  Both convolutions of the block are causal (left padding 2), so a tile needs two rows
// of ELU(h) and two rows of ELU(h') from its predecessor: they are carried through two 544-byte side buffers; a workgroup that
// does not start at the head of an utterance runs the tile in front of its range as a warm-up (stores masked), which leaves
// exactly these rows.  Arithmetic as in the two kernels: operands x = hi + lo in bf16, lo*hi + hi*lo + hi*hi with fp32
// accumulation (PASSES 3), or hi*hi only (PASSES 1, the engine's bf16 mode; XH: x arrives as bf16 rows).
#include <type_traits>

#include "common.h"

namespace {

constexpr int FC = 128;                 // input channels
constexpr int FTI = 32;                 // input rows per tile
constexpr int FTS = 4 * FTI;            // samples per tile
constexpr int FXR = FTI + 1;            // staged x rows
constexpr int FXROW = 2 * FC * 2 + 16;  // 528 B: [128 hi | 128 lo] bf16 + pad
constexpr int FHROW = 272;              // h tile row: [64 hi | 64 lo] bf16 + pad; later 64 fp32 + pad
constexpr int FHLD = FHROW / 4;         // 68 floats
constexpr int FHR = FTS + 2;            // h tile rows: two carried rows in front
constexpr int FYROW = 144;              // y tile row: [32 hi | 32 lo] bf16 + pad
constexpr int FW1B = 2 * 6 * 2;         // first convolution: [column tile][k-step][hi | lo] blocks of 64 lanes x 16 B (16x16x32 B operands)
constexpr int FW2B = 2 * 2 * 2;         // second: [channel half][substep][hi | lo] (32x32x16 B operands)
constexpr int FCARRY = 2 * FHROW;       // 544 B: two rows
constexpr int FXBUF = FXR * FXROW;        // one staged x tile; two of them (tile parity)
constexpr int FUSE_LDS = 2 * FXBUF + FHR * FHROW + FTS * FYROW + (FW1B + FW2B) * 1024 + 2 * FCARRY;
static_assert(FUSE_LDS <= 160 * 1024, "LDS");
static_assert((FXR * FXROW) % 16 == 0 && (FHR * FHROW) % 16 == 0 && (FTS * FYROW) % 16 == 0, "16-byte aligned LDS regions");

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
  unsigned char* hs = xs + 2 * FXBUF;
  unsigned char* ys = hs + FHR * FHROW;
  uint4* w1s = reinterpret_cast<uint4*>(ys + FTS * FYROW);
  uint4* w2s = w1s + FW1B * 64;
  uint4* chs = w2s + FW2B * 64;   // carried rows of the split ELU(h) tile (samples s0-2, s0-1 of the next tile)
  uint4* cps = chs + FCARRY / 16;  // carried rows of the ELU(h') tile

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.y;
  const int frow = lane & 31, fg = lane >> 5;   // 32x32x16 operands: row / column, k half
  const int col = lane & 15, kq = lane >> 4;    // 16x16x32 operands: row / column, k quarter
  const int ph = wave >> 1, cbase = 32 * (wave & 1);  // this wave's columns of the transposed convolution: sample phase, first channel
  const bool mfma_first = ((wave >> 2) & 1) != 0;     // waves w and w + 4 share a SIMD: opposite orders within a phase
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
  // their results are never stored).  Two x buffers by tile parity: while the matrix cores read tile k+1, tile k+2 arrives one
  // piece per phase - requested at the head of a phase, split and stored at the head of the next - so that a whole phase covers
  // the memory latency and only four registers carry it.
  constexpr int NV = XH ? 2 : 3;  // pieces per thread: 33 rows x 16 (bf16) or x 32 (fp32)
  auto request = [&](int t0, int q) -> uint4 {
    const int idx = tid + q * 512;
    const int rr = XH ? idx >> 4 : idx >> 5, cc = XH ? idx & 15 : idx & 31;
    const int p = t0 + rr;
    const int pc = (rr < FXR && p <= T) ? p : 0;
    if constexpr (XH) return *reinterpret_cast<const uint4*>(xb16 + (int64_t)pc * FC + cc * 8);
    else return *reinterpret_cast<const uint4*>(xb + (int64_t)pc * FC + cc * 4);
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
  if (tid < 2 * FCARRY / 16) chs[tid] = make_uint4(0u, 0u, 0u, 0u);  // both carries: the zero padding at the head of an utterance
  const float b1v[2] = {b1[col], b1[16 + col]};
  const float b2v = b2[cbase + frow];
  const float wl0 = wf[lane], wl1 = wf[64 + lane], wl2 = wf[128 + lane];  // last layer: tap j of this lane's channel

  f32x16 accP;   // the transposed convolution's accumulators of the tile in flight (32 input rows x this wave's 32 columns)
  float hraw[16];  // raw h of the tile whose tail is running, in the accumulator layout
#pragma unroll
  for (int q = 0; q < 16; ++q) accP[q] = 0.f;

  // ---- substeps [LO, HI) of the transposed convolution on the staged x tile: K index = tap * 128 + channel; substep s covers tap
  // s / 8 (LDS row + tap), channels 16 (s % 8) .. + 15.  Fragment reads run two substeps ahead of their MFMAs (seanet_up.hip).
  auto up_part = [&](const unsigned char* xcur, auto lo_, auto hi_) {
    constexpr int LO = decltype(lo_)::value, HI = decltype(hi_)::value;
    const unsigned char* a0 = xcur + frow * FXROW + fg * 16;
    uint4 ah[2], al[2];
    auto fread = [&](int s, int slot) {
      const unsigned char* p = a0 + (s >> 3) * FXROW + (s & 7) * 32;
      ah[slot] = *reinterpret_cast<const uint4*>(p);
      if (PASSES == 3) al[slot] = *reinterpret_cast<const uint4*>(p + 2 * FC);
    };
    fread(LO, 0);
    if (LO + 1 < HI) fread(LO + 1, 1);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = LO; s < HI; ++s) {
      const uint4 ch = ah[(s - LO) & 1];
      if (PASSES == 3) {
        const uint4 cl = al[(s - LO) & 1];
        accP = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ffrag(cl), ffrag(wh[s]), accP, 0, 0, 0);
        accP = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ffrag(ch), ffrag(wl[PASSES == 3 ? s : 0]), accP, 0, 0, 0);
      }
      accP = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ffrag(ch), ffrag(wh[s]), accP, 0, 0, 0);
      if (s + 2 < HI) fread(s + 2, (s - LO) & 1);
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  using I0 = std::integral_constant<int, 0>;
  using I6 = std::integral_constant<int, 6>;
  using I11 = std::integral_constant<int, 11>;
  using I16 = std::integral_constant<int, 16>;

  // ---- I4: the finished accumulators become the h tile: raw (+ bias) into registers for the skip operand, ELU + split into LDS.
  // Register q of a lane is input row t = 8 (q / 4) + 4 (lane >> 5) + q % 4 = sample 4 t + phase, channel cbase + (lane & 31);
  // registers (q, q + 1) are input rows (t, t + 1) = tile rows 4 apart.
  auto h_out = [&]() {
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      hraw[q] = accP[q] + bv;
      accP[q] = 0.f;
    }
#pragma unroll
    for (int q = 0; q < 16; q += 2) {
      const int t = 8 * (q >> 2) + 4 * fg + (q & 3);
      unsigned hi, lo;
      split2_bf16(eluf_(hraw[q]), eluf_(hraw[q + 1]), hi, lo);
      unsigned char* hp = hs + (2 + 4 * t + ph) * FHROW + (cbase + frow) * 2;
      *reinterpret_cast<unsigned short*>(hp) = (unsigned short)(hi & 0xffffu);
      *reinterpret_cast<unsigned short*>(hp + 4 * FHROW) = (unsigned short)(hi >> 16);
      if (PASSES == 3) {
        *reinterpret_cast<unsigned short*>(hp + 128) = (unsigned short)(lo & 0xffffu);
        *reinterpret_cast<unsigned short*>(hp + 4 * FHROW + 128) = (unsigned short)(lo >> 16);
      }
    }
    if (wave == 0 && lane < FCARRY / 16) reinterpret_cast<uint4*>(hs)[lane] = chs[lane];  // rows 0, 1: ELU(h) of the two samples in front
  };

  // ---- I1: conv k=3, 64 -> 32.  y of tile sample m reads tile rows m, m+1, m+2 (row j = sample m - 2 + j); this wave: m = 16 wave + (0 .. 15).
  // K index = tap * 64 + channel; k-step s covers tap s / 2, channels 32 (s % 2) .. + 31.  Output rows go to y tile row (m % 4) * 32 + m / 4.
  auto conv1 = [&]() {
    ff32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    const unsigned char* a = hs + (16 * wave + col) * FHROW + kq * 16;
#pragma unroll
    for (int s = 0; s < 6; ++s) {
      const unsigned char* p = a + (s >> 1) * FHROW + (s & 1) * 64;
      const uint4 ah = *reinterpret_cast<const uint4*>(p);
      uint4 al = ah;
      if (PASSES == 3) al = *reinterpret_cast<const uint4*>(p + 128);
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        const uint4 bh = w1s[((nt * 6 + s) * 2 + 0) * 64 + lane];
        if (PASSES == 3) {
          const uint4 bl = w1s[((nt * 6 + s) * 2 + 1) * 64 + lane];
          acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ffrag(al), ffrag(bh), acc[nt], 0, 0, 0);
          acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ffrag(ah), ffrag(bl), acc[nt], 0, 0, 0);
        }
        acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ffrag(ah), ffrag(bh), acc[nt], 0, 0, 0);
      }
    }
    // C: column 16 nt + col, rows m = 16 wave + 4 kq + i: phase i, row-in-phase 4 wave + kq
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int i = 0; i < 4; i += 2) {
        unsigned hi, lo;
        split2_bf16(eluf_(acc[nt][i] + b1v[nt]), eluf_(acc[nt][i + 1] + b1v[nt]), hi, lo);
        unsigned char* yp = ys + (i * 32 + 4 * wave + kq) * FYROW + (16 * nt + col) * 2;
        *reinterpret_cast<unsigned short*>(yp) = (unsigned short)(hi & 0xffffu);
        *reinterpret_cast<unsigned short*>(yp + 32 * FYROW) = (unsigned short)(hi >> 16);
        if (PASSES == 3) {
          *reinterpret_cast<unsigned short*>(yp + 64) = (unsigned short)(lo & 0xffffu);
          *reinterpret_cast<unsigned short*>(yp + 32 * FYROW + 64) = (unsigned short)(lo >> 16);
        }
      }
    // the last two rows of this tile's ELU(h) are the next tile's rows 0, 1
    if (wave == 7 && lane < FCARRY / 16) chs[lane] = reinterpret_cast<const uint4*>(hs + FTS * FHROW)[lane];
  };

  // ---- I2: conv k=1, 32 -> 64 in the producer's layout + skip operand (raw h in registers) -> ELU(h') as fp32 over the h tile
  auto conv2 = [&]() {
    f32x16 acc2;
#pragma unroll
    for (int q = 0; q < 16; ++q) acc2[q] = 0.f;
    const unsigned char* a = ys + (ph * 32 + frow) * FYROW + fg * 16;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const uint4 ah = *reinterpret_cast<const uint4*>(a + s * 32);
      const uint4 bh = w2s[(((wave & 1) * 2 + s) * 2 + 0) * 64 + lane];
      if (PASSES == 3) {
        const uint4 al = *reinterpret_cast<const uint4*>(a + 64 + s * 32);
        const uint4 bl = w2s[(((wave & 1) * 2 + s) * 2 + 1) * 64 + lane];
        acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ffrag(al), ffrag(bh), acc2, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ffrag(ah), ffrag(bl), acc2, 0, 0, 0);
      }
      acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ffrag(ah), ffrag(bh), acc2, 0, 0, 0);
    }
    float* hf = reinterpret_cast<float*>(hs);
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int t = 8 * (q >> 2) + 4 * fg + (q & 3);
      hf[(2 + 4 * t + ph) * FHLD + cbase + frow] = eluf_(hraw[q] + acc2[q] + b2v);
    }
    if (wave == 5 && lane < FCARRY / 16) reinterpret_cast<uint4*>(hs)[lane] = cps[lane];  // rows 0, 1: ELU(h') of the two samples in front
  };

  // ---- I3: last conv k=3, 64 -> 1 on the stored ELU(h'): output m reads tile rows m, m+1, m+2; lane = channel, this wave's 16
  // outputs, then a transpose-reduction over the 64 lanes (seanet_tail16_kernel)
  auto conv3 = [&](int s0, bool store) {
    const float* hf = reinterpret_cast<const float*>(hs);
    float xr[18], p16[16];
#pragma unroll
    for (int rr = 0; rr < 18; ++rr) xr[rr] = hf[(16 * wave + rr) * FHLD + lane];
#pragma unroll
    for (int i = 0; i < 16; ++i) p16[i] = fmaf(wl2, xr[i + 2], fmaf(wl1, xr[i + 1], wl0 * xr[i]));
    auto exchange = [&](auto n_, auto bit_) {
      constexpr int n = decltype(n_)::value, bit = decltype(bit_)::value;
      const bool up = (lane & bit) != 0;
#pragma unroll
      for (int k = 0; k < n; ++k) {
        const float keep = up ? p16[k + n] : p16[k], send = up ? p16[k] : p16[k + n];
        p16[k] = keep + __shfl_xor(send, bit, 64);
      }
    };
    exchange(std::integral_constant<int, 8>(), std::integral_constant<int, 32>());
    exchange(std::integral_constant<int, 4>(), std::integral_constant<int, 16>());
    exchange(std::integral_constant<int, 2>(), std::integral_constant<int, 8>());
    exchange(std::integral_constant<int, 1>(), std::integral_constant<int, 4>());
    float sum = p16[0];
    sum += __shfl_xor(sum, 1, 64);
    sum += __shfl_xor(sum, 2, 64);
    const int sidx = s0 + 16 * wave + (lane >> 2);
    if ((lane & 3) == 0 && store && sidx < S) wb[sidx] = sum + bf;
    // the last two rows of this tile's ELU(h') are the next tile's rows 0, 1 of that phase
    if (wave == 6 && lane < FCARRY / 16) cps[lane] = reinterpret_cast<const uint4*>(hs + FTS * FHROW)[lane];
  };

  // ---- prologue: h of the first tile
  __syncthreads();  // x tiles, weight fragments, zeroed carries
  up_part(xs + (kfirst & 1) * FXBUF, I0(), I16());
  h_out();
  __syncthreads();

  for (int k = kfirst; k <= klast; ++k) {
    const bool nxt = k < klast;     // tile k + 1 is accumulated while this tile's tail runs
    const bool store = k >= tile0;  // (the warm-up tile only leaves its carried rows)
    const unsigned char* xcur = xs + ((k + 1) & 1) * FXBUF;  // x of tile k + 1
    unsigned char* xnew = xs + (k & 1) * FXBUF;              // x of tile k: dead, becomes tile k + 2 (past the range: harmless rows)
    const int t2 = (k + 2) * FTI;
    uint4 vq = request(t2, 0);
    if (mfma_first) { if (nxt) up_part(xcur, I0(), I6()); conv1(); } else { conv1(); if (nxt) up_part(xcur, I0(), I6()); }
    __syncthreads();
    stage(xnew, 0, vq);
    vq = request(t2, 1);
    if (mfma_first) { if (nxt) up_part(xcur, I6(), I11()); conv2(); } else { conv2(); if (nxt) up_part(xcur, I6(), I11()); }
    __syncthreads();
    stage(xnew, 1, vq);
    if (NV == 3) vq = request(t2, 2);
    if (mfma_first) { if (nxt) up_part(xcur, I11(), I16()); conv3(k * FTS, store); } else { conv3(k * FTS, store); if (nxt) up_part(xcur, I11(), I16()); }
    if (!nxt) break;
    __syncthreads();
    if (NV == 3) stage(xnew, 2, vq);
    h_out();
    __syncthreads();
  }
}

int g_uptail_tiles = 0;

template <int PASSES, bool XH>
int launch_uptail(const void* x, int64_t x_seg_stride, const float* wu, const float* bu, const float* w1, const float* b1, const float* w2,
                  const float* b2, const float* wf, float bf, float* wav, int64_t wav_seg_stride, int32_t B, int32_t T, hipStream_t s) {
  const int ntile = (T + FTI - 1) / FTI;
  // ~4 workgroups per CU; a workgroup pays the weight fragments and one warm-up tile, so it should walk a few dozen tiles at least
  int tiles = g_uptail_tiles ? g_uptail_tiles : (int)(((int64_t)ntile * B + 1023) / 1024);
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
