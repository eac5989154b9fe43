// Last transposed convolution of the SEANet decoder, ConvTranspose1d(128 -> 64, k = 8, s = 4) at the 6 kHz -> 24 kHz level
// (HF:modeling_mimi.py:931-961, the fourth `MimiConvTranspose1d`), as the contraction it is after pack_convtr1d:
//     out[t, r*64 + co] = bias[co] + sum_k A[t, k] * W[r*64 + co, k],   A[t, :] = [ x[t-1, 0..127] | x[t, 0..127] ]   (K = 256)
// i.e. every input row makes one 256-float output row = its four output samples x 64 channels.  3.07 M rows for 32 x 200
// frames: 1.57 GB in, 3.1 GB out, 403 GFLOP.  On the generic tile kernel (gemm_bf16s.hip, 128x128 tiles) this shape is bound
// by its operand stream: with K = 256 a tile runs 8 K-steps, and two thirds of what it pulls through L2 is the SAME 256 KB of
// split weights again (2.26 ms, 178 TFLOP/s fp32-equivalent).
//
// WEIGHT-STATIONARY (the scheme of seanet_res.hip): the whole weight matrix lives in registers as split-bf16 MFMA B fragments
// for the life of a workgroup - eight waves, wave w owns output columns 32w .. 32w+31 over all of K: 16 substeps x (hi, lo)
// = 128 registers - and the workgroup walks `tiles` consecutive 64-row tiles of one utterance.  Per tile only the 65 input
// rows (33 KB, contiguous) come in and 64 KB go out: the kernel runs at the rate of its matrix-core work and its output
// stream.  x is the ACTIVATED input (the producer - sopro_seanet_res128_f32 - applied the ELU): it is split once per element
// while it is staged (LDS row = [128 hi | 128 lo] bf16 + 16 B pad = 528 B, so A row t is LDS rows t, t+1 and the 16-lane
// ds_read_b128 fragment reads are conflict free).  Two LDS tiles: the next tile is requested before and split after the
// current tile's matrix-core work, one barrier per tile.  Accumulators are stored straight from the MFMA layout: lanes 0-31
// of a register hold 32 consecutive columns of one row = one whole 128-byte line.
// Arithmetic: PASSES = 3: lo*hi + hi*lo + hi*hi on v_mfma_f32_32x32x16_bf16, fp32 accumulate (the decoder's three-pass class,
// gemm_bf16s.hip NPL = 2); PASSES = 1: hi*hi only (the engine's bf16 mode).
#include "common.h"

namespace {

constexpr int UC = 128;            // input channels
constexpr int UN = 256;            // output columns per input row (4 samples x 64 channels)
constexpr int UTO = 64;            // rows per tile
constexpr int UHR = UTO + 1;       // staged rows
constexpr int UROW = 2 * UC * 2 + 16;  // 528
constexpr int UV = (UHR * (UC / 4) + 511) / 512;  // float4 per thread and tile: 5

typedef __bf16 ubf16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ ubf16x8 ufrag(const uint4& v) { return *reinterpret_cast<const ubf16x8*>(&v); }

__device__ __forceinline__ void usplit8(const float* __restrict__ p, uint4& hi, uint4& lo) {
  const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
  split2_bf16(a.x, a.y, hi.x, lo.x);
  split2_bf16(a.z, a.w, hi.y, lo.y);
  split2_bf16(b.x, b.y, hi.z, lo.z);
  split2_bf16(b.z, b.w, hi.w, lo.w);
}

// x: row p of utterance b at x + b * x_seg_stride + p * 128; A row t = rows t, t+1 (the caller points x at the row BEFORE the
// first sample: a zero pad row).  out: row t at out + b * out_seg_stride + t * 256.
template <int PASSES>
__global__ __launch_bounds__(512, 1) void seanet_up128_kernel(const float* __restrict__ x, int64_t x_seg_stride,
                                                              const float* __restrict__ w, const float* __restrict__ bias,
                                                              float* __restrict__ out, int64_t out_seg_stride, int T, int tiles) {
  extern __shared__ __attribute__((aligned(16))) unsigned char es_all[];  // [2][UHR * UROW]: 68.6 KB, dynamic (over the 64 KB static limit)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.y;
  const int frow = lane & 31, fg = lane >> 5;
  const float* xb = x + (int64_t)b * x_seg_stride;
  float* ob = out + (int64_t)b * out_seg_stride;

  // ---- weights as (hi, lo) B fragments, once per workgroup: n = 32 * wave + (lane & 31), k = 16 * s + 8 * (lane >> 5) .. + 7
  uint4 wh[16], wl[PASSES == 3 ? 16 : 1];
#pragma unroll
  for (int s = 0; s < 16; ++s) {
    uint4 lo;
    usplit8(w + (int64_t)(wave * 32 + frow) * (2 * UC) + s * 16 + fg * 8, wh[s], lo);
    if (PASSES == 3) wl[s] = lo;
    if ((s & 3) == 3) asm volatile("" ::: "memory");  // four substeps per memory round: all 32 loads at once need 128 more registers
  }
  const float bv = bias[wave * 32 + frow];

  // tile request: rows t0 .. t0 + 64, all loads issued back to back; rows past the end are redirected to row 0 (results of
  // those A rows are never stored), so the loads are branch-free
  float4 v[UV];
  auto request = [&](int t0) {
#pragma unroll
    for (int q = 0; q < UV; ++q) {
      const int idx = tid + q * 512;  // float4 index: 32 per row
      const int r = idx >> 5, c4 = idx & 31;
      const int p = t0 + r;
      const int pc = (r < UHR && p <= T) ? p : 0;
      v[q] = *reinterpret_cast<const float4*>(xb + (int64_t)pc * UC + c4 * 4);
    }
  };
  auto stage = [&](int buf) {
#pragma unroll
    for (int q = 0; q < UV; ++q) {
      const int idx = tid + q * 512;
      const int r = idx >> 5, c4 = idx & 31;
      if (r < UHR) {
        uint2 hi, lo;
        split2_bf16(v[q].x, v[q].y, hi.x, lo.x);
        split2_bf16(v[q].z, v[q].w, hi.y, lo.y);
        *reinterpret_cast<uint2*>(es_all + buf * (UHR * UROW) + r * UROW + c4 * 8) = hi;
        if (PASSES == 3) *reinterpret_cast<uint2*>(es_all + buf * (UHR * UROW) + r * UROW + 2 * UC + c4 * 8) = lo;
      }
    }
  };

  const int tile0 = (int)blockIdx.x * tiles;
  if (tile0 * UTO >= T) return;
  request(tile0 * UTO);
  stage(0);
  __syncthreads();
  for (int it = 0; it < tiles; ++it) {
    const int t0 = (tile0 + it) * UTO;
    if (t0 >= T) break;  // uniform over the workgroup
    const int cur = it & 1;
    const bool more = it + 1 < tiles && t0 + UTO < T;
    if (more) request(t0 + UTO);

    f32x16 acc[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mt][r] = 0.f;
    // K index = tap * 128 + channel; substep s covers tap s / 8 (LDS row + tap), channels 16 * (s % 8) .. + 15
    const unsigned char* a0 = es_all + cur * (UHR * UROW) + frow * UROW + fg * 16;
    // the fragment reads run DEPTH substeps ahead of the MFMAs that consume them; the scheduling barriers pin that order (left
    // to itself - or with sched_group_barrier - the compiler emits read -> wait -> MFMA per substep: every LDS latency exposed)
    constexpr int DEPTH = 2;
    uint4 ah[DEPTH][2], al[DEPTH][2];
    auto fread = [&](int s, int slot) {
      const unsigned char* p = a0 + (s >> 3) * UROW + (s & 7) * 32;
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        ah[slot][mt] = *reinterpret_cast<const uint4*>(p + mt * 32 * UROW);
        if (PASSES == 3) al[slot][mt] = *reinterpret_cast<const uint4*>(p + mt * 32 * UROW + 2 * UC);
      }
    };
#pragma unroll
    for (int s = 0; s < DEPTH; ++s) fread(s, s);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      uint4 ch[2], cl[2];
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) { ch[mt] = ah[s % DEPTH][mt]; if (PASSES == 3) cl[mt] = al[s % DEPTH][mt]; }
      if (PASSES == 3) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ufrag(cl[mt]), ufrag(wh[s]), acc[mt], 0, 0, 0);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ufrag(ch[mt]), ufrag(wl[s]), acc[mt], 0, 0, 0);
      }
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ufrag(ch[mt]), ufrag(wh[s]), acc[mt], 0, 0, 0);
      if (s + DEPTH < 16) fread(s + DEPTH, s % DEPTH);
      __builtin_amdgcn_sched_barrier(0);
    }
    // ---- store: register r of a lane is row 8 * (r / 4) + 4 * (lane >> 5) + r % 4 of its 32-row block, column lane & 31
    float* ot = ob + (int64_t)t0 * UN;  // wave-uniform base + a 32-bit lane offset: no 64-bit address per store
    const int col = wave * 32 + frow + 4 * fg * UN;
    if (t0 + UTO <= T) {
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) ot[col + (mt * 32 + 8 * (r >> 2) + (r & 3)) * UN] = acc[mt][r] + bv;
    } else {
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = mt * 32 + 8 * (r >> 2) + (r & 3);
          if (t0 + row + 4 * fg < T) ot[col + row * UN] = acc[mt][r] + bv;
        }
    }
    if (more) stage(cur ^ 1);
    __syncthreads();  // the next tile is staged; every wave is done reading this one (it is rewritten one trip later)
  }
}

// HB (round 4: the bf16 mode's activation flow): x is the ACTIVATED input as bf16 rows (256 bytes per row: half the bytes in),
// the result leaves as bf16 rows (512 bytes per input row: half the bytes out), one MFMA pass on the rounded operands.  Staging is
// a pure 16-byte copy into the piece-0 plane (LDS row = 128 bf16 + 16 B pad = 272 B: A row t is LDS rows t, t+1; the 16-lane
// ds_read_b128 fragment reads stay conflict free).  Stores: two lanes that hold neighbouring columns exchange one register of
// every pair, so that a lane writes 4 bytes (2 columns of one row) - 16 lanes = one 64-byte segment per row and instruction.
constexpr int UROW_H = UC * 2 + 16;               // 272
constexpr int UVH = (UHR * (UC / 8) + 511) / 512;  // 16-byte pieces per thread and tile: 3
__global__ __launch_bounds__(512, 1) void seanet_up128_hb_kernel(const unsigned short* __restrict__ x, int64_t x_seg_stride,
                                                                 const float* __restrict__ w, const float* __restrict__ bias,
                                                                 unsigned short* __restrict__ out, int64_t out_seg_stride, int T, int tiles) {
  extern __shared__ __attribute__((aligned(16))) unsigned char es_all[];  // [2][UHR * UROW_H]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.y;
  const int frow = lane & 31, fg = lane >> 5;
  const unsigned short* xb = x + (int64_t)b * x_seg_stride;
  unsigned short* ob = out + (int64_t)b * out_seg_stride;

  uint4 wh[16];
#pragma unroll
  for (int s = 0; s < 16; ++s) {
    uint4 lo;
    usplit8(w + (int64_t)(wave * 32 + frow) * (2 * UC) + s * 16 + fg * 8, wh[s], lo);
    if ((s & 3) == 3) asm volatile("" ::: "memory");
  }
  const float bv = bias[wave * 32 + frow];

  uint4 v[UVH];
  auto request = [&](int t0) {
#pragma unroll
    for (int q = 0; q < UVH; ++q) {
      const int idx = tid + q * 512;  // 16-byte piece index: 16 per row
      const int r = idx >> 4, c8 = idx & 15;
      const int p = t0 + r;
      const int pc = (r < UHR && p <= T) ? p : 0;
      v[q] = *reinterpret_cast<const uint4*>(xb + (int64_t)pc * UC + c8 * 8);
    }
  };
  auto stage = [&](int buf) {
#pragma unroll
    for (int q = 0; q < UVH; ++q) {
      const int idx = tid + q * 512;
      const int r = idx >> 4, c8 = idx & 15;
      if (r < UHR) *reinterpret_cast<uint4*>(es_all + buf * (UHR * UROW_H) + r * UROW_H + c8 * 16) = v[q];
    }
  };

  const int tile0 = (int)blockIdx.x * tiles;
  if (tile0 * UTO >= T) return;
  request(tile0 * UTO);
  stage(0);
  __syncthreads();
  for (int it = 0; it < tiles; ++it) {
    const int t0 = (tile0 + it) * UTO;
    if (t0 >= T) break;  // uniform over the workgroup
    const int cur = it & 1;
    const bool more = it + 1 < tiles && t0 + UTO < T;
    if (more) request(t0 + UTO);

    f32x16 acc[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mt][r] = 0.f;
    const unsigned char* a0 = es_all + cur * (UHR * UROW_H) + frow * UROW_H + fg * 16;
    constexpr int DEPTH = 2;
    uint4 ah[DEPTH][2];
    auto fread = [&](int s, int slot) {
      const unsigned char* p = a0 + (s >> 3) * UROW_H + (s & 7) * 32;
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) ah[slot][mt] = *reinterpret_cast<const uint4*>(p + mt * 32 * UROW_H);
    };
#pragma unroll
    for (int s = 0; s < DEPTH; ++s) fread(s, s);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      uint4 ch[2];
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) ch[mt] = ah[s % DEPTH][mt];
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ufrag(ch[mt]), ufrag(wh[s]), acc[mt], 0, 0, 0);
      if (s + DEPTH < 16) fread(s + DEPTH, s % DEPTH);
      __builtin_amdgcn_sched_barrier(0);
    }
    // ---- store: register r of a lane is row 8 * (r / 4) + 4 * (lane >> 5) + r % 4 of its 32-row block, column lane & 31.
    // Registers (r, r + 1) are rows (R, R + 1): the even lane of a column pair keeps row R and takes its neighbour's column for
    // it, the odd lane keeps row R + 1 - each then holds 2 consecutive columns of one row.
    unsigned short* ot = ob + (int64_t)t0 * UN;
    const bool odd = (lane & 1) != 0;
    const int colp = wave * 32 + (frow & ~1);  // first column of the pair
    const bool full = t0 + UTO <= T;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        const float mine0 = acc[mt][r] + bv, mine1 = acc[mt][r + 1] + bv;
        const float got = __shfl_xor(odd ? mine0 : mine1, 1, 64);  // even lane receives row R of column c+1; odd lane row R+1 of column c-1
        const float c0 = odd ? got : mine0, c1 = odd ? mine1 : got;  // (column pair c, c+1) of this lane's row
        const int row = mt * 32 + 8 * (r >> 2) + (r & 3) + (odd ? 1 : 0) + 4 * fg;
        unsigned pk, lo_;
        split2_bf16(c0, c1, pk, lo_);
        if (full || t0 + row < T) *reinterpret_cast<unsigned*>(ot + (int64_t)row * UN + colp) = pk;
      }
    if (more) stage(cur ^ 1);
    __syncthreads();
  }
}

int g_up_tiles = 0;

}  // namespace

extern "C" int sopro_seanet_up_set_tiles(int tiles) {
  g_up_tiles = tiles > 0 ? tiles : 0;
  return 0;
}

// bf16 rows in, bf16 rows out (the bf16 mode's activation flow; one pass).  x: [B][>= 1 + T][128] bf16 (activated, row 0 of a
// segment = the zero row), out rows of 256 bf16; strides count bf16 elements.
extern "C" int sopro_seanet_up128_bf16(const void* x, int64_t x_seg_stride, const float* w, const float* bias, void* out,
                                       int64_t out_seg_stride, int32_t B, int32_t T, void* stream) {
  SOPRO_CHECK_ARG(x && w && bias && out && B > 0 && T > 0, "bad pointers or sizes");
  SOPRO_CHECK_ARG(aligned16(x) && aligned16(w) && (reinterpret_cast<uintptr_t>(out) & 3u) == 0 && (x_seg_stride & 7) == 0 && (out_seg_stride & 1) == 0,
                  "x, w 16-byte aligned, x segment stride % 8 == 0 (16-byte row pieces), out 4-byte aligned");
  SOPRO_CHECK_ARG(B == 1 || (x_seg_stride >= (int64_t)(T + 1) * UC && out_seg_stride >= (int64_t)T * UN),
                  "segment strides: x holds T + 1 rows of 128 per utterance (one pad row in front), out T rows of 256");
  const int ntile = (T + UTO - 1) / UTO;
  int tiles = g_up_tiles ? g_up_tiles : (int)(((int64_t)ntile * B + 1023) / 1024);
  if (tiles < 1) tiles = 1;
  const dim3 grid((unsigned)((ntile + tiles - 1) / tiles), (unsigned)B);
  constexpr int lds = 2 * UHR * UROW_H;
  SOPRO_SET_MAX_LDS_ONCE(seanet_up128_hb_kernel, lds);
  hipLaunchKernelGGL(seanet_up128_hb_kernel, grid, dim3(512), lds, (hipStream_t)stream, reinterpret_cast<const unsigned short*>(x), x_seg_stride, w, bias,
                     reinterpret_cast<unsigned short*>(out), out_seg_stride, T, tiles);
  SOPRO_LAUNCH_CHECK();
}

extern "C" int sopro_seanet_up128_f32(const float* x, int64_t x_seg_stride, const float* w, const float* bias, float* out,
                                      int64_t out_seg_stride, int32_t B, int32_t T, int32_t passes, void* stream) {
  SOPRO_CHECK_ARG(x && w && bias && out && B > 0 && T > 0, "bad pointers or sizes");
  SOPRO_CHECK_ARG(passes == 1 || passes == 3, "passes must be 3 (three-pass split-bf16) or 1 (bf16 mode)");
  SOPRO_CHECK_ARG(aligned16(x) && aligned16(w) && aligned16(out) && (x_seg_stride & 3) == 0 && (out_seg_stride & 3) == 0,
                  "x, w, out must be 16-byte aligned with segment strides % 4 == 0");
  SOPRO_CHECK_ARG(B == 1 || (x_seg_stride >= (int64_t)(T + 1) * UC && out_seg_stride >= (int64_t)T * UN),
                  "segment strides: x holds T + 1 rows of 128 per utterance (one pad row in front), out T rows of 256");
  const int ntile = (T + UTO - 1) / UTO;
  // enough workgroups for ~4 per CU, as many tiles per workgroup as that leaves (the weight fragments are split once per workgroup)
  int tiles = g_up_tiles ? g_up_tiles : (int)(((int64_t)ntile * B + 1023) / 1024);
  if (tiles < 1) tiles = 1;
  const dim3 grid((unsigned)((ntile + tiles - 1) / tiles), (unsigned)B);
  constexpr int lds = 2 * UHR * UROW;
  SOPRO_SET_MAX_LDS_ONCE(seanet_up128_kernel<3>, lds);
  SOPRO_SET_MAX_LDS_ONCE(seanet_up128_kernel<1>, lds);
  if (passes == 3)
    hipLaunchKernelGGL(seanet_up128_kernel<3>, grid, dim3(512), lds, (hipStream_t)stream, x, x_seg_stride, w, bias, out, out_seg_stride, T, tiles);
  else
    hipLaunchKernelGGL(seanet_up128_kernel<1>, grid, dim3(512), lds, (hipStream_t)stream, x, x_seg_stride, w, bias, out, out_seg_stride, T, tiles);
  SOPRO_LAUNCH_CHECK();
}
