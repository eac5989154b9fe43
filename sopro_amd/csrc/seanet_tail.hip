// Fused tail of the SEANet decoder at the 24 kHz / 64-channel level (HF:modeling_mimi.py:408-447, 957-960):
//     h' = h + Conv1d(32->64, k=1)(ELU(Conv1d(64->32, k=3)(ELU(h))))        (last MimiResnetBlock)
//     wav = Conv1d(64->1, k=3)(ELU(h'))                                       (last layer)
// h is the 64-channel activation written by the last transposed convolution: 3.1 GB for 32 x 200 frames.
// Unfused this level moves ~19 GB through HBM (h is read three times and rewritten once, the 32-channel
// intermediate makes a round trip); fused, h is read once (plus one L2-resident re-read for the skip operand)
// and only the waveform (1/64 of it) is written.
//
// One workgroup = `tiles` consecutive tiles of 126 output samples of one utterance (the weight fragments - 128 registers per
// lane, ~400 VALU instructions per wave to split - are prepared once and stay in registers for all of them; short inputs
// keep one tile per workgroup so that streaming chunks still spread over the chip).  Both convolutions run on the split-bf16 matrix-core path of
// the rest of the decoder (operands x = hi + lo in bf16, lo*hi + hi*lo + hi*hi on v_mfma_f32_32x32x16_bf16, fp32
// accumulate): ELU(h) is activated and split ONCE per element while the tile is staged into LDS (row = [64 hi | 64 lo]
// bf16 + 16 B pad = 272 B, so the k=3 window of a sample is three consecutive rows and 16-lane ds_read_b128 fragment
// reads are conflict free), the small weight matrices are split once per lane into register B fragments, the 32-channel
// intermediate goes through LDS in the same split form, and ELU(h') is stored once (fp32, over the dead h tile) for the
// last 64 -> 1 convolution, which reads every row three times.  The skip operand (an L2-resident re-read of h in the
// accumulator layout) and the last layer's weights are requested in the tile's memory round.  2 workgroups per CU (55 KB LDS).
#include <type_traits>

#include "common.h"

namespace {

constexpr int TO = 126;        // output samples per tile
constexpr int HR = TO + 4;     // h rows in LDS (samples s0-4 .. s0+TO-1)
constexpr int EROW = 272;      // bytes per row of the split h tile: 64 hi | 64 lo | pad
constexpr int YR = TO + 2;     // rows of the 32-channel intermediate (samples s0-2 .. s0+TO-1) == 128 == 4 MFMA row tiles
constexpr int YROW = 144;      // bytes per row of the split intermediate: 32 hi | 32 lo | pad
constexpr int HLD = 68;        // floats per row of the ELU(h') tile (written over the split h tile)

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ bf16x8 frag(const uint4& v) { return *reinterpret_cast<const bf16x8*>(&v); }

// 8 consecutive fp32 weights -> (hi, lo) fragments
__device__ __forceinline__ void split8(const float* __restrict__ p, uint4& hi, uint4& lo) {
  const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
  split2_bf16(a.x, a.y, hi.x, lo.x);
  split2_bf16(a.z, a.w, hi.y, lo.y);
  split2_bf16(b.x, b.y, hi.z, lo.z);
  split2_bf16(b.z, b.w, hi.w, lo.w);
}

template <bool LOOP>
__global__ __launch_bounds__(256, 2) void seanet_tail_kernel(const float* __restrict__ h, int64_t h_seg_stride,
                                                          const float* __restrict__ w1, const float* __restrict__ b1,
                                                          const float* __restrict__ w2, const float* __restrict__ b2,
                                                          const float* __restrict__ wf, float bf, float* __restrict__ wav,
                                                          int64_t wav_seg_stride, int T, int tiles_arg) {
  const int tiles = LOOP ? tiles_arg : 1;
  __shared__ __attribute__((aligned(16))) unsigned char es[HR * EROW];  // split ELU(h); later ELU(h') as fp32 [HR][HLD]
  __shared__ __attribute__((aligned(16))) unsigned char ys[YR * YROW];  // split ELU(intermediate)
  static_assert(HR * HLD * 4 <= HR * EROW, "the fp32 ELU(h') tile must fit over the split h tile");
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.y;
  const int frow = lane & 31, fg = lane >> 5;
  const float* hb = h + (int64_t)b * h_seg_stride;

  // ---- h tile: local row r holds sample s0-4+r, which is padded row s0-2+r of the buffer (2 zero rows in front).  The first
  // tile is requested before the weights (one memory round with them).
  float4 v[9];
  auto load_tile = [&](int s0) {
#pragma unroll
    for (int q = 0; q < 9; ++q) {
      const int idx = tid + q * 256;       // float4 index: 16 per row
      const int r = idx >> 4, c4 = idx & 15;
      const int p = s0 - 2 + r;            // padded row
      const int pc = (r < HR && p >= 0 && p < T + 2) ? p : 0;  // outside the segment: padded row 0, a zero row (branch-free loads)
      v[q] = *reinterpret_cast<const float4*>(hb + (int64_t)pc * 64 + c4 * 4);
    }
  };
  if (!LOOP) load_tile((int)blockIdx.x * TO);  // looped variant: requested per tile (the weight fragments fill the registers here)

  // ---- weights as (hi, lo) B fragments: n = lane&31, k = 16*substep + 8*(lane>>5) .. +7
  uint4 w1h[12], w1l[12];
#pragma unroll
  for (int s = 0; s < 12; ++s) split8(w1 + frow * 192 + s * 16 + fg * 8, w1h[s], w1l[s]);
  uint4 w2h[2][2], w2l[2][2];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int s = 0; s < 2; ++s) split8(w2 + (j * 32 + frow) * 32 + s * 16 + fg * 8, w2h[j][s], w2l[j][s]);
  const float b1v = b1[frow];
  const float b2v[2] = {b2[frow], b2[32 + frow]};
  __shared__ __attribute__((aligned(16))) float wfs[192];
  if (tid < 48) *reinterpret_cast<float4*>(wfs + tid * 4) = *reinterpret_cast<const float4*>(wf + tid * 4);

  for (int it = 0; it < tiles; ++it) {
  const int s0 = ((int)blockIdx.x * tiles + it) * TO;
  if (s0 >= T) break;  // uniform over the workgroup
  if (it > 0) __syncthreads();  // the previous tile's last phase is done with the LDS tiles
  if (LOOP) load_tile(s0);
  // ELU + split once per element
#pragma unroll
  for (int q = 0; q < 9; ++q) {
    const int idx = tid + q * 256;
    const int r = idx >> 4, c4 = idx & 15;
    if (r < HR) {
      uint2 hi, lo;
      split2_bf16(eluf_(v[q].x), eluf_(v[q].y), hi.x, lo.x);
      split2_bf16(eluf_(v[q].z), eluf_(v[q].w), hi.y, lo.y);
      *reinterpret_cast<uint2*>(es + r * EROW + c4 * 8) = hi;
      *reinterpret_cast<uint2*>(es + r * EROW + 128 + c4 * 8) = lo;
    }
  }
  __syncthreads();

  float skip[2][16];
  // ---- conv k=3, 64 -> 32: intermediate row m (sample s0-2+m) reads tile rows m, m+1, m+2; wave w owns rows 32w..32w+31.
  // K index = tap*64 + channel; substep s covers tap s/4, channels 16*(s%4) .. +15
  {
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const unsigned char* a = es + (wave * 32 + frow) * EROW + fg * 16;
    // the fragment reads run DEPTH substeps ahead of the MFMAs that consume them (left to itself the compiler emits
    // read -> wait -> MFMA per substep); the scheduling barriers pin that order
    constexpr int DEPTH = 4;
    uint4 ah[DEPTH], al[DEPTH];
#pragma unroll
    for (int s = 0; s < DEPTH; ++s) {
      const unsigned char* p = a + (s >> 2) * EROW + (s & 3) * 32;
      ah[s] = *reinterpret_cast<const uint4*>(p);
      al[s] = *reinterpret_cast<const uint4*>(p + 128);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < 12; ++s) {
      const uint4 ch = ah[s % DEPTH], cl = al[s % DEPTH];
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag(cl), frag(w1h[s]), acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag(ch), frag(w1l[s]), acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag(ch), frag(w1h[s]), acc, 0, 0, 0);
      if (s + DEPTH < 12) {
        const unsigned char* p = a + ((s + DEPTH) >> 2) * EROW + ((s + DEPTH) & 3) * 32;
        ah[s % DEPTH] = *reinterpret_cast<const uint4*>(p);
        al[s % DEPTH] = *reinterpret_cast<const uint4*>(p + 128);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    // skip operand of the residual block in the accumulator layout of the second convolution (row mr, column frow / 32+frow):
    // an L2-resident re-read of rows the tile request brought in, requested here - behind this phase's matrix-core work, so that
    // its 32 registers are free for the fragment reads that run ahead of the MFMAs - and consumed behind the next phase's
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int mr = wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * fg;
      const int p = s0 + mr;  // padded row of sample s0-2+mr
      const bool in = p >= 2 && p < T + 2;
      const float* hp = hb + (int64_t)(in ? p : 0) * 64 + frow;  // outside: padded row 0, a zero row
      skip[0][r] = hp[0];
      skip[1][r] = hp[32];
    }
    // D[r]: row (r&3) + 8*(r>>2) + 4*(lane>>5), column lane&31 -> ELU, split, one bf16 per plane
#pragma unroll
    for (int r = 0; r < 16; r += 1) {
      const int mr = wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * fg;
      unsigned hi, lo;
      split2_bf16(eluf_(acc[r] + b1v), 0.f, hi, lo);
      *reinterpret_cast<unsigned short*>(ys + mr * YROW + frow * 2) = (unsigned short)(hi & 0xffffu);
      *reinterpret_cast<unsigned short*>(ys + mr * YROW + 64 + frow * 2) = (unsigned short)(lo & 0xffffu);
    }
  }
  __syncthreads();  // every wave is done with the split h tile: its memory becomes the ELU(h') tile below

  // ---- conv k=1, 32 -> 64 on the intermediate, + skip operand (raw h, re-read: L2 resident), ELU(h') -> fp32 tile rows 2..129
  {
    f32x16 acc2[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc2[j][r] = 0.f;
    const unsigned char* a = ys + (wave * 32 + frow) * YROW + fg * 16;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const uint4 ah = *reinterpret_cast<const uint4*>(a + s * 32), al = *reinterpret_cast<const uint4*>(a + 64 + s * 32);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        acc2[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag(al), frag(w2h[j][s]), acc2[j], 0, 0, 0);
        acc2[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag(ah), frag(w2l[j][s]), acc2[j], 0, 0, 0);
        acc2[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag(ah), frag(w2h[j][s]), acc2[j], 0, 0, 0);
      }
    }
    float* hs = reinterpret_cast<float*>(es);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int mr = wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * fg;
      const bool real = (s0 - 2 + mr) >= 0;  // samples before the utterance start are the zero padding of the last conv
#pragma unroll
      for (int j = 0; j < 2; ++j)
        hs[(mr + 2) * HLD + j * 32 + frow] = real ? eluf_(skip[j][r] + acc2[j][r] + b2v[j]) : 0.f;
    }
  }
  __syncthreads();

  // ---- last conv k=3, 64 -> 1 on the stored ELU(h'): output i (sample s0+i) reads tile rows i+2 .. i+4; two threads per output
  {
    const float* hs = reinterpret_cast<const float*>(es);
    const int i = tid >> 1, half = tid & 1;  // half: channels 0..31 / 32..63
    float s = 0.f;
    if (i < TO) {
#pragma unroll
      for (int j = 0; j < 3; ++j) {
#pragma unroll
        for (int c4 = 0; c4 < 8; ++c4) {
          const float4 x = *reinterpret_cast<const float4*>(hs + (i + 2 + j) * HLD + half * 32 + c4 * 4);
          const float4 wv = *reinterpret_cast<const float4*>(wfs + j * 64 + half * 32 + c4 * 4);
          s += x.x * wv.x + x.y * wv.y + x.z * wv.z + x.w * wv.w;
        }
      }
    }
    s += __shfl_xor(s, 1, 64);
    if (half == 0 && i < TO && s0 + i < T) wav[(int64_t)b * wav_seg_stride + s0 + i] = s + bf;
  }
  }  // tiles of this workgroup
}

// ---------------------------------------------------------------------------------------------------------------------------
// Round 3: the same tail for long inputs as SIXTEEN AUTONOMOUS WAVES per CU.  Shader-clock stamps inside the kernel above (one
// workgroup of four waves, two per CU; tools/tail_timeline.py) showed ~9 cycles per instruction for a wave that has a SIMD to
// itself - its instruction streams are dependent chains (ELU, split, pack) - and a first sixteen-wave form with workgroup
// barriers lost 30 % of its cycles in them.  Here the weight fragments live in LDS (written once per workgroup, shared by its
// sixteen waves; <= 128 registers per lane, four waves per SIMD) and every wave owns whole tiles of 14 output samples: it
// stages the 18 rows it needs itself (two rows of halo per convolution), so no phase of a tile waits for another wave - only
// wavefront-scope fences between its own LDS writes and reads.  Products on v_mfma_f32_16x16x32_bf16, the same three passes
// (lo*hi + hi*lo + hi*hi), the same operand rounding as the kernel above.
constexpr int T3O = 14;            // output samples per wave tile
constexpr int H3R = T3O + 4;       // staged h rows (samples s0-4 .. s0+13)
constexpr int ES3 = H3R * EROW;    // 4896 B: split ELU(h); later ELU(h') as fp32 [16][HLD]
constexpr int YS3 = 16 * YROW;     // 2304 B: split ELU(intermediate), rows = samples s0-2 .. s0+13
constexpr int W1F = 2 * 6 * 2;     // first convolution: [column tile][k-step][hi | lo] fragment blocks of 64 lanes x 16 B
constexpr int W2F = 4 * 2;         // second: [column tile][hi | lo]
constexpr int TAIL3_LDS = (W1F + W2F) * 1024 + 16 * (ES3 + YS3);
static_assert(16 * HLD * 4 <= ES3, "the fp32 ELU(h') tile must fit over the split h tile");
typedef float f32x4v __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void tail_wave_sync() {  // a wave's own LDS writes -> its other lanes' reads (see ar_driver.hip wave_lds_sync)
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// HB (round 4: the bf16 mode's activation flow; PASSES == 1): h is bf16 rows (128 bytes per sample row: the 3.1 GB level is read at
// half the bytes); the tile request is three 16-byte pieces per lane, ELU is applied to the widened values and the result is
// rounded again while it is staged; the skip operand is the raw bf16 h widened to fp32.  Strides count bf16 elements then.
template <int PASSES, bool HB = false>
__global__ __launch_bounds__(1024, 1) void seanet_tail16_kernel(const float* __restrict__ h, int64_t h_seg_stride, const float* __restrict__ w1,
                                                                const float* __restrict__ b1, const float* __restrict__ w2,
                                                                const float* __restrict__ b2, const float* __restrict__ wf, float bf,
                                                                float* __restrict__ wav, int64_t wav_seg_stride, int T, int trips) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds3[];
  uint4* w1s = reinterpret_cast<uint4*>(lds3);
  uint4* w2s = w1s + W1F * 64;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  unsigned char* es = lds3 + (W1F + W2F) * 1024 + wave * (ES3 + YS3);
  unsigned char* ys = es + ES3;
  const int b = blockIdx.y;
  const int col = lane & 15, kq = lane >> 4;  // MFMA 16x16x32: operand row / column, k quarter (8 consecutive k); C: column, rows 4 kq + i
  static_assert(!HB || PASSES == 1, "bf16 rows are a one-pass (bf16 mode) input form");
  const float* hb = HB ? nullptr : h + (int64_t)b * h_seg_stride;
  const unsigned short* hb16 = HB ? reinterpret_cast<const unsigned short*>(h) + (int64_t)b * h_seg_stride : nullptr;

  // ---- weight fragments -> LDS, once per workgroup: waves 0-11 one (column tile, k-step) block pair of the first convolution
  // each (B operand: n = 16 nt + col, k = 32 s + 8 kq .. + 7), waves 12-15 one column tile of the second (n = 16 nt + col, k = 8 kq ..)
  {
    uint4 hi, lo;
    if (wave < 12) {
      const int nt = wave / 6, s = wave % 6;
      split8(w1 + (16 * nt + col) * 192 + s * 32 + kq * 8, hi, lo);
      w1s[((nt * 6 + s) * 2 + 0) * 64 + lane] = hi;
      w1s[((nt * 6 + s) * 2 + 1) * 64 + lane] = lo;
    } else {
      const int nt = wave - 12;
      split8(w2 + (16 * nt + col) * 32 + kq * 8, hi, lo);
      w2s[(nt * 2 + 0) * 64 + lane] = hi;
      w2s[(nt * 2 + 1) * 64 + lane] = lo;
    }
  }
  const float b1v[2] = {b1[col], b1[16 + col]};
  const float b2v[4] = {b2[col], b2[16 + col], b2[32 + col], b2[48 + col]};
  const float wl0 = wf[lane], wl1 = wf[64 + lane], wl2 = wf[128 + lane];  // last layer: tap j of this lane's channel

  // tile request of this wave: local row r holds sample s0-4+r = padded row s0-2+r (two zero rows in front of the segment);
  // rows outside the segment are redirected to padded row 0, a zero row (branch-free loads)
  float4 v[5];  // HB: three 16-byte pieces (8 bf16 each) per lane, carried as bit patterns
  auto request = [&](int s0) {
    if constexpr (HB) {
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        const int idx = lane + q * 64;  // 16-byte piece index: 8 per row
        const int r = idx >> 3, c8 = idx & 7;
        const int p = s0 - 2 + r;
        const int pc = (r < H3R && p >= 0 && p < T + 2) ? p : 0;
        const uint4 u = *reinterpret_cast<const uint4*>(hb16 + (int64_t)pc * 64 + c8 * 8);
        v[q] = make_float4(__uint_as_float(u.x), __uint_as_float(u.y), __uint_as_float(u.z), __uint_as_float(u.w));
      }
    } else {
#pragma unroll
      for (int q = 0; q < 5; ++q) {
        const int idx = lane + q * 64;  // float4 index: 16 per row
        const int r = idx >> 4, c4 = idx & 15;
        const int p = s0 - 2 + r;
        const int pc = (r < H3R && p >= 0 && p < T + 2) ? p : 0;
        v[q] = *reinterpret_cast<const float4*>(hb + (int64_t)pc * 64 + c4 * 4);
      }
    }
  };
  const int tile0 = (int)blockIdx.x * trips * 16 + wave;  // this wave's tiles: tile0, tile0 + 16, ...
  request(tile0 * T3O);
  __syncthreads();  // the weight fragments are in place (the only workgroup barrier)
  for (int it = 0; it < trips; ++it) {
    const int s0 = (tile0 + it * 16) * T3O;
    if (s0 >= T) break;  // uniform over the wave
    // ---- ELU + split once per element
    if constexpr (HB) {
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        const int idx = lane + q * 64;
        const int r = idx >> 3, c8 = idx & 7;
        if (r < H3R) {
          const unsigned u[4] = {__float_as_uint(v[q].x), __float_as_uint(v[q].y), __float_as_uint(v[q].z), __float_as_uint(v[q].w)};
          unsigned o[4], lo_;
#pragma unroll
          for (int e = 0; e < 4; ++e) split2_bf16(eluf_(__uint_as_float(u[e] << 16)), eluf_(__uint_as_float(u[e] & 0xffff0000u)), o[e], lo_);
          *reinterpret_cast<uint4*>(es + r * EROW + c8 * 16) = make_uint4(o[0], o[1], o[2], o[3]);
        }
      }
    } else {
#pragma unroll
      for (int q = 0; q < 5; ++q) {
        const int idx = lane + q * 64;
        const int r = idx >> 4, c4 = idx & 15;
        if (r < H3R) {
          uint2 hi, lo;
          split2_bf16(eluf_(v[q].x), eluf_(v[q].y), hi.x, lo.x);
          split2_bf16(eluf_(v[q].z), eluf_(v[q].w), hi.y, lo.y);
          *reinterpret_cast<uint2*>(es + r * EROW + c4 * 8) = hi;
          if (PASSES == 3) *reinterpret_cast<uint2*>(es + r * EROW + 128 + c4 * 8) = lo;
        }
      }
    }
    tail_wave_sync();
    // the staging registers are free: the next tile's rows travel while this one computes
    if (it + 1 < trips && s0 + 16 * T3O < T) request(s0 + 16 * T3O);
    // skip operand of the residual block in the accumulator layout of the second convolution (row 4 kq + i of the intermediate
    // = sample s0-2+row, columns 16 nt + col): an L2-resident re-read of rows the tile request brought in
    float skip[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int p = s0 + 4 * kq + i;  // padded row of sample s0-2+row
      const int64_t off = (int64_t)((p >= 2 && p < T + 2) ? p : 0) * 64 + col;
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        if constexpr (HB) skip[nt][i] = __uint_as_float((unsigned)hb16[off + 16 * nt] << 16);
        else skip[nt][i] = hb[off + 16 * nt];
      }
    }

    // ---- conv k=3, 64 -> 32: intermediate row m (sample s0-2+m) reads tile rows m, m+1, m+2.
    // K index = tap * 64 + channel; k-step s covers tap s / 2, channels 32 (s % 2) .. + 31
    {
      f32x4v acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
      const unsigned char* a = es + col * EROW + kq * 16;
#pragma unroll
      for (int s = 0; s < 6; ++s) {
        const unsigned char* p = a + (s >> 1) * EROW + (s & 1) * 64;
        const uint4 ah = *reinterpret_cast<const uint4*>(p);
        uint4 al = ah;
        if (PASSES == 3) al = *reinterpret_cast<const uint4*>(p + 128);
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
          const uint4 wh = w1s[((nt * 6 + s) * 2 + 0) * 64 + lane];
          if (PASSES == 3) {
            const uint4 wl = w1s[((nt * 6 + s) * 2 + 1) * 64 + lane];
            acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag(al), frag(wh), acc[nt], 0, 0, 0);
            acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag(ah), frag(wl), acc[nt], 0, 0, 0);
          }
          acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag(ah), frag(wh), acc[nt], 0, 0, 0);
        }
      }
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int i = 0; i < 4; i += 2) {  // two rows per conversion (the packed convert takes a pair)
          const int m = 4 * kq + i;
          unsigned hi, lo;
          split2_bf16(eluf_(acc[nt][i] + b1v[nt]), eluf_(acc[nt][i + 1] + b1v[nt]), hi, lo);
          unsigned char* yp = ys + m * YROW + (16 * nt + col) * 2;
          *reinterpret_cast<unsigned short*>(yp) = (unsigned short)(hi & 0xffffu);
          *reinterpret_cast<unsigned short*>(yp + YROW) = (unsigned short)(hi >> 16);
          if (PASSES == 3) {
            *reinterpret_cast<unsigned short*>(yp + 64) = (unsigned short)(lo & 0xffffu);
            *reinterpret_cast<unsigned short*>(yp + YROW + 64) = (unsigned short)(lo >> 16);
          }
        }
    }
    tail_wave_sync();  // (this wave is also done reading the split h tile: its memory becomes the ELU(h') tile below)

    // ---- conv k=1, 32 -> 64 on the intermediate, + skip operand, ELU(h') -> fp32 tile rows 0 .. 15 (row m = sample s0-2+m)
    {
      f32x4v acc2[4];
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) acc2[nt] = (f32x4v){0.f, 0.f, 0.f, 0.f};
      const unsigned char* a = ys + col * YROW + kq * 16;
      const uint4 ah = *reinterpret_cast<const uint4*>(a);
      uint4 al = ah;
      if (PASSES == 3) al = *reinterpret_cast<const uint4*>(a + 64);
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        const uint4 wh = w2s[(nt * 2 + 0) * 64 + lane];
        if (PASSES == 3) {
          const uint4 wl = w2s[(nt * 2 + 1) * 64 + lane];
          acc2[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag(al), frag(wh), acc2[nt], 0, 0, 0);
          acc2[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag(ah), frag(wl), acc2[nt], 0, 0, 0);
        }
        acc2[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag(ah), frag(wh), acc2[nt], 0, 0, 0);
      }
      float* hs = reinterpret_cast<float*>(es);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int m = 4 * kq + i;
        const bool real = (s0 - 2 + m) >= 0;  // samples before the utterance start are the zero padding of the last conv
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) hs[m * HLD + 16 * nt + col] = real ? eluf_(skip[nt][i] + acc2[nt][i] + b2v[nt]) : 0.f;
      }
    }
    tail_wave_sync();

    // ---- last conv k=3, 64 -> 1 on the stored ELU(h'): lane = channel; 16 column reads (the whole tile: 4 KB of LDS traffic
    // instead of 48 KB with one output per lane group), three FMAs per output, then a transpose-reduction over the 64 lanes:
    // every exchange halves the values a lane still carries, after four of them lane l holds output l >> 2 summed over 16 lanes
    {
      const float* hs = reinterpret_cast<const float*>(es);
      float x[16], p16[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) x[r] = hs[r * HLD + lane];
#pragma unroll
      for (int i = 0; i < T3O; ++i) p16[i] = fmaf(wl2, x[i + 2], fmaf(wl1, x[i + 1], wl0 * x[i]));
      p16[14] = p16[15] = 0.f;
      auto exchange = [&](auto n_, auto bit_) {  // n values kept per lane after this exchange; the lane bit that picks the half
        constexpr int n = decltype(n_)::value, bit = decltype(bit_)::value;  // (compile-time: the value arrays must stay in registers)
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
      const int i = lane >> 2;
      if ((lane & 3) == 0 && i < T3O && s0 + i < T) wav[(int64_t)b * wav_seg_stride + s0 + i] = sum + bf;
    }
    tail_wave_sync();  // the last phase is done with the LDS tiles before the next trip's staging
  }
}

}  // namespace

static int g_tail_tiles = 0;  // developer probe / tests: tiles per workgroup of the four-wave kernel, 0 = by size, < 0 = the sixteen-wave kernel
extern "C" int sopro_seanet_tail_set_tiles(int tiles) {
  g_tail_tiles = tiles;
  return 0;
}

extern "C" int sopro_seanet_tail_p_f32(const float* h, int64_t h_seg_stride, const float* w1, const float* b1, const float* w2,
                                        const float* b2, const float* wf, float bf, float* wav, int64_t wav_seg_stride, int32_t B,
                                        int32_t T, int32_t passes, void* stream) {
  SOPRO_CHECK_ARG(passes == 1 || passes == 3, "passes must be 3 (three-pass split-bf16) or 1 (bf16 mode)");
  SOPRO_CHECK_ARG(h && w1 && b1 && w2 && b2 && wf && wav && B > 0 && T > 0, "bad pointers or sizes");
  SOPRO_CHECK_ARG(aligned16(h) && aligned16(w1) && aligned16(w2) && aligned16(wf) && (h_seg_stride & 3) == 0, "alignment");
  // long inputs: sixteen autonomous waves per CU over LDS-resident weight fragments (a wave owns whole tiles of 14 samples)
  const int64_t all16 = (int64_t)((T + T3O - 1) / T3O) * B;
  if (g_tail_tiles < 0 || (g_tail_tiles == 0 && all16 >= 16 * 1024)) {
    const int n16 = (int)(((T + T3O - 1) / T3O + 15) / 16);  // sixteen-tile rounds per utterance
    int trips = all16 >= 64 * 4096 ? 32 : (all16 >= 16 * 4096 ? 8 : 2);  // rounds per workgroup (the fragments are made once per workgroup)
    if (g_tail_tiles < -1) trips = -g_tail_tiles;                         // (developer probe)
    SOPRO_SET_MAX_LDS_ONCE(seanet_tail16_kernel<3>, TAIL3_LDS);
    SOPRO_SET_MAX_LDS_ONCE(seanet_tail16_kernel<1>, TAIL3_LDS);
    if (passes == 3)
      hipLaunchKernelGGL(seanet_tail16_kernel<3>, dim3((n16 + trips - 1) / trips, B), dim3(1024), TAIL3_LDS, (hipStream_t)stream, h, h_seg_stride, w1, b1,
                         w2, b2, wf, bf, wav, wav_seg_stride, T, trips);
    else  // the engine's bf16 mode: hi * hi only (short inputs below keep the three-pass four-wave kernel)
      hipLaunchKernelGGL(seanet_tail16_kernel<1>, dim3((n16 + trips - 1) / trips, B), dim3(1024), TAIL3_LDS, (hipStream_t)stream, h, h_seg_stride, w1, b1,
                         w2, b2, wf, bf, wav, wav_seg_stride, T, trips);
    SOPRO_LAUNCH_CHECK();
  }
  const int ntile = (T + TO - 1) / TO;
  // several tiles per workgroup once there are enough of them to keep every CU supplied (>= 8 workgroups per CU after the grouping)
  const int tiles = g_tail_tiles > 0 ? g_tail_tiles : ((int64_t)ntile * B >= 16 * 2048 ? 16 : ((int64_t)ntile * B >= 8 * 1024 ? 8 : 1));
  dim3 grid((ntile + tiles - 1) / tiles, B);
  if (tiles > 1)
    hipLaunchKernelGGL(seanet_tail_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, h, h_seg_stride, w1, b1, w2, b2, wf, bf, wav,
                       wav_seg_stride, T, tiles);
  else
    hipLaunchKernelGGL(seanet_tail_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, h, h_seg_stride, w1, b1, w2, b2, wf, bf, wav,
                       wav_seg_stride, T, 1);
  SOPRO_LAUNCH_CHECK();
}

// The tail on bf16 rows (the bf16 mode's activation flow; one pass): h [B][2 + T][64] bf16 (two zero rows in front of each segment),
// h_seg_stride in bf16 elements; the sixteen-wave kernel at any size.
extern "C" int sopro_seanet_tail_bf16(const void* h, int64_t h_seg_stride, const float* w1, const float* b1, const float* w2,
                                       const float* b2, const float* wf, float bf, float* wav, int64_t wav_seg_stride, int32_t B,
                                       int32_t T, void* stream) {
  SOPRO_CHECK_ARG(h && w1 && b1 && w2 && b2 && wf && wav && B > 0 && T > 0, "bad pointers or sizes");
  SOPRO_CHECK_ARG(aligned16(h) && aligned16(w1) && aligned16(w2) && aligned16(wf) && (h_seg_stride & 7) == 0, "alignment (h rows in 16-byte pieces)");
  const int64_t all16 = (int64_t)((T + T3O - 1) / T3O) * B;
  const int n16 = (int)(((T + T3O - 1) / T3O + 15) / 16);
  int trips = all16 >= 64 * 4096 ? 32 : (all16 >= 16 * 4096 ? 8 : (all16 >= 1024 ? 2 : 1));
  if (g_tail_tiles < -1) trips = -g_tail_tiles;
  auto kern = seanet_tail16_kernel<1, true>;
  SOPRO_SET_MAX_LDS_ONCE(kern, TAIL3_LDS);
  hipLaunchKernelGGL(kern, dim3((n16 + trips - 1) / trips, B), dim3(1024), TAIL3_LDS, (hipStream_t)stream, reinterpret_cast<const float*>(h), h_seg_stride, w1, b1,
                     w2, b2, wf, bf, wav, wav_seg_stride, T, trips);
  SOPRO_LAUNCH_CHECK();
}

extern "C" int sopro_seanet_tail_f32(const float* h, int64_t h_seg_stride, const float* w1, const float* b1, const float* w2,
                                      const float* b2, const float* wf, float bf, float* wav, int64_t wav_seg_stride, int32_t B,
                                      int32_t T, void* stream) {
  return sopro_seanet_tail_p_f32(h, h_seg_stride, w1, b1, w2, b2, wf, bf, wav, wav_seg_stride, B, T, 3, stream);
}
