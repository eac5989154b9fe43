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

}  // namespace

static int g_tail_tiles = 0;  // developer probe / tests: tiles per workgroup, 0 = heuristic
extern "C" int sopro_seanet_tail_set_tiles(int tiles) {
  g_tail_tiles = tiles > 0 ? tiles : 0;
  return 0;
}

extern "C" int sopro_seanet_tail_f32(const float* h, int64_t h_seg_stride, const float* w1, const float* b1, const float* w2,
                                      const float* b2, const float* wf, float bf, float* wav, int64_t wav_seg_stride, int32_t B,
                                      int32_t T, void* stream) {
  SOPRO_CHECK_ARG(h && w1 && b1 && w2 && b2 && wf && wav && B > 0 && T > 0, "bad pointers or sizes");
  SOPRO_CHECK_ARG(aligned16(h) && aligned16(w1) && aligned16(w2) && aligned16(wf) && (h_seg_stride & 3) == 0, "alignment");
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
