// Fused tail of the SEANet decoder at the 24 kHz / 64-channel level (HF:modeling_mimi.py:408-447, 957-960):
//     h' = h + Conv1d(32->64, k=1)(ELU(Conv1d(64->32, k=3)(ELU(h))))        (last MimiResnetBlock)
//     wav = Conv1d(64->1, k=3)(ELU(h'))                                       (last layer)
// h is the 64-channel activation written by the last transposed convolution: 3.1 GB for 32 x 200 frames.
// Unfused this level moves ~19 GB through HBM (h is read three times and rewritten once, the 32-channel
// intermediate makes a round trip); fused, h is read once and only the waveform (1/64 of it) is written.
//
// One workgroup = 126 output samples of one utterance.  The h tile (130 rows incl. the 4-sample halo the two
// k=3 convolutions need) sits in LDS with rows padded to 68 floats, so the k=3 window of a sample is three
// consecutive LDS rows and 16-lane ds_read_b128 fragment reads are conflict free.  Both convolutions run on
// v_mfma_f32_32x32x2_f32 (exact fp32) with the small weight matrices held in registers as B fragments
// (24 + 8 float4 per lane); ELU is applied on the operand read; the residual and the last 64->1 convolution
// read the same LDS tile.  3 workgroups fit per CU (53 KB of LDS each).
#include "common.h"

namespace {

constexpr int TO = 126;        // output samples per workgroup
constexpr int HR = TO + 4;     // h rows in LDS (samples s0-4 .. s0+TO-1)
constexpr int HLD = 68;        // padded h row
constexpr int YR = TO + 2;     // rows of the 32-channel intermediate (samples s0-2 .. s0+TO-1) == 128 == 4 MFMA row tiles
constexpr int YLD = 36;

__global__ __launch_bounds__(256) void seanet_tail_kernel(const float* __restrict__ h, int64_t h_seg_stride,
                                                          const float* __restrict__ w1, const float* __restrict__ b1,
                                                          const float* __restrict__ w2, const float* __restrict__ b2,
                                                          const float* __restrict__ wf, float bf, float* __restrict__ wav,
                                                          int64_t wav_seg_stride, int T) {
  __shared__ float hs[HR * HLD];
  __shared__ float ys[YR * YLD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.y;
  const int s0 = blockIdx.x * TO;
  const int frow = lane & 31, fk = (lane >> 5) * 4;

  // ---- weights as B fragments (n = lane&31, the lane half picks k = 8c+4h .. +3), one memory round with the tile
  float4 w1f[24];
#pragma unroll
  for (int c = 0; c < 24; ++c) w1f[c] = *reinterpret_cast<const float4*>(w1 + frow * 192 + c * 8 + fk);
  float4 w2f[2][4];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int c = 0; c < 4; ++c) w2f[j][c] = *reinterpret_cast<const float4*>(w2 + (j * 32 + frow) * 32 + c * 8 + fk);
  const float b1v = b1[frow];
  const float b2v[2] = {b2[frow], b2[32 + frow]};

  // ---- h tile: local row r holds sample s0-4+r, which is padded row s0-2+r of the buffer (2 zero rows in front)
  const float* hb = h + (int64_t)b * h_seg_stride;
  {
    float4 v[9];
#pragma unroll
    for (int q = 0; q < 9; ++q) {
      const int idx = tid + q * 256;       // float4 index: 16 per row
      const int r = idx >> 4, c4 = idx & 15;
      const int p = s0 - 2 + r;            // padded row
      v[q] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (r < HR && p >= 0 && p < T + 2) v[q] = *reinterpret_cast<const float4*>(hb + (int64_t)p * 64 + c4 * 4);
    }
#pragma unroll
    for (int q = 0; q < 9; ++q) {
      const int idx = tid + q * 256;
      const int r = idx >> 4, c4 = idx & 15;
      if (r < HR) *reinterpret_cast<float4*>(hs + r * HLD + c4 * 4) = v[q];
    }
  }
  __syncthreads();

  // ---- conv k=3, 64 -> 32 on ELU(h): intermediate row m (sample s0-2+m) reads LDS rows m, m+1, m+2; wave w owns rows 32w..32w+31
  {
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const int m = wave * 32 + frow;
#pragma unroll
    for (int c = 0; c < 24; ++c) {
      const int j = c >> 3, cc = (c & 7) * 8 + fk;  // tap, channel
      float4 a4 = *reinterpret_cast<const float4*>(hs + (m + j) * HLD + cc);
      a4.x = eluf_(a4.x); a4.y = eluf_(a4.y); a4.z = eluf_(a4.z); a4.w = eluf_(a4.w);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, w1f[c].x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, w1f[c].y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, w1f[c].z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, w1f[c].w, acc, 0, 0, 0);
    }
    // D[r]: row (r&3) + 8*(r>>2) + 4*(lane>>5), column lane&31
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int mr = wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      ys[mr * YLD + frow] = eluf_(acc[r] + b1v);
    }
  }
  __syncthreads();

  // ---- conv k=1, 32 -> 64 on the ELU'd intermediate, + residual; ELU(h') is written back over the h tile (rows 2..129):
  // the last convolution reads every row three times, so its activation is applied once here
  {
    f32x16 acc2[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc2[j][r] = 0.f;
    const int m = wave * 32 + frow;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float4 a4 = *reinterpret_cast<const float4*>(ys + m * YLD + c * 8 + fk);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        acc2[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, w2f[j][c].x, acc2[j], 0, 0, 0);
        acc2[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, w2f[j][c].y, acc2[j], 0, 0, 0);
        acc2[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, w2f[j][c].z, acc2[j], 0, 0, 0);
        acc2[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, w2f[j][c].w, acc2[j], 0, 0, 0);
      }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int mr = wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      const bool real = (s0 - 2 + mr) >= 0;  // samples before the utterance start are the zero padding of the last conv
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        float* p = hs + (mr + 2) * HLD + j * 32 + frow;
        *p = real ? eluf_(*p + (acc2[j][r] + b2v[j])) : 0.f;
      }
    }
  }
  __syncthreads();

  // ---- last conv k=3, 64 -> 1 on the stored ELU(h'): output i (sample s0+i) reads LDS rows i+2 .. i+4; two threads per output
  {
    const int i = tid >> 1, half = tid & 1;  // half: channels 0..31 / 32..63
    float s = 0.f;
    if (i < TO) {
#pragma unroll
      for (int j = 0; j < 3; ++j) {
#pragma unroll
        for (int c4 = 0; c4 < 8; ++c4) {
          const float4 v = *reinterpret_cast<const float4*>(hs + (i + 2 + j) * HLD + half * 32 + c4 * 4);
          const float4 wv = *reinterpret_cast<const float4*>(wf + j * 64 + half * 32 + c4 * 4);
          s += v.x * wv.x + v.y * wv.y + v.z * wv.z + v.w * wv.w;
        }
      }
    }
    s += __shfl_xor(s, 1, 64);
    if (half == 0 && i < TO && s0 + i < T) wav[(int64_t)b * wav_seg_stride + s0 + i] = s + bf;
  }
}

}  // namespace

extern "C" int sopro_seanet_tail_f32(const float* h, int64_t h_seg_stride, const float* w1, const float* b1, const float* w2,
                                      const float* b2, const float* wf, float bf, float* wav, int64_t wav_seg_stride, int32_t B,
                                      int32_t T, void* stream) {
  SOPRO_CHECK_ARG(h && w1 && b1 && w2 && b2 && wf && wav && B > 0 && T > 0, "bad pointers or sizes");
  SOPRO_CHECK_ARG(aligned16(h) && aligned16(w1) && aligned16(w2) && aligned16(wf) && (h_seg_stride & 3) == 0, "alignment");
  dim3 grid((T + TO - 1) / TO, B);
  hipLaunchKernelGGL(seanet_tail_kernel, grid, dim3(256), 0, (hipStream_t)stream, h, h_seg_stride, w1, b1, w2, b2, wf, bf, wav,
                     wav_seg_stride, T);
  SOPRO_LAUNCH_CHECK();
}
