// Batch-of-a-few-dozen-rows contraction for the per-frame autoregressive step.
//
// One workgroup = 16 output columns (weights are read exactly once per step, spread over the
// chip), 4 waves split K in 32-wide chunks, v_mfma_f32_16x16x4_f32 with the batch rows as the
// A operand (16 rows per tile, NBT tiles) and the weight rows as the B operand.  Partial sums are
// combined across the 4 waves in a fixed order through LDS (deterministic, no atomics).
// Optional fusions: RMSNorm in front (row statistics recomputed per workgroup: the input is
// B x K fp32, L2 resident), and the GLU -> ring-buffer write -> dilated depthwise taps ->
// residual tail of SSMLiteBlock.forward_step (channels are independent, so the workgroup that
// owns 16 channels of the GLU output also owns their ring-buffer columns).
#include "common.h"

namespace {

template <int NBT, int NWB>
__global__ __launch_bounds__(256) void skinny_kernel(const sopro_skinny_args a) {
  __shared__ float red[4][NBT * NWB][4][64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 15, g = lane >> 4;
  const int ntile = blockIdx.x;
  const int bbase = blockIdx.y * 16 * NBT;
  const int K = a.K;
  const int D = a.N / 2;  // GLU_DW only

  const float* wrow[NWB];
  {
    const int n0 = ntile * 16 + i;
    if (NWB == 1) {
      wrow[0] = (n0 < a.N) ? a.W + (int64_t)n0 * a.ldw : nullptr;
    } else {
      wrow[0] = (n0 < D) ? a.W + (int64_t)n0 * a.ldw : nullptr;
      wrow[NWB - 1] = (n0 < D) ? a.W + (int64_t)(D + n0) * a.ldw : nullptr;
    }
  }
  const float* xrow[NBT];
  float rstd[NBT];
#pragma unroll
  for (int bt = 0; bt < NBT; ++bt) {
    const int b = bbase + bt * 16 + i;
    xrow[bt] = (b < a.B) ? a.X + (int64_t)b * a.ldx : nullptr;
    rstd[bt] = 1.0f;
  }
  const bool do_norm = a.norm_w != nullptr;
  if (do_norm) {
#pragma unroll
    for (int bt = 0; bt < NBT; ++bt) {
      float ss = 0.f;
      if (xrow[bt]) {
        for (int k4 = g; k4 < K / 4; k4 += 4) {
          const float4 v = *reinterpret_cast<const float4*>(xrow[bt] + k4 * 4);
          ss += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
        }
      }
      ss += __shfl_xor(ss, 16, 64);
      ss += __shfl_xor(ss, 32, 64);
      rstd[bt] = rsqrtf(ss / (float)K + a.eps);
    }
  }

  f32x4 acc[NBT][NWB];
#pragma unroll
  for (int bt = 0; bt < NBT; ++bt)
#pragma unroll
    for (int wb = 0; wb < NWB; ++wb) acc[bt][wb] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int nch = K / 32;
#pragma unroll 3
  for (int c = wave; c < nch; c += 4) {
    const int kb = c * 32 + g * 8;
    float4 w0[NWB], w1[NWB];
#pragma unroll
    for (int wb = 0; wb < NWB; ++wb) {
      w0[wb] = make_float4(0.f, 0.f, 0.f, 0.f);
      w1[wb] = w0[wb];
      if (wrow[wb]) {
        w0[wb] = *reinterpret_cast<const float4*>(wrow[wb] + kb);
        w1[wb] = *reinterpret_cast<const float4*>(wrow[wb] + kb + 4);
      }
    }
    float4 nw0 = make_float4(1.f, 1.f, 1.f, 1.f), nw1 = nw0;
    if (do_norm) {
      nw0 = *reinterpret_cast<const float4*>(a.norm_w + kb);
      nw1 = *reinterpret_cast<const float4*>(a.norm_w + kb + 4);
    }
#pragma unroll
    for (int bt = 0; bt < NBT; ++bt) {
      float4 x0 = make_float4(0.f, 0.f, 0.f, 0.f), x1 = x0;
      if (xrow[bt]) {
        x0 = *reinterpret_cast<const float4*>(xrow[bt] + kb);
        x1 = *reinterpret_cast<const float4*>(xrow[bt] + kb + 4);
      }
      if (do_norm) {
        const float s = rstd[bt];
        x0.x = (x0.x * s) * nw0.x; x0.y = (x0.y * s) * nw0.y; x0.z = (x0.z * s) * nw0.z; x0.w = (x0.w * s) * nw0.w;
        x1.x = (x1.x * s) * nw1.x; x1.y = (x1.y * s) * nw1.y; x1.z = (x1.z * s) * nw1.z; x1.w = (x1.w * s) * nw1.w;
      }
#pragma unroll
      for (int wb = 0; wb < NWB; ++wb) {
        f32x4 c4 = acc[bt][wb];
        c4 = __builtin_amdgcn_mfma_f32_16x16x4f32(x0.x, w0[wb].x, c4, 0, 0, 0);
        c4 = __builtin_amdgcn_mfma_f32_16x16x4f32(x0.y, w0[wb].y, c4, 0, 0, 0);
        c4 = __builtin_amdgcn_mfma_f32_16x16x4f32(x0.z, w0[wb].z, c4, 0, 0, 0);
        c4 = __builtin_amdgcn_mfma_f32_16x16x4f32(x0.w, w0[wb].w, c4, 0, 0, 0);
        c4 = __builtin_amdgcn_mfma_f32_16x16x4f32(x1.x, w1[wb].x, c4, 0, 0, 0);
        c4 = __builtin_amdgcn_mfma_f32_16x16x4f32(x1.y, w1[wb].y, c4, 0, 0, 0);
        c4 = __builtin_amdgcn_mfma_f32_16x16x4f32(x1.z, w1[wb].z, c4, 0, 0, 0);
        c4 = __builtin_amdgcn_mfma_f32_16x16x4f32(x1.w, w1[wb].w, c4, 0, 0, 0);
        acc[bt][wb] = c4;
      }
    }
  }

#pragma unroll
  for (int bt = 0; bt < NBT; ++bt)
#pragma unroll
    for (int wb = 0; wb < NWB; ++wb)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[wave][bt * NWB + wb][r][lane] = acc[bt][wb][r];
  __syncthreads();
  if (wave >= NBT) return;

  // wave `bt` finishes batch tile bt: D[r] = row (lane>>4)*4 + r, column lane&15
  const int bt = wave;
  float v[NWB][4];
#pragma unroll
  for (int wb = 0; wb < NWB; ++wb)
#pragma unroll
    for (int r = 0; r < 4; ++r)
      v[wb][r] = ((red[0][bt * NWB + wb][r][lane] + red[1][bt * NWB + wb][r][lane]) + red[2][bt * NWB + wb][r][lane]) +
                 red[3][bt * NWB + wb][r][lane];

  const int n = ntile * 16 + i;
  const int epi = a.epilogue;
  if (NWB == 1) {
    if (n >= a.N) return;
    const float bias = a.bias ? a.bias[n] : 0.f;
    const float sc = a.scale ? a.scale[n] : 1.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int b = bbase + bt * 16 + g * 4 + r;
      if (b >= a.B) continue;
      float y = v[0][r] + bias;
      if (epi == SOPRO_EPI_GELU) y = gelu_erf(y);
      else if (epi == SOPRO_EPI_TANH) y = tanhf(y);
      else if (epi == SOPRO_EPI_RES) y = a.R[(int64_t)b * a.ldr + n] + (a.scale ? sc * y : y);
      a.Y[(int64_t)b * a.ldy + n] = y;
    }
  } else {
    if (n >= D) return;
    const float bias_v = a.bias ? a.bias[n] : 0.f;
    const float bias_g = a.bias ? a.bias[D + n] : 0.f;
    const unsigned t = (unsigned)(*a.step);
    const unsigned L = (unsigned)a.ring_len;
    const int ks = a.ksize;
    const unsigned slot_now = t % L;
    const float wlast = a.dw_w[(int64_t)(ks - 1) * D + n];
    const float dwb = a.dw_b[n];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int b = bbase + bt * 16 + g * 4 + r;
      if (b >= a.B) continue;
      const float h = (v[0][r] + bias_v) * sigmoidf_(v[NWB - 1][r] + bias_g);
      a.ring[((int64_t)slot_now * a.ring_bcap + b) * D + n] = h;
      float y = 0.f;
      for (int j = 0; j < ks - 1; ++j) {
        const unsigned slot = (t + 1u + (unsigned)(j * a.dil)) % L;
        y += a.dw_w[(int64_t)j * D + n] * a.ring[((int64_t)slot * a.ring_bcap + b) * D + n];
      }
      y += wlast * h;
      y += dwb;
      a.Y[(int64_t)b * a.ldy + n] = a.X[(int64_t)b * a.ldx + n] + y;
    }
  }
}

}  // namespace

extern "C" int sopro_skinny_f32(const sopro_skinny_args* p, void* stream) {
  SOPRO_CHECK_ARG(p != nullptr, "args is NULL");
  const sopro_skinny_args& a = *p;
  SOPRO_CHECK_ARG(a.B > 0 && a.N > 0 && a.K > 0, "B, N, K must be positive");
  SOPRO_CHECK_ARG((a.K % 32) == 0, "K must be a multiple of 32");
  SOPRO_CHECK_ARG(a.X && a.W && a.Y, "X, W, Y must be non-NULL");
  SOPRO_CHECK_ARG(aligned16(a.X) && aligned16(a.W) && (a.ldx & 3) == 0 && (a.ldw & 3) == 0, "X/W must be 16-byte aligned with ld % 4 == 0");
  SOPRO_CHECK_ARG(!a.norm_w || aligned16(a.norm_w), "norm_w must be 16-byte aligned");
  SOPRO_CHECK_ARG(a.epilogue != SOPRO_EPI_RES || a.R, "EPI_RES needs R");
  SOPRO_CHECK_ARG(a.epilogue != SOPRO_EPI_GLU, "EPI_GLU is not a skinny epilogue (use EPI_GLU_DW)");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const bool dw = a.epilogue == SOPRO_EPI_GLU_DW;
  if (dw) {
    SOPRO_CHECK_ARG((a.N & 1) == 0 && a.ring && a.dw_w && a.dw_b && a.step, "EPI_GLU_DW needs even N, ring, dw_w, dw_b, step");
    SOPRO_CHECK_ARG(a.ksize >= 1 && a.dil >= 1 && a.ring_len == (a.ksize - 1) * a.dil + 1, "ring_len must be (ksize-1)*dil+1");
    SOPRO_CHECK_ARG(a.ring_bcap >= a.B, "ring_bcap < B");
  }
  const int ncols = dw ? a.N / 2 : a.N;
  const int ntiles = (ncols + 15) / 16;
  if (a.B <= 16) {
    dim3 grid(ntiles, 1);
    if (dw) hipLaunchKernelGGL((skinny_kernel<1, 2>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((skinny_kernel<1, 1>), grid, dim3(256), 0, s, a);
  } else {
    dim3 grid(ntiles, (a.B + 31) / 32);
    if (dw) hipLaunchKernelGGL((skinny_kernel<2, 2>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((skinny_kernel<2, 1>), grid, dim3(256), 0, s, a);
  }
  SOPRO_LAUNCH_CHECK();
}
