// Batch-of-a-few-dozen-rows contraction for the per-frame autoregressive step (latency design).
//
// One workgroup = 16 output columns x one 384-wide K slice x 16 batch rows (grid.z walks the batch in 16s).
// Weights are spread over the chip (every weight byte is read once per frame and batch group); the K slices of
// FF2 (K = 1536) go to different workgroups, which write fp32 partial sums that the *next* kernel adds up in a
// fixed order while it stages its input (deterministic: no atomics), so every stage of the frame has >= 96
// workgroups streaming weights.  What the in-kernel clock stamps (args.dbg) showed matters on MI355X:
//   * a dependent global round trip costs ~2000 cycles here, so the kernel is written as ONE round: the code is
//     branch-free up to the MFMA phase (clamped addresses instead of predicated loads, template parameters
//     instead of runtime options) so that the compiler can issue every load - weight fragments, the input
//     slice and its partial sums, the epilogue operands - back to back before the first wait;
//   * the frame index (needed only for ring-buffer addresses) is fetched with a scalar load, whose counter is
//     independent of the vector loads, and the ring-buffer taps it addresses are consumed in the epilogue only;
//   * the RMSNorm weight is folded into the weight matrix on the host and the row scale 1/rms is applied to the
//     accumulator, so the staged slice is the raw input (also the residual of the GLU tail) and nothing but
//     unique bytes goes through the CU's L1.
// Phases: stage [16 x 384] input (+ partial sums) into LDS with padded rows (388 floats: 16-lane ds_read_b128
// fragment reads hit 16 distinct slots) -> v_mfma_f32_16x16x4_f32 (exact fp32; batch rows = A operand from
// LDS, weight rows = B operand from registers, 4 waves x 3 chunks of 32 k) -> fixed-order cross-wave sum
// through LDS -> epilogue (bias, GELU, residual*scale, or GLU -> ring write -> 13 dilated taps -> residual of
// SSMLiteBlock.forward_step, src/sopro/nn/blocks.py:150-157, 76-110).
#include "common.h"

namespace {

constexpr int KS = 384;        // K slice == d_model of the checkpoint family
constexpr int XLD = KS + 4;    // padded LDS row
constexpr int MAXTAPS = 13;
constexpr int SQ = 6;          // float4 per staging thread and source (24 floats)

template <int NWB, int NP, bool NORM>
__global__ __launch_bounds__(256) void skinny_kernel(const sopro_skinny_args a) {
  __shared__ float xs[16 * XLD];          // raw (combined) input slice
  __shared__ float red[4 * NWB * 4 * 64];
  __shared__ float rstd_s[16];
  __shared__ float taps_s[NWB == 2 ? 4 * (MAXTAPS - 1) * 64 : 1];  // [row-in-quad r][tap j][lane]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 15, g = lane >> 4;
  const int ntile = blockIdx.x;
  const int bbase = blockIdx.z * 16;
  const int nslices = a.K / KS;
  const bool partial_out = gridDim.y > 1;
  const int D = a.N / 2;  // GLU_DW only
  constexpr bool dw = (NWB == 2);
  long long* dbg = a.dbg ? a.dbg + ((int64_t)(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 8 : nullptr;
  if (dbg && tid == 0) dbg[0] = clock64();

  // frame index: scalar load (own counter), waited for only where the tap addresses are formed
  int t_now = 0;
  if (dw) asm volatile("s_load_dword %0, %1, 0x0" : "=s"(t_now) : "s"(a.step) : "memory");

  const int ncols = dw ? D : a.N;
  const int n_col = ntile * 16 + i;
  const bool col_ok = n_col < ncols;
  const int n_ld = col_ok ? n_col : ncols - 1;  // clamped: loads stay in bounds, results are discarded
  const bool res_here = a.epilogue == SOPRO_EPI_RES && (!partial_out || blockIdx.y == 0);
  const bool bias_here = !partial_out || (blockIdx.y == 0 && a.epilogue == SOPRO_EPI_RES);

  // ---- epilogue operands (wave 0 finishes the tile: D[r] = row (lane>>4)*4 + r, column lane&15)
  float e_bias = 0.f, e_bias_g = 0.f, e_scale = 1.f, e_dwb = 0.f;
  float e_res[4] = {0.f, 0.f, 0.f, 0.f};
  float tapw[MAXTAPS];
  int brow[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) brow[r] = min(bbase + g * 4 + r, a.B - 1);
  if (wave == 0) {
    if (a.bias && bias_here) {
      e_bias = a.bias[n_ld];
      if (dw) e_bias_g = a.bias[D + n_ld];
    }
    if (a.scale) e_scale = a.scale[n_ld];
    if (res_here) {
#pragma unroll
      for (int r = 0; r < 4; ++r) e_res[r] = a.R[(int64_t)brow[r] * a.ldr + n_ld];
    }
    if (dw) {
      e_dwb = a.dw_b[n_ld];
#pragma unroll
      for (int j = 0; j < MAXTAPS; ++j) tapw[j] = a.dw_w[(int64_t)min(j, a.ksize - 1) * D + n_ld];
    }
  }

  const float* wrow[NWB];
  wrow[0] = a.W + (int64_t)n_ld * a.ldw;
  if (dw) wrow[NWB - 1] = a.W + (int64_t)(D + n_ld) * a.ldw;

  f32x4 acc[NWB];
#pragma unroll
  for (int wb = 0; wb < NWB; ++wb) acc[wb] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int srow = tid >> 4, spart = tid & 15;  // staging: row, 24-float part
  const int b_ld = min(bbase + srow, a.B - 1);
  float tapv[MAXTAPS - 1];  // wave w fetches the taps of accumulator row r == w; wave 0 picks them up from LDS

  for (int ks = blockIdx.y; ks < nslices; ks += gridDim.y) {
    const int k0 = ks * KS;
    // ---- all weight fragments of this slice for this wave (3 chunks x 2 float4 x NWB)
    float4 wf[NWB][3][2];
#pragma unroll
    for (int wb = 0; wb < NWB; ++wb)
#pragma unroll
      for (int cc = 0; cc < 3; ++cc) {
        const int kb = k0 + (wave * 3 + cc) * 32 + g * 8;
        wf[wb][cc][0] = *reinterpret_cast<const float4*>(wrow[wb] + kb);
        wf[wb][cc][1] = *reinterpret_cast<const float4*>(wrow[wb] + kb + 4);
      }
    // ---- input slice (+ producer's partial sums, fixed order) -> LDS
    {
      float4 xv[SQ], pv[NP > 0 ? NP : 1][SQ];
      const float* xp = a.X + (int64_t)b_ld * a.ldx + k0 + spart * 24;
#pragma unroll
      for (int q = 0; q < SQ; ++q) xv[q] = *reinterpret_cast<const float4*>(xp + q * 4);
#pragma unroll
      for (int sidx = 0; sidx < NP; ++sidx) {
        const float* pp = a.Xp + (int64_t)sidx * a.xp_stride + (int64_t)b_ld * a.ldx + k0 + spart * 24;
#pragma unroll
        for (int q = 0; q < SQ; ++q) pv[sidx][q] = *reinterpret_cast<const float4*>(pp + q * 4);
      }
#pragma unroll
      for (int sidx = 0; sidx < NP; ++sidx)
#pragma unroll
        for (int q = 0; q < SQ; ++q) {
          xv[q].x += pv[sidx][q].x; xv[q].y += pv[sidx][q].y; xv[q].z += pv[sidx][q].z; xv[q].w += pv[sidx][q].w;
        }
      if (NORM) {
        float ss = 0.f;
#pragma unroll
        for (int q = 0; q < SQ; ++q) ss += xv[q].x * xv[q].x + xv[q].y * xv[q].y + xv[q].z * xv[q].z + xv[q].w * xv[q].w;
        ss += __shfl_xor(ss, 1, 64);
        ss += __shfl_xor(ss, 2, 64);
        ss += __shfl_xor(ss, 4, 64);
        ss += __shfl_xor(ss, 8, 64);
        if (spart == 0) rstd_s[srow] = rsqrtf(ss / (float)KS + a.eps);
      }
      float* dst = xs + srow * XLD + spart * 24;
#pragma unroll
      for (int q = 0; q < SQ; ++q) *reinterpret_cast<float4*>(dst + q * 4) = xv[q];
    }
    // ---- ring-buffer taps of earlier frames: addresses need the frame index; values are used in the epilogue
    if (dw) asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(t_now));  // uniform: every wave waits for the scalar load
    if (dw) {
      const unsigned L = (unsigned)a.ring_len;
      const int bw = min(bbase + g * 4 + wave, a.B - 1);
#pragma unroll
      for (int j = 0; j < MAXTAPS - 1; ++j) {
        const unsigned slot = ((unsigned)t_now + 1u + (unsigned)(j * a.dil)) % L;
        tapv[j] = a.ring[((int64_t)slot * a.ring_bcap + bw) * D + n_ld];
      }
    }
    if (dbg && tid == 0) dbg[1] = clock64();
    __syncthreads();
    if (dbg && tid == 0) dbg[2] = clock64();
    // ---- MFMA over this wave's 3 chunks
#pragma unroll
    for (int cc = 0; cc < 3; ++cc) {
      const int kl = (wave * 3 + cc) * 32 + g * 8;
      const float4 x0 = *reinterpret_cast<const float4*>(xs + i * XLD + kl);
      const float4 x1 = *reinterpret_cast<const float4*>(xs + i * XLD + kl + 4);
#pragma unroll
      for (int wb = 0; wb < NWB; ++wb) {
        f32x4 c4 = acc[wb];
        c4 = __builtin_amdgcn_mfma_f32_16x16x4f32(x0.x, wf[wb][cc][0].x, c4, 0, 0, 0);
        c4 = __builtin_amdgcn_mfma_f32_16x16x4f32(x0.y, wf[wb][cc][0].y, c4, 0, 0, 0);
        c4 = __builtin_amdgcn_mfma_f32_16x16x4f32(x0.z, wf[wb][cc][0].z, c4, 0, 0, 0);
        c4 = __builtin_amdgcn_mfma_f32_16x16x4f32(x0.w, wf[wb][cc][0].w, c4, 0, 0, 0);
        c4 = __builtin_amdgcn_mfma_f32_16x16x4f32(x1.x, wf[wb][cc][1].x, c4, 0, 0, 0);
        c4 = __builtin_amdgcn_mfma_f32_16x16x4f32(x1.y, wf[wb][cc][1].y, c4, 0, 0, 0);
        c4 = __builtin_amdgcn_mfma_f32_16x16x4f32(x1.z, wf[wb][cc][1].z, c4, 0, 0, 0);
        c4 = __builtin_amdgcn_mfma_f32_16x16x4f32(x1.w, wf[wb][cc][1].w, c4, 0, 0, 0);
        acc[wb] = c4;
      }
    }
    if (ks + (int)gridDim.y < nslices) __syncthreads();  // xs is restaged by the next slice
  }

  if (dbg && tid == 0) dbg[3] = clock64();
  // ---- fixed-order cross-wave reduction
#pragma unroll
  for (int wb = 0; wb < NWB; ++wb)
#pragma unroll
    for (int r = 0; r < 4; ++r) red[((wave * NWB + wb) * 4 + r) * 64 + lane] = acc[wb][r];
  __syncthreads();
  if (wave != 0) return;
  if (dbg && tid == 0) dbg[4] = clock64();
  float v[NWB][4];
#pragma unroll
  for (int wb = 0; wb < NWB; ++wb)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int o = (wb * 4 + r) * 64 + lane;
      constexpr int ws = NWB * 4 * 64;
      v[wb][r] = ((red[o] + red[ws + o]) + red[2 * ws + o]) + red[3 * ws + o];
      if (NORM) v[wb][r] *= rstd_s[g * 4 + r];  // RMSNorm row scale (its weight is folded into W)
    }

  const int epi = a.epilogue;
  if (!dw) {
    float* yp = partial_out ? a.Y + (int64_t)blockIdx.y * a.y_part_stride : a.Y;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int b = bbase + g * 4 + r;
      float y = v[0][r] + e_bias;
      if (!partial_out) {
        if (epi == SOPRO_EPI_GELU) y = gelu_erf(y);
        else if (epi == SOPRO_EPI_TANH) y = tanhf(y);
      }
      if (res_here) y = e_res[r] + e_scale * y;
      if (col_ok && b < a.B) yp[(int64_t)b * a.ldy + n_col] = y;
    }
  } else {
    const unsigned L = (unsigned)a.ring_len;
    const unsigned slot_now = (unsigned)t_now % L;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int b = bbase + g * 4 + r;
      const float h = (v[0][r] + e_bias) * sigmoidf_(v[NWB - 1][r] + e_bias_g);
      float y = 0.f;
#pragma unroll
      for (int j = 0; j < MAXTAPS - 1; ++j)
        if (j < a.ksize - 1) y += tapw[j] * taps_s[(r * (MAXTAPS - 1) + j) * 64 + lane];
      y += tapw[MAXTAPS - 1] * h;  // tapw[j >= ksize-1] all hold the newest tap's weight
      y += e_dwb;
      if (col_ok && b < a.B) {
        a.ring[((int64_t)slot_now * a.ring_bcap + b) * D + n_col] = h;
        a.Y[(int64_t)b * a.ldy + n_col] = xs[(g * 4 + r) * XLD + n_col] + y;
      }
    }
  }
  if (dbg && tid == 0) dbg[5] = clock64();
}

template <int NWB, int NP, bool NORM>
int launch(const sopro_skinny_args& a, dim3 grid, hipStream_t s) {
  hipLaunchKernelGGL((skinny_kernel<NWB, NP, NORM>), grid, dim3(256), 0, s, a);
  SOPRO_LAUNCH_CHECK();
}

}  // namespace

extern "C" int sopro_skinny_f32(const sopro_skinny_args* p, void* stream) {
  SOPRO_CHECK_ARG(p != nullptr, "args is NULL");
  const sopro_skinny_args& a = *p;
  SOPRO_CHECK_ARG(a.B > 0 && a.N > 0 && a.K > 0, "B, N, K must be positive");
  SOPRO_CHECK_ARG((a.K % KS) == 0, "K must be a multiple of 384");
  SOPRO_CHECK_ARG(a.X && a.W && a.Y, "X, W, Y must be non-NULL");
  SOPRO_CHECK_ARG(aligned16(a.X) && aligned16(a.W) && (a.ldx & 3) == 0 && (a.ldw & 3) == 0, "X/W must be 16-byte aligned with ld % 4 == 0");
  SOPRO_CHECK_ARG(!a.rms_norm || a.K == KS, "rms_norm needs K == 384");
  SOPRO_CHECK_ARG(a.epilogue != SOPRO_EPI_RES || a.R, "EPI_RES needs R");
  SOPRO_CHECK_ARG(a.epilogue != SOPRO_EPI_GLU, "EPI_GLU is not a skinny epilogue (use EPI_GLU_DW)");
  SOPRO_CHECK_ARG(a.np == 0 || (a.np == 3 && a.Xp && aligned16(a.Xp) && (a.xp_stride & 3) == 0), "np must be 0 or 3 (with Xp)");
  SOPRO_CHECK_ARG(a.ksplit == 0 || a.ksplit == 1, "ksplit must be 0 or 1");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const bool dw = a.epilogue == SOPRO_EPI_GLU_DW;
  if (dw) {
    SOPRO_CHECK_ARG(a.K == KS && a.N == 2 * KS && a.ring && a.dw_w && a.dw_b && a.step,
                    "EPI_GLU_DW needs N == 2*K == 768, ring, dw_w, dw_b, step");
    SOPRO_CHECK_ARG(a.ksize >= 1 && a.ksize <= MAXTAPS && a.dil >= 1 && a.ring_len == (a.ksize - 1) * a.dil + 1, "ring_len must be (ksize-1)*dil+1, ksize <= 13");
    SOPRO_CHECK_ARG(a.ring_bcap >= a.B && a.rms_norm, "ring_bcap < B, or rms_norm not set (the GLU tail always follows an RMSNorm)");
  }
  const int nslices = a.K / KS;
  const int gy = (a.ksplit && nslices > 1) ? nslices : 1;
  SOPRO_CHECK_ARG(gy == 1 || a.epilogue == SOPRO_EPI_NONE || a.epilogue == SOPRO_EPI_RES,
                  "K-split output takes EPI_NONE or EPI_RES (slice 0 then carries bias + residual; the consumer sums the slices)");
  const int ncols = dw ? a.N / 2 : a.N;
  dim3 grid((ncols + 15) / 16, gy, (a.B + 15) / 16);
  const bool np3 = a.np == 3, nrm = a.rms_norm != 0;
  if (dw) return np3 ? launch<2, 3, true>(a, grid, s) : launch<2, 0, true>(a, grid, s);
  if (nrm) return np3 ? launch<1, 3, true>(a, grid, s) : launch<1, 0, true>(a, grid, s);
  return np3 ? launch<1, 3, false>(a, grid, s) : launch<1, 0, false>(a, grid, s);
}
