// Batch-of-a-few-dozen-rows contraction for the per-frame autoregressive step (latency design).
//
// One workgroup = 16 output columns x one 384-wide K slice x 16 batch rows (grid.z walks the batch in 16s).  Weights are spread
// over the chip (every weight byte is read once per frame); K slices of FF2 (K = 1536) go to different
// workgroups, which write fp32 partial sums that the *next* kernel adds up in a fixed order while it
// stages its input (deterministic: no atomics), so that every stage of the frame has >= 96 workgroups
// streaming weights.  Per workgroup:
//   P0  every lane issues ALL of its weight-fragment loads (and, for the GLU/ring-buffer tail, all of
//       its ring-buffer tap loads) before anything waits: one memory latency per kernel, not one per loop trip;
//   P1  256 threads stage the [16 x 384] input slice through registers into LDS, summing the partial
//       buffers / bias of the producer on the way, computing RMSNorm row statistics with 16-lane shuffles
//       and writing normalised values; rows are padded to 388 floats so the 16 lanes of a fragment read
//       hit 16 distinct 16-byte slots;
//   P2  v_mfma_f32_16x16x4_f32 (exact fp32), batch rows = A operand from LDS, weight rows = B operand from
//       registers, 4 waves x 3 chunks of 32 k;
//   P3  fixed-order cross-wave sum through LDS and the epilogue (bias, GELU, residual*scale, or the
//       GLU -> ring write -> 13 dilated taps -> residual tail of SSMLiteBlock.forward_step).
#include "common.h"

namespace {

constexpr int KS = 384;        // K slice == d_model of the checkpoint family
constexpr int XLD = KS + 4;    // padded LDS row
constexpr int MAXTAPS = 13;
constexpr int MAXNP = 3;
constexpr int SQ = 6;          // float4 per staging thread and source (24 floats)       // partial-sum buffers a consumer can add on top of X

template <int NWB>
__global__ __launch_bounds__(256) void skinny_kernel(const sopro_skinny_args a) {
  constexpr int NBT = 1;
  extern __shared__ float4 smem4[];
  float* xs = reinterpret_cast<float*>(smem4);            // [NBT*16][XLD] normalised / combined input slice
  float* red = xs + NBT * 16 * XLD;                       // [4][NBT*NWB][4][64]
  float* xraw = red + 4 * NBT * NWB * 4 * 64;             // [NBT*16][16] raw (combined) input of this tile's columns
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 15, g = lane >> 4;
  const int ntile = blockIdx.x;
  const int bbase = blockIdx.z * 16 * NBT;
  const int nslices = a.K / KS;
  const bool partial_out = gridDim.y > 1;
  const int D = a.N / 2;  // GLU_DW only
  const bool dw = (NWB == 2);

  // ---- P0a: everything the epilogue needs that does not depend on this frame's activations is requested now
  const int n_col = ntile * 16 + i;
  const int ncols = dw ? D : a.N;
  const bool col_ok = n_col < ncols;
  const bool epi_wave = wave < NBT;
  unsigned t_now = 0;
  if (dw) t_now = (unsigned)(*a.step);
  float e_bias = 0.f, e_bias_g = 0.f, e_scale = 1.f, e_dwb = 0.f;
  float e_res[4] = {0.f, 0.f, 0.f, 0.f};
  float tapw[MAXTAPS];
  const bool res_here = a.epilogue == SOPRO_EPI_RES && (!partial_out || blockIdx.y == 0);
  const bool bias_here = !partial_out || (blockIdx.y == 0 && a.epilogue == SOPRO_EPI_RES);
  if (epi_wave && col_ok) {
    if (a.bias && bias_here) {
      e_bias = a.bias[n_col];
      if (dw) e_bias_g = a.bias[D + n_col];
    }
    if (a.scale) e_scale = a.scale[n_col];
    if (res_here) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int b = bbase + wave * 16 + g * 4 + r;
        if (b < a.B) e_res[r] = a.R[(int64_t)b * a.ldr + n_col];
      }
    }
    if (dw) {
      e_dwb = a.dw_b[n_col];
#pragma unroll
      for (int j = 0; j < MAXTAPS; ++j) tapw[j] = (j < a.ksize) ? a.dw_w[(int64_t)j * D + n_col] : 0.f;
    }
  }

  const float* wrow[NWB];
  {
    if (!dw) {
      wrow[0] = col_ok ? a.W + (int64_t)n_col * a.ldw : nullptr;
    } else {
      wrow[0] = col_ok ? a.W + (int64_t)n_col * a.ldw : nullptr;
      wrow[NWB - 1] = col_ok ? a.W + (int64_t)(D + n_col) * a.ldw : nullptr;
    }
  }

  f32x4 acc[NBT][NWB];
#pragma unroll
  for (int bt = 0; bt < NBT; ++bt)
#pragma unroll
    for (int wb = 0; wb < NWB; ++wb) acc[bt][wb] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int srow = tid >> 4, spart = tid & 15;  // staging: row, 24-float part
  const bool do_norm = a.norm_w != nullptr;
  float tapv[4][MAXTAPS];

  for (int ks = blockIdx.y; ks < nslices; ks += gridDim.y) {
    const int k0 = ks * KS;
    // ---- P0b: all weight fragments of this slice for this wave (3 chunks x 2 float4 x NWB)
    float4 wf[NWB][3][2];
#pragma unroll
    for (int wb = 0; wb < NWB; ++wb)
#pragma unroll
      for (int cc = 0; cc < 3; ++cc) {
        const int kb = k0 + (wave * 3 + cc) * 32 + g * 8;
        wf[wb][cc][0] = make_float4(0.f, 0.f, 0.f, 0.f);
        wf[wb][cc][1] = wf[wb][cc][0];
        if (wrow[wb]) {
          wf[wb][cc][0] = *reinterpret_cast<const float4*>(wrow[wb] + kb);
          wf[wb][cc][1] = *reinterpret_cast<const float4*>(wrow[wb] + kb + 4);
        }
      }
    // ---- P1: stage the input slice (rows bbase .. bbase+16*NBT) into LDS; every source is requested before any is used
    {
      const int b = bbase + srow;
      float4 xv[SQ], pv[MAXNP][SQ], nwv[SQ];
#pragma unroll
      for (int q = 0; q < SQ; ++q) xv[q] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (b < a.B) {
        const float* xp = a.X + (int64_t)b * a.ldx + k0 + spart * 24;
#pragma unroll
        for (int q = 0; q < SQ; ++q) xv[q] = *reinterpret_cast<const float4*>(xp + q * 4);
#pragma unroll
        for (int sidx = 0; sidx < MAXNP; ++sidx) {
          if (sidx < a.np) {
            const float* pp = a.Xp + (int64_t)sidx * a.xp_stride + (int64_t)b * a.ldx + k0 + spart * 24;
#pragma unroll
            for (int q = 0; q < SQ; ++q) pv[sidx][q] = *reinterpret_cast<const float4*>(pp + q * 4);
          }
        }
      }
      if (do_norm) {
        const float* nw = a.norm_w + spart * 24;
#pragma unroll
        for (int q = 0; q < SQ; ++q) nwv[q] = *reinterpret_cast<const float4*>(nw + q * 4);
      }
      if (b < a.B) {
        if (a.xbias) {
          const float* bp = a.xbias + k0 + spart * 24;
#pragma unroll
          for (int q = 0; q < SQ; ++q) {
            const float4 t4 = *reinterpret_cast<const float4*>(bp + q * 4);
            xv[q].x += t4.x; xv[q].y += t4.y; xv[q].z += t4.z; xv[q].w += t4.w;
          }
        }
#pragma unroll
        for (int sidx = 0; sidx < MAXNP; ++sidx) {  // producer's K-slice partial sums, fixed order
          if (sidx < a.np) {
#pragma unroll
            for (int q = 0; q < SQ; ++q) {
              xv[q].x += pv[sidx][q].x; xv[q].y += pv[sidx][q].y; xv[q].z += pv[sidx][q].z; xv[q].w += pv[sidx][q].w;
            }
          }
        }
      }
      // raw (combined) values of this tile's 16 columns: side output + GLU residual
      if (nslices == 1) {
#pragma unroll
        for (int q = 0; q < SQ; ++q) {
          const int gq = spart * SQ + q;  // float4 index inside the 384-wide row
          if ((gq >> 2) == ntile) {
            *reinterpret_cast<float4*>(xraw + srow * 16 + (gq & 3) * 4) = xv[q];
            if (a.Xc && b < a.B) *reinterpret_cast<float4*>(a.Xc + (int64_t)b * a.ldxc + gq * 4) = xv[q];
          }
        }
      }
      if (do_norm) {
        float ss = 0.f;
#pragma unroll
        for (int q = 0; q < SQ; ++q) ss += xv[q].x * xv[q].x + xv[q].y * xv[q].y + xv[q].z * xv[q].z + xv[q].w * xv[q].w;
        ss += __shfl_xor(ss, 1, 64);
        ss += __shfl_xor(ss, 2, 64);
        ss += __shfl_xor(ss, 4, 64);
        ss += __shfl_xor(ss, 8, 64);
        const float rstd = rsqrtf(ss / (float)KS + a.eps);
#pragma unroll
        for (int q = 0; q < SQ; ++q) {
          xv[q].x = (xv[q].x * rstd) * nwv[q].x; xv[q].y = (xv[q].y * rstd) * nwv[q].y;
          xv[q].z = (xv[q].z * rstd) * nwv[q].z; xv[q].w = (xv[q].w * rstd) * nwv[q].w;
        }
      }
      float* dst = xs + srow * XLD + spart * 24;
#pragma unroll
      for (int q = 0; q < SQ; ++q) *reinterpret_cast<float4*>(dst + q * 4) = xv[q];
    }
    // ---- ring-buffer taps of earlier frames (their addresses need the frame index, which has arrived by now);
    //      they are consumed only in the epilogue, so this latency hides behind the MFMA phase
    if (dw && epi_wave) {
      const unsigned L = (unsigned)a.ring_len;
#pragma unroll
      for (int j = 0; j < MAXTAPS; ++j) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int b = bbase + wave * 16 + g * 4 + r;
          float v = 0.f;
          if (j < a.ksize - 1 && col_ok && b < a.B) {
            const unsigned slot = (t_now + 1u + (unsigned)(j * a.dil)) % L;
            v = a.ring[((int64_t)slot * a.ring_bcap + b) * D + n_col];
          }
          tapv[r][j] = v;
        }
      }
    }
    __syncthreads();
    // ---- P2: MFMA over this wave's 3 chunks
#pragma unroll
    for (int cc = 0; cc < 3; ++cc) {
      const int kl = (wave * 3 + cc) * 32 + g * 8;
#pragma unroll
      for (int bt = 0; bt < NBT; ++bt) {
        const float4 x0 = *reinterpret_cast<const float4*>(xs + (bt * 16 + i) * XLD + kl);
        const float4 x1 = *reinterpret_cast<const float4*>(xs + (bt * 16 + i) * XLD + kl + 4);
#pragma unroll
        for (int wb = 0; wb < NWB; ++wb) {
          f32x4 c4 = acc[bt][wb];
          c4 = __builtin_amdgcn_mfma_f32_16x16x4f32(x0.x, wf[wb][cc][0].x, c4, 0, 0, 0);
          c4 = __builtin_amdgcn_mfma_f32_16x16x4f32(x0.y, wf[wb][cc][0].y, c4, 0, 0, 0);
          c4 = __builtin_amdgcn_mfma_f32_16x16x4f32(x0.z, wf[wb][cc][0].z, c4, 0, 0, 0);
          c4 = __builtin_amdgcn_mfma_f32_16x16x4f32(x0.w, wf[wb][cc][0].w, c4, 0, 0, 0);
          c4 = __builtin_amdgcn_mfma_f32_16x16x4f32(x1.x, wf[wb][cc][1].x, c4, 0, 0, 0);
          c4 = __builtin_amdgcn_mfma_f32_16x16x4f32(x1.y, wf[wb][cc][1].y, c4, 0, 0, 0);
          c4 = __builtin_amdgcn_mfma_f32_16x16x4f32(x1.z, wf[wb][cc][1].z, c4, 0, 0, 0);
          c4 = __builtin_amdgcn_mfma_f32_16x16x4f32(x1.w, wf[wb][cc][1].w, c4, 0, 0, 0);
          acc[bt][wb] = c4;
        }
      }
    }
    if (ks + (int)gridDim.y < nslices) __syncthreads();  // xs is restaged by the next slice
  }

  // ---- P3: fixed-order cross-wave reduction
#pragma unroll
  for (int bt = 0; bt < NBT; ++bt)
#pragma unroll
    for (int wb = 0; wb < NWB; ++wb)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[((wave * NBT * NWB + bt * NWB + wb) * 4 + r) * 64 + lane] = acc[bt][wb][r];
  __syncthreads();
  if (!epi_wave) return;
  const int bt = wave;
  float v[NWB][4];
#pragma unroll
  for (int wb = 0; wb < NWB; ++wb)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int o = ((bt * NWB + wb) * 4 + r) * 64 + lane;
      const int ws = NBT * NWB * 4 * 64;
      v[wb][r] = ((red[o] + red[ws + o]) + red[2 * ws + o]) + red[3 * ws + o];
    }

  // D[r] of tile bt: batch row bbase + bt*16 + (lane>>4)*4 + r, column ntile*16 + (lane&15)
  if (!col_ok) return;
  const int n = n_col;
  const int epi = a.epilogue;
  if (!dw) {
    float* yp = partial_out ? a.Y + (int64_t)blockIdx.y * a.y_part_stride : a.Y;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int b = bbase + bt * 16 + g * 4 + r;
      if (b >= a.B) continue;
      float y = v[0][r] + e_bias;
      if (!partial_out) {
        if (epi == SOPRO_EPI_GELU) y = gelu_erf(y);
        else if (epi == SOPRO_EPI_TANH) y = tanhf(y);
      }
      if (res_here) y = e_res[r] + (a.scale ? e_scale * y : y);
      yp[(int64_t)b * a.ldy + n] = y;
    }
  } else {
    const unsigned L = (unsigned)a.ring_len;
    const unsigned slot_now = t_now % L;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int b = bbase + bt * 16 + g * 4 + r;
      if (b >= a.B) continue;
      const float h = (v[0][r] + e_bias) * sigmoidf_(v[NWB - 1][r] + e_bias_g);
      a.ring[((int64_t)slot_now * a.ring_bcap + b) * D + n] = h;
      float y = 0.f;
#pragma unroll
      for (int j = 0; j < MAXTAPS; ++j) {
        if (j < a.ksize - 1) y += tapw[j] * tapv[r][j];
        else if (j == a.ksize - 1) y += tapw[j] * h;
      }
      y += e_dwb;
      a.Y[(int64_t)b * a.ldy + n] = xraw[(bt * 16 + g * 4 + r) * 16 + i] + y;
    }
  }
}

template <int NWB>
int launch(const sopro_skinny_args& a, dim3 grid, hipStream_t s) {
  constexpr size_t lds = sizeof(float) * ((size_t)16 * XLD + 4 * NWB * 4 * 64 + 16 * 16);
  auto kern = skinny_kernel<NWB>;
  hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, a);
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
  SOPRO_CHECK_ARG(!a.norm_w || (aligned16(a.norm_w) && a.K == KS), "norm_w needs 16-byte alignment and K == 384");
  SOPRO_CHECK_ARG(a.epilogue != SOPRO_EPI_RES || a.R, "EPI_RES needs R");
  SOPRO_CHECK_ARG(a.epilogue != SOPRO_EPI_GLU, "EPI_GLU is not a skinny epilogue (use EPI_GLU_DW)");
  SOPRO_CHECK_ARG(a.np >= 0 && a.np <= MAXNP && (a.np == 0 || (a.Xp && aligned16(a.Xp) && (a.xp_stride & 3) == 0)), "bad partial-sum inputs");
  SOPRO_CHECK_ARG(!a.xbias || aligned16(a.xbias), "xbias must be 16-byte aligned");
  SOPRO_CHECK_ARG(!a.Xc || (a.K == KS && aligned16(a.Xc) && (a.ldxc & 3) == 0), "Xc needs K == 384 and 16-byte alignment");
  SOPRO_CHECK_ARG(a.ksplit == 0 || a.ksplit == 1, "ksplit must be 0 or 1");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const bool dw = a.epilogue == SOPRO_EPI_GLU_DW;
  if (dw) {
    SOPRO_CHECK_ARG((a.N & 1) == 0 && a.K == KS && a.ring && a.dw_w && a.dw_b && a.step, "EPI_GLU_DW needs even N, K == 384, ring, dw_w, dw_b, step");
    SOPRO_CHECK_ARG(a.ksize >= 1 && a.ksize <= MAXTAPS && a.dil >= 1 && a.ring_len == (a.ksize - 1) * a.dil + 1, "ring_len must be (ksize-1)*dil+1, ksize <= 13");
    SOPRO_CHECK_ARG(a.ring_bcap >= a.B, "ring_bcap < B");
  }
  const int nslices = a.K / KS;
  const int gy = (a.ksplit && nslices > 1) ? nslices : 1;
  SOPRO_CHECK_ARG(gy == 1 || a.epilogue == SOPRO_EPI_NONE || a.epilogue == SOPRO_EPI_RES,
                  "K-split output takes EPI_NONE or EPI_RES (slice 0 then carries bias + residual; the consumer sums the slices)");
  const int ncols = dw ? a.N / 2 : a.N;
  const int ntiles = (ncols + 15) / 16;
  dim3 grid(ntiles, gy, (a.B + 15) / 16);
  return dw ? launch<2>(a, grid, s) : launch<1>(a, grid, s);
}
