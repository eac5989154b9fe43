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

typedef __bf16 sk_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 sk_bf16x2 __attribute__((ext_vector_type(2)));
typedef float sk_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned cvt2_bf16(float x, float y) {
  const sk_f32x2 v = {x, y};
  const sk_bf16x2 h = __builtin_convertvector(v, sk_bf16x2);
  return *reinterpret_cast<const unsigned*>(&h);
}

// WB (bf16 mode of the engine): the weights are bf16 in the fragment order of sopro_pack_skinny_w_bf16 (half the bytes of the
// frame's dominant stream), the staged activations are rounded to bf16 as they leave LDS, and a 32-wide K chunk is ONE
// v_mfma_f32_16x16x32_bf16 (fp32 accumulate) instead of eight exact-fp32 MFMAs.  Norm statistics, biases, the ring buffer,
// the taps and the residual stay fp32.
//
// MT x NT (round 3): one workgroup owns MT 16-row groups of the batch and NT 16-column tiles (NT 8-channel tiles in the GLU
// form).  A wave's weight fragments (registers) serve all MT row groups and its activation fragments (LDS) all NT column
// tiles, so a workgroup ingests 16*NT weight rows + 16*MT activation rows for MT*NT output tiles.  1 x 1 is the latency form
// (most workgroups, least bytes per workgroup: one frame alone on the whole chip); 2 x 2 halves both the weight bytes and
// the activation re-reads of a 32-row batch, which is what bounds two frames sharing a 64-CU partition (a CU ingests
// ~17-23 B/clk however many workgroups ask: profiles/r03_experiments.md).  Same arithmetic per output element in every form
// (same K order inside a wave, same cross-wave order), so the forms are bit-identical to each other.
// RB (bf16 mode, round 4; GLU tail only): the ring buffer holds bf16 (h is rounded once when it is written; the 12 older taps
// are read as bf16 and widened) - half the bytes of the frame's second-largest activation stream.  The newest tap, the tap
// weights and the accumulation stay fp32.
template <bool GLU, int NP, bool NORM, bool WB, int MT, int NT, bool RB = false>
__global__ __launch_bounds__(256) void skinny_kernel(const sopro_skinny_args a) {
  __shared__ float xs[16 * MT * XLD];          // raw (combined) input slice
  __shared__ float red[MT * NT * 4 * 4 * 64];
  __shared__ float rstd_s[16 * MT];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 15, g = lane >> 4;
  // AUX column tiles (round 4; plain form only): the workgroups behind the main ones run the SAME input rows against a second
  // weight / bias / residual / output set - the query projection of the following cross-attention block, emitted by the
  // feed-forward launches that already stage its operands (q = Wq' x is linear in x = out + b2 + W2 u: the `Wq' out` columns ride
  // on FF1, the `(Wq' W2) u` columns on FF2's K-slices) - so the cached keys stay UNFOLDED ([S, 96] per head instead of the
  // folded [S, 384]) without an extra launch.  aux_flags bit 0: no RMSNorm row scale; bit 1: epilogue NONE.
  const int main_x = (int)gridDim.x - a.aux_tiles / NT;
  const bool aux = !GLU && a.aux_tiles > 0 && (int)blockIdx.x >= main_x;
  const float* const W_ = aux ? a.aux_W : a.W;
  const float* const bias_ = aux ? a.aux_bias : a.bias;
  const float* const R_ = aux ? a.aux_R : a.R;
  float* const Y_ = aux ? a.aux_Y : a.Y;
  const int N_ = aux ? a.aux_tiles * 16 : a.N;
  const int64_t ldy_ = aux ? a.aux_ldy : a.ldy, ldr_ = aux ? a.aux_ldr : a.ldr, yps_ = aux ? a.aux_y_part_stride : a.y_part_stride;
  const bool norm_ = NORM && !(aux && (a.aux_flags & 1));
  const int epi_ = (aux && (a.aux_flags & 2)) ? SOPRO_EPI_NONE : a.epilogue;
  const int tile0 = (aux ? (int)blockIdx.x - main_x : (int)blockIdx.x) * NT;
  const int bbase = blockIdx.z * 16 * MT;
  const int nslices = a.K / KS;
  const bool partial_out = gridDim.y > 1;
  const int D = a.N / 2;  // GLU tail only (no aux tiles there: N_ == a.N)
  long long* dbg = a.dbg ? a.dbg + ((int64_t)(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 8 : nullptr;
  if (dbg && tid == 0) dbg[0] = clock64();

  // frame index: scalar load (own counter), waited for only where the tap addresses are formed
  int t_now = 0;
  if (GLU) asm volatile("s_load_dword %0, %1, 0x0" : "=s"(t_now) : "s"(a.step) : "memory");

  // Columns owned by this lane.  Plain: 16 output columns per tile.  GLU tail: 8 channels per tile, lanes 0-7 carry the
  // value rows and lanes 8-15 the gate rows of the same channels (one B operand, paired by a shuffle).
  const int ncols = GLU ? D : N_;
  const int ntiles = GLU ? (D + 7) / 8 : (N_ + 15) / 16;
  int n_col[NT], n_ld[NT], w_row[NT];
  bool col_ok[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int tile = tile0 + nt;
    n_col[nt] = GLU ? tile * 8 + (i & 7) : tile * 16 + i;
    col_ok[nt] = n_col[nt] < ncols && (!GLU || i < 8);
    n_ld[nt] = min(n_col[nt], ncols - 1);  // clamped: loads stay in bounds, results are discarded
    w_row[nt] = GLU ? ((i < 8) ? n_ld[nt] : D + n_ld[nt]) : n_ld[nt];
  }
  const bool res_here = epi_ == SOPRO_EPI_RES && (!partial_out || blockIdx.y == 0);
  const bool bias_here = !partial_out || (blockIdx.y == 0 && epi_ == SOPRO_EPI_RES);

  // ---- epilogue operands: wave w finishes accumulator row r == w, i.e. batch row bbase + mt*16 + (lane>>4)*4 + w, column lane&15
  int b_row[MT], b_cl[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    b_row[mt] = bbase + mt * 16 + g * 4 + wave;
    b_cl[mt] = min(b_row[mt], a.B - 1);
  }
  float e_bias[NT], e_scale[NT], e_dwb[NT], e_res[MT][NT];
  float tapw[NT][MAXTAPS];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    e_bias[nt] = (bias_ && bias_here) ? bias_[w_row[nt]] : 0.f;
    e_scale[nt] = a.scale ? a.scale[n_ld[nt]] : 1.f;
    e_dwb[nt] = 0.f;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) e_res[mt][nt] = res_here ? R_[(int64_t)b_cl[mt] * ldr_ + n_ld[nt]] : 0.f;
    if (GLU) {
      e_dwb[nt] = a.dw_b[n_ld[nt]];
#pragma unroll
      for (int j = 0; j < MAXTAPS; ++j) {
        // slots 0..11: weights of the older taps (0 beyond ksize-1), slot 12: weight of the newest tap
        const float wv = a.dw_w[(int64_t)min(j, a.ksize - 1) * D + n_ld[nt]];
        tapw[nt][j] = (j == MAXTAPS - 1 || j < a.ksize - 1) ? wv : 0.f;
      }
    }
  }

  // weight fragment addresses: row-major rows of W, or the fragment order of sopro_pack_skinny_w (1 KiB per load instruction)
  const bool packed = a.w_layout == 1;
  const int kchunks = a.K >> 5;
  const float* wbase[NT];
  const uint4* wbase16[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int tile_ld = min(tile0 + nt, ntiles - 1);
    wbase[nt] = packed ? W_ + ((int64_t)tile_ld * kchunks * 2) * 256 + lane * 4 : W_ + (int64_t)w_row[nt] * a.ldw + g * 8;
    wbase16[nt] = reinterpret_cast<const uint4*>(W_) + (int64_t)tile_ld * kchunks * 64 + lane;  // WB: [tile][chunk][lane] x 16 B
  }
  const int64_t w_chunk = packed ? 512 : 32, w_half = packed ? 256 : 4;
  f32x4 acc[MT][NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int srow = tid >> 4, spart = tid & 15;  // staging: row within a 16-row group, float4 column within each 64-float group
  unsigned slot0 = 0;  // ring slot of the oldest tap, (t + 1) mod L: one division, then increments
  float tapv[MT][NT][MAXTAPS - 1];

  for (int ks = blockIdx.y; ks < nslices; ks += gridDim.y) {
    const int k0 = ks * KS;
    // ---- all weight fragments of this slice for this wave (3 chunks x 2 float4 per column tile)
    float4 wf[NT][3][2];
    uint4 wh[NT][3];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int cc = 0; cc < 3; ++cc) {
        if constexpr (WB) {
          wh[nt][cc] = wbase16[nt][(int64_t)((k0 >> 5) + wave * 3 + cc) * 64];
        } else {
          const float* wp = wbase[nt] + (int64_t)((k0 >> 5) + wave * 3 + cc) * w_chunk;
          wf[nt][cc][0] = *reinterpret_cast<const float4*>(wp);
          wf[nt][cc][1] = *reinterpret_cast<const float4*>(wp + w_half);
        }
      }
    // ---- input slice (+ producer's partial sums, fixed order) -> LDS
    {
      float4 xv[MT][SQ], pv[MT][NP > 0 ? NP : 1][SQ];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const int b_ld = min(bbase + mt * 16 + srow, a.B - 1);
        // lane-contiguous: float4 q*16 + spart of the row, so that the 16 lanes of a row cover whole 128-byte lines per instruction
        const float* xp = a.X + (int64_t)b_ld * a.ldx + k0 + spart * 4;
#pragma unroll
        for (int q = 0; q < SQ; ++q) xv[mt][q] = *reinterpret_cast<const float4*>(xp + q * 64);
#pragma unroll
        for (int sidx = 0; sidx < NP; ++sidx) {
          const float* pp = a.Xp + (int64_t)sidx * a.xp_stride + (int64_t)b_ld * a.ldx + k0 + spart * 4;
#pragma unroll
          for (int q = 0; q < SQ; ++q) pv[mt][sidx][q] = *reinterpret_cast<const float4*>(pp + q * 64);
        }
      }
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
        for (int sidx = 0; sidx < NP; ++sidx)
#pragma unroll
          for (int q = 0; q < SQ; ++q) {
            xv[mt][q].x += pv[mt][sidx][q].x; xv[mt][q].y += pv[mt][sidx][q].y;
            xv[mt][q].z += pv[mt][sidx][q].z; xv[mt][q].w += pv[mt][sidx][q].w;
          }
        if (NORM) {
          float ss = 0.f;
#pragma unroll
          for (int q = 0; q < SQ; ++q)
            ss += xv[mt][q].x * xv[mt][q].x + xv[mt][q].y * xv[mt][q].y + xv[mt][q].z * xv[mt][q].z + xv[mt][q].w * xv[mt][q].w;
          ss += __shfl_xor(ss, 1, 64);
          ss += __shfl_xor(ss, 2, 64);
          ss += __shfl_xor(ss, 4, 64);
          ss += __shfl_xor(ss, 8, 64);
          if (spart == 0) rstd_s[mt * 16 + srow] = rsqrtf(ss / (float)KS + a.eps);
        }
        float* dst = xs + (mt * 16 + srow) * XLD + spart * 4;
#pragma unroll
        for (int q = 0; q < SQ; ++q) *reinterpret_cast<float4*>(dst + q * 64) = xv[mt][q];
      }
    }
    // ---- ring-buffer taps of earlier frames: addresses need the frame index; values are used in the epilogue
    if (GLU) {
      asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(t_now));
      const unsigned L = (unsigned)a.ring_len;
      slot0 = ((unsigned)t_now + 1u) % L;
      unsigned slot = slot0;
#pragma unroll
      for (int j = 0; j < MAXTAPS - 1; ++j) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            const int64_t ri = ((int64_t)slot * a.ring_bcap + b_cl[mt]) * D + n_ld[nt];
            if constexpr (RB) tapv[mt][nt][j] = __uint_as_float((unsigned)reinterpret_cast<const unsigned short*>(a.ring)[ri] << 16);
            else tapv[mt][nt][j] = a.ring[ri];
          }
        slot += (unsigned)a.dil;
        if (slot >= L) slot -= L;
      }
    }
    if (dbg && tid == 0) dbg[1] = clock64();
    __syncthreads();
    if (dbg && tid == 0) dbg[2] = clock64();
    // ---- MFMA over this wave's 3 chunks: A fragments (LDS) serve NT column tiles, B fragments (registers) MT row groups
#pragma unroll
    for (int cc = 0; cc < 3; ++cc) {
      const int kl = (wave * 3 + cc) * 32 + g * 8;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const float4 x0 = *reinterpret_cast<const float4*>(xs + (mt * 16 + i) * XLD + kl);
        const float4 x1 = *reinterpret_cast<const float4*>(xs + (mt * 16 + i) * XLD + kl + 4);
        if constexpr (WB) {
          const uint4 xa = make_uint4(cvt2_bf16(x0.x, x0.y), cvt2_bf16(x0.z, x0.w), cvt2_bf16(x1.x, x1.y), cvt2_bf16(x1.z, x1.w));
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const sk_bf16x8*>(&xa),
                                                                  *reinterpret_cast<const sk_bf16x8*>(&wh[nt][cc]), acc[mt][nt], 0, 0, 0);
        } else {
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            f32x4 c = acc[mt][nt];
            c = __builtin_amdgcn_mfma_f32_16x16x4f32(x0.x, wf[nt][cc][0].x, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x4f32(x0.y, wf[nt][cc][0].y, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x4f32(x0.z, wf[nt][cc][0].z, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x4f32(x0.w, wf[nt][cc][0].w, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x4f32(x1.x, wf[nt][cc][1].x, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x4f32(x1.y, wf[nt][cc][1].y, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x4f32(x1.z, wf[nt][cc][1].z, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x4f32(x1.w, wf[nt][cc][1].w, c, 0, 0, 0);
            acc[mt][nt] = c;
          }
        }
      }
    }
    if (ks + (int)gridDim.y < nslices) __syncthreads();  // xs is restaged by the next slice
  }

  if (dbg && tid == 0) dbg[3] = clock64();
  // ---- fixed-order cross-wave reduction; afterwards wave w owns accumulator row r == w of every 4-row group
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[(((mt * NT + nt) * 4 + wave) * 4 + r) * 64 + lane] = acc[mt][nt][r];
  __syncthreads();
  if (dbg && tid == 0) dbg[4] = clock64();
  const int epi = epi_;
  const unsigned L = (unsigned)a.ring_len;
  const unsigned slot_now = slot0 == 0 ? L - 1 : slot0 - 1;  // t mod L (GLU only)
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const float* rp = red + ((mt * NT + nt) * 16 + wave) * 64 + lane;
      float v = ((rp[0 * 256] + rp[1 * 256]) + rp[2 * 256]) + rp[3 * 256];
      if (NORM && norm_) v *= rstd_s[mt * 16 + g * 4 + wave];  // RMSNorm row scale (its weight is folded into W)
      const bool ok = col_ok[nt] && b_row[mt] < a.B;
      if (!GLU) {
        float* yp = partial_out ? Y_ + (int64_t)blockIdx.y * yps_ : Y_;
        float y = v + e_bias[nt];
        if (!partial_out) {
          if (epi == SOPRO_EPI_GELU) y = gelu_erf(y);
          else if (epi == SOPRO_EPI_TANH) y = tanhf(y);
        }
        if (res_here) y = fmaf(e_scale[nt], y, e_res[mt][nt]);  // (explicit: every workgroup shape must round alike)
        if (ok) yp[(int64_t)b_row[mt] * ldy_ + n_col[nt]] = y;
      } else {
        const float pre = v + e_bias[nt];                // lanes 0-7: value, lanes 8-15: gate pre-activation
        const float gate = __shfl_xor(pre, 8, 64);
        const float h = pre * sigmoidf_(gate);
        float y = 0.f;
#pragma unroll
        for (int j = 0; j < MAXTAPS - 1; ++j) y = fmaf(tapw[nt][j], tapv[mt][nt][j], y);  // explicit fma chain: the compiler must
        y = fmaf(tapw[nt][MAXTAPS - 1], h, y);                                            // not pick mul + add in one shape
        y += e_dwb[nt];
        if (ok) {
          if constexpr (RB) reinterpret_cast<unsigned short*>(a.ring)[((int64_t)slot_now * a.ring_bcap + b_row[mt]) * D + n_col[nt]] = (unsigned short)(cvt2_bf16(h, 0.f) & 0xffffu);
          else a.ring[((int64_t)slot_now * a.ring_bcap + b_row[mt]) * D + n_col[nt]] = h;
          a.Y[(int64_t)b_row[mt] * a.ldy + n_col[nt]] = xs[(mt * 16 + g * 4 + wave) * XLD + n_col[nt]] + y;
        }
      }
    }
  if (dbg && tid == 0) dbg[5] = clock64();
}

// one thread per float4 of the packed image
__global__ __launch_bounds__(256) void pack_skinny_kernel(const float* __restrict__ W, int64_t ldw, int N, int K, int glu,
                                                          float4* __restrict__ out, int64_t total) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int lane = (int)(idx & 63), half = (int)((idx >> 6) & 1);
  const int64_t tc = idx >> 7;
  const int kchunks = K >> 5;
  const int chunk = (int)(tc % kchunks), t = (int)(tc / kchunks);
  const int i = lane & 15, g = lane >> 4;
  const int D = N / 2;
  int row;
  bool ok;
  if (glu) {
    const int n = t * 8 + (i & 7);
    ok = n < D;
    row = (i < 8) ? n : D + n;
  } else {
    row = t * 16 + i;
    ok = row < N;
  }
  const int k = chunk * 32 + g * 8 + half * 4;
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (ok) v = *reinterpret_cast<const float4*>(W + (int64_t)row * ldw + k);
  out[idx] = v;
}

// bf16 form: one thread per 16-byte fragment piece (8 consecutive k of one weight row)
__global__ __launch_bounds__(256) void pack_skinny_bf16_kernel(const float* __restrict__ W, int64_t ldw, int N, int K, int glu,
                                                               uint4* __restrict__ out, int64_t total) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int lane = (int)(idx & 63);
  const int64_t tc = idx >> 6;
  const int kchunks = K >> 5;
  const int chunk = (int)(tc % kchunks), t = (int)(tc / kchunks);
  const int i = lane & 15, g = lane >> 4;
  const int D = N / 2;
  int row;
  bool ok;
  if (glu) {
    const int n = t * 8 + (i & 7);
    ok = n < D;
    row = (i < 8) ? n : D + n;
  } else {
    row = t * 16 + i;
    ok = row < N;
  }
  uint4 v = make_uint4(0u, 0u, 0u, 0u);
  if (ok) {
    const float* p = W + (int64_t)row * ldw + chunk * 32 + g * 8;
    const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
    v = make_uint4(cvt2_bf16(a.x, a.y), cvt2_bf16(a.z, a.w), cvt2_bf16(b.x, b.y), cvt2_bf16(b.z, b.w));
  }
  out[idx] = v;
}

template <bool GLU, int NP, bool NORM, int MT, int NT>
int launch_t(const sopro_skinny_args& a, dim3 grid, hipStream_t s) {
  if constexpr (GLU) {
    if (a.w_layout == 2 && a.ring_format == 1) {
      hipLaunchKernelGGL((skinny_kernel<GLU, NP, NORM, true, MT, NT, true>), grid, dim3(256), 0, s, a);
      SOPRO_LAUNCH_CHECK();
    }
  }
  if (a.w_layout == 2)
    hipLaunchKernelGGL((skinny_kernel<GLU, NP, NORM, true, MT, NT>), grid, dim3(256), 0, s, a);
  else
    hipLaunchKernelGGL((skinny_kernel<GLU, NP, NORM, false, MT, NT>), grid, dim3(256), 0, s, a);
  SOPRO_LAUNCH_CHECK();
}

template <bool GLU, int NP, bool NORM>
int launch(const sopro_skinny_args& a, int ntiles, int gy, hipStream_t s) {
  const int mt = a.mt == 2 ? 2 : 1, nt = a.nt == 2 ? 2 : 1;
  dim3 grid((ntiles + nt - 1) / nt + a.aux_tiles / nt, gy, (a.B + 16 * mt - 1) / (16 * mt));
  if (mt == 2 && nt == 2) return launch_t<GLU, NP, NORM, 2, 2>(a, grid, s);
  if (mt == 2) return launch_t<GLU, NP, NORM, 2, 1>(a, grid, s);
  if (nt == 2) return launch_t<GLU, NP, NORM, 1, 2>(a, grid, s);
  return launch_t<GLU, NP, NORM, 1, 1>(a, grid, s);
}

}  // namespace

extern "C" int64_t sopro_skinny_packed_floats(int32_t N, int32_t K, int32_t glu) {
  if (N <= 0 || K <= 0 || (K & 31) || (glu && (N & 1))) return 0;
  const int64_t tiles = glu ? (N / 2 + 7) / 8 : (N + 15) / 16;
  return tiles * (K >> 5) * 512;
}

extern "C" int sopro_pack_skinny_w(const float* W, int64_t ldw, int32_t N, int32_t K, int32_t glu, float* out, void* stream) {
  SOPRO_CHECK_ARG(W && out && N > 0 && K > 0 && ldw >= K, "bad pointers or sizes");
  SOPRO_CHECK_ARG((K & 31) == 0 && (ldw & 3) == 0 && aligned16(W) && aligned16(out), "K % 32 == 0, ldw % 4 == 0, 16-byte aligned W / out");
  SOPRO_CHECK_ARG(!glu || (N & 1) == 0, "glu: N must be even");
  const int64_t total = sopro_skinny_packed_floats(N, K, glu) / 4;
  hipLaunchKernelGGL(pack_skinny_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, W, ldw, N, K, glu,
                     reinterpret_cast<float4*>(out), total);
  SOPRO_LAUNCH_CHECK();
}

extern "C" int sopro_pack_skinny_w_bf16(const float* W, int64_t ldw, int32_t N, int32_t K, int32_t glu, void* out, void* stream) {
  SOPRO_CHECK_ARG(W && out && N > 0 && K > 0 && ldw >= K, "bad pointers or sizes");
  SOPRO_CHECK_ARG((K & 31) == 0 && (ldw & 3) == 0 && aligned16(W) && aligned16(out), "K % 32 == 0, ldw % 4 == 0, 16-byte aligned W / out");
  SOPRO_CHECK_ARG(!glu || (N & 1) == 0, "glu: N must be even");
  const int64_t total = sopro_skinny_packed_floats(N, K, glu) / 8;  // 16-byte pieces: half the bytes of the fp32 form
  hipLaunchKernelGGL(pack_skinny_bf16_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, W, ldw, N, K, glu,
                     reinterpret_cast<uint4*>(out), total);
  SOPRO_LAUNCH_CHECK();
}

extern "C" int sopro_skinny_f32(const sopro_skinny_args* p, void* stream) {
  SOPRO_CHECK_ARG(p != nullptr, "args is NULL");
  const sopro_skinny_args& a = *p;
  SOPRO_CHECK_ARG(a.B > 0 && a.N > 0 && a.K > 0, "B, N, K must be positive");
  SOPRO_CHECK_ARG((a.K % KS) == 0, "K must be a multiple of 384");
  SOPRO_CHECK_ARG(a.X && a.W && a.Y, "X, W, Y must be non-NULL");
  SOPRO_CHECK_ARG(aligned16(a.X) && aligned16(a.W) && (a.ldx & 3) == 0 && (a.ldw & 3) == 0, "X/W must be 16-byte aligned with ld % 4 == 0");
  SOPRO_CHECK_ARG(a.w_layout >= 0 && a.w_layout <= 2, "w_layout must be 0 (row-major), 1 (sopro_pack_skinny_w) or 2 (sopro_pack_skinny_w_bf16)");
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
    SOPRO_CHECK_ARG(a.ring_format == 0 || (a.ring_format == 1 && a.w_layout == 2), "ring_format: 0 (fp32), or 1 (bf16 ring) with the bf16 weights of w_layout 2");
  }
  const int nslices = a.K / KS;
  const int gy = (a.ksplit && nslices > 1) ? nslices : 1;
  SOPRO_CHECK_ARG(gy == 1 || a.epilogue == SOPRO_EPI_NONE || a.epilogue == SOPRO_EPI_RES,
                  "K-split output takes EPI_NONE or EPI_RES (slice 0 then carries bias + residual; the consumer sums the slices)");
  SOPRO_CHECK_ARG(a.mt >= 0 && a.mt <= 2 && a.nt >= 0 && a.nt <= 2, "mt / nt must be 0 (= 1), 1 or 2");
  if (a.aux_tiles != 0) {
    SOPRO_CHECK_ARG(a.aux_tiles > 0 && (a.aux_tiles & 1) == 0 && !dw && a.w_layout != 0 && (a.N & 31) == 0,
                    "aux tiles: an even count, plain form, fragment-ordered weights, N a multiple of 32 (whole main tiles)");
    SOPRO_CHECK_ARG(a.aux_W && a.aux_Y && aligned16(a.aux_W) && ((a.aux_flags & 2) || a.epilogue != SOPRO_EPI_RES || a.aux_R), "aux tiles: aux_W, aux_Y (and aux_R for EPI_RES)");
  }
  const int ntiles = dw ? (a.N / 2 + 7) / 8 : (a.N + 15) / 16;
  const bool np3 = a.np == 3, nrm = a.rms_norm != 0;
  if (dw) return np3 ? launch<true, 3, true>(a, ntiles, gy, s) : launch<true, 0, true>(a, ntiles, gy, s);
  if (nrm) return np3 ? launch<false, 3, true>(a, ntiles, gy, s) : launch<false, 0, true>(a, ntiles, gy, s);
  return np3 ? launch<false, 3, false>(a, ntiles, gy, s) : launch<false, 0, false>(a, ntiles, gy, s);
}
