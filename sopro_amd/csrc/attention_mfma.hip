// Attention on the matrix cores: causal sliding-window self-attention with head dim 64 (Mimi decoder / encoder transformers,
// HF:modeling_mimi.py MimiAttention + sliding-window mask) and the dense cross-attention over the reference voice with head dim
// 192 / 96 (src/sopro/nn/xattn.py RefXAttnBlock; per-row key counts).  Exact fp32 v_mfma_f32_32x32x2_f32 in "transposed" form
// so that a lane owns ONE query for the whole kernel:
//     S^T [32 keys x 32 queries] = K_tile . Q^T        (A = K rows, B = Q rows: both read straight from HBM/L2)
//     O^T [DH dims x 32 queries] += V_tile^T . P       (A = V columns; B = P = exp(S^T - m), the S^T accumulators as they are)
// In the MFMA C layout a lane holds column (lane & 31) = its query and 16 of the 32 key rows, which is exactly the B
// operand layout of the second product (MFMA step r multiplies key rows k(r), k(r)+4), so P never leaves the registers;
// the online-softmax statistics of a query are per lane plus one exchange with lane ^ 32; no LDS is used at all.
// One wave = 32 queries of one (batch, head); a workgroup = 4 consecutive query tiles.
#include "common.h"

namespace {

template <int DH, bool CAUSAL>
__global__ __launch_bounds__(256) void attn_mfma_kernel(const sopro_attn_args a) {
  constexpr int NC = DH / 8, NT = DH / 32;  // float4 fragments of a query / key row per lane; 32-wide output tiles
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int qt = blockIdx.x * 4 + wave, h = blockIdx.y, b = blockIdx.z;
  const int q0 = qt * 32;
  if (q0 >= a.Tq) return;  // whole wave
  const int col = lane & 31, half = lane >> 5;
  const float* Qb = a.Q + (int64_t)b * a.q_bstride + h * DH;
  const int kb = a.kv_index ? a.kv_index[b] : b;  // rows that share a voice share one copy of its keys / values
  const float* Kb = a.K + (int64_t)kb * a.k_bstride + h * DH;
  const float* Vb = a.V + (int64_t)kb * a.v_bstride + h * DH;
  const int klen = a.klens ? min(a.klens[b], a.Tk) : a.Tk;

  // Q fragments of this lane's query: k = 8c + 4*half + s  (the same k order as the K fragments below)
  const int qi = min(q0 + col, a.Tq - 1);
  const int qabs = a.q_pos0 + qi;
  float4 qf[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) qf[c] = *reinterpret_cast<const float4*>(Qb + (int64_t)qi * a.ldq + c * 8 + half * 4);

  f32x16 o[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[t][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;

  // key range any query of the tile can see: q_abs - window < k_abs <= q_abs (causal), every valid key otherwise
  int k_first = 0, k_last = klen - 1;
  if constexpr (CAUSAL) {
    const int q_lo_abs = a.q_pos0 + q0, q_hi_abs = a.q_pos0 + min(q0 + 31, a.Tq - 1);
    k_first = max(q_lo_abs - a.window + 1 - a.k_pos0, 0);
    k_last = min(q_hi_abs - a.k_pos0, klen - 1);
  }
  for (int k0 = (k_first / 32) * 32; k0 <= k_last; k0 += 32) {
    // ---- S^T = K_tile . Q^T
    const int kr = min(k0 + col, klen - 1);  // clamped: rows past the end are masked below
    const float* kp = Kb + (int64_t)kr * a.ldk + half * 4;
    float4 kf[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) kf[c] = *reinterpret_cast<const float4*>(kp + c * 8);
    // V^T fragments: MFMA step r of the second product multiplies key rows k(r) + 4*half; lane reads column `col`
    float vf[NT][16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = min(k0 + (r & 3) + 8 * (r >> 2) + 4 * half, klen - 1);
      const float* vp = Vb + (int64_t)key * a.ldv + col;
#pragma unroll
      for (int t = 0; t < NT; ++t) vf[t][r] = vp[32 * t];
    }
    f32x16 st;
#pragma unroll
    for (int r = 0; r < 16; ++r) st[r] = 0.f;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      st = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[c].x, qf[c].x, st, 0, 0, 0);
      st = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[c].y, qf[c].y, st, 0, 0, 0);
      st = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[c].z, qf[c].z, st, 0, 0, 0);
      st = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[c].w, qf[c].w, st, 0, 0, 0);
    }
    // ---- mask + online softmax for this lane's query (rows of st = keys k0 + (r&3) + 8(r>>2) + 4*half)
    float mx = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = k0 + (r & 3) + 8 * (r >> 2) + 4 * half;
      const int kabs = a.k_pos0 + key;
      const bool ok = key < klen && (!CAUSAL || (kabs <= qabs && kabs > qabs - a.window));
      st[r] = ok ? st[r] * a.scale : -INFINITY;
      mx = fmaxf(mx, st[r]);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m_run, mx);
    // a query that sees nothing in this tile keeps its state (m_new == -inf only before its first visible key)
    const float alpha = (m_new == -INFINITY) ? 1.f : expf(m_run - m_new);
    float ps = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float p = (st[r] == -INFINITY) ? 0.f : expf(st[r] - m_new);
      st[r] = p;
      ps += p;
    }
    ps += __shfl_xor(ps, 32, 64);
    l_run = l_run * alpha + ps;
    m_run = m_new;
#pragma unroll
    for (int r = 0; r < 16; ++r)
#pragma unroll
      for (int t = 0; t < NT; ++t) o[t][r] *= alpha;
    // ---- O^T += V_tile^T . P
#pragma unroll
    for (int r = 0; r < 16; ++r)
#pragma unroll
      for (int t = 0; t < NT; ++t) o[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(vf[t][r], st[r], o[t], 0, 0, 0);
  }
  // ---- O[q][d]: lane holds d = 32t + (r&3) + 8(r>>2) + 4*half of its query: four consecutive d per (t, r>>2)
  if (q0 + col < a.Tq) {
    const float inv = l_run > 0.f ? 1.f / l_run : 0.f;
    float* op = a.O + (int64_t)b * a.o_bstride + (int64_t)(q0 + col) * a.ldo + h * DH + half * 4;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        *reinterpret_cast<float4*>(op + 32 * t + g * 8) =
            make_float4(o[t][g * 4] * inv, o[t][g * 4 + 1] * inv, o[t][g * 4 + 2] * inv, o[t][g * 4 + 3] * inv);
  }
}

// ---------------------------------------------------------------------------------------------
// The same transposed form on v_mfma_f32_32x32x16_bf16 for the Mimi DECODER's window attention (waveform path: the decoder's
// contractions already multiply two-piece bf16 operands in three passes, 16 mantissa bits, under the 1e-4-of-peak contract;
// the encoder and the conditioning keep the exact kernel above - their outputs are discrete codes / token logits).
// fp32 MFMA runs at 1/16 of the bf16 rate: a 32 x 32 key / query tile costs 64 exact steps of 64 cycles, or 24 bf16 steps
// of 32.  Operand slots: an MFMA sums over (lane >> 5, i = 0..7); A and B only have to agree on what a slot means:
//     S^T:  slot (g, i) of step j = head dim 16 j + 8 g + i            (K rows and Q rows, 32 contiguous bytes per lane)
//     O^T:  slot (g, i) of step j = key row (r & 3) + 8 (r >> 2) + 4 g, r = 8 j + i  = the C layout of S^T, so that
//           P is used from the accumulators it was computed in (split once, never moved).
typedef __bf16 abf16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ abf16x8 afrag(const uint4& v) { return *reinterpret_cast<const abf16x8*>(&v); }

template <int PASSES>
__device__ __forceinline__ void asplit8(const float* v, uint4& hi, uint4& lo) {
  split2_bf16(v[0], v[1], hi.x, lo.x);
  split2_bf16(v[2], v[3], hi.y, lo.y);
  split2_bf16(v[4], v[5], hi.z, lo.z);
  split2_bf16(v[6], v[7], hi.w, lo.w);
}

template <int PASSES>
__device__ __forceinline__ f32x16 mma_split(const uint4& ah, const uint4& al, const uint4& bh, const uint4& bl, f32x16 c) {
  if constexpr (PASSES == 3) {  // small terms first
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afrag(al), afrag(bh), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afrag(ah), afrag(bl), c, 0, 0, 0);
  }
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(afrag(ah), afrag(bh), c, 0, 0, 0);
}

template <int PASSES>
__global__ __launch_bounds__(256) void attn_mfma_split_kernel(const sopro_attn_args a) {
  constexpr int DH = 64, NJ = DH / 16, NT = DH / 32;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int qt = blockIdx.x * 4 + wave, h = blockIdx.y, b = blockIdx.z;
  const int q0 = qt * 32;
  if (q0 >= a.Tq) return;  // whole wave
  const int col = lane & 31, half = lane >> 5;
  const float* Qb = a.Q + (int64_t)b * a.q_bstride + h * DH;
  const int kb = a.kv_index ? a.kv_index[b] : b;
  const float* Kb = a.K + (int64_t)kb * a.k_bstride + h * DH;
  const float* Vb = a.V + (int64_t)kb * a.v_bstride + h * DH;
  const int klen = a.klens ? min(a.klens[b], a.Tk) : a.Tk;

  const int qi = min(q0 + col, a.Tq - 1);
  const int qabs = a.q_pos0 + qi;
  const float qscale = a.scale * 1.44269504088896340736f;
  uint4 qh[NJ], ql[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const float* qp = Qb + (int64_t)qi * a.ldq + 16 * j + 8 * half;
    const float4 x = *reinterpret_cast<const float4*>(qp), y = *reinterpret_cast<const float4*>(qp + 4);
    // the softmax scale and log2(e) go into Q once: scores come out of the MFMAs in the exp2 domain
    const float v[8] = {x.x * qscale, x.y * qscale, x.z * qscale, x.w * qscale, y.x * qscale, y.y * qscale, y.z * qscale, y.w * qscale};
    asplit8<PASSES>(v, qh[j], ql[j]);
  }

  f32x16 o[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[t][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;

  const int q_lo_abs = a.q_pos0 + q0, q_hi_abs = a.q_pos0 + min(q0 + 31, a.Tq - 1);
  const int k_first = max(q_lo_abs - a.window + 1 - a.k_pos0, 0);
  const int k_last = min(q_hi_abs - a.k_pos0, klen - 1);
  const int64_t v_lane = (int64_t)(4 * half) * a.ldv + col;  // this lane's place inside a tile's value rows (loop-invariant)
  for (int k0 = (k_first / 32) * 32; k0 <= k_last; k0 += 32) {
    // ---- operands of this tile: the lane's key row (its 32 head dims) and its value column (16 key rows per 32-wide tile)
    const bool inrange = k0 + 31 < klen;  // wave-uniform: no clamping, row addresses = uniform row base + the lane's place
    const int kr = min(k0 + col, klen - 1);  // clamped: rows past the end are masked below
    const float* kp = Kb + (int64_t)kr * a.ldk + 8 * half;
    float4 kraw[2 * NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      kraw[2 * j] = *reinterpret_cast<const float4*>(kp + 16 * j);
      kraw[2 * j + 1] = *reinterpret_cast<const float4*>(kp + 16 * j + 4);
    }
    float vraw[NT][16];
    if (inrange) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float* vrow = Vb + (int64_t)(k0 + (r & 3) + 8 * (r >> 2)) * a.ldv;  // uniform
#pragma unroll
        for (int t = 0; t < NT; ++t) vraw[t][r] = vrow[v_lane + 32 * t];
      }
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = min(k0 + (r & 3) + 8 * (r >> 2) + 4 * half, klen - 1);
        const float* vp = Vb + (int64_t)key * a.ldv + col;
#pragma unroll
        for (int t = 0; t < NT; ++t) vraw[t][r] = vp[32 * t];
      }
    }
    // ---- S^T = K_tile . Q^T (already in the exp2 domain: Q carries scale * log2 e)
    f32x16 st;
#pragma unroll
    for (int r = 0; r < 16; ++r) st[r] = 0.f;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const float v[8] = {kraw[2 * j].x, kraw[2 * j].y, kraw[2 * j].z, kraw[2 * j].w, kraw[2 * j + 1].x, kraw[2 * j + 1].y, kraw[2 * j + 1].z, kraw[2 * j + 1].w};
      uint4 kh, kl;
      asplit8<PASSES>(v, kh, kl);
      st = mma_split<PASSES>(kh, kl, qh[j], ql[j], st);
    }
    // ---- online softmax for this lane's query (rows of st = keys k0 + (r&3) + 8(r>>2) + 4*half).
    // Most tiles lie wholly inside every query's window: no masks, no infinities (wave-uniform test).
    const bool full = inrange && (a.k_pos0 + k0 + 31 <= q_lo_abs) && (a.k_pos0 + k0 > q_hi_abs - a.window);
    float mx = -INFINITY;
    if (full) {
#pragma unroll
      for (int r = 0; r < 16; ++r) mx = fmaxf(mx, st[r]);
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = k0 + (r & 3) + 8 * (r >> 2) + 4 * half;
        const int kabs = a.k_pos0 + key;
        const bool ok = key < klen && kabs <= qabs && kabs > qabs - a.window;
        st[r] = ok ? st[r] : -INFINITY;
        mx = fmaxf(mx, st[r]);
      }
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    // Lazy reference: the running reference m_run only moves when some query's maximum outgrows it by more than 2^8 (or has
    // none yet); until then the weights are taken against the old reference (at most 2^8, exact in fp32 and in the split) and
    // the accumulators (AGPRs: a read, a multiply and a write each) are left alone.  O / l at the end is the same quotient.
    const bool grow = mx > m_run + 8.f || m_run == -INFINITY;
    if (__builtin_amdgcn_ballot_w64(grow) != 0) {
      const float m_new = fmaxf(m_run, mx);
      // a query that has seen nothing yet keeps its (empty) state: m_new == -inf only before its first visible key
      const float alpha = (m_new == -INFINITY || m_new == m_run) ? 1.f : __builtin_amdgcn_exp2f(m_run - m_new);
      l_run *= alpha;
      m_run = m_new;
#pragma unroll
      for (int r = 0; r < 16; ++r)
#pragma unroll
        for (int t = 0; t < NT; ++t) o[t][r] *= alpha;
    }
    float ps = 0.f;
    if (full) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        st[r] = __builtin_amdgcn_exp2f(st[r] - m_run);
        ps += st[r];
      }
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float p = (st[r] == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(st[r] - m_run);
        st[r] = p;
        ps += p;
      }
    }
    ps += __shfl_xor(ps, 32, 64);
    l_run += ps;
    // ---- O^T += V_tile^T . P   (step j: key rows of accumulator registers 8 j .. 8 j + 7)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const float pv[8] = {st[8 * j], st[8 * j + 1], st[8 * j + 2], st[8 * j + 3], st[8 * j + 4], st[8 * j + 5], st[8 * j + 6], st[8 * j + 7]};
      uint4 ph, pl;
      asplit8<PASSES>(pv, ph, pl);
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        uint4 vh, vl;
        asplit8<PASSES>(&vraw[t][8 * j], vh, vl);
        o[t] = mma_split<PASSES>(vh, vl, ph, pl, o[t]);
      }
    }
  }
  // ---- O[q][d]: lane holds d = 32t + (r&3) + 8(r>>2) + 4*half of its query: four consecutive d per (t, r>>2)
  if (q0 + col < a.Tq) {
    const float inv = l_run > 0.f ? 1.f / l_run : 0.f;
    float* op = a.O + (int64_t)b * a.o_bstride + (int64_t)(q0 + col) * a.ldo + h * DH + half * 4;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        *reinterpret_cast<float4*>(op + 32 * t + g * 8) =
            make_float4(o[t][g * 4] * inv, o[t][g * 4 + 1] * inv, o[t][g * 4 + 2] * inv, o[t][g * 4 + 3] * inv);
  }
}

}  // namespace

// called by sopro_attention_f32 for problems with 16-byte aligned rows: dh == 64 causal window, dh in {64, 96, 192} dense
int sopro_attn_mfma(const sopro_attn_args& a, hipStream_t s) {
  dim3 grid(((a.Tq + 31) / 32 + 3) / 4, a.H, a.B);
  if (a.causal) {
    hipLaunchKernelGGL((attn_mfma_kernel<64, true>), grid, dim3(256), 0, s, a);
  } else if (a.dh == 64) {
    hipLaunchKernelGGL((attn_mfma_kernel<64, false>), grid, dim3(256), 0, s, a);
  } else if (a.dh == 96) {
    hipLaunchKernelGGL((attn_mfma_kernel<96, false>), grid, dim3(256), 0, s, a);
  } else {
    hipLaunchKernelGGL((attn_mfma_kernel<192, false>), grid, dim3(256), 0, s, a);
  }
  SOPRO_LAUNCH_CHECK();
}

// decoder-only form: causal window, dh == 64, operands as two bf16 pieces (passes = 3) or one (passes = 1, bf16 mode)
int sopro_attn_mfma_split(const sopro_attn_args& a, int passes, hipStream_t s) {
  dim3 grid(((a.Tq + 31) / 32 + 3) / 4, a.H, a.B);
  if (passes == 1) hipLaunchKernelGGL((attn_mfma_split_kernel<1>), grid, dim3(256), 0, s, a);
  else hipLaunchKernelGGL((attn_mfma_split_kernel<3>), grid, dim3(256), 0, s, a);
  SOPRO_LAUNCH_CHECK();
}
