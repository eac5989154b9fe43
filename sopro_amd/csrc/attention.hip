// fp32 attention for the three attention shapes of the path: text cross-attention of the AR step
// (1 query, S keys, 4 x 96), reference cross-attention of prepare_conditioning (Tar queries, Tr keys,
// 2 x 192) and the Mimi decoder transformer (causal sliding window 250, 8 x 64).
//
// One workgroup = one (batch, head, 16-query tile); K/V tiles of 64 keys are staged in LDS
// (coalesced row reads), scores are computed 4 keys per thread with the query row broadcast, the
// softmax is the online (running max / running sum) form in fp32, P goes through LDS into the
// P.V accumulation where each thread owns dh/16 output columns of one query.
#include "common.h"

namespace {

constexpr int TQ = 16, TK = 64;

template <int DH>
__global__ __launch_bounds__(256) void attention_kernel(const sopro_attn_args a) {
  constexpr int KP = DH + 1;   // padded K row: lanes walk keys -> distinct banks
  constexpr int NE = DH / 16;  // output columns per thread
  extern __shared__ float4 smem4[];
  float* Qs = reinterpret_cast<float*>(smem4);  // [TQ][DH]
  float* Ks = Qs + TQ * DH;                     // [TK][KP]
  float* Vs = Ks + TK * KP;                     // [TK][DH]
  float* Ps = Vs + TK * DH;                     // [TQ][TK+1]

  const int tid = threadIdx.x;
  const int q = tid >> 4, sub = tid & 15;
  const int qt = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
  const int q0 = qt * TQ;
  const float* Qb = a.Q + (int64_t)b * a.q_bstride + h * DH;
  const int kb = a.kv_index ? a.kv_index[b] : b;  // rows that share a voice share one copy of its keys / values
  const float* Kb = a.K + (int64_t)kb * a.k_bstride + h * DH;
  const float* Vb = a.V + (int64_t)kb * a.v_bstride + h * DH;
  const int klen = a.klens ? min(a.klens[b], a.Tk) : a.Tk;

  for (int idx = tid; idx < TQ * DH; idx += 256) {
    const int r = idx / DH, e = idx - r * DH;
    Qs[idx] = (q0 + r < a.Tq) ? Qb[(int64_t)(q0 + r) * a.ldq + e] : 0.f;
  }

  const int qi = q0 + q;
  const int qabs = a.q_pos0 + qi;
  // key range this query tile can see (causal window): skip whole tiles outside it
  int k_begin = 0, k_end = klen;
  if (a.causal) {
    const int lo = a.q_pos0 + q0 - a.window + 1 - a.k_pos0;                 // first key index visible to the first query
    const int hi = a.q_pos0 + min(q0 + TQ, a.Tq) - 1 - a.k_pos0 + 1;        // one past the last key visible to the last query
    k_begin = max(0, lo);
    k_end = min(klen, hi);
  }
  const int kt_begin = k_begin / TK;

  float m_run = -INFINITY, l_run = 0.f;
  float o[NE];
#pragma unroll
  for (int j = 0; j < NE; ++j) o[j] = 0.f;

  for (int k0 = kt_begin * TK; k0 < k_end; k0 += TK) {
    __syncthreads();  // previous tile fully consumed (also orders the Q staging before first use)
    {
      // K/V tile: all of a thread's float4 loads are issued before the first LDS store (one memory latency per tile)
      constexpr int F4 = TK * DH / 4 / 256;  // float4 per thread and matrix
      float4 kreg[F4], vreg[F4];
#pragma unroll
      for (int f = 0; f < F4; ++f) {
        const int idx4 = tid + f * 256;
        const int r = idx4 / (DH / 4), e4 = idx4 - r * (DH / 4);
        kreg[f] = make_float4(0.f, 0.f, 0.f, 0.f);
        vreg[f] = kreg[f];
        if ((k0 + r) < klen) {
          kreg[f] = *reinterpret_cast<const float4*>(Kb + (int64_t)(k0 + r) * a.ldk + e4 * 4);
          vreg[f] = *reinterpret_cast<const float4*>(Vb + (int64_t)(k0 + r) * a.ldv + e4 * 4);
        }
      }
#pragma unroll
      for (int f = 0; f < F4; ++f) {
        const int idx4 = tid + f * 256;
        const int r = idx4 / (DH / 4), e4 = idx4 - r * (DH / 4);
        float* kd = Ks + r * KP + e4 * 4;
        kd[0] = kreg[f].x; kd[1] = kreg[f].y; kd[2] = kreg[f].z; kd[3] = kreg[f].w;
        *reinterpret_cast<float4*>(Vs + r * DH + e4 * 4) = vreg[f];
      }
    }
    __syncthreads();

    float s[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) s[i] = 0.f;
    const float* qr = Qs + q * DH;
#pragma unroll 8
    for (int e = 0; e < DH; ++e) {
      const float qv = qr[e];
#pragma unroll
      for (int i = 0; i < 4; ++i) s[i] += qv * Ks[(sub + 16 * i) * KP + e];
    }
    float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int kk = k0 + sub + 16 * i;
      bool ok = kk < klen && qi < a.Tq;
      if (a.causal) {
        const int kabs = a.k_pos0 + kk;
        ok = ok && kabs <= qabs && kabs > qabs - a.window;
      }
      s[i] = ok ? s[i] * a.scale : -INFINITY;
      mx = fmaxf(mx, s[i]);
    }
#pragma unroll
    for (int off = 8; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
    const float m_new = fmaxf(m_run, mx);
    float alpha = 1.f, psum = 0.f;
    float p[4];
    if (m_new == -INFINITY) {
#pragma unroll
      for (int i = 0; i < 4; ++i) p[i] = 0.f;
    } else {
      alpha = expf(m_run - m_new);  // m_run == -inf -> 0
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        p[i] = expf(s[i] - m_new);
        psum += p[i];
      }
    }
#pragma unroll
    for (int off = 8; off > 0; off >>= 1) psum += __shfl_xor(psum, off, 64);
    l_run = l_run * alpha + psum;
    m_run = m_new;
#pragma unroll
    for (int i = 0; i < 4; ++i) Ps[q * (TK + 1) + sub + 16 * i] = p[i];
    __syncthreads();
#pragma unroll
    for (int j = 0; j < NE; ++j) o[j] *= alpha;
    const float* pr = Ps + q * (TK + 1);
#pragma unroll 4
    for (int kk = 0; kk < TK; ++kk) {
      const float pv = pr[kk];
#pragma unroll
      for (int j = 0; j < NE; ++j) o[j] += pv * Vs[kk * DH + sub + 16 * j];
    }
  }

  if (qi < a.Tq) {
    const float inv = l_run > 0.f ? 1.f / l_run : 0.f;
    float* orow = a.O + (int64_t)b * a.o_bstride + (int64_t)qi * a.ldo + h * DH;
#pragma unroll
    for (int j = 0; j < NE; ++j) orow[sub + 16 * j] = o[j] * inv;
  }
}

template <int DH>
int launch_attn(const sopro_attn_args& a, hipStream_t s) {
  constexpr size_t lds = sizeof(float) * (TQ * DH + TK * (DH + 1) + TK * DH + TQ * (TK + 1));
  auto kern = attention_kernel<DH>;
  SOPRO_SET_MAX_LDS_ONCE(kern, lds);
  dim3 grid((a.Tq + TQ - 1) / TQ, a.H, a.B);
  hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, a);
  SOPRO_LAUNCH_CHECK();
}

}  // namespace

int sopro_attn_mfma(const sopro_attn_args& a, hipStream_t s);  // attention_mfma.hip
int sopro_attn_mfma_split(const sopro_attn_args& a, int passes, hipStream_t s);

extern "C" int sopro_attention_f32(const sopro_attn_args* p, void* stream) {
  SOPRO_CHECK_ARG(p != nullptr, "args is NULL");
  const sopro_attn_args& a = *p;
  SOPRO_CHECK_ARG(a.Q && a.K && a.V && a.O, "Q, K, V, O must be non-NULL");
  SOPRO_CHECK_ARG(a.B > 0 && a.H > 0 && a.Tq > 0 && a.Tk > 0, "B, H, Tq, Tk must be positive");
  SOPRO_CHECK_ARG(!a.causal || a.window > 0, "causal attention needs window > 0");
  SOPRO_CHECK_ARG(aligned16(a.K) && aligned16(a.V) && (a.ldk & 3) == 0 && (a.ldv & 3) == 0 && (a.k_bstride & 3) == 0 &&
                      (a.v_bstride & 3) == 0,
                  "K/V must be 16-byte aligned with strides % 4 == 0");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  // causal sliding-window self-attention with 64-wide heads (the codec transformers) and dense attention (the reference
  // cross-attention of the conditioning) run on the matrix cores
  if ((a.causal ? a.dh == 64 : (a.dh == 64 || a.dh == 96 || a.dh == 192)) && a.Tq >= 16 && aligned16(a.Q) && aligned16(a.O) &&
      (a.ldq & 3) == 0 && (a.ldo & 3) == 0 && (a.q_bstride & 3) == 0 && (a.o_bstride & 3) == 0 && !SOPRO_DEV_ENV("SOPRO_ATTN_VALU"))
    return sopro_attn_mfma(a, s);
  switch (a.dh) {
    case 64: return launch_attn<64>(a, s);
    case 96: return launch_attn<96>(a, s);
    case 192: return launch_attn<192>(a, s);
    default: sopro_set_error("sopro_attention_f32: unsupported head dim %d (64, 96, 192)", a.dh); return -2;
  }
}

// Waveform-path form of the codec decoder's causal window attention: Q, K, V, P as two bf16 pieces multiplied in three
// v_mfma_f32_32x32x16_bf16 passes (passes == 3; 16 mantissa bits like the decoder's contractions) or as one piece
// (passes == 1, bf16 mode).  Shapes it is not written for (dh != 64, no causal window, < 16 queries, unaligned rows) run
// the exact kernel.  HF:modeling_mimi.py:657-726.
extern "C" int sopro_attention_split_bf16(const sopro_attn_args* p, int32_t passes, void* stream) {
  SOPRO_CHECK_ARG(p != nullptr, "args is NULL");
  SOPRO_CHECK_ARG(passes == 1 || passes == 3, "passes must be 1 or 3");
  const sopro_attn_args& a = *p;
  const bool fits = a.Q && a.K && a.V && a.O && a.B > 0 && a.H > 0 && a.Tk > 0 && a.causal && a.window > 0 && a.dh == 64 && a.Tq >= 16 &&
                    aligned16(a.Q) && aligned16(a.K) && aligned16(a.V) && aligned16(a.O) && ((a.ldq | a.ldk | a.ldv | a.ldo) & 3) == 0 &&
                    ((a.q_bstride | a.k_bstride | a.v_bstride | a.o_bstride) & 3) == 0;
  if (!fits) return sopro_attention_f32(p, stream);
  return sopro_attn_mfma_split(a, passes, reinterpret_cast<hipStream_t>(stream));
}

// ---------------------------------------------------------------------------------------------
// Single-query ("decode") attention for the AR frame: text cross-attention with cached K/V
// (src/sopro/nn/text.py:85-132).  One workgroup per (batch row, head); 4 lanes share a key, every
// lane issues all of its q/K/V loads up front (one memory latency), scores are reduced with two
// shuffles, the softmax statistics go through LDS once per 64-key tile, V is re-laid in LDS so that
// P.V is a conflict-free column walk.
// ---------------------------------------------------------------------------------------------
namespace {

template <int DH>
__global__ __launch_bounds__(256) void attn_decode_kernel(const sopro_attn_args a) {
  constexpr int PF = DH / 16;  // float4 per lane (a quarter of the head)
  __shared__ float Vs[64][DH + 4];
  __shared__ float ps[64];
  __shared__ float wred[4];
  __shared__ float opart[2][DH];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = blockIdx.x, b = blockIdx.y;
  const int kq = tid >> 2, part = tid & 3;
  const int klen = a.klens ? min(a.klens[b], a.Tk) : a.Tk;
  const float* qp = a.Q + (int64_t)b * a.q_bstride + h * DH + part * (DH / 4);
  const float* Kb = a.K + (int64_t)b * a.k_bstride + h * DH + part * (DH / 4);
  const float* Vb = a.V + (int64_t)b * a.v_bstride + h * DH + part * (DH / 4);
  float4 qv[PF];
#pragma unroll
  for (int f = 0; f < PF; ++f) qv[f] = *reinterpret_cast<const float4*>(qp + f * 4);

  const int e = tid % DH, kg = tid / DH;  // P.V mapping: column e, key group kg (threads >= 2*DH idle there)
  float m_run = -INFINITY, l_run = 0.f, o = 0.f;

  for (int k0 = 0; k0 < klen; k0 += 64) {
    const int key = k0 + kq;
    const bool in = key < klen;
    float4 kv[PF], vv[PF];
#pragma unroll
    for (int f = 0; f < PF; ++f) {
      kv[f] = make_float4(0.f, 0.f, 0.f, 0.f);
      vv[f] = kv[f];
      if (in) {
        kv[f] = *reinterpret_cast<const float4*>(Kb + (int64_t)key * a.ldk + f * 4);
        vv[f] = *reinterpret_cast<const float4*>(Vb + (int64_t)key * a.ldv + f * 4);
      }
    }
    float s = 0.f;
#pragma unroll
    for (int f = 0; f < PF; ++f) s += qv[f].x * kv[f].x + qv[f].y * kv[f].y + qv[f].z * kv[f].z + qv[f].w * kv[f].w;
    s += __shfl_xor(s, 1, 64);
    s += __shfl_xor(s, 2, 64);
    s = in ? s * a.scale : -INFINITY;
    float mx = wave_max(s);
    if (k0 > 0) __syncthreads();  // previous tile's LDS fully consumed
    if (lane == 0) wred[wave] = mx;
#pragma unroll
    for (int f = 0; f < PF; ++f) *reinterpret_cast<float4*>(&Vs[kq][part * (DH / 4) + f * 4]) = vv[f];
    __syncthreads();
    const float m_new = fmaxf(m_run, fmaxf(fmaxf(wred[0], wred[1]), fmaxf(wred[2], wred[3])));
    const float p = in ? expf(s - m_new) : 0.f;
    const float alpha = (m_run == -INFINITY) ? 0.f : expf(m_run - m_new);
    if (part == 0) ps[kq] = p;
    float psum = wave_sum(part == 0 ? p : 0.f);
    __syncthreads();
    if (lane == 0) wred[wave] = psum;  // safe: every thread read wred before the barrier above
    float acc = 0.f;
    if (kg < 2) {
#pragma unroll 8
      for (int kk = 0; kk < 32; ++kk) acc += ps[kg * 32 + kk] * Vs[kg * 32 + kk][e];
    }
    o = o * alpha + acc;
    __syncthreads();
    l_run = l_run * alpha + ((wred[0] + wred[1]) + (wred[2] + wred[3]));
    m_run = m_new;
  }
  if (kg < 2) opart[kg][e] = o;
  __syncthreads();
  if (tid < DH) {
    const float inv = l_run > 0.f ? 1.f / l_run : 0.f;
    a.O[(int64_t)b * a.o_bstride + h * DH + tid] = (opart[0][tid] + opart[1][tid]) * inv;
  }
}

}  // namespace

extern "C" int sopro_attn_decode_f32(const sopro_attn_args* p, void* stream) {
  SOPRO_CHECK_ARG(p == nullptr || p->kv_index == nullptr, "kv_index is a sopro_attention_f32 feature");
  SOPRO_CHECK_ARG(p != nullptr, "args is NULL");
  const sopro_attn_args& a = *p;
  SOPRO_CHECK_ARG(a.Q && a.K && a.V && a.O, "Q, K, V, O must be non-NULL");
  SOPRO_CHECK_ARG(a.B > 0 && a.H > 0 && a.Tq == 1 && a.Tk > 0 && !a.causal, "decode attention: Tq == 1, no causal mask");
  SOPRO_CHECK_ARG(aligned16(a.Q) && aligned16(a.K) && aligned16(a.V) && (a.ldk & 3) == 0 && (a.ldv & 3) == 0 &&
                      (a.k_bstride & 3) == 0 && (a.v_bstride & 3) == 0 && (a.q_bstride & 3) == 0,
                  "Q/K/V must be 16-byte aligned with strides % 4 == 0");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  dim3 grid(a.H, a.B);
  switch (a.dh) {
    case 96: hipLaunchKernelGGL(attn_decode_kernel<96>, grid, dim3(256), 0, s, a); break;
    case 64: hipLaunchKernelGGL(attn_decode_kernel<64>, grid, dim3(256), 0, s, a); break;
    default: sopro_set_error("sopro_attn_decode_f32: unsupported head dim %d (64, 96)", a.dh); return -2;
  }
  SOPRO_LAUNCH_CHECK();
}

// ---------------------------------------------------------------------------------------------
// Whole cached text cross-attention block of the AR frame in ONE launch (src/sopro/nn/text.py:85-132):
//     x + tanh(gate) * out_proj( softmax( q_proj(RMSNorm(x)) K^T / sqrt(dh) ) V )
// with the two projections folded, once per utterance, into the cached operands
//     K'_h = K_h Wq_h   [S, D]   (score_h[k] = <RMSNorm(x), K'_h[k]>)
//     V'_h = V_h Wo_h^T [S, D]   (head h's contribution to the block output, already in model space)
// so that per frame there is no q / out projection kernel at all: three launches become one.  One
// workgroup (8 waves) per (batch row, head); every K'/V' load is issued before the first wait; the four
// heads write four partial outputs (slice 0 carries the residual) which the next kernel of the frame sums
// while staging its input, exactly like the K-slices of the feed-forward output (deterministic, no atomics).
// ---------------------------------------------------------------------------------------------
namespace {

constexpr int XD = 384;

// NT: the folded operands are read once per frame and not again for a whole frame time (hundreds of MB per pass): non-temporal
// requests keep them from displacing what the throughput partition's kernels reuse in the XCDs' L2s (SOPRO_XATTN_NT=0: plain loads;
// measured in the pipeline: AR phase 38.6 -> 36.9 ms per step, 21.1 -> 21.5 k audio-s/s, profiles/r03_experiments.md).
typedef float xf32x4 __attribute__((ext_vector_type(4)));
template <bool NT>
__device__ __forceinline__ float4 ld_stream4(const float* p) {
  if constexpr (NT) {
    const xf32x4 v = __builtin_nontemporal_load(reinterpret_cast<const xf32x4*>(p));
    return make_float4(v.x, v.y, v.z, v.w);
  } else {
    return *reinterpret_cast<const float4*>(p);
  }
}

// HB (the engine's bf16 mode, round 4): K' / V' are stored as bf16 (half the bytes of the block's dominant stream: 98 KB instead
// of 196 KB per (row, head) at S = 64); scores and the weighted sum are accumulated in fp32 as before.
template <bool NT>
__device__ __forceinline__ uint4 ld_stream_u4(const void* p) {
  typedef unsigned xu32x4 __attribute__((ext_vector_type(4)));
  if constexpr (NT) {
    const xu32x4 v = __builtin_nontemporal_load(reinterpret_cast<const xu32x4*>(p));
    return make_uint4(v.x, v.y, v.z, v.w);
  } else {
    return *reinterpret_cast<const uint4*>(p);
  }
}
template <bool NT>
__device__ __forceinline__ uint2 ld_stream_u2(const void* p) {
  typedef unsigned xu32x2 __attribute__((ext_vector_type(2)));
  if constexpr (NT) {
    const xu32x2 v = __builtin_nontemporal_load(reinterpret_cast<const xu32x2*>(p));
    return make_uint2(v.x, v.y);
  } else {
    return *reinterpret_cast<const uint2*>(p);
  }
}
__device__ __forceinline__ float bf_lo(unsigned u) { return __uint_as_float(u << 16); }          // element 0 of a packed pair
__device__ __forceinline__ float bf_hi(unsigned u) { return __uint_as_float(u & 0xffff0000u); }  // element 1

// UK (round 4): UNFOLDED keys.  Kp then points at K [B, S_cap, D] (the k projection itself, head h in columns 96 h ..: 4x fewer
// bytes than the folded K'_h [S_cap, D] per head) and the query comes from the feed-forward launches in front of this block
// (sopro_skinny_args aux tiles): Qp = the K-slice partials of q_raw = Wq' x, summed here in slice order and scaled by the
// RMSNorm row scale of x.  V' stays folded (its output projection has no launch to ride on).
template <bool NT, bool HB, bool UK>
__global__ __launch_bounds__(512) void xattn_step_kernel(const sopro_xattn_args a) {
  __shared__ float xsum[XD];
  __shared__ float xn[XD];
  __shared__ float sc[64];
  __shared__ float ps[64];
  __shared__ float opart[4][XD];
  __shared__ float red[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = blockIdx.x, b = blockIdx.y;
  const int klen = a.klens ? min(a.klens[b], a.S_cap) : a.S_cap;
  constexpr int ES = HB ? 2 : 4;  // bytes per stored element
  constexpr int XDH = 96;  // head width of the unfolded keys (D / H)
  const char* Kb = UK ? reinterpret_cast<const char*>(a.Kp) + (((int64_t)b * a.S_cap) * XD + h * XDH) * ES
                      : reinterpret_cast<const char*>(a.Kp) + ((int64_t)(b * a.H + h) * a.S_cap) * XD * ES;
  const char* Vb = reinterpret_cast<const char*>(a.Vp) + ((int64_t)(b * a.H + h) * a.S_cap) * XD * ES;
  const int key = tid >> 3, part = tid & 7;   // score mapping: 8 lanes share a key; 16-byte piece f*8+part of its row each, so that one
                                              // load instruction covers whole 128-byte lines (8 per wave instead of 64)
  const int vd4 = tid % 96, vg = tid / 96;    // P.V' mapping: 4-column group, 16-key group (vg < 4)
  constexpr int KF = UK ? 3 : (HB ? 6 : 12);  // pieces per lane of a key row: 16-byte pieces of a K' row (768 / 1536 bytes over 8 lanes);
                                              // UK: 3 pieces of 4 elements (16 bytes fp32 / 8 bytes bf16) of the head's 96

  float m_run = -INFINITY, l_run = 0.f;
  float4 o4 = make_float4(0.f, 0.f, 0.f, 0.f);

  for (int k0 = 0; k0 < klen; k0 += 64) {
    // ---- every operand of this 64-key tile is requested up front
    uint4 kreg[KF];
    uint4 vreg[HB ? 8 : 16];  // HB: 16 keys x 4 bf16 (8 bytes) = two keys per register
    const bool kin = (k0 + key) < klen;
#pragma unroll
    for (int f = 0; f < KF; ++f) {
      kreg[f] = make_uint4(0u, 0u, 0u, 0u);
      if constexpr (UK && HB) {
        if (kin) { const uint2 u = ld_stream_u2<NT>(Kb + ((int64_t)(k0 + key) * XD) * ES + (f * 8 + part) * 8); kreg[f].x = u.x; kreg[f].y = u.y; }
      } else {
        if (kin) kreg[f] = ld_stream_u4<NT>(Kb + ((int64_t)(k0 + key) * XD) * ES + (f * 8 + part) * 16);
      }
    }
    if constexpr (HB) {
#pragma unroll
      for (int kk = 0; kk < 16; kk += 2) {
        uint2 v0 = make_uint2(0u, 0u), v1 = make_uint2(0u, 0u);
        if (vg < 4 && (k0 + vg * 16 + kk) < klen) v0 = ld_stream_u2<NT>(Vb + ((int64_t)(k0 + vg * 16 + kk) * XD + vd4 * 4) * 2);
        if (vg < 4 && (k0 + vg * 16 + kk + 1) < klen) v1 = ld_stream_u2<NT>(Vb + ((int64_t)(k0 + vg * 16 + kk + 1) * XD + vd4 * 4) * 2);
        vreg[kk >> 1] = make_uint4(v0.x, v0.y, v1.x, v1.y);
      }
    } else {
#pragma unroll
      for (int kk = 0; kk < 16; ++kk) {
        vreg[kk] = make_uint4(0u, 0u, 0u, 0u);
        if (vg < 4 && (k0 + vg * 16 + kk) < klen) vreg[kk] = ld_stream_u4<NT>(Vb + ((int64_t)(k0 + vg * 16 + kk) * XD + vd4 * 4) * 4);
      }
    }
    if (k0 == 0) {
      // ---- input stream (+ the producer's partial sums, fixed order), RMSNorm (src/sopro/nn/blocks.py:26-37)
      float ss = 0.f;
      if (tid < 96) {
        float4 xv = *reinterpret_cast<const float4*>(a.X + (int64_t)b * a.ldx + tid * 4);
        float4 pv[3];
#pragma unroll
        for (int s = 0; s < 3; ++s)
          if (s < a.np) pv[s] = *reinterpret_cast<const float4*>(a.Xp + (int64_t)s * a.xp_stride + (int64_t)b * a.ldx + tid * 4);
#pragma unroll
        for (int s = 0; s < 3; ++s)
          if (s < a.np) { xv.x += pv[s].x; xv.y += pv[s].y; xv.z += pv[s].z; xv.w += pv[s].w; }
        *reinterpret_cast<float4*>(xsum + tid * 4) = xv;
        ss = xv.x * xv.x + xv.y * xv.y + xv.z * xv.z + xv.w * xv.w;
      }
      ss = wave_sum(ss);
      if (lane == 0 && wave < 2) red[wave] = ss;
      __syncthreads();
      const float rstd = rsqrtf((red[0] + red[1]) / (float)XD + a.eps);
      if (tid < 96) {
        const float4 xv = *reinterpret_cast<const float4*>(xsum + tid * 4);
        float4 nw = make_float4(1.f, 1.f, 1.f, 1.f);  // NULL: the norm weight is folded into Kp by the host
        if (a.norm_w) nw = *reinterpret_cast<const float4*>(a.norm_w + tid * 4);
        float4 y;
        y.x = (xv.x * rstd) * nw.x; y.y = (xv.y * rstd) * nw.y; y.z = (xv.z * rstd) * nw.z; y.w = (xv.w * rstd) * nw.w;
        if constexpr (!UK) *reinterpret_cast<float4*>(xn + tid * 4) = y;
      }
      if constexpr (UK) {  // this head's query: the slices of q_raw in slice order, times the row scale  (xn[0 .. 95])
        if (tid < XDH / 4) {
          const float* qp = a.Qp + (int64_t)b * XD + h * XDH + tid * 4;
          float4 q = *reinterpret_cast<const float4*>(qp);
#pragma unroll
          for (int s = 1; s < 4; ++s)
            if (s < a.nqp) {
              const float4 t = *reinterpret_cast<const float4*>(qp + (int64_t)s * a.qp_stride);
              q.x += t.x; q.y += t.y; q.z += t.z; q.w += t.w;
            }
          q.x *= rstd; q.y *= rstd; q.z *= rstd; q.w *= rstd;
          *reinterpret_cast<float4*>(xn + tid * 4) = q;
        }
      }
      __syncthreads();
    }
    // ---- scores of this tile
    float s = 0.f;
    if constexpr (UK) {
#pragma unroll
      for (int f = 0; f < KF; ++f) {  // elements 4 * (f*8+part) .. + 3 of the head's 96
        const float4 q4 = *reinterpret_cast<const float4*>(xn + (f * 8 + part) * 4);
        if constexpr (HB) s += q4.x * bf_lo(kreg[f].x) + q4.y * bf_hi(kreg[f].x) + q4.z * bf_lo(kreg[f].y) + q4.w * bf_hi(kreg[f].y);
        else s += q4.x * __uint_as_float(kreg[f].x) + q4.y * __uint_as_float(kreg[f].y) + q4.z * __uint_as_float(kreg[f].z) + q4.w * __uint_as_float(kreg[f].w);
      }
    } else if constexpr (HB) {
#pragma unroll
      for (int f = 0; f < KF; ++f) {  // piece f*8+part of the row = elements 8 * (f*8+part) .. + 7
        const float4 q0 = *reinterpret_cast<const float4*>(xn + (f * 8 + part) * 8), q1 = *reinterpret_cast<const float4*>(xn + (f * 8 + part) * 8 + 4);
        s += q0.x * bf_lo(kreg[f].x) + q0.y * bf_hi(kreg[f].x) + q0.z * bf_lo(kreg[f].y) + q0.w * bf_hi(kreg[f].y);
        s += q1.x * bf_lo(kreg[f].z) + q1.y * bf_hi(kreg[f].z) + q1.z * bf_lo(kreg[f].w) + q1.w * bf_hi(kreg[f].w);
      }
    } else {
#pragma unroll
      for (int f = 0; f < KF; ++f) {
        const float4 q4 = *reinterpret_cast<const float4*>(xn + (f * 8 + part) * 4);
        s += q4.x * __uint_as_float(kreg[f].x) + q4.y * __uint_as_float(kreg[f].y) + q4.z * __uint_as_float(kreg[f].z) + q4.w * __uint_as_float(kreg[f].w);
      }
    }
    s += __shfl_xor(s, 1, 64);
    s += __shfl_xor(s, 2, 64);
    s += __shfl_xor(s, 4, 64);
    if (part == 0) sc[key] = kin ? s * a.scale : -INFINITY;
    __syncthreads();
    if (wave == 0) {
      const float v = sc[lane];
      const float m_new = fmaxf(m_run, wave_max(v));
      const float p = (v == -INFINITY) ? 0.f : expf(v - m_new);
      ps[lane] = p;
      const float lt = wave_sum(p);
      if (lane == 0) { red[0] = m_new; red[1] = lt; }
    }
    __syncthreads();
    const float m_new = red[0];
    const float alpha = (m_run == -INFINITY) ? 0.f : expf(m_run - m_new);
    l_run = l_run * alpha + red[1];
    m_run = m_new;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (vg < 4) {
      if constexpr (HB) {
#pragma unroll
        for (int kk = 0; kk < 16; kk += 2) {
          const float p0 = ps[vg * 16 + kk], p1 = ps[vg * 16 + kk + 1];
          const uint4 v = vreg[kk >> 1];
          acc.x += p0 * bf_lo(v.x); acc.y += p0 * bf_hi(v.x); acc.z += p0 * bf_lo(v.y); acc.w += p0 * bf_hi(v.y);
          acc.x += p1 * bf_lo(v.z); acc.y += p1 * bf_hi(v.z); acc.z += p1 * bf_lo(v.w); acc.w += p1 * bf_hi(v.w);
        }
      } else {
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
          const float p = ps[vg * 16 + kk];
          acc.x += p * __uint_as_float(vreg[kk].x); acc.y += p * __uint_as_float(vreg[kk].y);
          acc.z += p * __uint_as_float(vreg[kk].z); acc.w += p * __uint_as_float(vreg[kk].w);
        }
      }
    }
    o4.x = o4.x * alpha + acc.x; o4.y = o4.y * alpha + acc.y; o4.z = o4.z * alpha + acc.z; o4.w = o4.w * alpha + acc.w;
    __syncthreads();  // sc / ps / red are rewritten by the next tile
  }
  if (vg < 4) *reinterpret_cast<float4*>(&opart[vg][vd4 * 4]) = o4;
  __syncthreads();
  if (tid < 96) {
    const float inv = l_run > 0.f ? a.gate / l_run : 0.f;
    float4 y;
    const float4 p0 = *reinterpret_cast<const float4*>(&opart[0][tid * 4]);
    const float4 p1 = *reinterpret_cast<const float4*>(&opart[1][tid * 4]);
    const float4 p2 = *reinterpret_cast<const float4*>(&opart[2][tid * 4]);
    const float4 p3 = *reinterpret_cast<const float4*>(&opart[3][tid * 4]);
    y.x = (((p0.x + p1.x) + p2.x) + p3.x) * inv; y.y = (((p0.y + p1.y) + p2.y) + p3.y) * inv;
    y.z = (((p0.z + p1.z) + p2.z) + p3.z) * inv; y.w = (((p0.w + p1.w) + p2.w) + p3.w) * inv;
    if (h == 0) {
      const float4 xv = *reinterpret_cast<const float4*>(xsum + tid * 4);
      y.x += xv.x; y.y += xv.y; y.z += xv.z; y.w += xv.w;
    }
    *reinterpret_cast<float4*>(a.Y + (int64_t)h * a.y_part_stride + (int64_t)b * a.ldx + tid * 4) = y;
  }
}

}  // namespace

extern "C" int sopro_xattn_step_f32(const sopro_xattn_args* p, void* stream) {
  SOPRO_CHECK_ARG(p != nullptr, "args is NULL");
  const sopro_xattn_args& a = *p;
  SOPRO_CHECK_ARG(a.X && a.Kp && a.Vp && a.Y, "X, Kp, Vp, Y must be non-NULL");
  SOPRO_CHECK_ARG(a.D == XD && a.H >= 1 && a.B > 0 && a.S_cap > 0, "D must be 384, H >= 1");
  SOPRO_CHECK_ARG(a.np >= 0 && a.np <= 3 && (a.np == 0 || a.Xp), "np in 0..3");
  SOPRO_CHECK_ARG(aligned16(a.X) && aligned16(a.Kp) && aligned16(a.Vp) && aligned16(a.Y) && (!a.norm_w || aligned16(a.norm_w)) && (a.ldx & 3) == 0 &&
                      (a.xp_stride & 3) == 0 && (a.y_part_stride & 3) == 0,
                  "16-byte alignment / strides % 4");
  static const bool nt_on = !(SOPRO_DEV_ENV("SOPRO_XATTN_NT") != nullptr && SOPRO_DEV_ENV("SOPRO_XATTN_NT")[0] == '0');  // default on (r03: +1.7 %)
  // only where the operands cannot stay cached from one frame to the next anyway (> 8 MB per layer: the eight L2s hold 32 MB
  // for three layers); a single utterance's 0.8 MB per layer is L2-resident across frames and is asked for normally
  SOPRO_CHECK_ARG(a.kv_format == 0 || a.kv_format == 1, "kv_format: 0 (fp32 Kp / Vp) or 1 (bf16)");
  SOPRO_CHECK_ARG(a.k_unfolded == 0 || (a.k_unfolded == 1 && a.H * 96 == XD && a.Qp && aligned16(a.Qp) && a.nqp >= 1 && a.nqp <= 4 && (a.qp_stride & 3) == 0 && !a.norm_w),
                  "k_unfolded: Kp = K [B, S_cap, D] with 96-wide heads, Qp = 1..4 K-slice partials of the raw query (norm weight folded into it)");
  const bool hb = a.kv_format == 1, uk = a.k_unfolded == 1;
  const bool nt = nt_on && (int64_t)a.B * a.H * a.S_cap * XD * (hb ? 4 : 8) > ((int64_t)8 << 20);
  const dim3 grid(a.H, a.B), blk(512);
  hipStream_t st = (hipStream_t)stream;
#define SOPRO_XL(NTv, HBv, UKv) hipLaunchKernelGGL((xattn_step_kernel<NTv, HBv, UKv>), grid, blk, 0, st, a)
  if (uk) {
    if (hb) { if (nt) SOPRO_XL(true, true, true); else SOPRO_XL(false, true, true); }
    else { if (nt) SOPRO_XL(true, false, true); else SOPRO_XL(false, false, true); }
  } else {
    if (hb) { if (nt) SOPRO_XL(true, true, false); else SOPRO_XL(false, true, false); }
    else { if (nt) SOPRO_XL(true, false, false); else SOPRO_XL(false, false, false); }
  }
#undef SOPRO_XL
  SOPRO_LAUNCH_CHECK();
}
