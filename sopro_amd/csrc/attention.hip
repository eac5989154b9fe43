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
  const float* Kb = a.K + (int64_t)b * a.k_bstride + h * DH;
  const float* Vb = a.V + (int64_t)b * a.v_bstride + h * DH;
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
  static bool attr_done = false;
  auto kern = attention_kernel<DH>;
  if (!attr_done) {
    SOPRO_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_done = true;
  }
  dim3 grid((a.Tq + TQ - 1) / TQ, a.H, a.B);
  hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, a);
  SOPRO_LAUNCH_CHECK();
}

}  // namespace

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
  switch (a.dh) {
    case 64: return launch_attn<64>(a, s);
    case 96: return launch_attn<96>(a, s);
    case 192: return launch_attn<192>(a, s);
    default: sopro_set_error("sopro_attention_f32: unsupported head dim %d (64, 96, 192)", a.dh); return -2;
  }
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
