// Device-resident driver state of the autoregressive loop: the sampler, the anti-loop policy,
// the EOS rule and the construction of the next frame's input all run on the GPU so that one
// frame is a fixed launch sequence with no host round trip (the reference synchronises the host
// several times per frame in sample_token, src/sopro/sampling.py:64-93) and can be replayed
// from a hipGraph.
#include "common.h"

namespace {

constexpr int SAMP_THREADS = 256;

__device__ __forceinline__ unsigned ord_f32(float v) {
  const unsigned u = __float_as_uint(v);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__device__ __forceinline__ void philox_round(unsigned& c0, unsigned& c1, unsigned& c2, unsigned& c3, unsigned k0, unsigned k1) {
  const unsigned long long p0 = 0xD2511F53ull * c0, p1 = 0xCD9E8D57ull * c2;
  const unsigned h0 = (unsigned)(p0 >> 32), l0 = (unsigned)p0, h1 = (unsigned)(p1 >> 32), l1 = (unsigned)p1;
  c0 = h1 ^ c1 ^ k0; c1 = l1; c2 = h0 ^ c3 ^ k1; c3 = l0;
}
__device__ float philox_uniform(unsigned long long seed, unsigned t, unsigned b) {
  unsigned c0 = t, c1 = b, c2 = 0x5090u, c3 = 0u;
  unsigned k0 = (unsigned)seed, k1 = (unsigned)(seed >> 32);
  for (int i = 0; i < 10; ++i) {
    philox_round(c0, c1, c2, c3, k0, k1);
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  return (float)(c0 >> 8) * (1.0f / 16777216.0f);
}

__global__ void ar_init_kernel(const sopro_ar_state st) {
  const int b = blockIdx.x;
  for (int c = threadIdx.x; c < st.D; c += blockDim.x)
    st.x_cur[(int64_t)b * st.D + c] = st.cond[((int64_t)b * st.Tar) * st.D + c] + st.emb[(int64_t)st.bos_row * st.D + c];
  if (threadIdx.x == 0) {
    st.first_eos[b] = -1;
    st.stop_t[b] = -1;
    if (b == 0) { *st.step = 0; *st.arrive = 0; *st.n_stopped = 0; }
  }
}

// wave-wide bitonic sort (descending) of one 64-bit key per lane
__device__ __forceinline__ unsigned long long wave_sort_desc(unsigned long long key, int lane) {
#pragma unroll
  for (int k = 2; k <= 64; k <<= 1) {
#pragma unroll
    for (int j = k >> 1; j > 0; j >>= 1) {
      const unsigned long long other = __shfl_xor(key, j, 64);
      const bool desc = (lane & k) == 0;
      const bool lower = (lane & j) == 0;
      const unsigned long long mx = key > other ? key : other, mn = key > other ? other : key;
      key = (lower == desc) ? mx : mn;
    }
  }
  return key;
}

__global__ __launch_bounds__(SAMP_THREADS) void ar_sample_kernel(const sopro_ar_state st, const float* __restrict__ logits,
                                                                 int64_t ld) {
  __shared__ float xs[2049 + 7];
  __shared__ unsigned hist8[256];
  __shared__ unsigned wtot[4];
  __shared__ unsigned long long redk[4];
  __shared__ float redf[4];
  __shared__ unsigned long long sel[64];
  __shared__ float selp[64];
  __shared__ int sh_flag, sh_tok, sh_t;
  __shared__ unsigned sh_digit, sh_need, sh_cnt;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.x;
  const int V1 = st.V + 1;
  if (tid == 0) { sh_t = *st.step; sh_flag = 0; sh_cnt = 0; }
  __syncthreads();
  const int t = sh_t;
  if (t >= st.max_steps) return;  // uniform
  const int* hist = st.hist + (int64_t)b * st.max_steps;

  // ---- anti-loop policy (src/sopro/model.py:274-279, sampling.py:16-21) on the t tokens so far
  const bool anti = st.params[2] != 0.f;
  if (anti) {
    if (tid >= 3 && tid <= 16) {
      const int n = tid;
      if (2 * n <= t) {
        bool same = true;
        for (int i = 0; i < n; ++i) same = same && (hist[t - n + i] == hist[t - 2 * n + i]);
        if (same) atomicOr(&sh_flag, 1);
      }
    } else if (tid == 17 && t >= 9) {
      bool same = true;
      for (int i = 1; i < 9; ++i) same = same && (hist[t - 1 - i] == hist[t - 1]);
      if (same) atomicOr(&sh_flag, 1);
    }
  }
  // ---- nan_to_num (sampling.py:33-35); temperature needs the policy flag, applied after the barrier
  const float* lg = logits + (int64_t)b * ld;
  constexpr int PER = (2049 + SAMP_THREADS - 1) / SAMP_THREADS;
  float xv[PER];
#pragma unroll
  for (int q = 0; q < PER; ++q) {
    const int i = tid + q * SAMP_THREADS;
    float v = (i < V1) ? lg[i] : -INFINITY;
    if (v != v) v = -1e9f;
    else if (v == INFINITY) v = 1e9f;
    else if (v == -INFINITY && i < V1) v = -1e9f;
    xv[q] = v;
  }
  __syncthreads();
  const bool recover = sh_flag != 0;
  const float top_p = recover ? st.params[3] : st.params[0];
  const float temp = recover ? st.params[4] : st.params[1];
  const float rep = st.params[5];
  const int top_k = (int)st.params[6];
  const int min_gen = (int)st.params[7];
#pragma unroll
  for (int q = 0; q < PER; ++q) {
    const int i = tid + q * SAMP_THREADS;
    if (i < V1) xs[i] = (temp != 0.f && temp != 1.0f) ? xv[q] / temp : xv[q];  // sampling.py:37-38
  }
  __syncthreads();
  // ---- repetition penalty on the unique ids among the last 50 tokens (sampling.py:40-50)
  if (rep != 1.0f && tid < 50 && tid < t) {
    const int id = hist[t - 1 - tid];
    bool first = true;
    for (int j = 0; j < tid; ++j) first = first && (hist[t - 1 - j] != id);
    if (first && id >= 0 && id < V1) {
      const float v = xs[id];
      xs[id] = v < 0.f ? v * rep : v / rep;
    }
  }
  __syncthreads();

  // 44-bit unique sort key: (order-preserving logit bits, 4095 - index): larger == better, ties -> lower index
  unsigned long long key[PER];
  unsigned long long best = 0ull;
  float xmax_l = -INFINITY;
#pragma unroll
  for (int q = 0; q < PER; ++q) {
    const int i = tid + q * SAMP_THREADS;
    key[q] = (i < V1) ? (((unsigned long long)ord_f32(xs[i]) << 12) | (unsigned long long)(4095 - i)) : 0ull;
    best = key[q] > best ? key[q] : best;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const unsigned long long ok = __shfl_xor(best, o, 64);
    best = ok > best ? ok : best;
  }
  if (lane == 0) redk[wave] = best;
  __syncthreads();
  {
    unsigned long long m = redk[0];
#pragma unroll
    for (int w = 1; w < SAMP_THREADS / 64; ++w) m = redk[w] > m ? redk[w] : m;
    best = m;
  }
  const int top_i = 4095 - (int)(best & 4095ull);

  const bool greedy = !(top_p > 0.f);
  const int kk = (top_k > 0) ? min(top_k, V1) : V1;
  if (greedy || kk > 64) {
    // top_p <= 0 keeps only the head of the sorted distribution == arg-max of the penalised logits.
    // (top_k > 64 or "no top-k" is not used by the reference policy, model.py:289; it falls back to the head too.)
    if (tid == 0) sh_tok = top_i;
  } else {
    // ---- radix select of the kk-th largest key: 6 digits (8,8,8,8,8,4 bits) of the 44-bit key
    unsigned long long prefix = 0ull, mask = 0ull;
    unsigned need = (unsigned)kk;
#pragma unroll 1
    for (int pass = 0; pass < 6; ++pass) {
      const int shift = pass < 5 ? 36 - 8 * pass : 0;
      const unsigned dmask = pass < 5 ? 255u : 15u;
      hist8[tid] = 0u;
      __syncthreads();
#pragma unroll
      for (int q = 0; q < PER; ++q) {
        const int i = tid + q * SAMP_THREADS;
        if (i < V1 && (key[q] & mask) == prefix) atomicAdd(&hist8[(unsigned)(key[q] >> shift) & dmask], 1u);
      }
      __syncthreads();
      // suffix counts S[d] = #keys with digit >= d (wave suffix scan + cross-wave offsets)
      unsigned v = hist8[tid];
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const unsigned u = __shfl_down(v, o, 64);
        if (lane + o < 64) v += u;
      }
      if (lane == 0) wtot[wave] = v;
      __syncthreads();
      unsigned above = 0u;
      for (int w = wave + 1; w < 4; ++w) above += wtot[w];
      const unsigned S = v + above;                 // digits >= tid
      const unsigned Snext = S - hist8[tid];        // digits >  tid
      if (S >= need && Snext < need) { sh_digit = (unsigned)tid; sh_need = need - Snext; }
      __syncthreads();
      prefix |= (unsigned long long)sh_digit << shift;
      mask |= (unsigned long long)dmask << shift;
      need = sh_need;
    }
    // prefix is now exactly the kk-th largest key: collect everything >= it (kk unique keys)
#pragma unroll
    for (int q = 0; q < PER; ++q) {
      const int i = tid + q * SAMP_THREADS;
      if (i < V1 && key[q] >= prefix) {
        const unsigned slot = atomicAdd(&sh_cnt, 1u);
        if (slot < 64u) sel[slot] = key[q];
      }
    }
    // softmax denominator over the whole row (sampling.py:52)
    const float xmax = xs[top_i];
    float z = 0.f;
#pragma unroll
    for (int q = 0; q < PER; ++q) {
      const int i = tid + q * SAMP_THREADS;
      if (i < V1) z += expf(xs[i] - xmax);
    }
    z = wave_sum(z);
    if (lane == 0) redf[wave] = z;
    __syncthreads();
    if (wave == 0) {
      const float Z = (redf[0] + redf[1]) + (redf[2] + redf[3]);
      unsigned long long k64 = (lane < kk) ? sel[lane] : 0ull;
      k64 = wave_sort_desc(k64, lane);
      const int id = 4095 - (int)(k64 & 4095ull);
      const float p = (lane < kk) ? expf(xs[min(id, V1 - 1)] - xmax) / Z : 0.f;
      sel[lane] = k64;
      selp[lane] = p;
      __builtin_amdgcn_wave_barrier();
      if (lane == 0) {
        // top-k renormalisation (sampling.py:56-66), serial like torch's cumsum
        float s = 0.f;
        for (int j = 0; j < kk; ++j) s += selp[j];
        int tok = top_i;
        if (s > 1e-12f) {
          // top-p: drop entry j when the cumulative mass *before* it already exceeds top_p (sampling.py:68-76)
          float cum = 0.f, kept = 0.f;
          int nkeep = 0;
          for (int j = 0; j < kk; ++j) {
            const float pj = selp[j] / s;
            const bool remove = (top_p < 1.0f) && (j > 0) && (cum > top_p);
            cum += pj;
            if (remove) break;  // cum is monotone: everything after is removed too
            kept += pj;
            nkeep = j + 1;
          }
          if (kept > 1e-12f) {
            const float u = philox_uniform(st.seed, (unsigned)t, (unsigned)b) * kept;
            float c2 = 0.f;
            int pick = nkeep - 1;
            for (int j = 0; j < nkeep; ++j) {
              c2 += selp[j] / s;
              if (u < c2) { pick = j; break; }
            }
            tok = 4095 - (int)(sel[pick] & 4095ull);
          }
        }
        sh_tok = tok;
      }
    }
  }
  __syncthreads();
  const int tok = sh_tok;

  // ---- bookkeeping: history, EOS rule (model.py:293-305), next input (model.py:266-272)
  if (tid == 0) {
    st.hist[(int64_t)b * st.max_steps + t] = tok;
    if (tok == st.V) {
      if (st.first_eos[b] < 0) st.first_eos[b] = t;
      if (st.stop_t[b] < 0 && (t + 1) >= min_gen) {
        st.stop_t[b] = t;
        atomicAdd(st.n_stopped, 1);
      }
    }
  }
  if (t + 1 < st.Tar) {
    const float* c = st.cond + ((int64_t)b * st.Tar + (t + 1)) * st.D;
    const float* e = st.emb + (int64_t)tok * st.D;
    for (int d = tid; d < st.D; d += SAMP_THREADS) st.x_cur[(int64_t)b * st.D + d] = c[d] + e[d];
  }
  if (tid == 0) {
    __threadfence();
    const int old = atomicAdd(st.arrive, 1);
    if (old == st.B - 1) {
      *st.arrive = 0;
      *st.step = t + 1;
    }
  }
}

}  // namespace

extern "C" {

static int check_state(const sopro_ar_state* st) {
  SOPRO_CHECK_ARG(st != nullptr, "state is NULL");
  SOPRO_CHECK_ARG(st->x_cur && st->cond && st->emb && st->hist && st->step && st->arrive && st->first_eos && st->stop_t &&
                      st->n_stopped && st->params,
                  "state has NULL pointers");
  SOPRO_CHECK_ARG(st->B > 0 && st->D > 0 && st->Tar > 0 && st->max_steps > 0 && st->V > 0 && st->V + 1 <= 2049,
                  "bad sizes (V <= 2048)");
  return 0;
}

int sopro_ar_init(const sopro_ar_state* st, void* stream) {
  if (int rc = check_state(st)) return rc;
  hipLaunchKernelGGL(ar_init_kernel, dim3(st->B), dim3(128), 0, (hipStream_t)stream, *st);
  SOPRO_LAUNCH_CHECK();
}

int sopro_ar_sample(const sopro_ar_state* st, const float* logits, int64_t ld_logits, void* stream) {
  if (int rc = check_state(st)) return rc;
  SOPRO_CHECK_ARG(logits && ld_logits >= st->V + 1, "bad logits");
  hipLaunchKernelGGL(ar_sample_kernel, dim3(st->B), dim3(SAMP_THREADS), 0, (hipStream_t)stream, *st, logits, ld_logits);
  SOPRO_LAUNCH_CHECK();
}

}  // extern "C"
