// Device-resident driver state of the autoregressive loop: the sampler, the anti-loop policy,
// the EOS rule and the construction of the next frame's input all run on the GPU so that one
// frame is a fixed launch sequence with no host round trip (the reference synchronises the host
// several times per frame in sample_token, src/sopro/sampling.py:64-93) and can be replayed
// from a hipGraph.
#include "common.h"

namespace {

constexpr int SAMP_THREADS = 1024;
constexpr int SORT_N = 4096;

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

__global__ __launch_bounds__(SAMP_THREADS) void ar_sample_kernel(const sopro_ar_state st, const float* __restrict__ logits,
                                                                 int64_t ld) {
  __shared__ unsigned long long keys[SORT_N];
  __shared__ float xs[2049 + 7];
  __shared__ float redf[SAMP_THREADS / 64];
  __shared__ unsigned long long redk[SAMP_THREADS / 64];
  __shared__ int sh_flag, sh_tok, sh_t;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.x;
  const int V1 = st.V + 1;
  if (tid == 0) { sh_t = *st.step; sh_flag = 0; }
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
  __syncthreads();
  const bool recover = sh_flag != 0;
  const float top_p = recover ? st.params[3] : st.params[0];
  const float temp = recover ? st.params[4] : st.params[1];
  const float rep = st.params[5];
  const int top_k = (int)st.params[6];
  const int min_gen = (int)st.params[7];

  // ---- nan_to_num, temperature (sampling.py:33-38)
  const float* lg = logits + (int64_t)b * ld;
  for (int i = tid; i < V1; i += SAMP_THREADS) {
    float v = lg[i];
    if (v != v) v = -1e9f;
    else if (v == INFINITY) v = 1e9f;
    else if (v == -INFINITY) v = -1e9f;
    if (temp != 0.f && temp != 1.0f) v = v / temp;
    xs[i] = v;
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

  const bool greedy = !(top_p > 0.f);
  if (greedy) {
    // top_p <= 0 keeps only the head of the sorted distribution == arg-max of the penalised logits
    unsigned long long best = 0ull;
    for (int i = tid; i < V1; i += SAMP_THREADS) {
      const unsigned long long k = ((unsigned long long)ord_f32(xs[i]) << 32) | (unsigned)(0xFFFFFFFFu - (unsigned)i);
      best = k > best ? k : best;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const unsigned long long ok = __shfl_xor(best, o, 64);
      best = ok > best ? ok : best;
    }
    if (lane == 0) redk[wave] = best;
    __syncthreads();
    if (tid == 0) {
      unsigned long long m = redk[0];
      for (int w = 1; w < SAMP_THREADS / 64; ++w) m = redk[w] > m ? redk[w] : m;
      sh_tok = (int)(0xFFFFFFFFu - (unsigned)(m & 0xFFFFFFFFull));
    }
  } else {
    // ---- full descending sort of (logit, index): equivalent to sorting the softmax (monotone)
    for (int i = tid; i < SORT_N; i += SAMP_THREADS)
      keys[i] = (i < V1) ? (((unsigned long long)ord_f32(xs[i]) << 32) | (unsigned)(0xFFFFFFFFu - (unsigned)i)) : 0ull;
    __syncthreads();
    for (int k = 2; k <= SORT_N; k <<= 1) {
      for (int j = k >> 1; j > 0; j >>= 1) {
        for (int i = tid; i < SORT_N; i += SAMP_THREADS) {
          const int ixj = i ^ j;
          if (ixj > i) {
            const unsigned long long a = keys[i], c = keys[ixj];
            const bool desc = (i & k) == 0;
            if (desc ? (a < c) : (a > c)) { keys[i] = c; keys[ixj] = a; }
          }
        }
        __syncthreads();
      }
    }
    // softmax denominator over the whole row (sampling.py:52)
    const int top_i = (int)(0xFFFFFFFFu - (unsigned)(keys[0] & 0xFFFFFFFFull));
    const float xmax = xs[top_i];
    float z = 0.f;
    for (int i = tid; i < V1; i += SAMP_THREADS) z += expf(xs[i] - xmax);
    z = wave_sum(z);
    if (lane == 0) redf[wave] = z;
    __syncthreads();
    if (tid == 0) {
      float Z = 0.f;
      for (int w = 0; w < SAMP_THREADS / 64; ++w) Z += redf[w];
      const int kk = (top_k > 0) ? min(top_k, V1) : V1;
      // top-k renormalisation (sampling.py:56-66); the candidates are the first kk sorted entries
      float s = 0.f;
      for (int j = 0; j < kk; ++j) {
        const int id = (int)(0xFFFFFFFFu - (unsigned)(keys[j] & 0xFFFFFFFFull));
        const float p = expf(xs[id] - xmax) / Z;
        s += p;
      }
      int tok = top_i;
      if (s > 1e-12f) {
        // top-p: drop entry j when the cumulative mass *before* it already exceeds top_p (sampling.py:68-76)
        float cum = 0.f, kept = 0.f;
        int nkeep = 0;
        for (int j = 0; j < kk; ++j) {
          const int id = (int)(0xFFFFFFFFu - (unsigned)(keys[j] & 0xFFFFFFFFull));
          const float p = (expf(xs[id] - xmax) / Z) / s;
          const bool remove = (top_p < 1.0f) && (j > 0) && (cum > top_p);
          cum += p;
          if (remove) break;  // cum is monotone: everything after is removed too
          kept += p;
          nkeep = j + 1;
        }
        if (kept > 1e-12f) {
          const float u = philox_uniform(st.seed, (unsigned)t, (unsigned)b) * kept;
          float c2 = 0.f;
          int pick = nkeep - 1;
          for (int j = 0; j < nkeep; ++j) {
            const int id = (int)(0xFFFFFFFFu - (unsigned)(keys[j] & 0xFFFFFFFFull));
            c2 += (expf(xs[id] - xmax) / Z) / s;
            if (u < c2) { pick = j; break; }
          }
          tok = (int)(0xFFFFFFFFu - (unsigned)(keys[pick] & 0xFFFFFFFFull));
        }
      }
      sh_tok = tok;
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
