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
    for (int j = 0; j < 64; ++j) st.recent[(int64_t)b * 64 + j] = -1;
    if (st.start) st.start[b] = -1;  // slot mode: every slot starts free
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

// number of entries of a descending 64-entry array that are greater than x (binary search, <= 7 probes)
__device__ __forceinline__ int count_greater64(const unsigned long long* arr, unsigned long long x) {
  int lo = 0, hi = 64;  // invariant: arr[0..lo) > x >= arr[hi..64)
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (arr[mid] > x) lo = mid + 1; else hi = mid;
  }
  return lo;
}

constexpr int RECENT = 64;  // rolling token window per row: slot j = token sampled j+1 frames ago (-1 = none)
constexpr int MAXCAND = 9 * 64;  // PER * max kk: the candidate bound of the scheme below

// Top-k without a sort: the kk-th largest of the 256 per-thread maxima is a lower bound of the kk-th largest
// logit, so "key >= that bound" keeps at most 9*kk candidates (all of the true top-kk among them); exact ranks
// among the candidates come from an all-pairs count, which also lays them out in descending order.
__global__ __launch_bounds__(SAMP_THREADS) void ar_sample_kernel(const sopro_ar_state st, const float* __restrict__ logits,
                                                                 int64_t ld) {
  __shared__ float xs[2049 + 7];
  __shared__ unsigned long long lmax[SAMP_THREADS];
  __shared__ unsigned long long cand[MAXCAND];
  __shared__ unsigned long long selk[64];
  __shared__ float selp[64];
  __shared__ float redf[4];
  __shared__ int rs[RECENT];
  __shared__ unsigned long long sh_thr, sh_best;
  __shared__ unsigned sh_cnt;
  __shared__ int sh_tok;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.x;
  const int V1 = st.V + 1;
  constexpr int PER = (2049 + SAMP_THREADS - 1) / SAMP_THREADS;

  // ---- the only up-front memory round: frame index, policy parameters, this row's logits, its recent tokens
  // Classic mode: every row started at global frame 0.  Slot mode (st.start != NULL, continuous batching): row b was
  // admitted at global frame start[b] (-1 = free slot) with its own frame budget and parameters; its time is local.
  const int tg = *st.step;
  const int s0 = st.start ? st.start[b] : 0;
  const int t = tg - s0;
  const int tmax = (st.start && st.row_max) ? st.row_max[b] : st.Tar;
  const float* prm = st.row_params ? st.row_params + (int64_t)b * 8 : st.params;
  const float p_top_p = prm[0], p_temp = prm[1], p_anti = prm[2], p_rec_p = prm[3];
  const float p_rec_t = prm[4], rep = prm[5];
  const int top_k = (int)prm[6], min_gen = (int)prm[7];
  const float* lg = logits + (int64_t)b * ld;
  float xv[PER];
#pragma unroll
  for (int q = 0; q < PER; ++q) {
    const int i = tid + q * SAMP_THREADS;
    xv[q] = (i < V1) ? lg[i] : 0.f;
  }
  int* recent = st.recent + (int64_t)b * RECENT;
  const int r_mine = (tid < RECENT) ? recent[tid] : -1;
  if (!st.start && tg >= st.max_steps) return;  // classic mode, uniform over the grid: the loop is over
  if (st.start && (s0 < 0 || t >= min(tmax, st.max_steps))) {
    // free slot, or a row past its budget waiting to be harvested: nothing to sample, but the frame still needs its ticket
    if (tid == 0) {
      __threadfence();
      const int old = atomicAdd(st.arrive, 1);
      if (old == st.B - 1) {
        *st.arrive = 0;
        *st.step = tg + 1;
      }
    }
    return;
  }
  // next frame's conditioning row: address needs t only, consumed at the very end
  float cnext[2] = {0.f, 0.f};
  if (t + 1 < tmax) {
    const float* c = st.cond + ((int64_t)b * st.Tar + (t + 1)) * st.D;
#pragma unroll
    for (int q = 0; q < 2; ++q)
      if (tid + q * SAMP_THREADS < st.D) cnext[q] = c[tid + q * SAMP_THREADS];
  }
  if (tid < RECENT) rs[tid] = r_mine;
  if (tid == 0) sh_cnt = 0u;
  __syncthreads();

  // ---- anti-loop policy (src/sopro/model.py:274-279, sampling.py:16-21): repeated tail of length 3..16, or 9 equal tokens
  int loopy = 0;
  if (p_anti != 0.f) {
    if (tid >= 3 && tid <= 16) {
      const int n = tid;
      if (rs[2 * n - 1] >= 0) {
        bool same = true;
        for (int m2 = 0; m2 < n; ++m2) same = same && (rs[m2] == rs[m2 + n]);
        loopy = same;
      }
    } else if (tid == 17 && rs[8] >= 0) {
      bool same = true;
      for (int m2 = 1; m2 < 9; ++m2) same = same && (rs[m2] == rs[0]);
      loopy = same;
    }
  }
  const bool recover = __syncthreads_or(loopy) != 0;
  const float top_p = recover ? p_rec_p : p_top_p;
  const float temp = recover ? p_rec_t : p_temp;
  // ---- nan_to_num, temperature (sampling.py:33-38)
#pragma unroll
  for (int q = 0; q < PER; ++q) {
    const int i = tid + q * SAMP_THREADS;
    float v = xv[q];
    if (v != v) v = -1e9f;
    else if (v == INFINITY) v = 1e9f;
    else if (v == -INFINITY) v = -1e9f;
    if (temp != 0.f && temp != 1.0f) v = v / temp;
    if (i < V1) xs[i] = v;
  }
  __syncthreads();
  // ---- repetition penalty on the unique ids among the last 50 tokens (sampling.py:40-50)
  if (rep != 1.0f && tid < 50) {
    const int id = rs[tid];
    if (id >= 0 && id < V1) {
      bool first = true;
      for (int j = 0; j < tid; ++j) first = first && (rs[j] != id);
      if (first) {
        const float v = xs[id];
        xs[id] = v < 0.f ? v * rep : v / rep;
      }
    }
  }
  __syncthreads();

  // 44-bit unique sort key: (order-preserving logit bits, 4095 - index): larger == better, ties -> lower index
  unsigned long long key[PER];
  unsigned long long lm = 0ull;
#pragma unroll
  for (int q = 0; q < PER; ++q) {
    const int i = tid + q * SAMP_THREADS;
    key[q] = (i < V1) ? (((unsigned long long)ord_f32(xs[i]) << 12) | (unsigned long long)(4095 - i)) : 0ull;
    lm = key[q] > lm ? key[q] : lm;
  }
  // per-wave sort of the 64 per-thread maxima, then each lane ranks its entry against the other three sorted lists
  const unsigned long long lms = wave_sort_desc(lm, lane);  // lane i: i-th largest maximum of this wave
  lmax[tid] = lms;
  __syncthreads();
  const bool greedy = !(top_p > 0.f);
  int kk = (top_k > 0) ? min(top_k, V1) : V1;
  const bool head_only = greedy || kk > 64;  // top_k > 64 / "no top-k" is outside the reference policy (model.py:289)
  {
    int rank = lane;
#pragma unroll
    for (int w2 = 0; w2 < SAMP_THREADS / 64; ++w2)
      if (w2 != wave) rank += count_greater64(lmax + w2 * 64, lms);
    if (rank == 0) sh_best = lms;
    if (!head_only && rank == kk - 1) sh_thr = lms;
  }
  __syncthreads();
  const int top_i = 4095 - (int)(sh_best & 4095ull);
  if (head_only) {
    // top_p <= 0 keeps only the head of the sorted distribution == arg-max of the penalised logits
    if (tid == 0) sh_tok = top_i;
  } else {
    const unsigned long long thr = sh_thr;
    const float xmax = xs[top_i];
    float z = 0.f;
#pragma unroll
    for (int q = 0; q < PER; ++q) {
      const int i = tid + q * SAMP_THREADS;
      if (i < V1) {
        z += expf(xs[i] - xmax);  // softmax denominator over the whole row (sampling.py:52)
        if (key[q] >= thr) {
          const unsigned slot = atomicAdd(&sh_cnt, 1u);
          if (slot < (unsigned)MAXCAND) cand[slot] = key[q];
        }
      }
    }
    z = wave_sum(z);
    if (lane == 0) redf[wave] = z;
    __syncthreads();
    const int C = min((int)sh_cnt, MAXCAND);
    const float Z = (redf[0] + redf[1]) + (redf[2] + redf[3]);
    if (C <= 64) {
      // usual case: one wave sorts the candidates directly (lane j ends up with the j-th largest)
      if (wave == 0) {
        const unsigned long long ks = wave_sort_desc(lane < C ? cand[lane] : 0ull, lane);
        if (lane < kk) {
          selk[lane] = ks;
          selp[lane] = expf(xs[4095 - (int)(ks & 4095ull)] - xmax) / Z;
        }
      }
    } else {
      for (int c = tid; c < C; c += SAMP_THREADS) {
        const unsigned long long kc = cand[c];
        int rank = 0;
        for (int j = 0; j < C; ++j) rank += (cand[j] > kc) ? 1 : 0;
        if (rank < kk) {
          selk[rank] = kc;
          selp[rank] = expf(xs[4095 - (int)(kc & 4095ull)] - xmax) / Z;
        }
      }
    }
    __syncthreads();
    if (wave == 0) {
      // top-k renormalisation, top-p cut and the draw on one wave (sampling.py:56-93): lane j owns sorted entry j
      const float pj_raw = (lane < kk) ? selp[lane] : 0.f;
      const float s = wave_sum(pj_raw);
      int tok = top_i;
      if (s > 1e-12f) {
        const float pj = pj_raw / s;
        float inc = pj;  // inclusive prefix sum over lanes
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
          const float u2 = __shfl_up(inc, o, 64);
          if (lane >= o) inc += u2;
        }
        const float before = inc - pj;
        // drop entry j when the cumulative mass *before* it already exceeds top_p; entry 0 always stays (sampling.py:68-76)
        const bool keep = lane < kk && (lane == 0 || !(top_p < 1.0f && before > top_p));
        const float kp = keep ? pj : 0.f;
        const float kept = wave_sum(kp);
        if (kept > 1e-12f) {
          float kinc = kp;
#pragma unroll
          for (int o = 1; o < 64; o <<= 1) {
            const float u2 = __shfl_up(kinc, o, 64);
            if (lane >= o) kinc += u2;
          }
          const float u = philox_uniform(st.seed, (unsigned)t, (unsigned)b) * kept;
          const unsigned long long hit = __ballot(keep && u < kinc);
          const unsigned long long kmask = __ballot(keep);
          const int pick = hit ? (int)__ffsll((long long)hit) - 1 : 63 - __clzll((long long)kmask);
          tok = 4095 - (int)(selk[pick] & 4095ull);
        }
      }
      if (lane == 0) sh_tok = tok;
    }
  }
  __syncthreads();
  const int tok = sh_tok;

  // ---- bookkeeping: history, EOS rule (model.py:293-305), next input (model.py:266-272)
  if (t + 1 < tmax) {
    const float* e = st.emb + (int64_t)tok * st.D;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int d = tid + q * SAMP_THREADS;
      if (d < st.D) st.x_cur[(int64_t)b * st.D + d] = cnext[q] + e[d];
    }
  }
  if (tid < RECENT) recent[tid] = (tid == 0) ? tok : rs[tid - 1];
  if (tid == 0) {
    st.hist[(int64_t)b * st.max_steps + t] = tok;
    if (tok == st.V) {
      if (st.first_eos[b] < 0) st.first_eos[b] = t;
      if (st.stop_t[b] < 0 && (t + 1) >= min_gen) {
        st.stop_t[b] = t;
        atomicAdd(st.n_stopped, 1);
      }
    }
    __threadfence();
    const int old = atomicAdd(st.arrive, 1);
    if (old == st.B - 1) {
      *st.arrive = 0;
      *st.step = tg + 1;
    }
  }
}

// Slot mode: (re)start row `row` at the current global frame.  Stream-ordered between two frames, so *step is stable.
__global__ void ar_admit_kernel(const sopro_ar_state st, int row) {
  const int b = row;
  for (int c = threadIdx.x; c < st.D; c += blockDim.x)
    st.x_cur[(int64_t)b * st.D + c] = st.cond[((int64_t)b * st.Tar) * st.D + c] + st.emb[(int64_t)st.bos_row * st.D + c];
  for (int j = threadIdx.x; j < 64; j += blockDim.x) st.recent[(int64_t)b * 64 + j] = -1;
  if (threadIdx.x == 0) {
    st.first_eos[b] = -1;
    st.stop_t[b] = -1;
    st.start[b] = *st.step;
  }
}

}  // namespace

extern "C" {

static int check_state(const sopro_ar_state* st) {
  SOPRO_CHECK_ARG(st != nullptr, "state is NULL");
  SOPRO_CHECK_ARG(st->x_cur && st->cond && st->emb && st->hist && st->step && st->arrive && st->first_eos && st->stop_t &&
                      st->n_stopped && st->params && st->recent,
                  "state has NULL pointers");
  SOPRO_CHECK_ARG(st->B > 0 && st->D > 0 && st->Tar > 0 && st->max_steps > 0 && st->V > 0 && st->V + 1 <= 2049 && st->D <= 512,
                  "bad sizes (V <= 2048, D <= 512)");
  return 0;
}

int sopro_ar_init(const sopro_ar_state* st, void* stream) {
  if (int rc = check_state(st)) return rc;
  hipLaunchKernelGGL(ar_init_kernel, dim3(st->B), dim3(128), 0, (hipStream_t)stream, *st);
  SOPRO_LAUNCH_CHECK();
}

int sopro_ar_admit(const sopro_ar_state* st, int32_t row, void* stream) {
  if (int rc = check_state(st)) return rc;
  SOPRO_CHECK_ARG(st->start != nullptr, "sopro_ar_admit needs slot mode (state.start)");
  SOPRO_CHECK_ARG(row >= 0 && row < st->B, "row out of range");
  hipLaunchKernelGGL(ar_admit_kernel, dim3(1), dim3(128), 0, (hipStream_t)stream, *st, (int)row);
  SOPRO_LAUNCH_CHECK();
}

int sopro_ar_sample(const sopro_ar_state* st, const float* logits, int64_t ld_logits, void* stream) {
  if (int rc = check_state(st)) return rc;
  SOPRO_CHECK_ARG(logits && ld_logits >= st->V + 1, "bad logits");
  hipLaunchKernelGGL(ar_sample_kernel, dim3(st->B), dim3(SAMP_THREADS), 0, (hipStream_t)stream, *st, logits, ld_logits);
  SOPRO_LAUNCH_CHECK();
}

}  // extern "C"
