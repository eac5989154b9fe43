// Device-resident driver state of the autoregressive loop: the sampler, the anti-loop policy,
// the EOS rule and the construction of the next frame's input all run on the GPU so that one
// frame is a fixed launch sequence with no host round trip (the reference synchronises the host
// several times per frame in sample_token, src/sopro/sampling.py:64-93) and can be replayed
// from a hipGraph.
#include <type_traits>

#include "common.h"

namespace {

constexpr int SAMP_THREADS = 256;  // four wavefronts per row

typedef unsigned long long u64;

__device__ __forceinline__ unsigned ord_f32(float v) {
  const unsigned u = __float_as_uint(v);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float unord_f32(unsigned o) { return __uint_as_float((o & 0x80000000u) ? (o & 0x7fffffffu) : ~o); }

__device__ __forceinline__ void philox_round(unsigned& c0, unsigned& c1, unsigned& c2, unsigned& c3, unsigned k0, unsigned k1) {
  const unsigned long long p0 = 0xD2511F53ull * c0, p1 = 0xCD9E8D57ull * c2;
  const unsigned h0 = (unsigned)(p0 >> 32), l0 = (unsigned)p0, h1 = (unsigned)(p1 >> 32), l1 = (unsigned)p1;
  c0 = h1 ^ c1 ^ k0; c1 = l1; c2 = h0 ^ c3 ^ k1; c3 = l0;
}
// counter = (frame, row, run nonce), key = seed: a new take per run / admission, reproducible for a given (seed, nonce)
__device__ float philox_uniform(unsigned long long seed, unsigned t, unsigned b, unsigned nonce) {
  unsigned c0 = t, c1 = b, c2 = 0x5090u, c3 = nonce;
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
    st.row_step[b] = 0;
    if (b == 0) { *st.step = 0; *st.n_stopped = 0; }
  }
}

// ---- wave-wide all-reduce on the DPP data path (no LDS round trips): butterflies inside the quads (quad_perm) and the
// rows of 16 lanes (row_ror), then the four row results through scalar registers.  The result is wave-uniform.
template <int CTRL>
__device__ __forceinline__ unsigned dpp_u32(unsigned v) {
  return (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, 0xf, 0xf, false);
}
constexpr int DPP_QUAD_XOR1 = 0xB1;   // quad_perm:[1,0,3,2]
constexpr int DPP_QUAD_XOR2 = 0x4E;   // quad_perm:[2,3,0,1]
constexpr int DPP_ROW_ROR4 = 0x124;   // row_ror:4
constexpr int DPP_ROW_ROR8 = 0x128;   // row_ror:8
__device__ __forceinline__ unsigned wave_max_u32(unsigned v) {
  v = max(v, dpp_u32<DPP_QUAD_XOR1>(v));
  v = max(v, dpp_u32<DPP_QUAD_XOR2>(v));
  v = max(v, dpp_u32<DPP_ROW_ROR4>(v));
  v = max(v, dpp_u32<DPP_ROW_ROR8>(v));
  const unsigned r0 = __builtin_amdgcn_readlane(v, 0), r1 = __builtin_amdgcn_readlane(v, 16);
  const unsigned r2 = __builtin_amdgcn_readlane(v, 32), r3 = __builtin_amdgcn_readlane(v, 48);
  return max(max(r0, r1), max(r2, r3));
}
__device__ __forceinline__ float wave_sum_dpp(float v) {
  v += __uint_as_float(dpp_u32<DPP_QUAD_XOR1>(__float_as_uint(v)));
  v += __uint_as_float(dpp_u32<DPP_QUAD_XOR2>(__float_as_uint(v)));
  v += __uint_as_float(dpp_u32<DPP_ROW_ROR4>(__float_as_uint(v)));
  v += __uint_as_float(dpp_u32<DPP_ROW_ROR8>(__float_as_uint(v)));
  const float r0 = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(v), 0));
  const float r1 = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(v), 16));
  const float r2 = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(v), 32));
  const float r3 = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(v), 48));
  return (r0 + r1) + (r2 + r3);
}
// the n-th largest (n >= 1) of the 64 lanes' values, built bit by bit from ballots: 32 scalar steps, no data movement
__device__ __forceinline__ unsigned wave_nth_largest_u32(unsigned v, int n) {
  unsigned r = 0u;
#pragma unroll
  for (int bit = 31; bit >= 0; --bit) {
    const unsigned c = r | (1u << bit);
    if (__popcll(__ballot(v >= c)) >= n) r = c;
  }
  return r;
}

// Lanes of ONE wave exchanging data through LDS: the hardware executes a wave's LDS operations in issue order, so no barrier
// is needed - but the compiler reasons per thread (it may forward a thread's own store to its later load, or move a load into
// a branch).  A wavefront-scope fence pair + a scheduling barrier pins the order; it emits no instruction beyond a wait.
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

constexpr int RECENT = 64;   // rolling token window per row: slot j = token sampled j+1 frames ago (-1 = none)
constexpr int PER = 9;       // logits per thread: ceil(2049 / 256), element q*256 + tid
constexpr int WREG = PER * 64;             // candidate slots of one wave's region (every logit of the wave may qualify)
constexpr int MAXCAND = 4 * WREG;
constexpr int FAST = 128;    // candidates the register all-pairs stage ranks (two per lane)

// sample_token (src/sopro/sampling.py:24-93) + the loop policy of src/sopro/model.py:274-305 for one row per workgroup.
// Nothing is sorted.  What the reference does with sort / top-k / cumsum is done with counting:
//   * top-k: every wave takes the ceil(kk/4)-th largest of its 64 per-thread maxima (32 ballot steps); the smallest of the
//     four is a lower bound of the kk-th largest logit, so "logit >= bound" keeps a superset of the top kk (~1.4 kk
//     entries).  Every wave compacts its own into its own LDS region in (q, lane) order: positions do not depend on
//     which wave gets there first (same seed -> same draw, bit for bit);
//   * an all-pairs pass gives every candidate its exact rank and the probability mass ahead of it - i.e. its place in the
//     sorted order and the cumulative sum the top-p rule needs.  Round 3: the candidates sit two per lane in registers of
//     EVERY wave and wave w broadcasts candidates 32w .. 32w+31 with v_readlane (no LDS traffic, no dependent LDS
//     latencies); the four partial (rank, mass) pairs meet through LDS in wave order.  The top kk then land in rank order
//     in wave 0, which renormalises, cuts and draws with two wave reductions and two ballots.
// The repetition penalty needs no workgroup-wide bitmap: every wave holds the token window in its lanes and marks the
// entries that name its own threads' logits in 64 words of LDS.  Three workgroup barriers (one when decoding greedily).
// Frame bookkeeping without a ticket: row b keeps its own frame counter (row_step[b]); *step - read by the ring-buffer
// kernels of the NEXT frame only, i.e. behind a kernel boundary - is advanced by row 0's workgroup.
__global__ __launch_bounds__(SAMP_THREADS) void ar_sample_kernel(const sopro_ar_state st, const float* __restrict__ logits,
                                                                 int64_t ld) {
  __shared__ u64 cand_k[MAXCAND];
  __shared__ float cand_e[MAXCAND];
  __shared__ int p_rank[4 * FAST];
  __shared__ float p_ahead[4 * FAST];
  __shared__ u64 selk[64];
  __shared__ float sele[64], selb[64];
  __shared__ unsigned w_thr[4], w_best_v[4], w_best_i[4], w_cnt[4];
  __shared__ unsigned pen_bits[4 * 64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.x;
  const int V1 = st.V + 1;
  long long* dbg = st.dbg ? st.dbg + (int64_t)b * 12 : nullptr;
#define SAMP_STAMP(i) do { if (dbg && tid == 0) dbg[i] = clock64(); } while (0)
  SAMP_STAMP(0);

  // ---- the only up-front memory round: frame index, policy parameters, this row's logits, its recent tokens
  // Classic mode: every row started at global frame 0.  Slot mode (st.start != NULL, continuous batching): row b was
  // admitted at global frame start[b] (-1 = free slot) with its own frame budget and parameters; its time is local.
  const int tg = st.row_step[b];
  const int s0 = st.start ? st.start[b] : 0;
  const int t = tg - s0;
  const int tmax = (st.start && st.row_max) ? st.row_max[b] : st.Tar;
  const float* prm = st.row_params ? st.row_params + (int64_t)b * 8 : st.params;
  const float p_top_p = prm[0], p_temp = prm[1], p_anti = prm[2], p_rec_p = prm[3];
  const float p_rec_t = prm[4], rep = prm[5];
  const int top_k = (int)prm[6], min_gen = (int)prm[7];
  const unsigned nonce = st.nonce ? st.nonce[b] : 0u;
  const unsigned rid = st.row_id ? (unsigned)st.row_id[b] : (unsigned)b;  // the row's identity in the Philox counter
  const unsigned long long seed = st.key ? ((unsigned long long)st.key[0] | ((unsigned long long)st.key[1] << 32)) : st.seed;
  const float* lg = logits + (int64_t)b * ld;
  float xv[PER];
#pragma unroll
  for (int q = 0; q < PER; ++q) {
    const int i = q * SAMP_THREADS + tid;
    xv[q] = (i < V1) ? lg[i] : 0.f;
  }
  int* recent = st.recent + (int64_t)b * RECENT;
  const int r_mine = recent[lane];  // every wave holds the window: lane j = token sampled j+1 frames ago
  if (!st.start && tg >= st.max_steps) return;  // classic mode, uniform over the grid: the loop is over
  if (st.start && (s0 < 0 || t >= min(tmax, st.max_steps))) {
    // free slot, or a row past its budget waiting to be harvested: nothing to sample, the frame still ticks
    if (tid == 0) {
      st.row_step[b] = tg + 1;
      if (b == 0) *st.step = tg + 1;
    }
    return;
  }
  // next frame's conditioning row (wave 0 writes the next input): address needs t only, consumed at the very end
  float cnext[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (wave == 0 && t + 1 < tmax) {
    const float* c = st.cond + ((int64_t)b * st.Tar + (t + 1)) * st.D;
#pragma unroll
    for (int q = 0; q < 8; ++q)
      if (q * 64 + lane < st.D) cnext[q] = c[q * 64 + lane];
  }

  // ---- anti-loop policy (src/sopro/model.py:274-279, sampling.py:16-21): repeated tail of length 3..16, or 9 equal
  // tokens; evaluated by every wave on its own copy of the window (wave-uniform result, no exchange)
  bool recover = false;
  if (p_anti != 0.f) {
    const u64 valid = __ballot(r_mine >= 0);
#pragma unroll
    for (int n = 3; n <= 16; ++n) {
      const int other = __shfl_down(r_mine, n, 64);
      const u64 same = __ballot(r_mine == other);
      const u64 need = (1ull << n) - 1ull;
      recover = recover || (((valid >> (2 * n - 1)) & 1ull) && (same & need) == need);
    }
    const int newest = __builtin_amdgcn_readlane(r_mine, 0);
    const u64 eq0 = __ballot(r_mine == newest);
    recover = recover || (((valid >> 8) & 1ull) && (eq0 & 0x1FFull) == 0x1FFull);
  }
  const float top_p = recover ? p_rec_p : p_top_p;
  const float temp = recover ? p_rec_t : p_temp;
  SAMP_STAMP(1);
  // ---- nan_to_num, temperature (sampling.py:33-38)
#pragma unroll
  for (int q = 0; q < PER; ++q) {
    float v = xv[q];
    if (v != v) v = -1e9f;
    else if (v == INFINITY) v = 1e9f;
    else if (v == -INFINITY) v = -1e9f;
    if (temp != 0.f && temp != 1.0f) v = v / temp;
    xv[q] = v;
  }
  // ---- repetition penalty on the unique ids among the last 50 tokens (sampling.py:40-50).  Token tok is logit (tok >> 8) of
  // thread (tok & 255): lanes holding a window entry of THIS wave's threads mark it in the wave's own 64 words of LDS
  // (duplicates OR the same bit; one wave's LDS operations execute in issue order: clear, mark, read - no barrier)
  if (rep != 1.0f) {
    const bool in_win = lane < 50 && r_mine >= 0 && r_mine < V1;
    unsigned* mine = pen_bits + wave * 64;
    mine[lane] = 0u;
    wave_lds_sync();  // lanes talk to each other through LDS: the compiler must neither reorder nor forward across these points
    if (in_win && ((r_mine >> 6) & 3) == wave) atomicOr(&mine[r_mine & 63], 1u << (r_mine >> 8));
    wave_lds_sync();
    const unsigned pm = mine[lane];
#pragma unroll
    for (int q = 0; q < PER; ++q)
      if ((pm >> q) & 1u) xv[q] = xv[q] < 0.f ? xv[q] * rep : xv[q] / rep;
  }

  SAMP_STAMP(2);
  // ---- per-thread maximum as (order-preserving value bits, 4095 - index): larger == better, ties -> lower index
  unsigned ov[PER];
  unsigned tm_v = 0u, tm_i = 0u;
#pragma unroll
  for (int q = 0; q < PER; ++q) {
    const int i = q * SAMP_THREADS + tid;
    ov[q] = (i < V1) ? ord_f32(xv[q]) : 0u;  // padding sorts below every real logit (ord_f32 of a finite value is > 0)
    const unsigned ii = (unsigned)(4095 - i);
    const bool better = (i < V1) && (ov[q] > tm_v || (ov[q] == tm_v && ii > tm_i));
    tm_v = better ? ov[q] : tm_v;
    tm_i = better ? ii : tm_i;
  }
  const bool greedy = !(top_p > 0.f);
  const int kk = (top_k > 0) ? min(top_k, V1) : V1;
  // top_p <= 0 keeps only the head of the sorted distribution == arg-max of the penalised logits.  top_k > 64 or "no
  // top-k" is outside the reference policy (model.py:289) and refused by the host; here it degrades to the arg-max.
  const bool sampling = !greedy && kk <= 64;
  {
    const unsigned wv = wave_max_u32(tm_v);
    const unsigned wi = wave_max_u32(tm_v == wv ? tm_i : 0u);
    unsigned thr = 0u;
    if (sampling) thr = wave_nth_largest_u32(tm_v, (kk + 3) >> 2);
    if (lane == 0) { w_best_v[wave] = wv; w_best_i[wave] = wi; w_thr[wave] = thr; }
  }
  __syncthreads();
  SAMP_STAMP(3);
  unsigned best_v = w_best_v[0], best_i = w_best_i[0], thr = w_thr[0];
#pragma unroll
  for (int w2 = 1; w2 < 4; ++w2) {
    const unsigned v2 = w_best_v[w2], i2 = w_best_i[w2];
    const bool better = v2 > best_v || (v2 == best_v && i2 > best_i);
    best_v = better ? v2 : best_v;
    best_i = better ? i2 : best_i;
    thr = min(thr, w_thr[w2]);
  }
  const int top_i = 4095 - (int)best_i;
  int tok = top_i;
  // the arg-max is the token under greedy decoding and the likeliest one otherwise: its embedding row (a dependent, far
  // load at the very end of the frame otherwise) is requested now
  float espec[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (wave == 0 && t + 1 < tmax) {
    const float* e = st.emb + (int64_t)top_i * st.D;
#pragma unroll
    for (int q = 0; q < 8; ++q)
      if (q * 64 + lane < st.D) espec[q] = e[q * 64 + lane];
  }
  if (sampling) {
    // ---- candidates: every logit >= thr (at least kk of them), with exp(x - max) next to the key; wave w fills its own
    // region [w * WREG, ...) in (q, lane) order
    const float xmax = unord_f32(best_v);
    {
      const u64 lt = (1ull << lane) - 1ull;
      unsigned pos = (unsigned)(wave * WREG);
#pragma unroll
      for (int q = 0; q < PER; ++q) {
        const u64 m = __ballot((q * SAMP_THREADS + tid < V1) && ov[q] >= thr);
        if ((m >> lane) & 1ull) {
          const unsigned at = pos + (unsigned)__popcll(m & lt);
          cand_k[at] = ((u64)ov[q] << 12) | (u64)(4095 - (q * SAMP_THREADS + tid));
          cand_e[at] = xv[q];  // the penalised logit; exp(x - max) is taken by whoever ranks the candidate (two per lane)
        }
        pos += (unsigned)__popcll(m);
      }
      if (lane == 0) w_cnt[wave] = pos - (unsigned)(wave * WREG);
    }
    __syncthreads();
    SAMP_STAMP(4);
    const int c0 = (int)w_cnt[0], c1 = c0 + (int)w_cnt[1], c2 = c1 + (int)w_cnt[2], C = c2 + (int)w_cnt[3];
    // candidate c of the concatenated order lives at region slot:
    auto slot_of = [&](int c) { return c < c0 ? c : (c < c1 ? WREG + (c - c0) : (c < c2 ? 2 * WREG + (c - c1) : 3 * WREG + (c - c2))); };
    if (C <= FAST) {
      // ---- all pairs in registers: lane l of every wave holds candidates l and l + 64; wave w broadcasts 32w .. 32w + 31
      const int ca = lane, cb = lane + 64;
      const u64 ka = ca < C ? cand_k[slot_of(ca)] : 0ull, kb = cb < C ? cand_k[slot_of(cb)] : 0ull;   // 0 sorts below every real key
      const float ea = ca < C ? expf(cand_e[slot_of(ca)] - xmax) : 0.f, eb = cb < C ? expf(cand_e[slot_of(cb)] - xmax) : 0.f;
      const unsigned src_lo = (wave & 2) ? (unsigned)kb : (unsigned)ka, src_hi = (wave & 2) ? (unsigned)(kb >> 32) : (unsigned)(ka >> 32);
      const float src_e = (wave & 2) ? eb : ea;
      int ra = 0, rb = 0;
      float aa = 0.f, ab = 0.f;
      // (the broadcast lane of v_readlane is an immediate: one unrolled copy of the loop per 32-lane half)
      auto pairs = [&](auto half) {
        constexpr int H0 = decltype(half)::value * 32;
#pragma unroll
        for (int j = 0; j < 32; ++j) {
          const unsigned lo = __builtin_amdgcn_readlane(src_lo, H0 + j), hi = __builtin_amdgcn_readlane(src_hi, H0 + j);
          const float ej = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(src_e), H0 + j));
          const u64 kj = ((u64)hi << 32) | lo;
          const bool ga = kj > ka, gb = kj > kb;
          ra += ga ? 1 : 0; aa += ga ? ej : 0.f;
          rb += gb ? 1 : 0; ab += gb ? ej : 0.f;
        }
      };
      if (__builtin_amdgcn_readfirstlane(wave) & 1) pairs(std::integral_constant<int, 1>{});
      else pairs(std::integral_constant<int, 0>{});
      p_rank[wave * FAST + ca] = ra; p_rank[wave * FAST + cb] = rb;
      p_ahead[wave * FAST + ca] = aa; p_ahead[wave * FAST + cb] = ab;
      __syncthreads();
      SAMP_STAMP(5);
      if (wave != 0) return;
      {
        const int rka = ((p_rank[ca] + p_rank[FAST + ca]) + p_rank[2 * FAST + ca]) + p_rank[3 * FAST + ca];
        const int rkb = ((p_rank[cb] + p_rank[FAST + cb]) + p_rank[2 * FAST + cb]) + p_rank[3 * FAST + cb];
        const float aha = ((p_ahead[ca] + p_ahead[FAST + ca]) + p_ahead[2 * FAST + ca]) + p_ahead[3 * FAST + ca];
        const float ahb = ((p_ahead[cb] + p_ahead[FAST + cb]) + p_ahead[2 * FAST + cb]) + p_ahead[3 * FAST + cb];
        if (ca < C && rka < kk) { selk[rka] = ka; sele[rka] = ea; selb[rka] = aha; }
        if (cb < C && rkb < kk) { selk[rkb] = kb; sele[rkb] = eb; selb[rkb] = ahb; }
      }
      // one wave wrote, the same wave reads: LDS executes a wave's operations in issue order (no barrier - the other waves are gone)
      wave_lds_sync();
    } else {
      // ---- many ties at the bound (degenerate logits): the same ranks from LDS, any number of candidates
      for (int c = tid; c < C; c += SAMP_THREADS) {
        const u64 kc = cand_k[slot_of(c)];
        int rank = 0;
        float ahead = 0.f;
        for (int j = 0; j < C; ++j) {
          const int sj = slot_of(j);
          const bool gt = cand_k[sj] > kc;
          rank += gt ? 1 : 0;
          ahead += gt ? expf(cand_e[sj] - xmax) : 0.f;
        }
        if (rank < kk) { selk[rank] = kc; sele[rank] = expf(cand_e[slot_of(c)] - xmax); selb[rank] = ahead; }
      }
      __syncthreads();
      SAMP_STAMP(5);
      if (wave != 0) return;
    }
    {
      // top-k renormalisation, top-p cut and the draw (sampling.py:56-93): lane j owns sorted entry j.  The softmax
      // denominator cancels in the renormalisation: p_j / sum_topk p = e_j / sum_topk e  (and sum_topk p >= 1/2049, so
      // the reference's "mass <= 1e-12 -> arg-max" fallback cannot trigger on finite logits).
      const float ej = (lane < kk) ? sele[lane] : 0.f;
      const float S = wave_sum_dpp(ej);
      const float pj = ej / S;
      const float before = (lane < kk) ? selb[lane] / S : 0.f;
      // drop entry j when the cumulative mass *before* it already exceeds top_p; entry 0 always stays (sampling.py:68-76)
      const bool keep = lane < kk && (lane == 0 || !(top_p < 1.0f && before > top_p));
      const float kept = wave_sum_dpp(keep ? pj : 0.f);
      if (kept > 1e-12f) {
        const float u = philox_uniform(seed, (unsigned)t, rid, nonce) * kept;
        const u64 hit = __ballot(keep && u < before + pj);
        const u64 kmask = __ballot(keep);
        const int pick = hit ? (int)__ffsll((long long)hit) - 1 : 63 - __clzll((long long)kmask);
        tok = 4095 - (int)(selk[pick] & 4095ull);
      }
    }
  }
  SAMP_STAMP(6);
  if (wave != 0) return;  // wave 0 holds the token (every wave does when it is the arg-max) and does the bookkeeping

  // ---- bookkeeping: history, EOS rule (model.py:293-305), next input (model.py:266-272)
  if (t + 1 < tmax) {
    if (tok != top_i) {  // wave-uniform
      const float* e = st.emb + (int64_t)tok * st.D;
#pragma unroll
      for (int q = 0; q < 8; ++q)
        if (q * 64 + lane < st.D) espec[q] = e[q * 64 + lane];
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int d = q * 64 + lane;
      if (d < st.D) st.x_cur[(int64_t)b * st.D + d] = cnext[q] + espec[q];
    }
  }
  SAMP_STAMP(7);
  const int older = __shfl_up(r_mine, 1, 64);
  recent[lane] = (lane == 0) ? tok : older;
  if (lane == 0) {
    st.hist[(int64_t)b * st.max_steps + t] = tok;
    if (tok == st.V) {
      if (st.first_eos[b] < 0) st.first_eos[b] = t;
      if (st.stop_t[b] < 0 && (t + 1) >= min_gen) {
        st.stop_t[b] = t;
        atomicAdd(st.n_stopped, 1);
      }
    }
    // this row's frame counter; the global one (ring-buffer slots of the next frame's kernels) follows row 0.  Plain
    // stores: the next reader is behind a kernel boundary.
    st.row_step[b] = tg + 1;
    if (b == 0) *st.step = tg + 1;
    if (dbg) dbg[10] = clock64();
  }
  SAMP_STAMP(8);
#undef SAMP_STAMP
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
  SOPRO_CHECK_ARG(st->x_cur && st->cond && st->emb && st->hist && st->step && st->row_step && st->first_eos && st->stop_t &&
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
