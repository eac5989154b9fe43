// HBM-bound row / elementwise kernels of the Sopro hot path (fp32, channels-last).
// All are one-pass, float4-vectorised where the layout allows, one wave per row for the
// row-statistics kernels (64-lane shuffles, no LDS).
#include "common.h"

namespace {

// ---------------------------------------------------------------------------------------------
// RMSNorm / LayerNorm with optional per-segment FiLM-style modulation
// ---------------------------------------------------------------------------------------------
constexpr int NORM_MAX_PER_LANE = 16;  // C <= 1024

__global__ __launch_bounds__(256) void norm_kernel(const float* __restrict__ x, int64_t ldx, int64_t x_seg_stride, float* __restrict__ out,
                                                   int64_t ldo, const float* __restrict__ w, const float* __restrict__ b,
                                                   const float* __restrict__ mul, const float* __restrict__ add, int rows,
                                                   int rows_per_seg, int C, float eps, int kind) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int seg = row / rows_per_seg;
  const float* xr = x + (int64_t)seg * x_seg_stride + (int64_t)(row - seg * rows_per_seg) * ldx;
  float v[NORM_MAX_PER_LANE];
  const int per = (C + 63) / 64;
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < NORM_MAX_PER_LANE; ++j) {
    const int c = lane + j * 64;
    v[j] = (j < per && c < C) ? xr[c] : 0.f;
    s += v[j];
  }
  float mean = 0.f, var;
  if (kind == SOPRO_NORM_LN) {
    mean = wave_sum(s) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < NORM_MAX_PER_LANE; ++j) {
      const int c = lane + j * 64;
      const float d = (j < per && c < C) ? v[j] - mean : 0.f;
      q += d * d;
    }
    var = wave_sum(q) / (float)C;
  } else {
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < NORM_MAX_PER_LANE; ++j) q += v[j] * v[j];
    var = wave_sum(q) / (float)C;
  }
  const float rstd = rsqrtf(var + eps);
  const float* mr = mul ? mul + (int64_t)seg * C : nullptr;
  const float* ar = add ? add + (int64_t)seg * C : nullptr;
  float* orow = out + (int64_t)row * ldo;
#pragma unroll
  for (int j = 0; j < NORM_MAX_PER_LANE; ++j) {
    const int c = lane + j * 64;
    if (j < per && c < C) {
      float y = ((v[j] - mean) * rstd) * w[c];
      if (b) y += b[c];
      if (mr) y *= mr[c];
      if (ar) y += ar[c];
      orow[c] = y;
    }
  }
}

__global__ __launch_bounds__(256) void rms_match_kernel(const float* __restrict__ a, const float* __restrict__ x,
                                                        float* __restrict__ out, int rows, int C) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* ar = a + (int64_t)row * C;
  const float* xr = x + (int64_t)row * C;
  float sa = 0.f, sx = 0.f;
  for (int c = lane; c < C; c += 64) {
    sa += ar[c] * ar[c];
    sx += xr[c] * xr[c];
  }
  sa = wave_sum(sa);
  sx = wave_sum(sx);
  const float ra = sqrtf(sa / (float)C + 1e-6f), rx = sqrtf(sx / (float)C + 1e-6f);
  const float sc = fminf(fmaxf(rx / ra, 0.f), 10.f);
  for (int c = lane; c < C; c += 64) out[(int64_t)row * C + c] = ar[c] * sc;
}

__global__ void tanh_affine_kernel(const float* __restrict__ in, float* __restrict__ out, float c0, float c1, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = c0 + c1 * tanhf(in[i]);
}

__global__ void add_pos_kernel(const float* __restrict__ rowvec, const float* __restrict__ table, float* __restrict__ out,
                               int B, int T, int C, int pos0) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t n = (int64_t)B * T * C;
  if (i >= n) return;
  const int c = (int)(i % C);
  const int64_t bt = i / C;
  const int t = (int)(bt % T), b = (int)(bt / T);
  out[i] = rowvec[(int64_t)b * C + c] + table[(int64_t)(pos0 + t) * C + c];
}

__global__ void masked_mean_kernel(const float* __restrict__ x, const int* __restrict__ lens, float* __restrict__ out,
                                   int B, int T, int C) {
  const int b = blockIdx.y;
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const int len = lens ? min(lens[b], T) : T;
  float s = 0.f;
  for (int t = 0; t < len; ++t) s += x[((int64_t)b * T + t) * C + c];
  out[(int64_t)b * C + c] = s / ((float)len + 1e-6f);
}

// ---------------------------------------------------------------------------------------------
// depthwise conv over time
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void dwconv_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                     const float* __restrict__ bias, const float* __restrict__ res,
                                                     float* __restrict__ out, const int* __restrict__ lens, int B, int T,
                                                     int C, int ksize, int dil, int left, int mode) {
  const int c4n = C >> 2;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)B * T * c4n) return;
  const int c4 = (int)(i % c4n);
  const int64_t bt = i / c4n;
  const int t = (int)(bt % T), b = (int)(bt / T);
  const int len = lens ? min(lens[b], T) : T;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int j = 0; j < ksize; ++j) {
    const int ts = t - left + j * dil;
    if (ts < 0 || ts >= len) continue;
    const float4 xv = *reinterpret_cast<const float4*>(x + ((int64_t)b * T + ts) * C + c4 * 4);
    const float4 wv = *reinterpret_cast<const float4*>(w + (int64_t)j * C + c4 * 4);
    acc.x += xv.x * wv.x; acc.y += xv.y * wv.y; acc.z += xv.z * wv.z; acc.w += xv.w * wv.w;
  }
  if (bias) {
    const float4 bv = *reinterpret_cast<const float4*>(bias + c4 * 4);
    acc.x += bv.x; acc.y += bv.y; acc.z += bv.z; acc.w += bv.w;
  }
  const int64_t o = ((int64_t)b * T + t) * C + c4 * 4;
  if (mode == 1) {
    const float4 rv = *reinterpret_cast<const float4*>(res + o);
    acc.x += rv.x; acc.y += rv.y; acc.z += rv.z; acc.w += rv.w;
  } else if (mode == 2) {
    acc.x = gelu_erf(acc.x); acc.y = gelu_erf(acc.y); acc.z = gelu_erf(acc.z); acc.w = gelu_erf(acc.w);
  }
  *reinterpret_cast<float4*>(out + o) = acc;
}

// The same convolution with every input row loaded ONCE per CI outputs instead of once per tap (round 5).  Outputs t, t + dil,
// t + 2 dil, ... read the same rows: a thread takes a COMB of CI = 8 outputs of one residue class (t = r + (i0 + i) dil) of one
// channel quad; their ksize taps each touch only the CI + ksize - 1 rows r - left + (i0 + u) dil, u = 0 .. CI + ksize - 2: 18 row
// loads for 8 outputs at ksize 11 where dwconv_kernel issues 88 (the refinement's 24 launches per 64-row pass ran at 1.6 TB/s of
// algorithmic traffic, bound by those requests, not by memory).  Per output the taps are accumulated in the same order with the
// same skips, so results are bit-identical to dwconv_kernel.
template <int KS>
__global__ __launch_bounds__(256) void dwconv_comb_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                          const float* __restrict__ bias, const float* __restrict__ res,
                                                          float* __restrict__ out, const int* __restrict__ lens, int B, int T,
                                                          int C, int dil, int left, int mode, int per_r) {
  constexpr int CI = 8, NW = CI + KS - 1;
  const int c4n = C >> 2;
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)B * dil * per_r * c4n) return;
  const int c4 = (int)(idx % c4n);
  int64_t k = idx / c4n;
  const int chunk = (int)(k % per_r);
  k /= per_r;
  const int r = (int)(k % dil), b = (int)(k / dil);
  const int i0 = chunk * CI;
  if (r + (int64_t)i0 * dil >= T) return;
  const int len = lens ? min(lens[b], T) : T;
  const float* xb = x + (int64_t)b * T * C + c4 * 4;
  float4 xv[NW];
  unsigned okm = 0u;
#pragma unroll
  for (int u = 0; u < NW; ++u) {
    const int64_t ts = (int64_t)r - left + (int64_t)(i0 + u) * dil;
    const bool ok = ts >= 0 && ts < len;
    okm |= ok ? (1u << u) : 0u;
    xv[u] = ok ? *reinterpret_cast<const float4*>(xb + ts * C) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  float4 wv[KS];
#pragma unroll
  for (int j = 0; j < KS; ++j) wv[j] = *reinterpret_cast<const float4*>(w + (int64_t)j * C + c4 * 4);
  float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
  if (bias) bv = *reinterpret_cast<const float4*>(bias + c4 * 4);
#pragma unroll
  for (int i = 0; i < CI; ++i) {
    const int t = r + (i0 + i) * dil;
    if (t >= T) break;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int j = 0; j < KS; ++j) {
      if (!((okm >> (i + j)) & 1u)) continue;
      const float4 xq = xv[i + j];
      acc.x += xq.x * wv[j].x; acc.y += xq.y * wv[j].y; acc.z += xq.z * wv[j].z; acc.w += xq.w * wv[j].w;
    }
    if (bias) { acc.x += bv.x; acc.y += bv.y; acc.z += bv.z; acc.w += bv.w; }
    const int64_t o = ((int64_t)b * T + t) * C + c4 * 4;
    if (mode == 1) {
      const float4 rv = *reinterpret_cast<const float4*>(res + o);
      acc.x += rv.x; acc.y += rv.y; acc.z += rv.z; acc.w += rv.w;
    } else if (mode == 2) {
      acc.x = gelu_erf(acc.x); acc.y = gelu_erf(acc.y); acc.z = gelu_erf(acc.z); acc.w = gelu_erf(acc.w);
    }
    *reinterpret_cast<float4*>(out + o) = acc;
  }
}

// ---------------------------------------------------------------------------------------------
// embedding gathers
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void codebook_sum_kernel(const int* __restrict__ tok, int ldt, const int* __restrict__ col,
                                                           const int* __restrict__ off, const float* __restrict__ wq, int nq,
                                                           const float* __restrict__ table, int64_t table_rows,
                                                           const float* __restrict__ base, float alpha, float beta,
                                                           float* __restrict__ out, int64_t ldo, int64_t o_seg_stride,
                                                           int rows, int rows_per_seg, int D) {
  const int d4n = D >> 2;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)rows * d4n) return;
  const int d4 = (int)(i % d4n);
  const int row = (int)(i / d4n);
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int q = 0; q < nq; ++q) {
    int64_t r = (int64_t)off[q] + tok[(int64_t)row * ldt + col[q]];
    r = r < 0 ? 0 : (r >= table_rows ? table_rows - 1 : r);
    const float4 e = *reinterpret_cast<const float4*>(table + r * D + d4 * 4);
    const float s = wq[q];
    acc.x += s * e.x; acc.y += s * e.y; acc.z += s * e.z; acc.w += s * e.w;
  }
  float4 y = make_float4(beta * acc.x, beta * acc.y, beta * acc.z, beta * acc.w);
  if (base) {
    const float4 bv = *reinterpret_cast<const float4*>(base + (int64_t)row * D + d4 * 4);
    y.x += alpha * bv.x; y.y += alpha * bv.y; y.z += alpha * bv.z; y.w += alpha * bv.w;
  }
  const int seg = row / rows_per_seg;
  const int rr = row - seg * rows_per_seg;
  *reinterpret_cast<float4*>(out + (int64_t)seg * o_seg_stride + (int64_t)rr * ldo + d4 * 4) = y;
}

__global__ __launch_bounds__(256) void text_embed_kernel(const int* __restrict__ ids, const int* __restrict__ lens,
                                                         const float* __restrict__ table, int64_t table_rows,
                                                         const float* __restrict__ pe, float* __restrict__ out, int B, int T,
                                                         int C) {
  const int c4n = C >> 2;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)B * T * c4n) return;
  const int c4 = (int)(i % c4n);
  const int64_t bt = i / c4n;
  const int t = (int)(bt % T), b = (int)(bt / T);
  const int len = lens ? min(lens[b], T) : T;
  float4 y = make_float4(0.f, 0.f, 0.f, 0.f);
  if (t < len) {
    int64_t r = ids[bt];
    r = r < 0 ? 0 : (r >= table_rows ? table_rows - 1 : r);
    const float4 e = *reinterpret_cast<const float4*>(table + r * C + c4 * 4);
    const float4 p = *reinterpret_cast<const float4*>(pe + (int64_t)t * C + c4 * 4);
    y = make_float4(e.x + p.x, e.y + p.y, e.z + p.z, e.w + p.w);
  }
  *reinterpret_cast<float4*>(out + bt * C + c4 * 4) = y;
}

// first maximum wins (torch.argmax returns the first index on CPU for exact ties)
__global__ __launch_bounds__(256) void argmax_rows_kernel(const float* __restrict__ x, int64_t ldx, int* __restrict__ out,
                                                          int64_t ldo, int inner, int rows, int N) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* xr = x + (int64_t)row * ldx;
  float best = -INFINITY;
  int bi = 0x7fffffff;
  for (int n = lane; n < N; n += 64) {
    const float v = xr[n];
    if (v > best || (v == best && n < bi)) { best = v; bi = n; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(best, o, 64);
    const int oi = __shfl_xor(bi, o, 64);
    if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
  }
  if (lane == 0) out[(int64_t)(row / inner) * ldo + row % inner] = (bi == 0x7fffffff) ? 0 : bi;
}

// second half of the arg-max fused into a projection (gemm c_mode 5): head h of row r = the best of its `per_head` partial
// (value, global column) pairs, in column order, first maximum wins; the token is the column within the head
// (round 6) EIGHT lanes per (row, head): lane l takes pairs l, l + 8, .. and the eight reduce with three xor exchanges - the 8 lanes of a
// group read 64 consecutive bytes per step and a wave whole 256-byte runs, where one thread per head walked its 32 pairs alone with
// the lanes of a wave 256 bytes apart (74 us per refinement stage for 51 MB).  The winner - largest value, smallest column among
// equals - does not depend on the order of the comparisons.
__global__ __launch_bounds__(256) void argmax_partials_kernel(const float* __restrict__ part, int64_t ldp, int* __restrict__ out,
                                                              int64_t ldo, int heads, int per_head, int V, int64_t total) {
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t i = gid >> 3;
  const int l8 = (int)(gid & 7);
  if (i >= total) return;  // (whole groups of eight: the exchanges below stay inside a group)
  const int64_t row = i / heads;
  const int h = (int)(i % heads);
  const float2* p = reinterpret_cast<const float2*>(part) + row * ldp + (int64_t)h * per_head;
  float best = -INFINITY;
  int bi = 0x7fffffff;
  for (int j = l8; j < per_head; j += 8) {
    const float2 v = p[j];
    const int idx = __float_as_int(v.y);
    if (v.x > best || (v.x == best && idx < bi)) { best = v.x; bi = idx; }
  }
#pragma unroll
  for (int o = 1; o < 8; o <<= 1) {
    const float ov = __shfl_xor(best, o, 64);
    const int oi = __shfl_xor(bi, o, 64);
    if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
  }
  if (l8 == 0) out[row * ldo + h] = (bi == 0x7fffffff) ? 0 : bi - h * V;
}

// ---------------------------------------------------------------------------------------------
// Mimi encode side: single-channel FIR bank (first SEANet conv, polyphase resampler) and the
// residual-VQ assignment step
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void fir1_kernel(const float* __restrict__ x, int64_t x_seg_stride, int n_in,
                                                   const float* __restrict__ w, const float* __restrict__ bias,
                                                   float* __restrict__ out, int64_t ldo, int64_t o_seg_stride, int n_out, int C,
                                                   int K, int stride, int left) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)n_out * C) return;
  const int c = (int)(i % C);
  const int t = (int)(i / C);
  const float* xb = x + (int64_t)blockIdx.y * x_seg_stride;
  const float* wc = w + (int64_t)c * K;
  const int s0 = t * stride - left;
  float acc = 0.0f;  // taps in ascending order, one fma each
  for (int k = 0; k < K; ++k) {
    const int s = s0 + k;
    const float xv = (s >= 0 && s < n_in) ? xb[s] : 0.0f;
    acc = fmaf(wc[k], xv, acc);
  }
  out[(int64_t)blockIdx.y * o_seg_stride + (int64_t)t * ldo + c] = acc + (bias ? bias[c] : 0.0f);
}

// one 256-thread workgroup per row: first maximum of the score row -> code; residual -= codebook row
__global__ __launch_bounds__(256) void rvq_assign_kernel(const float* __restrict__ scores, int64_t lds_, int V,
                                                         const float* __restrict__ table, float* __restrict__ res, int64_t ldr,
                                                         int D, int* __restrict__ codes, int64_t ldc) {
  __shared__ float sv[4];
  __shared__ int si[4];
  const int row = blockIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float* xr = scores + (int64_t)row * lds_;
  float best = -INFINITY;
  int bi = 0x7fffffff;
  for (int n = threadIdx.x; n < V; n += 256) {
    const float v = xr[n];
    if (v > best || (v == best && n < bi)) { best = v; bi = n; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(best, o, 64);
    const int oi = __shfl_xor(bi, o, 64);
    if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
  }
  if (lane == 0) { sv[wave] = best; si[wave] = bi; }
  __syncthreads();
  best = sv[0]; bi = si[0];
#pragma unroll
  for (int k = 1; k < 4; ++k)
    if (sv[k] > best || (sv[k] == best && si[k] < bi)) { best = sv[k]; bi = si[k]; }
  if (bi == 0x7fffffff) bi = 0;  // all-NaN row
  if (threadIdx.x == 0) codes[(int64_t)row * ldc] = bi;
  const float* e = table + (int64_t)bi * D;
  float* r = res + (int64_t)row * ldr;
  for (int d = threadIdx.x; d < D; d += 256) r[d] -= e[d];
}

// ---------------------------------------------------------------------------------------------
// RoPE (rotate-half), Mimi upsample, last SEANet conv
// ---------------------------------------------------------------------------------------------
__global__ void rope_kernel(float* __restrict__ x, int64_t ldx, const float* __restrict__ cos_t,
                            const float* __restrict__ sin_t, int rows, int rows_per_seg, int pos0, int H, int dh) {
  const int half = dh >> 1;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)rows * H * half) return;
  const int e = (int)(i % half);
  const int64_t rh = i / half;
  const int h = (int)(rh % H);
  const int row = (int)(rh / H);
  const int pos = pos0 + row % rows_per_seg;
  const float c = cos_t[(int64_t)pos * half + e], s = sin_t[(int64_t)pos * half + e];
  float* p = x + (int64_t)row * ldx + h * dh;
  const float a = p[e], b = p[e + half];
  p[e] = a * c - b * s;
  p[e + half] = b * c + a * s;
}

__global__ void upsample2_kernel(const float* __restrict__ x, const float* __restrict__ w, float* __restrict__ y,
                                 int64_t y_seg_stride, int B, int T, int C) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)B * T * 2 * C) return;
  const int c = (int)(i % C);
  const int64_t bn = i / C;
  const int n = (int)(bn % (2 * T)), b = (int)(bn / (2 * T));
  const int t = n >> 1, r = n & 1;
  float v = x[((int64_t)b * T + t) * C + c] * w[c * 4 + r];
  if (t > 0) v += x[((int64_t)b * T + t - 1) * C + c] * w[c * 4 + r + 2];
  y[(int64_t)b * y_seg_stride + (int64_t)n * C + c] = v;
}

// wav[n] = bias + d0[n-2] + d1[n-1] + d2[n] with d_j[r] = sum_c elu(h[r, c]) w[j, c]; 16 lanes per row.
__global__ __launch_bounds__(256) void final_conv_kernel(const float* __restrict__ h, int64_t h_seg_stride,
                                                         const float* __restrict__ w, float bias, float* __restrict__ wav,
                                                         int64_t wav_seg_stride, int B, int T) {
  constexpr int ROWS = 256;  // output samples per workgroup
  __shared__ float d[3][ROWS + 2];
  const int b = blockIdx.y;
  const int n0 = blockIdx.x * ROWS;
  const int sub = threadIdx.x & 15, grp = threadIdx.x >> 4;  // 16 groups of 16 lanes
  const float4 w0 = *reinterpret_cast<const float4*>(w + 0 * 64 + sub * 4);
  const float4 w1 = *reinterpret_cast<const float4*>(w + 1 * 64 + sub * 4);
  const float4 w2 = *reinterpret_cast<const float4*>(w + 2 * 64 + sub * 4);
  // h rows are addressed relative to the first padded row: padded row p = sample index + 2
  const float* hb = h + (int64_t)b * h_seg_stride;
  for (int rr = grp; rr < ROWS + 2; rr += 16) {
    const int p = n0 + rr;  // padded-row index; sample index = p - 2
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
    if (p < T + 2) {
      float4 v = *reinterpret_cast<const float4*>(hb + (int64_t)p * 64 + sub * 4);
      v.x = eluf_(v.x); v.y = eluf_(v.y); v.z = eluf_(v.z); v.w = eluf_(v.w);
      s0 = v.x * w0.x + v.y * w0.y + v.z * w0.z + v.w * w0.w;
      s1 = v.x * w1.x + v.y * w1.y + v.z * w1.z + v.w * w1.w;
      s2 = v.x * w2.x + v.y * w2.y + v.z * w2.z + v.w * w2.w;
    }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) {
      s0 += __shfl_xor(s0, o, 64);
      s1 += __shfl_xor(s1, o, 64);
      s2 += __shfl_xor(s2, o, 64);
    }
    if (sub == 0) { d[0][rr] = s0; d[1][rr] = s1; d[2][rr] = s2; }
  }
  __syncthreads();
  const int n = n0 + threadIdx.x;
  if (n < T) wav[(int64_t)b * wav_seg_stride + n] = ((d[0][threadIdx.x] + d[1][threadIdx.x + 1]) + d[2][threadIdx.x + 2]) + bias;
}

// attentive statistics pooling (src/sopro/nn/blocks.py:174-188): a = softmax_t(logit), mu = sum a h,
// std = sqrt(clamp_min(sum a (h-mu)^2, 1e-6)); out[b] = [mu | std].  One workgroup per batch row.
__global__ __launch_bounds__(256) void stats_pool_kernel(const float* __restrict__ h, const float* __restrict__ logit,
                                                         const int* __restrict__ lens, float* __restrict__ out, int B, int T,
                                                         int C) {
  __shared__ float red[4];
  __shared__ float sh_max, sh_sum;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int len = lens ? min(lens[b], T) : T;
  const float* lg = logit + (int64_t)b * T;
  float m = -INFINITY;
  for (int t = tid; t < len; t += 256) m = fmaxf(m, lg[t]);
  m = wave_max(m);
  if (lane == 0) red[wave] = m;
  __syncthreads();
  if (tid == 0) sh_max = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  const float mx = sh_max;
  float s = 0.f;
  for (int t = tid; t < len; t += 256) s += expf(lg[t] - mx);
  s = wave_sum(s);
  __syncthreads();
  if (lane == 0) red[wave] = s;
  __syncthreads();
  if (tid == 0) sh_sum = (red[0] + red[1]) + (red[2] + red[3]);
  __syncthreads();
  const float inv = 1.f / sh_sum;
  for (int c = tid; c < C; c += 256) {
    float mu = 0.f;
    for (int t = 0; t < len; ++t) mu += (expf(lg[t] - mx) * inv) * h[((int64_t)b * T + t) * C + c];
    float var = 0.f;
    for (int t = 0; t < len; ++t) {
      const float d = h[((int64_t)b * T + t) * C + c] - mu;
      var += (expf(lg[t] - mx) * inv) * d * d;
    }
    out[(int64_t)b * 2 * C + c] = mu;
    out[(int64_t)b * 2 * C + C + c] = sqrtf(fmaxf(var, 1e-6f));
  }
}

// x / max(||x||_2, eps) per row (torch F.normalize, src/sopro/nn/speaker.py:60)
__global__ __launch_bounds__(256) void l2norm_kernel(const float* __restrict__ x, float* __restrict__ out, int rows, int C,
                                                     float eps) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  float s = 0.f;
  for (int c = lane; c < C; c += 64) s += x[(int64_t)row * C + c] * x[(int64_t)row * C + c];
  s = wave_sum(s);
  const float d = fmaxf(sqrtf(s), eps);
  for (int c = lane; c < C; c += 64) out[(int64_t)row * C + c] = x[(int64_t)row * C + c] / d;
}

inline unsigned nblk(int64_t n, int bs) { return (unsigned)((n + bs - 1) / bs); }


// rows x width 32-bit words with row pitches: the stage sequences' pad clears, ticket resets and strided copies as KERNELS
// (a recorded launch sequence must not hold memset / memcpy nodes: a hipMemsetAsync captured into a hipGraph was measured to
// corrupt the split-K tickets of the replayed sequence once other allocations had happened - profiles/r03_experiments.md)
__global__ __launch_bounds__(256) void fill2d_kernel(unsigned* __restrict__ p, int64_t pitch, int rows, int width, unsigned v) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)rows * width) return;
  p[(i / width) * pitch + (i % width)] = v;
}
__global__ __launch_bounds__(256) void copy2d_kernel(unsigned* __restrict__ d, int64_t dpitch, const unsigned* __restrict__ s, int64_t spitch, int rows,
                                                     int width) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)rows * width) return;
  d[(i / width) * dpitch + (i % width)] = s[(i / width) * spitch + (i % width)];
}

// codebook 0 of a batch, as the AR loop left it (rows of a longer history buffer; EOS = V in rows that stopped early), into column 0
// of the refinement's token matrix [B T, Q], clamped to valid codes (src/sopro/model.py:385-390: frames past a row's own length are
// ignored downstream, they only have to be valid indices)
__global__ __launch_bounds__(256) void nar_seed_kernel(int* __restrict__ tokens, int Q, const int* __restrict__ cb0, int64_t bstride, int B, int T, int vmax) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)B * T) return;
  const int b = (int)(i / T), t = (int)(i - (int64_t)b * T);
  tokens[i * Q] = min(max(cb0[(int64_t)b * bstride + t], 0), vmax);
}

// fp32 <-> bf16 images of a tensor (round to nearest even), 4 elements per thread
__global__ __launch_bounds__(256) void cvt_f32_bf16_kernel(const float* __restrict__ src, uint2* __restrict__ dst, int64_t n4) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n4) return;
  const float4 v = reinterpret_cast<const float4*>(src)[i];
  unsigned a, b, lo;
  split2_bf16(v.x, v.y, a, lo);
  split2_bf16(v.z, v.w, b, lo);
  dst[i] = make_uint2(a, b);
}
__global__ __launch_bounds__(256) void cvt_bf16_f32_kernel(const uint2* __restrict__ src, float* __restrict__ dst, int64_t n4) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n4) return;
  const uint2 v = src[i];
  reinterpret_cast<float4*>(dst)[i] = make_float4(__uint_as_float(v.x << 16), __uint_as_float(v.x & 0xffff0000u), __uint_as_float(v.y << 16),
                                                  __uint_as_float(v.y & 0xffff0000u));
}

// (mean, sum of squared deviations) of every 64-column group of a row: one thread per float4, 16 lanes per group - the same reduction, in
// the same order, as the contractions' EPI_RES epilogue writes for ln_stats_out (gemm_epilogue.h)
__global__ __launch_bounds__(256) void row_stats_kernel(const float* __restrict__ x, int64_t ldx, int64_t seg_stride, int rows, int rps, int C,
                                                        float* __restrict__ stats) {
  const int per_row = C >> 2;  // threads per row
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t r = i / per_row;
  if (r >= rows) return;  // (whole 16-lane groups leave together: per_row is a multiple of 16)
  const int c4 = (int)(i - r * per_row);
  const int64_t seg = r / rps, rr = r - seg * rps;
  const float4 v = *reinterpret_cast<const float4*>(x + seg * seg_stride + rr * ldx + c4 * 4);
  float s = (v.x + v.y) + (v.z + v.w);
  s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 4, 64); s += __shfl_xor(s, 8, 64);
  const float mu = s * (1.0f / 64.0f);
  const float dx = v.x - mu, dy = v.y - mu, dz = v.z - mu, dw = v.w - mu;
  float q = (dx * dx + dy * dy) + (dz * dz + dw * dw);
  q += __shfl_xor(q, 1, 64); q += __shfl_xor(q, 2, 64); q += __shfl_xor(q, 4, 64); q += __shfl_xor(q, 8, 64);
  if ((c4 & 15) == 0) *reinterpret_cast<float2*>(stats + (r * (C >> 6) + (c4 >> 4)) * 2) = make_float2(mu, q);
}

}  // namespace

extern "C" {

int sopro_row_stats_f32(const float* x, int64_t ldx, int64_t x_seg_stride, int32_t rows, int32_t rows_per_seg, int32_t C, float* stats, void* stream) {
  SOPRO_CHECK_ARG(x && stats && rows > 0 && rows_per_seg > 0 && C > 0 && (C & 63) == 0, "x, stats non-NULL; rows > 0; C a multiple of 64");
  SOPRO_CHECK_ARG(aligned16(x) && (ldx & 3) == 0 && (x_seg_stride & 3) == 0 && (reinterpret_cast<uintptr_t>(stats) & 7u) == 0, "16-byte aligned rows, 8-byte aligned stats");
  hipLaunchKernelGGL(row_stats_kernel, dim3(nblk((int64_t)rows * (C >> 2), 256)), dim3(256), 0, (hipStream_t)stream, x, ldx,
                     x_seg_stride ? x_seg_stride : (int64_t)rows_per_seg * ldx, rows, rows_per_seg, C, stats);
  SOPRO_LAUNCH_CHECK();
}

int sopro_norm_f32(const float* x, int64_t ldx, int64_t x_seg_stride, float* out, int64_t ldo, const float* w, const float* b,
                   const float* mul, const float* add, int32_t rows, int32_t rows_per_seg, int32_t C, float eps,
                   int32_t kind, void* stream) {
  SOPRO_CHECK_ARG(x && out && w, "x, out, w must be non-NULL");
  SOPRO_CHECK_ARG(rows > 0 && C > 0 && C <= 64 * NORM_MAX_PER_LANE && rows_per_seg > 0, "rows > 0, 0 < C <= 1024");
  SOPRO_CHECK_ARG(kind == SOPRO_NORM_RMS || kind == SOPRO_NORM_LN, "unknown norm kind");
  hipLaunchKernelGGL(norm_kernel, dim3(nblk(rows, 4)), dim3(256), 0, (hipStream_t)stream, x, ldx,
                     x_seg_stride ? x_seg_stride : (int64_t)rows_per_seg * ldx, out, ldo, w, b, mul, add, rows, rows_per_seg, C, eps, kind);
  SOPRO_LAUNCH_CHECK();
}

int sopro_rms_match_f32(const float* a, const float* x, float* out, int32_t rows, int32_t C, void* stream) {
  SOPRO_CHECK_ARG(a && x && out && rows > 0 && C > 0, "bad pointers or sizes");
  hipLaunchKernelGGL(rms_match_kernel, dim3(nblk(rows, 4)), dim3(256), 0, (hipStream_t)stream, a, x, out, rows, C);
  SOPRO_LAUNCH_CHECK();
}

int sopro_tanh_affine_f32(const float* in, float* out, float c0, float c1, int64_t n, void* stream) {
  SOPRO_CHECK_ARG(in && out && n > 0, "bad pointers or sizes");
  hipLaunchKernelGGL(tanh_affine_kernel, dim3(nblk(n, 256)), dim3(256), 0, (hipStream_t)stream, in, out, c0, c1, n);
  SOPRO_LAUNCH_CHECK();
}

int sopro_add_pos_f32(const float* rowvec, const float* table, float* out, int32_t B, int32_t T, int32_t C, int32_t pos0,
                      void* stream) {
  SOPRO_CHECK_ARG(rowvec && table && out && B > 0 && T > 0 && C > 0 && pos0 >= 0, "bad pointers or sizes");
  hipLaunchKernelGGL(add_pos_kernel, dim3(nblk((int64_t)B * T * C, 256)), dim3(256), 0, (hipStream_t)stream, rowvec, table,
                     out, B, T, C, pos0);
  SOPRO_LAUNCH_CHECK();
}

int sopro_masked_mean_f32(const float* x, const int32_t* lens, float* out, int32_t B, int32_t T, int32_t C, void* stream) {
  SOPRO_CHECK_ARG(x && out && B > 0 && T > 0 && C > 0, "bad pointers or sizes");
  hipLaunchKernelGGL(masked_mean_kernel, dim3(nblk(C, 128), B), dim3(128), 0, (hipStream_t)stream, x, lens, out, B, T, C);
  SOPRO_LAUNCH_CHECK();
}

int sopro_dwconv_f32(const float* x, const float* w, const float* bias, const float* res, float* out, const int32_t* lens,
                     int32_t B, int32_t T, int32_t C, int32_t ksize, int32_t dil, int32_t left, int32_t mode, void* stream) {
  SOPRO_CHECK_ARG(x && w && out && B > 0 && T > 0 && C > 0 && (C & 3) == 0, "bad pointers or sizes (C % 4 == 0)");
  SOPRO_CHECK_ARG(ksize >= 1 && dil >= 1 && left >= 0 && mode >= 0 && mode <= 2, "bad conv geometry or mode");
  SOPRO_CHECK_ARG(mode != 1 || res, "mode 1 needs res");
  SOPRO_CHECK_ARG(aligned16(x) && aligned16(w) && aligned16(out), "pointers must be 16-byte aligned");
  SOPRO_CHECK_ARG((!bias || aligned16(bias)) && (!res || aligned16(res)), "bias / res must be 16-byte aligned");
  // long inputs of the two kernel sizes the model has (refinement 11, encoders 7): the comb form (same results, 1 / 5 of the requests)
  static const bool comb_off = SOPRO_DEV_ENV("SOPRO_DWCONV_COMB") != nullptr && SOPRO_DEV_ENV("SOPRO_DWCONV_COMB")[0] == '0';  // developer A/B
  if (!comb_off && (ksize == 11 || ksize == 7) && (int64_t)B * T >= 1024 && T >= 8 * dil) {
    const int per_r = ((T + dil - 1) / dil + 7) / 8;
    const dim3 grid(nblk((int64_t)B * dil * per_r * (C / 4), 256));
    if (ksize == 11)
      hipLaunchKernelGGL(dwconv_comb_kernel<11>, grid, dim3(256), 0, (hipStream_t)stream, x, w, bias, res, out, lens, B, T, C, dil, left, mode, per_r);
    else
      hipLaunchKernelGGL(dwconv_comb_kernel<7>, grid, dim3(256), 0, (hipStream_t)stream, x, w, bias, res, out, lens, B, T, C, dil, left, mode, per_r);
    SOPRO_LAUNCH_CHECK();
  }
  hipLaunchKernelGGL(dwconv_kernel, dim3(nblk((int64_t)B * T * (C / 4), 256)), dim3(256), 0, (hipStream_t)stream, x, w, bias,
                     res, out, lens, B, T, C, ksize, dil, left, mode);
  SOPRO_LAUNCH_CHECK();
}

int sopro_codebook_sum_f32(const int32_t* tok, int32_t ldt, const int32_t* col, const int32_t* off, const float* wq,
                           int32_t nq, const float* table, int64_t table_rows, const float* base, float alpha, float beta,
                           float* out, int64_t ldo, int64_t o_seg_stride, int32_t rows, int32_t rows_per_seg, int32_t D,
                           void* stream) {
  SOPRO_CHECK_ARG(tok && col && off && wq && table && out, "NULL pointer");
  SOPRO_CHECK_ARG(rows > 0 && rows_per_seg > 0 && nq > 0 && D > 0 && (D & 3) == 0 && table_rows > 0, "bad sizes (D % 4 == 0)");
  SOPRO_CHECK_ARG(aligned16(table) && aligned16(out) && (ldo & 3) == 0 && (o_seg_stride & 3) == 0, "alignment");
  hipLaunchKernelGGL(codebook_sum_kernel, dim3(nblk((int64_t)rows * (D / 4), 256)), dim3(256), 0, (hipStream_t)stream, tok,
                     ldt, col, off, wq, nq, table, table_rows, base, alpha, beta, out, ldo, o_seg_stride, rows, rows_per_seg, D);
  SOPRO_LAUNCH_CHECK();
}

int sopro_text_embed_f32(const int32_t* ids, const int32_t* lens, const float* table, int64_t table_rows, const float* pe,
                         float* out, int32_t B, int32_t T, int32_t C, void* stream) {
  SOPRO_CHECK_ARG(ids && table && pe && out && B > 0 && T > 0 && C > 0 && (C & 3) == 0, "bad pointers or sizes");
  hipLaunchKernelGGL(text_embed_kernel, dim3(nblk((int64_t)B * T * (C / 4), 256)), dim3(256), 0, (hipStream_t)stream, ids, lens,
                     table, table_rows, pe, out, B, T, C);
  SOPRO_LAUNCH_CHECK();
}

int sopro_argmax_rows_f32(const float* x, int64_t ldx, int32_t* out, int64_t ldo, int32_t inner, int32_t rows, int32_t N,
                          void* stream) {
  SOPRO_CHECK_ARG(x && out && rows > 0 && N > 0 && inner > 0, "bad pointers or sizes");
  hipLaunchKernelGGL(argmax_rows_kernel, dim3(nblk(rows, 4)), dim3(256), 0, (hipStream_t)stream, x, ldx, out, ldo, inner, rows, N);
  SOPRO_LAUNCH_CHECK();
}

int sopro_fir1_f32(const float* x, int64_t x_seg_stride, int32_t n_in, const float* w, const float* bias, float* out, int64_t ldo,
                   int64_t o_seg_stride, int32_t B, int32_t n_out, int32_t C, int32_t K, int32_t stride, int32_t left, void* stream) {
  SOPRO_CHECK_ARG(x && w && out, "NULL pointer");
  SOPRO_CHECK_ARG(B > 0 && B <= 65535 && n_in > 0 && n_out > 0 && C > 0 && K > 0 && stride > 0 && left >= 0 && ldo >= C, "bad sizes");
  hipLaunchKernelGGL(fir1_kernel, dim3(nblk((int64_t)n_out * C, 256), B), dim3(256), 0, (hipStream_t)stream, x, x_seg_stride, n_in, w,
                     bias, out, ldo, o_seg_stride, n_out, C, K, stride, left);
  SOPRO_LAUNCH_CHECK();
}

int sopro_argmax_partials_i32(const float* partials, int64_t ldp, int32_t* out, int64_t ldo, int32_t heads, int32_t per_head, int32_t V,
                              int32_t rows, void* stream) {
  SOPRO_CHECK_ARG(partials && out && rows > 0 && heads > 0 && per_head > 0 && V > 0 && ldp >= (int64_t)heads * per_head && ldo >= heads,
                  "bad pointers or sizes");
  SOPRO_CHECK_ARG((reinterpret_cast<uintptr_t>(partials) & 7u) == 0, "partials must be 8-byte aligned");
  const int64_t total = (int64_t)rows * heads;
  hipLaunchKernelGGL(argmax_partials_kernel, dim3(nblk(total * 8, 256)), dim3(256), 0, (hipStream_t)stream, partials, ldp, out, ldo, heads,
                     per_head, V, total);
  SOPRO_LAUNCH_CHECK();
}

int sopro_rvq_assign_f32(const float* scores, int64_t lds, int32_t V, const float* table, float* res, int64_t ldr, int32_t D,
                         int32_t* codes, int64_t ldc, int32_t rows, void* stream) {
  SOPRO_CHECK_ARG(scores && table && res && codes, "NULL pointer");
  SOPRO_CHECK_ARG(rows > 0 && V > 0 && D > 0 && lds >= V && ldr >= D && ldc > 0, "bad sizes");
  hipLaunchKernelGGL(rvq_assign_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, scores, lds, V, table, res, ldr, D, codes, ldc);
  SOPRO_LAUNCH_CHECK();
}

int sopro_rope_f32(float* x, int64_t ldx, const float* cos_t, const float* sin_t, int32_t rows, int32_t rows_per_seg,
                   int32_t pos0, int32_t H, int32_t dh, void* stream) {
  SOPRO_CHECK_ARG(x && cos_t && sin_t && rows > 0 && rows_per_seg > 0 && H > 0 && dh > 0 && (dh & 1) == 0 && pos0 >= 0,
                  "bad pointers or sizes");
  hipLaunchKernelGGL(rope_kernel, dim3(nblk((int64_t)rows * H * (dh / 2), 256)), dim3(256), 0, (hipStream_t)stream, x, ldx, cos_t,
                     sin_t, rows, rows_per_seg, pos0, H, dh);
  SOPRO_LAUNCH_CHECK();
}

int sopro_upsample2_f32(const float* x, const float* w, float* y, int64_t y_seg_stride, int32_t B, int32_t T, int32_t C,
                        void* stream) {
  SOPRO_CHECK_ARG(x && w && y && B > 0 && T > 0 && C > 0, "bad pointers or sizes");
  hipLaunchKernelGGL(upsample2_kernel, dim3(nblk((int64_t)B * T * 2 * C, 256)), dim3(256), 0, (hipStream_t)stream, x, w, y,
                     y_seg_stride, B, T, C);
  SOPRO_LAUNCH_CHECK();
}

int sopro_final_conv_f32(const float* h, int64_t h_seg_stride, const float* w, float bias, float* wav,
                         int64_t wav_seg_stride, int32_t B, int32_t T, void* stream) {
  SOPRO_CHECK_ARG(h && w && wav && B > 0 && T > 0, "bad pointers or sizes");
  SOPRO_CHECK_ARG(aligned16(h) && aligned16(w) && (h_seg_stride & 3) == 0, "alignment");
  hipLaunchKernelGGL(final_conv_kernel, dim3(nblk(T, 256), B), dim3(256), 0, (hipStream_t)stream, h, h_seg_stride, w, bias, wav,
                     wav_seg_stride, B, T);
  SOPRO_LAUNCH_CHECK();
}

int sopro_stats_pool_f32(const float* h, const float* logit, const int32_t* lens, float* out, int32_t B, int32_t T, int32_t C,
                         void* stream) {
  SOPRO_CHECK_ARG(h && logit && out && B > 0 && T > 0 && C > 0, "bad pointers or sizes");
  hipLaunchKernelGGL(stats_pool_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, h, logit, lens, out, B, T, C);
  SOPRO_LAUNCH_CHECK();
}

int sopro_l2norm_f32(const float* x, float* out, int32_t rows, int32_t C, float eps, void* stream) {
  SOPRO_CHECK_ARG(x && out && rows > 0 && C > 0, "bad pointers or sizes");
  hipLaunchKernelGGL(l2norm_kernel, dim3(nblk(rows, 4)), dim3(256), 0, (hipStream_t)stream, x, out, rows, C, eps);
  SOPRO_LAUNCH_CHECK();
}

int sopro_fill2d_u32(void* p, int64_t pitch, int32_t rows, int32_t width, uint32_t value, void* stream) {
  SOPRO_CHECK_ARG(p && rows > 0 && width > 0 && pitch >= width, "bad pointers or sizes");
  hipLaunchKernelGGL(fill2d_kernel, dim3(nblk((int64_t)rows * width, 256)), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<unsigned*>(p), pitch, rows,
                     width, value);
  SOPRO_LAUNCH_CHECK();
}

int sopro_copy2d_u32(void* dst, int64_t dpitch, const void* src, int64_t spitch, int32_t rows, int32_t width, void* stream) {
  SOPRO_CHECK_ARG(dst && src && rows > 0 && width > 0 && dpitch >= width && spitch >= width, "bad pointers or sizes");
  hipLaunchKernelGGL(copy2d_kernel, dim3(nblk((int64_t)rows * width, 256)), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<unsigned*>(dst), dpitch,
                     reinterpret_cast<const unsigned*>(src), spitch, rows, width);
  SOPRO_LAUNCH_CHECK();
}

int sopro_nar_seed_i32(int32_t* tokens, int32_t Q, const int32_t* cb0, int64_t cb0_bstride, int32_t B, int32_t T, int32_t vmax, void* stream) {
  SOPRO_CHECK_ARG(tokens && cb0 && Q > 0 && B > 0 && T > 0 && cb0_bstride >= T && vmax >= 0, "bad pointers or sizes");
  hipLaunchKernelGGL(nar_seed_kernel, dim3(nblk((int64_t)B * T, 256)), dim3(256), 0, (hipStream_t)stream, tokens, Q, cb0, cb0_bstride, B, T, vmax);
  SOPRO_LAUNCH_CHECK();
}

int sopro_cvt_f32_bf16(const float* src, void* dst, int64_t n, void* stream) {
  SOPRO_CHECK_ARG(src && dst && n > 0 && (n & 3) == 0 && aligned16(src) && (reinterpret_cast<uintptr_t>(dst) & 7u) == 0, "n % 4 == 0, 16 / 8-byte aligned");
  hipLaunchKernelGGL(cvt_f32_bf16_kernel, dim3(nblk(n / 4, 256)), dim3(256), 0, (hipStream_t)stream, src, reinterpret_cast<uint2*>(dst), n / 4);
  SOPRO_LAUNCH_CHECK();
}

int sopro_cvt_bf16_f32(const void* src, float* dst, int64_t n, void* stream) {
  SOPRO_CHECK_ARG(src && dst && n > 0 && (n & 3) == 0 && aligned16(dst) && (reinterpret_cast<uintptr_t>(src) & 7u) == 0, "n % 4 == 0, 16 / 8-byte aligned");
  hipLaunchKernelGGL(cvt_bf16_f32_kernel, dim3(nblk(n / 4, 256)), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const uint2*>(src), dst, n / 4);
  SOPRO_LAUNCH_CHECK();
}

}  // extern "C"
