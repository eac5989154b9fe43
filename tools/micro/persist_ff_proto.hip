// Developer prototype (round 3, verdict item 1b): FF1 -> FF2 of one AR block for 32 rows as ONE resident kernel against the
// same two stages as two launches of a replayed hipGraph.  Not part of the library.
//
// The pair, with the workgroup shapes of the frame's skinny kernels (192 workgroups x 256 threads per stage, exact-fp32
// v_mfma_f32_16x16x4_f32, weight slices resident in registers for the whole run):
//   A (FF1): u[32, 1536] = gelu((sum of the 4 partial-sum slabs P[4][32, 384] of the previous stage) . W1^T + b1)
//            workgroup = (16-row group, 16-column tile): ingests 4 x 24 KB of partial sums, emits a 16 x 16 tile of u
//   B (FF2): P'[ks][32, 384] = u[:, 384 ks .. 384 ks + 383] . W2[:, same]^T     (4 K-slices, summed by the next consumer)
//            workgroup = (16-row group, K-slice, 16-column tile): ingests 24 KB of u, emits a 16 x 16 partial tile
// Resident form: every workgroup owns one A role and one B role and loops over `iters` pairs.  Hand-offs are the placement-
// independent counter protocol of the platform guide (per-wave vmcnt(0) -> barrier -> one-lane agent-scope release -> relaxed
// add on a counter sharded 8 ways by producer; consumer: one wave polls the 8 shards with relaxed agent-scope loads + s_sleep,
// one-lane agent-scope acquire, barrier, plain loads).  A's wait is the all-to-all edge (96 producers of its row group), B's a
// 24 -> 1 fan-in.  Every spin is bounded (a stuck run sets `abort`, never hangs the GPU).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/persist_ff tools/micro/persist_ff_proto.hip && /tmp/persist_ff
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// Round 4 (the no-go above was measured at 32 rows, where a hand-off moves 24-96 KB; at batch <= 4 it is <= 6 KB):
//   -DPROTO_ROWS=16  one 16-row group (96 workgroups per stage), -DPROTO_BR=<1|4|16> real rows of the group (the others are padding:
//   neither loaded nor stored, in the launched AND the resident form) - the frame of stream() / a batch-1 synthesize().
#ifndef PROTO_ROWS
#define PROTO_ROWS 32
#endif
#ifndef PROTO_BR
#define PROTO_BR 16
#endif
constexpr int ROWS = PROTO_ROWS, BR = PROTO_BR, D = 384, F = 1536, NWG = 96 * (ROWS / 16), ALD = 388;  // ALD: LDS row stride (floats), 16-byte rows, conflict-free b128 reads
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct Params {
  const float *W1, *b1, *W2, *b2;
  float* P;        // [2 (pair parity)][4 (K-slice)][32][384]
  float* U;        // [32][1536]
  unsigned* xcnt;  // [2 row groups][8 shards]       arrivals of B workgroups (96 per row group and pair)
  unsigned* ucnt;  // [2 row groups][4 K-slices][8]  arrivals of A workgroups (24 per (row group, K-slice) and pair)
  unsigned* abort;
  int iters, parity0;
};

__device__ __forceinline__ float gelu(float v) { return 0.5f * v * (1.f + erff(v * 0.70710678f)); }

// 16 x 16 output tile over K = 384: wave = K quarter, lane group g = lane / 16 owns k in [96 wave + 24 g, + 24)
__device__ __forceinline__ f32x4 tile_mma(const float* As, const float (&w)[24], int lane, int wave) {
  const float* a = As + (lane & 15) * ALD + wave * 96 + (lane >> 4) * 24;
  float av[24];
#pragma unroll
  for (int q = 0; q < 6; ++q) {
    const float4 t = *reinterpret_cast<const float4*>(a + q * 4);
    av[q * 4] = t.x; av[q * 4 + 1] = t.y; av[q * 4 + 2] = t.z; av[q * 4 + 3] = t.w;
  }
  f32x4 c = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int s = 0; s < 24; ++s) c = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s], w[s], c, 0, 0, 0);
  return c;
}

// the four K quarters meet in LDS and are added in a fixed order; thread t then owns element (row, col) of the tile
__device__ __forceinline__ float reduce_tile(float* red, f32x4 c, int tid, int& row, int& col) {
  const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
  for (int i = 0; i < 4; ++i) red[(wave * 4 + i) * 64 + lane] = c[i];
  __syncthreads();
  const int l = tid & 63, i = tid >> 6;
  row = 4 * (l >> 4) + i;
  col = l & 15;
  return ((red[(0 * 4 + i) * 64 + l] + red[(1 * 4 + i) * 64 + l]) + red[(2 * 4 + i) * 64 + l]) + red[(3 * 4 + i) * 64 + l];
}

// MODE 1 transport: payload with system-coherent-cache-bypassing (sc0 sc1, write-through / read-through) accesses and a drained
// flag, no fences (the guide's "drained sc1" hand-off form)
__device__ __forceinline__ f32x4 ld_sc1(const float* p) {
  f32x4 v;
  asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(v) : "v"(p) : "memory");
  return v;
}
__device__ __forceinline__ f32x4 ld_plain(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ void wait6(f32x4 (&v)[6], bool first) {  // ties the values to the wait: no use can be moved above it
  if (first)
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5])::"memory");
  else
    asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5])::"memory");
}
__device__ __forceinline__ void st_sc1(float* p, float v) { asm volatile("global_store_dword %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory"); }

template <int MODE>
__device__ __forceinline__ void publish(unsigned* ctr) {  // after the tile's stores
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    if (MODE == 0) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// wave 0 polls the 8 shards (lanes 0-7); false = spin limit or another workgroup's abort
template <int MODE>
__device__ __forceinline__ bool await(unsigned* shards, unsigned target, unsigned* abort, int* flag_lds) {
  if (threadIdx.x < 64) {
    const int lane = threadIdx.x;
    bool ok = true;
    unsigned spins = 0;
    for (;;) {
      const unsigned v = lane < 8 ? __hip_atomic_load(shards + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : target;
      if (__all((int)(v - target) >= 0)) break;
      if (++spins > (1u << 20) || (((spins & 1023u) == 0) && __hip_atomic_load(abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) { ok = false; break; }
      __builtin_amdgcn_s_sleep(1);
    }
    if (lane == 0) {
      if (!ok) __hip_atomic_store(abort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (MODE == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      *flag_lds = ok ? 1 : 0;
    }
  }
  __syncthreads();
  return *flag_lds != 0;
}

template <int MODE>
__device__ __forceinline__ void stage_a(const Params& p, int w, int parity, const float (&w1)[24], float b1v, float* As, float* red) {
  const int tid = threadIdx.x, ra = w / 96, ca = w % 96;
  const float* P = p.P + (size_t)parity * 4 * ROWS * D + (size_t)(16 * ra) * D;
  // (two rounds of two slabs: 48 staging registers, so that three workgroups fit a CU of the 64-CU partition)
  f32x4 v[2][6], acc[6];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int q = 0; q < 6; ++q) {
        const float* src = P + (size_t)(2 * h + j) * ROWS * D + (size_t)(tid + q * 256) * 4;
        v[j][q] = f32x4{0.f, 0.f, 0.f, 0.f};
        if ((tid + q * 256) * 4 / D < BR) v[j][q] = MODE == 1 ? ld_sc1(src) : ld_plain(src);  // (padding rows are not moved)
      }
    if (MODE == 1) { wait6(v[0], true); wait6(v[1], false); }
#pragma unroll
    for (int q = 0; q < 6; ++q) {
      if (h == 0) acc[q] = v[0][q] + v[1][q];
      else acc[q] = (acc[q] + v[0][q]) + v[1][q];
    }
    __builtin_amdgcn_sched_barrier(0);  // (keeps the two load rounds apart: all four slabs at once need 96 registers)
  }
#pragma unroll
  for (int q = 0; q < 6; ++q) {
    const int e = (tid + q * 256) * 4, r = e / D, k = e % D;
    const float4 bb = *reinterpret_cast<const float4*>(p.b2 + k);
    auto cl = [](float t) { return fminf(fmaxf(t, -4.f), 4.f); };  // (keeps the synthetic stream bounded over thousands of pairs)
    *reinterpret_cast<float4*>(As + r * ALD + k) = make_float4(cl(acc[q].x + bb.x), cl(acc[q].y + bb.y), cl(acc[q].z + bb.z), cl(acc[q].w + bb.w));
  }
  __syncthreads();
  const f32x4 c = tile_mma(As, w1, tid & 63, tid >> 6);
  int row, col;
  const float s = reduce_tile(red, c, tid, row, col);
  float* dst = p.U + (size_t)(16 * ra + row) * F + 16 * ca + col;
  if (row < BR) { if (MODE == 1) st_sc1(dst, gelu(s + b1v)); else *dst = gelu(s + b1v); }
}

template <int MODE>
__device__ __forceinline__ void stage_b(const Params& p, int w, int parity_out, const float (&w2)[24], float* As, float* red) {
  const int tid = threadIdx.x, rb = w / 96, ks = (w % 96) / 24, cb = w % 24;
  const float* U = p.U + (size_t)(16 * rb) * F + 384 * ks;
  f32x4 uv[6];
#pragma unroll
  for (int q = 0; q < 6; ++q) {
    const int e = (tid + q * 256) * 4, r = e / D, k = e % D;
    uv[q] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (r < BR) uv[q] = MODE == 1 ? ld_sc1(U + (size_t)r * F + k) : ld_plain(U + (size_t)r * F + k);
  }
  if (MODE == 1) wait6(uv, true);
#pragma unroll
  for (int q = 0; q < 6; ++q) {
    const int e = (tid + q * 256) * 4, r = e / D, k = e % D;
    *reinterpret_cast<f32x4*>(As + r * ALD + k) = uv[q];
  }
  __syncthreads();
  const f32x4 c = tile_mma(As, w2, tid & 63, tid >> 6);
  int row, col;
  const float s = reduce_tile(red, c, tid, row, col);
  float* dst = p.P + (size_t)parity_out * 4 * ROWS * D + (size_t)ks * ROWS * D + (size_t)(16 * rb + row) * D + 16 * cb + col;
  if (row < BR) { if (MODE == 1) st_sc1(dst, s); else *dst = s; }
}

__device__ __forceinline__ void load_w1(const Params& p, int w, float (&w1)[24], float& b1v) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, ca = w % 96;
  const float* a = p.W1 + (size_t)(16 * ca + (lane & 15)) * D + wave * 96 + (lane >> 4) * 24;
#pragma unroll
  for (int s = 0; s < 24; ++s) w1[s] = a[s];
  b1v = p.b1[16 * ca + (lane & 15)];
}
__device__ __forceinline__ void load_w2(const Params& p, int w, float (&w2)[24]) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, ks = (w % 96) / 24, cb = w % 24;
  const float* b = p.W2 + (size_t)(16 * cb + (lane & 15)) * F + 384 * ks + wave * 96 + (lane >> 4) * 24;
#pragma unroll
  for (int s = 0; s < 24; ++s) w2[s] = b[s];
}

template <int MODE>
__global__ __launch_bounds__(256, 3) void persist_pair(const Params p) {
  __shared__ __attribute__((aligned(16))) float As[16 * ALD];
  __shared__ float red[16 * 64];
  __shared__ int flag;
  const int w = blockIdx.x;
  float w1[24], w2[24], b1v;
  load_w1(p, w, w1, b1v);
  load_w2(p, w, w2);
  const int ra = w / 96, ks = (w % 96) / 24;
  for (int t = 0; t < p.iters; ++t) {
    const int parity = (p.parity0 + t) & 1;
    if (!await<MODE>(p.xcnt + ra * 8, 12u * (unsigned)t, p.abort, &flag)) return;  // all 96 B workgroups of my row group finished pair t-1
    stage_a<MODE>(p, w, parity, w1, b1v, As, red);
    publish<MODE>(p.ucnt + (ra * 4 + ks) * 8 + (w & 7));                            // (my A tile belongs to K-slice ca / 24 == ks)
    if (!await<MODE>(p.ucnt + (ra * 4 + ks) * 8, 3u * (unsigned)(t + 1), p.abort, &flag)) return;  // the 24 A tiles of my K-slice
    stage_b<MODE>(p, w, parity ^ 1, w2, As, red);
    publish<MODE>(p.xcnt + ra * 8 + (w & 7));
  }
}

__global__ __launch_bounds__(256) void launched_a(const Params p, int parity) {
  __shared__ __attribute__((aligned(16))) float As[16 * ALD];
  __shared__ float red[16 * 64];
  float w1[24], b1v;
  load_w1(p, blockIdx.x, w1, b1v);  // (the launched form re-reads its weight slice every launch, as the frame's kernels do)
  stage_a<0>(p, blockIdx.x, parity, w1, b1v, As, red);
}
__global__ __launch_bounds__(256) void launched_b(const Params p, int parity_out) {
  __shared__ __attribute__((aligned(16))) float As[16 * ALD];
  __shared__ float red[16 * 64];
  float w2[24];
  load_w2(p, blockIdx.x, w2);
  stage_b<0>(p, blockIdx.x, parity_out, w2, As, red);
}

__global__ void copy_load(const float4* __restrict__ a, float4* __restrict__ b, size_t n, int reps) {
  for (int r = 0; r < reps; ++r)
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}

static hipStream_t masked_stream(int first, int n, int total) {
  uint32_t mask[16] = {0};
  for (int c = first; c < first + n; ++c) mask[c >> 5] |= 1u << (c & 31);
  hipStream_t s;
  CK(hipExtStreamCreateWithCUMask(&s, (total + 31) / 32, mask));
  return s;
}

int main() {
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int total = prop.multiProcessorCount;
  std::vector<float> hW1((size_t)F * D), hW2((size_t)D * F), hb1(F), hb2(D), hx((size_t)ROWS * D);
  srand(7);
  auto rnd = [] { return (float)rand() / RAND_MAX * 2.f - 1.f; };
  for (auto& v : hW1) v = rnd() * 0.09f;
  for (auto& v : hW2) v = rnd() * 0.075f;
  for (auto& v : hb1) v = rnd() * 0.1f;
  for (auto& v : hb2) v = rnd() * 0.1f;
  for (auto& v : hx) v = rnd();
  float *W1, *W2, *b1, *b2, *P, *U, *Pref;
  unsigned* cnt;
  CK(hipMalloc(&W1, hW1.size() * 4)); CK(hipMalloc(&W2, hW2.size() * 4)); CK(hipMalloc(&b1, F * 4)); CK(hipMalloc(&b2, D * 4));
  CK(hipMalloc(&P, 2 * 4 * ROWS * D * 4)); CK(hipMalloc(&Pref, 2 * 4 * ROWS * D * 4)); CK(hipMalloc(&U, ROWS * F * 4)); CK(hipMalloc(&cnt, 4096));
  CK(hipMemcpy(W1, hW1.data(), hW1.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(W2, hW2.data(), hW2.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(b1, hb1.data(), F * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(b2, hb2.data(), D * 4, hipMemcpyHostToDevice));
  float4 *big_a, *big_b;
  const size_t big_n = (size_t)64 << 20;  // 1 GiB each
  CK(hipMalloc(&big_a, big_n * 16)); CK(hipMalloc(&big_b, big_n * 16));
  CK(hipMemset(big_a, 0, big_n * 16));
  Params p{W1, b1, W2, b2, P, U, cnt, cnt + 64, cnt + 512, 0, 0};
  auto reset = [&](hipStream_t s) {
    CK(hipMemsetAsync(P, 0, 2 * 4 * ROWS * D * 4, s));
    CK(hipMemcpyAsync(P, hx.data(), hx.size() * 4, hipMemcpyHostToDevice, s));  // parity 0, K-slice 0 = x0
    CK(hipMemsetAsync(cnt, 0, 4096, s));
  };
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int T = 2000, GN = 100;  // pairs per persistent launch; pairs per graph
  struct Cfg { const char* name; int first, n; bool load; };
  const Cfg cfgs[] = {{"whole chip, alone", 0, total, false}, {"64-CU partition, alone", 0, 64, false}, {"64-CU partition, copy loop on the other 192 CUs", 0, 64, true}};
  for (const Cfg& c : cfgs) {
    hipStream_t s = masked_stream(c.first, c.n, total), ls = nullptr;
    if (c.load) {
      ls = masked_stream(64, total - 64, total);
      hipLaunchKernelGGL(copy_load, dim3(768), dim3(256), 0, ls, big_a, big_b, big_n, 200);  // ~200 x 2 GiB of traffic: outlasts the measurements
    }
    // ---- launched: a graph of GN pairs, replayed
    reset(s);
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    for (int t = 0; t < GN; ++t) {
      hipLaunchKernelGGL(launched_a, dim3(NWG), dim3(256), 0, s, p, t & 1);
      hipLaunchKernelGGL(launched_b, dim3(NWG), dim3(256), 0, s, p, (t & 1) ^ 1);
    }
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CK(hipGraphLaunch(ge, s));  // warm
    reset(s);
    CK(hipEventRecord(e0, s));
    for (int r = 0; r < T / GN; ++r) CK(hipGraphLaunch(ge, s));
    CK(hipEventRecord(e1, s));
    CK(hipStreamSynchronize(s));
    float ms_l; CK(hipEventElapsedTime(&ms_l, e0, e1));
    CK(hipMemcpy(Pref, P, 2 * 4 * ROWS * D * 4, hipMemcpyDeviceToDevice));
    // ---- resident, two transports
    Params q = p; q.iters = T;
    float ms_p[2]; bool same[2]; unsigned ab[2]; double nrm = 0;
    for (int mode = 0; mode < 2; ++mode) {
      auto kern = mode == 0 ? persist_pair<0> : persist_pair<1>;
      reset(s);
      hipLaunchKernelGGL(kern, dim3(NWG), dim3(256), 0, s, q);  // warm (also the correctness run)
      CK(hipStreamSynchronize(s));
      std::vector<float> a(2 * 4 * ROWS * D), b(2 * 4 * ROWS * D);
      CK(hipMemcpy(a.data(), P, a.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(b.data(), Pref, b.size() * 4, hipMemcpyDeviceToHost));
      same[mode] = memcmp(a.data(), b.data(), a.size() * 4) == 0;
      nrm = 0; for (float v : a) nrm += (double)v * v;
      reset(s);
      CK(hipEventRecord(e0, s));
      hipLaunchKernelGGL(kern, dim3(NWG), dim3(256), 0, s, q);
      CK(hipEventRecord(e1, s));
      CK(hipStreamSynchronize(s));
      CK(hipEventElapsedTime(&ms_p[mode], e0, e1));
      CK(hipMemcpy(&ab[mode], cnt + 512, 4, hipMemcpyDeviceToHost));
    }
    printf("rows %d (%d real per group) | %-50s launched (graph of %d pairs): %6.2f us per pair | resident, fences: %6.2f (%.2fx, %s%s) | resident, sc1 payload + drained flag: %6.2f (%.2fx, %s%s) | |P| %.3e\n",
           ROWS, BR, c.name, GN, ms_l * 1e3 / T, ms_p[0] * 1e3 / T, ms_l / ms_p[0], same[0] ? "identical" : "DIFFER", ab[0] ? ", SPIN LIMIT" : "",
           ms_p[1] * 1e3 / T, ms_l / ms_p[1], same[1] ? "identical" : "DIFFER", ab[1] ? ", SPIN LIMIT" : "", sqrt(nrm));
    fflush(stdout);
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    if (ls) { CK(hipStreamSynchronize(ls)); CK(hipStreamDestroy(ls)); }
    CK(hipStreamDestroy(s));
  }
  return 0;
}
