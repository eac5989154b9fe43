// Three-pass split-bf16 contraction, LONG-K form (round 6): 256 x 256 tiles on eight waves, BOTH operands already split in memory
// ("split form": every aligned group of 32 k of a row = [32 hi bf16 | 32 lo bf16] = 128 bytes) and brought into LDS by LDS-DMA
// (global_load_lds_dwordx4) - no staging registers, no split work, no ds_write in the main loop.  Same products in the same order per
// output element as gemm_bf16s.hip's NPL = 2 kernel on a split-form A: lo*hi + hi*lo + hi*hi per 16-k substep, K ascending -
// bit-identical results (tests/test_gpu_ops.py).  For the SEANet decoder's transposed convolutions with K >= 1024
// (HF:modeling_mimi.py:350-405, MimiConvTranspose1d as a row-window contraction) and the k = 3 convolution of the first residual
// block (:408-447): measured on the 192-CU partition 320-337 TFLOP/s fp32-equivalent against 243-245 for the tile kernel
// (profiles/r06_experiments.md section 2; the bare MFMA loop on random operands runs 428 there - the chip clocks down to ~1.6 GHz
// under back-to-back MFMAs on real data).
//
// Geometry.  K-tile = 32 k = one 128-byte group per row.  LDS = 2 buffers x (A tile 256 rows x 128 B | W tile 256 rows x 128 B) =
// 128 KB; a HALF-tile = 128 rows = 16 KB = 16 wave-instructions of 1 KB, two per wave.  The DMA writes LDS lane-linearly, so the
// bank swizzle is applied on the SOURCE address: 16-byte chunk c of row r lands at chunk c ^ ((r >> 1) & 7) of its row, and the
// 16-lane groups of a ds_read_b128 fragment read (16 consecutive rows, one logical chunk) touch 16 distinct 16-byte slots.
// Wave (wm, wn) = (wave >> 2, wave & 3) owns the 32-row blocks 2 i + wm (i = 0..3) and the 32-column blocks 4 j + wn (j = 0, 1) -
// interleaved, so that each half-tile is read in ONE phase of a K-tile and can be re-staged soon after:
//     phase 0: quadrant (i 0-1, j 0)  reads A0 (8 x ds_read_b128) + B0 (4)      DMA: B1 of tile t + 1
//     phase 1: quadrant (i 0-1, j 1)  reads B1 (4)                              DMA: A1 of tile t + 1
//     phase 2: quadrant (i 2-3, j 1)  reads A1 (8)                              DMA: A0 of tile t + 2
//     phase 3: quadrant (i 2-3, j 0)  (B0 fragments still in registers)         DMA: B0 of tile t + 2
// A phase: [fragment reads] s_waitcnt vmcnt(6) s_barrier lgkmcnt(0) [12 MFMA, its two DMA instructions behind the 4th and the 8th]
// s_barrier.  The two waves of a SIMD (w, w + 4) belong to different groups (wm) and the second group runs ONE barrier behind the
// first: one group's MFMAs run under the other group's fragment reads.  Hazards, by barrier count: a half-tile is waited for (counted
// vmcnt: the three newest half-tiles stay in flight) in the phase BEFORE its first read, in front of a barrier every reader passes
// first; it is re-staged two or three phases after its last read.  Measured choices (r06_experiments.md): the stagger +6 %, DMA between
// the MFMAs instead of next to the fragment reads +5 %, without barriers -8 %.
#include "common.h"
#include "gemm_epilogue.h"

namespace {

constexpr int P_BM = 256, P_BN = 256, P_NT = 512;
constexpr int P_ROWB = 128;              // bytes per row of a K-tile
constexpr int P_TILE = 256 * P_ROWB;     // 32 KB
constexpr int P_BUF = 2 * P_TILE;        // A | W
constexpr int P_LDS = 2 * P_BUF;         // 128 KB
constexpr int P_CLD = 128 + 4;           // epilogue sub-tile: 64 rows x 128 columns, two buffers
static_assert(2 * 64 * P_CLD * 4 <= P_LDS, "epilogue tiles alias the operand buffers");

typedef __bf16 pbf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned pu32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void p_lds_void;
typedef const __attribute__((address_space(1))) void p_glb_void;

__device__ __forceinline__ pbf16x8 pfrag(const pu32x4& v) { return __builtin_bit_cast(pbf16x8, v); }
__device__ __forceinline__ void dma16(const void* g, unsigned lds_off) {  // 16 bytes per lane -> LDS (wave-uniform) lds_off + lane * 16
  __builtin_amdgcn_global_load_lds((p_glb_void*)(uintptr_t)g, (p_lds_void*)(uintptr_t)lds_off, 16, 0, 0);
}

// A: split-form rows with the fp32 geometry of sopro_gemm_args (row m at A + seg * a_seg_stride + r * lda floats; rows may overlap);
// Wr: [N][K / 32][32 hi | 32 lo] bf16 from sopro_pack_w_rows_bf16 (N % 256 == 0, K % 32 == 0)
template <int EPI, int OUT>
__global__ __launch_bounds__(P_NT, 1) void gemm_8p_kernel(const sopro_gemm_args g, const unsigned char* __restrict__ Wr, const sopro_gemm_split_ext ext) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int ntn = g.N / P_BN;
  const int bid = xcd_contiguous((int)blockIdx.x, (int)gridDim.x);
  int mt = bid / ntn, nt = bid % ntn;
  if (ext.group_m > 1) {  // grouped walk: group_m row tiles at a time, row tile fastest (sopro_gemm_split_ext.group_m)
    const int ntm_all = (g.M + P_BM - 1) / P_BM;
    const int per = ext.group_m * ntn;
    const int grp = bid / per, rem = bid - grp * per;
    const int first = grp * ext.group_m;
    const int gsz = min(ntm_all - first, ext.group_m);
    mt = first + rem % gsz;
    nt = rem / gsz;
  }
  const int m0 = mt * P_BM, n0 = nt * P_BN;
  const int KT = g.K >> 5;
  const int64_t ldw = (int64_t)g.K * 4;

  // ---- DMA sources: instruction q (0, 1) of this wave for half-tile h covers rows 128 h + 8 (wave + 8 q) + (lane >> 3); the lane
  // fetches the logical chunk that belongs at physical chunk lane & 7 of that row.  Rows >= M are clamped (never stored).
  const unsigned char* asrc[2][2];
  const unsigned char* wsrc[2][2];
  const int rps = g.rows_per_seg;
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int r = 128 * h + 8 * (wave + 8 * q) + (lane >> 3);
      const int ch = (lane & 7) ^ ((r >> 1) & 7);
      const int m = min(m0 + r, g.M - 1);
      const int seg = m / rps, rr = m - seg * rps;
      asrc[h][q] = reinterpret_cast<const unsigned char*>(g.A) + ((int64_t)seg * g.a_seg_stride + (int64_t)rr * g.lda) * 4 + ch * 16;
      wsrc[h][q] = Wr + (int64_t)(n0 + r) * ldw + ch * 16;
    }
  const unsigned lds0 = (unsigned)(uintptr_t)smem;
  auto issue_one = [&](bool isW, int h, int kt, int buf, int q) {
    const int ktc = min(kt, KT - 1);  // (beyond the last tile: a harmless re-read, never consumed)
    const unsigned char* src = (isW ? wsrc[h][q] : asrc[h][q]) + (int64_t)ktc * P_ROWB;
    dma16(src, lds0 + buf * P_BUF + (isW ? P_TILE : 0) + (128 * h + 8 * (wave + 8 * q)) * P_ROWB);
  };

  // ---- fragment read offsets: row (32-row block base + frow), logical chunk = piece * 4 + s * 2 + fg
  const int frow = lane & 31, fg = lane >> 5;
  const int swz = (frow >> 1) & 7;
  int foff[2][2];  // [piece][s]: byte offset within the row's 128 bytes, swizzled
#pragma unroll
  for (int p = 0; p < 2; ++p)
#pragma unroll
    for (int s = 0; s < 2; ++s) foff[p][s] = ((p * 4 + s * 2 + fg) ^ swz) * 16;
  const int arow = (wm * 32 + frow) * P_ROWB;  // + i * 64 rows
  const int brow = (wn * 32 + frow) * P_ROWB;  // + j * 128 rows

  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  pu32x4 af[2][2][2];  // [i within the quadrant][s][piece]
  pu32x4 bf[2][2][2];  // [j][s][piece]
  auto read_a = [&](int buf, int qi) {
    const unsigned char* base = smem + buf * P_BUF + arow + qi * 2 * 64 * P_ROWB;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int p = 0; p < 2; ++p) af[i][s][p] = *reinterpret_cast<const pu32x4*>(base + i * 64 * P_ROWB + foff[p][s]);
  };
  auto read_b = [&](int buf, int j) {
    const unsigned char* base = smem + buf * P_BUF + P_TILE + brow + j * 128 * P_ROWB;
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int p = 0; p < 2; ++p) bf[j][s][p] = *reinterpret_cast<const pu32x4*>(base + foff[p][s]);
  };
  // 12 MFMAs of a quadrant; (A piece, W piece): (lo, hi), (hi, lo), (hi, hi) per substep - the tile kernel's order.  The phase's two DMA
  // instructions go out behind the 4th and the 8th.
  auto phase_mfma = [&](int qi, int j, bool isW, int h, int kt, int buf) {
    constexpr int PA[3] = {1, 0, 0}, PB[3] = {0, 1, 0};
    int n = 0;
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int q = 0; q < 3; ++q)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          acc[qi * 2 + i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pfrag(af[i][s][PA[q]]), pfrag(bf[j][s][PB[q]]), acc[qi * 2 + i][j], 0, 0, 0);
          ++n;
          if (n == 4 || n == 8) {
            __builtin_amdgcn_sched_barrier(0);
            issue_one(isW, h, kt, buf, n == 4 ? 0 : 1);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
  };
#define P_PHASE(QI, J, W_, H_, KT_, B_)                     \
  asm volatile("s_waitcnt vmcnt(6)" ::: "memory");          \
  __builtin_amdgcn_s_barrier();                             \
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        \
  __builtin_amdgcn_sched_barrier(0);                        \
  __builtin_amdgcn_s_setprio(1);                            \
  phase_mfma(QI, J, W_, H_, KT_, B_);                       \
  __builtin_amdgcn_s_setprio(0);                            \
  __builtin_amdgcn_sched_barrier(0);                        \
  __builtin_amdgcn_s_barrier();

  // ---- prologue: what the steady state would have issued before tile 0, in its order: A0(0) B0(0) B1(0) A1(0) A0(1) B0(1)
#pragma unroll
  for (int q = 0; q < 2; ++q) issue_one(false, 0, 0, 0, q);
#pragma unroll
  for (int q = 0; q < 2; ++q) issue_one(true, 0, 0, 0, q);
#pragma unroll
  for (int q = 0; q < 2; ++q) issue_one(true, 1, 0, 0, q);
#pragma unroll
  for (int q = 0; q < 2; ++q) issue_one(false, 1, 0, 0, q);
#pragma unroll
  for (int q = 0; q < 2; ++q) issue_one(false, 0, 1, 1, q);
#pragma unroll
  for (int q = 0; q < 2; ++q) issue_one(true, 0, 1, 1, q);
  asm volatile("s_waitcnt vmcnt(8)" ::: "memory");  // A0(0), B0(0) have landed
  __builtin_amdgcn_s_barrier();
  if (wm == 1) __builtin_amdgcn_s_barrier();  // the second wave group runs one barrier behind

  for (int t = 0; t < KT; ++t) {
    const int b = t & 1;
    read_a(b, 0);
    read_b(b, 0);
    P_PHASE(0, 0, true, 1, t + 1, b ^ 1)
    read_b(b, 1);
    P_PHASE(0, 1, false, 1, t + 1, b ^ 1)
    read_a(b, 1);
    P_PHASE(1, 1, false, 0, t + 2, b)
    P_PHASE(1, 0, true, 0, t + 2, b)
  }
#undef P_PHASE
  if (wm == 0) __builtin_amdgcn_s_barrier();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // no DMA may land in LDS once the epilogue tiles are written there
  __syncthreads();

  // ---- epilogue: eight 64-row x 128-column sub-tiles (i, j) through gemm_store_tile (rows 64 i + 32 wm + .., columns 128 j + 32 wn + ..
  // = exactly this wave's blocks), two LDS tiles alternating: a sub-tile's buffer was last read two calls ago, behind the barrier of the
  // call in between
  float* Cs[2] = {reinterpret_cast<float*>(smem), reinterpret_cast<float*>(smem) + 64 * P_CLD};
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int nb = n0 + 128 * j;
      const int n = nb + wn * 32 + frow;
      const float bv[1] = {(g.bias && n < g.N) ? g.bias[n] : 0.f};
      gemm_store_tile<2, 4, 1, 1, EPI, OUT>(g, Cs[(i * 2 + j) & 1], reinterpret_cast<f32x16(&)[1][1]>(acc[i][j]), bv, m0 + 64 * i, nb, &ext, nullptr);
    }
}

// W [N, ldw] fp32 -> rows in split form; one thread per 16-byte chunk (8 k of one piece)
__global__ __launch_bounds__(256) void pack_w_rows_kernel(const float* __restrict__ W, int64_t ldw, int N, int K, uint4* __restrict__ out, int64_t total) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int c = (int)(i & 7);               // chunk of the 128-byte group: piece = c >> 2, k offset = (c & 3) * 8
  const int64_t gi = i >> 3;
  const int kg = (int)(gi % (K >> 5));
  const int n = (int)(gi / (K >> 5));
  const float* src = W + (int64_t)n * ldw + kg * 32 + (c & 3) * 8;
  unsigned w[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    unsigned hi, lo;
    split2_bf16(src[2 * e], src[2 * e + 1], hi, lo);
    w[e] = (c >> 2) ? lo : hi;
  }
  out[i] = make_uint4(w[0], w[1], w[2], w[3]);
}

template <int EPI, int OUT>
int launch_8p(const sopro_gemm_args& g, const void* wrows, const sopro_gemm_split_ext& ext, hipStream_t s) {
  auto kern = gemm_8p_kernel<EPI, OUT>;
  SOPRO_SET_MAX_LDS_ONCE(kern, P_LDS);
  const int ntm = (g.M + P_BM - 1) / P_BM, ntn = g.N / P_BN;
  hipLaunchKernelGGL(kern, dim3(ntm * ntn), dim3(P_NT), P_LDS, s, g, reinterpret_cast<const unsigned char*>(wrows), ext);
  SOPRO_LAUNCH_CHECK();
}

}  // namespace

extern "C" int64_t sopro_packed_w_rows_bytes(int32_t N, int32_t K) {
  if (N <= 0 || K <= 0 || (K & 31) != 0) return 0;
  return (int64_t)N * K * 4;
}

extern "C" int sopro_pack_w_rows_bf16(const float* W, int64_t ldw, int32_t N, int32_t K, void* packed, void* stream) {
  SOPRO_CHECK_ARG(W && packed && N > 0 && K > 0 && ldw >= K, "bad pointers or sizes");
  SOPRO_CHECK_ARG((K & 31) == 0 && (ldw & 1) == 0 && (reinterpret_cast<uintptr_t>(W) & 7u) == 0, "K % 32 == 0, 8-byte aligned rows");
  SOPRO_CHECK_ARG((reinterpret_cast<uintptr_t>(packed) & 127u) == 0, "packed must be 128-byte aligned");
  const int64_t total = (int64_t)N * (K >> 5) * 8;
  hipLaunchKernelGGL(pack_w_rows_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, W, ldw, N, K,
                     reinterpret_cast<uint4*>(packed), total);
  SOPRO_LAUNCH_CHECK();
}

extern "C" int sopro_gemm_8p_takes(const sopro_gemm_args* a, const sopro_gemm_split_ext* x) {
  if (!a || !x) return 0;
  if (x->a_format != 1 || a->prologue != SOPRO_PRO_NONE || a->epilogue != SOPRO_EPI_NONE || x->rms_norm || x->ln_stats || x->ln_stats_out) return 0;
  if (x->ksplit > 1) return 0;
  if (!(x->c_mode == 0 || x->c_mode == 1 || x->c_mode == 2 || x->c_mode == 4)) return 0;
  if ((a->N % P_BN) != 0 || (a->K & 31) != 0 || a->K < 1024) return 0;
  // enough tiles to occupy the chip (a streaming chunk or a single short utterance stays on the tile kernel and its split-K)
  const int64_t tiles = (int64_t)((a->M + P_BM - 1) / P_BM) * (a->N / P_BN);
  return tiles >= 128 ? 1 : 0;
}

extern "C" int sopro_gemm_bf16x3_8p(const sopro_gemm_args* a, const void* w_rows, const sopro_gemm_split_ext* x, void* stream) {
  SOPRO_CHECK_ARG(a != nullptr && x != nullptr && w_rows != nullptr, "args / ext / w_rows is NULL");
  sopro_gemm_args g = *a;
  sopro_gemm_split_ext ext = *x;
  if (g.a_seg_stride == 0) g.a_seg_stride = (int64_t)g.rows_per_seg * g.lda;
  if (g.c_seg_stride == 0) g.c_seg_stride = (int64_t)g.rows_per_seg * g.ldc;
  if (ext.c2_seg_stride == 0) ext.c2_seg_stride = (int64_t)g.rows_per_seg * ext.ldc2;
  SOPRO_CHECK_ARG(g.M > 0 && g.N > 0 && g.K > 0 && g.rows_per_seg > 0, "M, N, K, rows_per_seg must be positive");
  SOPRO_CHECK_ARG((g.N % P_BN) == 0 && (g.K & 31) == 0 && g.K >= 64, "the long-K form takes N % 256 == 0, K % 32 == 0, K >= 64");
  SOPRO_CHECK_ARG(ext.a_format == 1 && g.prologue == SOPRO_PRO_NONE, "A must be split-form rows (a_format 1, no prologue)");
  SOPRO_CHECK_ARG(g.epilogue == SOPRO_EPI_NONE && !ext.rms_norm && !ext.ln_stats && !ext.ln_stats_out && ext.ksplit <= 1,
                  "epilogue NONE, no fused norm, no split-K");
  SOPRO_CHECK_ARG(g.A && g.C && (reinterpret_cast<uintptr_t>(g.A) & 127u) == 0 && (g.lda & 31) == 0 && (g.a_seg_stride & 31) == 0,
                  "split-form A: 128-byte aligned base, lda / a_seg_stride multiples of 32");
  SOPRO_CHECK_ARG((reinterpret_cast<uintptr_t>(w_rows) & 127u) == 0, "w_rows must be 128-byte aligned (sopro_pack_w_rows_bf16)");
  SOPRO_CHECK_ARG(ext.c_mode == 0 || ext.c_mode == 1 || ext.c_mode == 2 || ext.c_mode == 4, "c_mode must be 0, 1, 2 or 4");
  if (ext.c_mode != 0) {
    const bool second = ext.c_mode == 2 || ext.c_mode == 4;
    float* d = second ? ext.C2 : g.C;
    const int64_t ldd = second ? ext.ldc2 : g.ldc, dseg = second ? ext.c2_seg_stride : g.c_seg_stride;
    if (ext.c_mode <= 2)
      SOPRO_CHECK_ARG(d && (reinterpret_cast<uintptr_t>(d) & 127u) == 0 && (ldd & 31) == 0 && (dseg & 31) == 0 && ldd >= g.N,
                      "split-form output rows must start on 128-byte boundaries (ld, seg stride multiples of 32)");
    else
      SOPRO_CHECK_ARG(d && aligned16(d) && (ldd & 3) == 0 && (dseg & 3) == 0 && ldd >= g.N,
                      "activated output rows must be 16-byte aligned (ld, seg stride multiples of 4)");
  }
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  switch (ext.c_mode) {
    case 0: return launch_8p<SOPRO_EPI_NONE, 0>(g, w_rows, ext, s);
    case 1: return launch_8p<SOPRO_EPI_NONE, 1>(g, w_rows, ext, s);
    case 2: return launch_8p<SOPRO_EPI_NONE, 2>(g, w_rows, ext, s);
    default: return launch_8p<SOPRO_EPI_NONE, 4>(g, w_rows, ext, s);
  }
}
