// Dense fp32 contraction on the gfx950 matrix cores (v_mfma_f32_32x32x2_f32: fp32 in, fp32
// accumulate, bit-equal to an fmaf chain) with segmented / overlapping-row addressing of A so
// that causal Conv1d and ConvTranspose1d over channels-last activations are the same kernel.
//
// Tile: BM x BN x 32 per workgroup, LDS double buffered, rows padded to 36 floats so that the
// ds_read_b128 fragment reads (lane -> row, 16 B) hit 16 distinct 16-byte slots per lane group.
// Fragment convention (K is summed, so its order inside a 8-wide chunk is free as long as A and
// B agree): lanes 0-31 carry k = 8c+s, lanes 32-63 carry k = 8c+4+s at MFMA step s.
#include "common.h"
#include "gemm_epilogue.h"

namespace {

constexpr int BK = 32;
constexpr int LDT = BK + 4;

template <int WM, int WN, int TM, int TN, int EPI>
__global__ __launch_bounds__(WM* WN * 64) void gemm_f32_kernel(const sopro_gemm_args g) {
  constexpr int NT = WM * WN * 64;
  constexpr int BM = WM * TM * 32;
  constexpr int BN = WN * TN * 32;
  constexpr int A_F4 = BM * 8 / NT;
  constexpr int W_F4 = BN * 8 / NT;
  constexpr int RSTEP = NT / 8;
  extern __shared__ float4 smem4[];
  float* As = reinterpret_cast<float*>(smem4);  // [2][BM][LDT]
  float* Ws = As + 2 * BM * LDT;                // [2][BN][LDT]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  long long* dbg = g.dbg ? g.dbg + (int64_t)blockIdx.x * 8 : nullptr;
  if (dbg && tid == 0) dbg[0] = clock64();
  const int ntn = (g.N + BN - 1) / BN;
  const int mt = blockIdx.x / ntn, nt = blockIdx.x % ntn;
  const int m0 = mt * BM, n0 = nt * BN;
  const int lrow = tid >> 3, lc4 = tid & 7;
  const int rps = g.rows_per_seg;

  const float* ap[A_F4];
  const float* wp[W_F4];
#pragma unroll
  for (int i = 0; i < A_F4; ++i) {
    const int m = m0 + lrow + i * RSTEP;
    if (m < g.M) {
      const int seg = m / rps;
      const int r = m - seg * rps;
      ap[i] = g.A + (int64_t)seg * g.a_seg_stride + (int64_t)r * g.lda + lc4 * 4;
    } else {
      ap[i] = nullptr;
    }
  }
#pragma unroll
  for (int i = 0; i < W_F4; ++i) {
    const int n = n0 + lrow + i * RSTEP;
    wp[i] = (n < g.N) ? g.W + (int64_t)n * g.ldw + lc4 * 4 : nullptr;
  }

  // epilogue operands that do not depend on the contraction are requested before the main loop
  float biasv[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int n = n0 + (wn * TN + j) * 32 + (lane & 31);
    biasv[j] = (g.bias && n < g.N) ? g.bias[n] : 0.f;
  }

  float4 ra[A_F4], rw[W_F4];
  const int KT = (g.K + BK - 1) / BK;
  const int pro = g.prologue;

  auto gload = [&](int kt) {
    const int k = kt * BK + lc4 * 4;
    const bool kin = k < g.K;
#pragma unroll
    for (int i = 0; i < A_F4; ++i) {
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (kin && ap[i]) v = *reinterpret_cast<const float4*>(ap[i] + kt * BK);
      ra[i] = v;
    }
#pragma unroll
    for (int i = 0; i < W_F4; ++i) {
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (kin && wp[i]) v = *reinterpret_cast<const float4*>(wp[i] + kt * BK);
      rw[i] = v;
    }
    if (pro == SOPRO_PRO_ELU) {
#pragma unroll
      for (int i = 0; i < A_F4; ++i) {
        ra[i].x = eluf_(ra[i].x); ra[i].y = eluf_(ra[i].y); ra[i].z = eluf_(ra[i].z); ra[i].w = eluf_(ra[i].w);
      }
    } else if (pro == SOPRO_PRO_ADDVEC) {
      if (kin) {
        const float4 pv = *reinterpret_cast<const float4*>(g.pro_vec + k);
#pragma unroll
        for (int i = 0; i < A_F4; ++i) {
          if (ap[i]) { ra[i].x += pv.x; ra[i].y += pv.y; ra[i].z += pv.z; ra[i].w += pv.w; }
        }
      }
    }
  };
  auto lstore = [&](int buf) {
    float* a = As + buf * BM * LDT;
    float* w = Ws + buf * BN * LDT;
#pragma unroll
    for (int i = 0; i < A_F4; ++i)
      *reinterpret_cast<float4*>(a + (lrow + i * RSTEP) * LDT + lc4 * 4) = ra[i];
#pragma unroll
    for (int i = 0; i < W_F4; ++i)
      *reinterpret_cast<float4*>(w + (lrow + i * RSTEP) * LDT + lc4 * 4) = rw[i];
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  gload(0);
  lstore(0);
  __syncthreads();
  if (dbg && tid == 0) dbg[1] = clock64();

  const int frow = lane & 31, fk = (lane >> 5) * 4;
  for (int kt = 0; kt < KT; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < KT) gload(kt + 1);
    const float* a = As + buf * BM * LDT + (wm * TM * 32 + frow) * LDT + fk;
    const float* w = Ws + buf * BN * LDT + (wn * TN * 32 + frow) * LDT + fk;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      float4 a4[TM], b4[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) a4[i] = *reinterpret_cast<const float4*>(a + i * 32 * LDT + c * 8);
#pragma unroll
      for (int j = 0; j < TN; ++j) b4[j] = *reinterpret_cast<const float4*>(w + j * 32 * LDT + c * 8);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[i].x, b4[j].x, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[i].y, b4[j].y, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[i].z, b4[j].z, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[i].w, b4[j].w, acc[i][j], 0, 0, 0);
        }
    }
    if (kt + 1 < KT) lstore(buf ^ 1);
    __syncthreads();
  }

  if (dbg && tid == 0) dbg[2] = clock64();
  // ---- epilogue through LDS (gemm_epilogue.h): the tile buffers are free now
  gemm_store_tile<WM, WN, TM, TN, EPI>(g, reinterpret_cast<float*>(smem4), acc, biasv, m0, n0);
  if (dbg && tid == 0) dbg[3] = clock64();
}

template <int WM, int WN, int TM, int TN, int EPI>
int launch_epi(const sopro_gemm_args& g, hipStream_t s) {
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  constexpr size_t lds = (size_t)2 * (BM + BN) * LDT * sizeof(float);
  static_assert((size_t)BM * (BN + 4) * sizeof(float) <= lds, "the epilogue tile must fit in the main-loop buffers");
  auto kern = gemm_f32_kernel<WM, WN, TM, TN, EPI>;
  SOPRO_SET_MAX_LDS_ONCE(kern, lds);
  const int ntm = (g.M + BM - 1) / BM, ntn = (g.N + BN - 1) / BN;
  hipLaunchKernelGGL(kern, dim3(ntm * ntn), dim3(WM * WN * 64), lds, s, g);
  SOPRO_LAUNCH_CHECK();
}

template <int WM, int WN, int TM, int TN>
int launch_cfg(const sopro_gemm_args& g, hipStream_t s) {
  switch (g.epilogue) {
    case SOPRO_EPI_NONE: return launch_epi<WM, WN, TM, TN, SOPRO_EPI_NONE>(g, s);
    case SOPRO_EPI_GELU: return launch_epi<WM, WN, TM, TN, SOPRO_EPI_GELU>(g, s);
    case SOPRO_EPI_RES: return launch_epi<WM, WN, TM, TN, SOPRO_EPI_RES>(g, s);
    case SOPRO_EPI_TANH: return launch_epi<WM, WN, TM, TN, SOPRO_EPI_TANH>(g, s);
    case SOPRO_EPI_GLU:
      if constexpr (WN * TN * 32 >= 64) return launch_epi<WM, WN, TM, TN, SOPRO_EPI_GLU>(g, s);
      break;
    default: break;
  }
  sopro_set_error("sopro_gemm_f32: epilogue %d is not available for this tile shape", g.epilogue);
  return -2;
}

}  // namespace

static int g_tile_override = 0;  // 0 = heuristic; 1: 128x128, 2: 64x128, 3: 256x64, 4: 256x32, 5: 64x64 (developer probe)
extern "C" int sopro_gemm_set_tile_override(int cfg) {
  g_tile_override = cfg;
  return 0;
}

extern "C" int sopro_gemm_f32(const sopro_gemm_args* a, void* stream) {
  SOPRO_CHECK_ARG(a != nullptr, "args is NULL");
  sopro_gemm_args g = *a;
  // a zero segment stride means "dense": segments follow each other without padding rows
  if (g.a_seg_stride == 0) g.a_seg_stride = (int64_t)g.rows_per_seg * g.lda;
  if (g.c_seg_stride == 0) g.c_seg_stride = (int64_t)g.rows_per_seg * g.ldc;
  if (g.r_seg_stride == 0) g.r_seg_stride = (int64_t)g.rows_per_seg * g.ldr;
  SOPRO_CHECK_ARG(g.M > 0 && g.N > 0 && g.K > 0, "M, N, K must be positive");
  SOPRO_CHECK_ARG((g.K & 3) == 0, "K must be a multiple of 4");
  SOPRO_CHECK_ARG(g.rows_per_seg > 0, "rows_per_seg must be positive");
  SOPRO_CHECK_ARG(g.A && g.W && g.C, "A, W, C must be non-NULL");
  SOPRO_CHECK_ARG(aligned16(g.A) && aligned16(g.W), "A and W must be 16-byte aligned");
  SOPRO_CHECK_ARG((g.lda & 3) == 0 && (g.ldw & 3) == 0 && (g.a_seg_stride & 3) == 0, "lda, ldw, a_seg_stride must be multiples of 4");
  SOPRO_CHECK_ARG(g.prologue >= SOPRO_PRO_NONE && g.prologue <= SOPRO_PRO_ADDVEC, "unknown prologue");
  SOPRO_CHECK_ARG(g.epilogue >= SOPRO_EPI_NONE && g.epilogue <= SOPRO_EPI_TANH, "unknown epilogue");
  SOPRO_CHECK_ARG(g.epilogue != SOPRO_EPI_RES || g.R != nullptr, "EPI_RES needs R");
  SOPRO_CHECK_ARG(g.prologue != SOPRO_PRO_ADDVEC || (g.pro_vec && aligned16(g.pro_vec)), "PRO_ADDVEC needs an aligned pro_vec");
  SOPRO_CHECK_ARG(g.epilogue != SOPRO_EPI_GLU || (g.N % 64) == 0, "EPI_GLU needs N % 64 == 0 (packed value/gate blocks)");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  switch (g_tile_override) {
    case 1: return launch_cfg<2, 2, 2, 2>(g, s);
    case 2: return launch_cfg<2, 2, 1, 2>(g, s);
    case 3: return launch_cfg<4, 1, 2, 2>(g, s);
    case 4: if (g.epilogue != SOPRO_EPI_GLU) return launch_cfg<4, 1, 2, 1>(g, s); break;
    case 5: if (g.epilogue != SOPRO_EPI_GLU) return launch_cfg<2, 2, 1, 1>(g, s); break;
    default: break;
  }
  // Tile shape (measured with tools/gemm_probe.py on the NAR / Mimi shapes): 128x128 only pays on very large grids or
  // long contractions; everywhere else 64x64 tiles (4x the workgroups, same LDS traffic per flop) hide the
  // prologue / epilogue latency better.  The GLU epilogue needs a value/gate tile pair per wave (64x128).
  if (g.N <= 32 && g.epilogue != SOPRO_EPI_GLU) return launch_cfg<4, 1, 2, 1>(g, s);
  const int64_t tiles128 = (int64_t)((g.M + 127) / 128) * ((g.N + 127) / 128);
  const bool big = g.N > 64 && g.K > 128 && (tiles128 >= 3000 || (g.K >= 1024 && tiles128 >= 512));
  if (big) return launch_cfg<2, 2, 2, 2>(g, s);
  if (g.epilogue == SOPRO_EPI_GLU) return launch_cfg<2, 2, 1, 2>(g, s);
  return launch_cfg<2, 2, 1, 1>(g, s);
}

