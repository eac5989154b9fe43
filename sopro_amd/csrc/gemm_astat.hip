// Activation-stationary form of the split-bf16 contraction for SHORT K (round 5; VERDICT r4 item 1: "contraction main loop v2").
//
// The tile kernel (gemm_bf16s.hip) re-stages its A rows for every 128 x 128 output tile - global -> registers -> split -> LDS, two
// workgroup barriers per 32-deep K-step - and at K = 512 a tile is only 16 such steps: its K loop offers the matrix cores nothing
// for ~35 % of a step (profiles/r04_gemm_kstep_stamps_after.txt: LDS wait 14-15 %, A split / store 12-13 %, barrier 6 %) and ~6 us of
// prologue + epilogue ride on a ~25 us tile.  Here a workgroup of eight waves stages a 64-row block of A ONCE - all of K, split into
// (hi, lo) bf16 planes, 2 x 64 x (2 K + 16) bytes of LDS: 133 KB at K = 512 - and then walks column tiles of 256 with NO barrier
// at all: wave w owns columns 32 w .. 32 w + 31 of the tile and all 64 rows (two 32 x 32 accumulators), reads its A fragments
// from LDS (conflict-free ds_read_b128: the row pitch is 4 dwords mod 64) and streams its W fragments from L2 in the packed
// fragment order of sopro_pack_w_bf16, four 16-deep substeps ahead, into registers.  Per substep and wave: 4 LDS reads, 2 global
// requests, 6 MFMAs (lo*hi, hi*lo, hi*hi for two row blocks) - the same products in the same order per output element as the tile
// kernel, so results are bit-identical to it.  Every W fragment is requested by exactly one wave of the CU (the tile kernel: two),
// no A byte travels per tile: at the matrix cores' rate the CU ingests ~43 B / clk of W from L2, inside what the path delivers
// (115-128 GB/s per CU, profiles/r04_vmem_bandwidth.txt) - the tile kernel would need 62.
// The accumulator layout (a lane = one column x 16 rows) is turned into row pieces IN REGISTERS (4 x 4 transposes over lane quads:
// DPP quad_perm + selects, no LDS - it is full - and no barrier), so stores and residual loads are 16-byte pieces of rows: a quad
// writes 4 rows x 16 bytes, the 8 quads of a half-wave one 128-byte line per row.
// Forms: fp32 rows in (AMODE 0 of the tile kernel; split while staged), fp32 rows out (c_mode 0), epilogues NONE / GELU / RES (+
// layer scale), K % 32 == 0, K <= 512, N % 4 == 0.  Everything else stays on the tile kernel (sopro_gemm_bf16x3 dispatches).
#include "common.h"

namespace {

constexpr int AS_BM = 64, AS_BN = 256, AS_NT = 512, AS_D = 4;
typedef __bf16 as_bf16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ as_bf16x8 as_frag(const uint4& v) { return *reinterpret_cast<const as_bf16x8*>(&v); }

// 4 x 4 transpose over the four lanes of a quad: lane j, register k <- lane k, register j.  Two butterfly stages (lane bit 0 with
// register pairs (0, 1) (2, 3); lane bit 1 with pairs (0, 2) (1, 3)), each one DPP move and selects.
__device__ __forceinline__ float as_dpp(float v, int ctrl_is_xor2) {
  const int x = __float_as_int(v);
  const int r = ctrl_is_xor2 ? __builtin_amdgcn_update_dpp(0, x, 0x4E /* quad_perm [2, 3, 0, 1] */, 0xF, 0xF, true)
                             : __builtin_amdgcn_update_dpp(0, x, 0xB1 /* quad_perm [1, 0, 3, 2] */, 0xF, 0xF, true);
  return __int_as_float(r);
}
__device__ __forceinline__ void as_transpose4(float& a0, float& a1, float& a2, float& a3, bool b0, bool b1) {
  {  // lanes differing in bit 0: even keeps (a0, a2), takes the odd lane's (a0, a2) into (a1, a3); odd the other way round
    const float s01 = as_dpp(b0 ? a0 : a1, 0), s23 = as_dpp(b0 ? a2 : a3, 0);
    if (b0) { a0 = s01; a2 = s23; } else { a1 = s01; a3 = s23; }
  }
  {  // lanes differing in bit 1: pairs (0, 2) and (1, 3)
    const float s02 = as_dpp(b1 ? a0 : a2, 1), s13 = as_dpp(b1 ? a1 : a3, 1);
    if (b1) { a0 = s02; a1 = s13; } else { a2 = s02; a3 = s13; }
  }
}

// KS = K / 16 substeps (16: K = 256, 24: 384, 32: 512).  grid.x = row blocks x column chunks (a chunk = tpw column tiles of 256).
template <int KS, int EPI>
__global__ __launch_bounds__(AS_NT, 1) void gemm_astat_kernel(const sopro_gemm_args g, const uint4* __restrict__ Wp, int tpw) {
  constexpr int K = KS * 16;
  constexpr int PITCH = K * 2 + 16;  // bytes per LDS row of one plane: 4 dwords mod 64 -> a 16-lane group hits 16 distinct 16-byte slots
  constexpr int PLANE = AS_BM * PITCH;
  constexpr int F4ROW = K / 4;                 // float4 pieces per A row
  constexpr int NLD = AS_BM * F4ROW / AS_NT;   // staging loads per thread (8, 12, 16)
  static_assert((AS_BM * F4ROW) % AS_NT == 0, "staging");
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nct = (g.N + AS_BN - 1) / AS_BN;
  const int nch = (nct + tpw - 1) / tpw;
  const int bid = xcd_contiguous((int)blockIdx.x, (int)gridDim.x);
  const int mt = bid / nch, ch = bid - mt * nch;  // chunk fastest: the workgroups an XCD runs together share their A rows in its L2
  const int m0 = mt * AS_BM;
  const int ct0 = ch * tpw, ct1 = min(nct, ct0 + tpw);
  const int rps = g.rows_per_seg;
  const int ntiles32 = (g.N + 31) >> 5;

  // ---- W fragment stream of this wave: tile t32(ct) = ct * 8 + wave (clamped), substep s, piece p at ((t32 KS + s) 2 + p) 64 + lane
  auto wbase = [&](int ct) -> const uint4* {
    const int t32 = min(ct * (AS_BN / 32) + wave, ntiles32 - 1);
    return Wp + (int64_t)t32 * KS * 2 * 64 + lane;
  };
  uint4 wq[AS_D][2];
  {
    const uint4* w0 = wbase(ct0);
#pragma unroll
    for (int d = 0; d < AS_D; ++d) {
      wq[d][0] = w0[(d * 2 + 0) * 64];
      wq[d][1] = w0[(d * 2 + 1) * 64];
    }
  }
  // ---- stage the A block: fp32 rows -> (hi, lo) planes.  Thread t takes float4 pieces t, t + 512, ... of the block (row-major)
  {
    float4 v[NLD];
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int f = tid + i * AS_NT;
      const int row = f / F4ROW, c4 = f - row * F4ROW;
      const int m = min(m0 + row, g.M - 1);
      const int seg = m / rps, r = m - seg * rps;
      v[i] = *reinterpret_cast<const float4*>(g.A + (int64_t)seg * g.a_seg_stride + (int64_t)r * g.lda + c4 * 4);
    }
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int f = tid + i * AS_NT;
      const int row = f / F4ROW, c4 = f - row * F4ROW;
      uint2 hi, lo;
      split2_bf16(v[i].x, v[i].y, hi.x, lo.x);
      split2_bf16(v[i].z, v[i].w, hi.y, lo.y);
      unsigned char* p = lds + row * PITCH + c4 * 8;
      *reinterpret_cast<uint2*>(p) = hi;
      *reinterpret_cast<uint2*>(p + PLANE) = lo;
    }
  }
  __syncthreads();  // the only barrier of the kernel

  const int frow = lane & 31, fg = lane >> 5;
  const unsigned char* abase = lds + frow * PITCH + fg * 16;  // fragment (row block i, substep s, plane p): + i * 32 PITCH + s * 32 + p * PLANE
  const bool b0 = (lane & 1) != 0, b1 = (lane & 2) != 0;
  const int colq = (lane & 31) & ~3;                           // after the transposes: this lane's 4 columns within the wave's 32
  const int rsub = lane & 3;                                   // ... and its row within a group of 4
  const bool big = (int64_t)g.M * g.N * 4 >= SOPRO_BIG_BYTES;

  for (int ct = ct0; ct < ct1; ++ct) {
    f32x16 acc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    const uint4* wcur = wbase(ct);
    const uint4* wnxt = wbase(min(ct + 1, ct1 - 1));  // (the last tile of the chunk re-requests its own head: harmless, never used)
    uint4 ah[2], al[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      ah[i] = *reinterpret_cast<const uint4*>(abase + i * 32 * PITCH);
      al[i] = *reinterpret_cast<const uint4*>(abase + i * 32 * PITCH + PLANE);
    }
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const uint4 wh = wq[s % AS_D][0], wl = wq[s % AS_D][1];
      {  // the slot's next occupant: substep s + D of this tile, or the head of the next one
        const int sn = s + AS_D;
        const uint4* src = sn < KS ? wcur + (int64_t)sn * 2 * 64 : wnxt + (int64_t)(sn - KS) * 2 * 64;
        wq[s % AS_D][0] = src[0];
        wq[s % AS_D][1] = src[64];
      }
      // (the requests stay HERE, four substeps ahead of their use: left alone, the scheduler sinks them to just before it - the first
      // build waited `vmcnt(0)` / `vmcnt(1)` at 36 of its 64 uses)
      __builtin_amdgcn_sched_barrier(0);
      const uint4 ch0 = ah[0], ch1 = ah[1], cl0 = al[0], cl1 = al[1];
      if (s + 1 < KS) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          ah[i] = *reinterpret_cast<const uint4*>(abase + i * 32 * PITCH + (s + 1) * 32);
          al[i] = *reinterpret_cast<const uint4*>(abase + i * 32 * PITCH + (s + 1) * 32 + PLANE);
        }
      }
      // smallest terms first, the two row blocks alternating (independent accumulators back to back): as the tile kernel
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_frag(cl0), as_frag(wh), acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_frag(cl1), as_frag(wh), acc[1], 0, 0, 0);
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_frag(ch0), as_frag(wl), acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_frag(ch1), as_frag(wl), acc[1], 0, 0, 0);
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_frag(ch0), as_frag(wh), acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_frag(ch1), as_frag(wh), acc[1], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    // ---- epilogue of this wave's 64 x 32 piece.  Register q of block i: row 32 i + 8 (q / 4) + 4 fg + q % 4, column lane & 31.
    // After the quad transposes the lane holds, per group of four registers, ROW 32 i + 8 grp + 4 fg + (lane & 3), columns colq .. + 3.
    const int n = ct * AS_BN + wave * 32 + colq;
    const bool col_ok = n < g.N;  // (N % 4 == 0: a lane's four columns are in range together)
    float4 bias4 = make_float4(0.f, 0.f, 0.f, 0.f), sc4 = make_float4(1.f, 1.f, 1.f, 1.f);
    if (col_ok) {
      if (g.bias) bias4 = *reinterpret_cast<const float4*>(g.bias + n);
      if (EPI == SOPRO_EPI_RES && g.scale) sc4 = *reinterpret_cast<const float4*>(g.scale + n);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      float4 v[4], rv[4];
      float* cp[4];
#pragma unroll
      for (int grp = 0; grp < 4; ++grp) {
        float a0 = acc[i][grp * 4 + 0], a1 = acc[i][grp * 4 + 1], a2 = acc[i][grp * 4 + 2], a3 = acc[i][grp * 4 + 3];
        as_transpose4(a0, a1, a2, a3, b0, b1);
        v[grp] = make_float4(a0 + bias4.x, a1 + bias4.y, a2 + bias4.z, a3 + bias4.w);
        const int m = m0 + 32 * i + 8 * grp + 4 * fg + rsub;
        cp[grp] = nullptr;
        rv[grp] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (col_ok && m < g.M) {
          const int seg = m / rps, r = m - seg * rps;
          cp[grp] = g.C + (int64_t)seg * g.c_seg_stride + (int64_t)r * g.ldc + n;
          // (R may alias C - in-place residual updates: every load of this batch is issued before its stores)
          if (EPI == SOPRO_EPI_RES) rv[grp] = *reinterpret_cast<const float4*>(g.R + (int64_t)seg * g.r_seg_stride + (int64_t)r * g.ldr + n);
        }
      }
#pragma unroll
      for (int grp = 0; grp < 4; ++grp) {
        if (!cp[grp]) continue;
        float4 o = v[grp];
        if (EPI == SOPRO_EPI_GELU) {
          o.x = gelu_fast(o.x); o.y = gelu_fast(o.y); o.z = gelu_fast(o.z); o.w = gelu_fast(o.w);  // (the three-pass tile kernel's form since round 6)
        } else if (EPI == SOPRO_EPI_RES) {
          o.x = rv[grp].x + sc4.x * o.x; o.y = rv[grp].y + sc4.y * o.y; o.z = rv[grp].z + sc4.z * o.z; o.w = rv[grp].w + sc4.w * o.w;
        }
        bulk_store4(cp[grp], o, big);
      }
    }
  }
}

template <int KS, int EPI>
int as_launch(const sopro_gemm_args& g, const uint4* wp, hipStream_t s) {
  constexpr int K = KS * 16;
  constexpr size_t lds = (size_t)2 * AS_BM * (K * 2 + 16);
  auto kern = gemm_astat_kernel<KS, EPI>;
  SOPRO_SET_MAX_LDS_ONCE(kern, lds);
  const int ntm = (g.M + AS_BM - 1) / AS_BM, nct = (g.N + AS_BN - 1) / AS_BN;
  // column tiles per workgroup: the A block is staged once per workgroup (~15 % of one tile's matrix-core time), so more tiles per
  // workgroup amortise it - but the launch should keep >= ~6 rounds of workgroups for an even last round
  int tpw = 1;
  for (int t = 2; t <= 4 && t <= nct; ++t)
    if ((int64_t)ntm * ((nct + t - 1) / t) >= 1100) tpw = t;
  hipLaunchKernelGGL(kern, dim3((unsigned)(ntm * ((nct + tpw - 1) / tpw))), dim3(AS_NT), lds, s, g, wp, tpw);
  SOPRO_LAUNCH_CHECK();
}

template <int KS>
int as_launch_epi(const sopro_gemm_args& g, const uint4* wp, hipStream_t s) {
  switch (g.epilogue) {
    case SOPRO_EPI_NONE: return as_launch<KS, SOPRO_EPI_NONE>(g, wp, s);
    case SOPRO_EPI_GELU: return as_launch<KS, SOPRO_EPI_GELU>(g, wp, s);
    case SOPRO_EPI_RES: return as_launch<KS, SOPRO_EPI_RES>(g, wp, s);
    default: break;
  }
  sopro_set_error("gemm_astat: epilogue %d has no activation-stationary form", g.epilogue);
  return -2;
}

}  // namespace

// Is (g, ext) a problem the activation-stationary form takes?  (The caller - sopro_gemm_bf16x3 - has run its own checks already.)
bool sopro_gemm_astat_takes(const sopro_gemm_args& g, const sopro_gemm_split_ext& ext) {
  if (!(g.K == 256 || g.K == 384 || g.K == 512)) return false;
  if (ext.a_format != 0 || ext.c_mode != 0 || ext.rms_norm || ext.ksplit > 1 || g.prologue != SOPRO_PRO_NONE || g.dbg) return false;
  if (!(g.epilogue == SOPRO_EPI_NONE || g.epilogue == SOPRO_EPI_GELU || g.epilogue == SOPRO_EPI_RES)) return false;
  if ((g.N & 3) || (g.ldc & 3) || (g.c_seg_stride & 3) || !aligned16(g.C)) return false;
  if (g.bias && !aligned16(g.bias)) return false;
  if (g.epilogue == SOPRO_EPI_RES && ((g.ldr & 3) || (g.r_seg_stride & 3) || !aligned16(g.R) || (g.scale && !aligned16(g.scale)))) return false;
  return true;
}

int sopro_gemm_astat_bf16x3(const sopro_gemm_args& g, const void* packed_w, hipStream_t s) {
  const uint4* wp = reinterpret_cast<const uint4*>(packed_w);
  switch (g.K) {
    case 256: return as_launch_epi<16>(g, wp, s);
    case 384: return as_launch_epi<24>(g, wp, s);
    case 512: return as_launch_epi<32>(g, wp, s);
    default: break;
  }
  sopro_set_error("gemm_astat: K = %d (256, 384 or 512)", g.K);
  return -2;
}
