// Shared epilogue of the tiled contractions (gemm_f32.hip, gemm_bf16s.hip): bias, activation / residual, and the
// transposition of the MFMA accumulator layout through LDS so that global traffic is coalesced dwordx4.
#pragma once
#include "common.h"

// OUT: 3 = ELU(C) as fp32 rows to C; 4 = fp32 rows to C AND ELU(C) as fp32 rows to ext.C2 (the consumer then needs no
// activation prologue; host guarantees N % 4 == 0 and 16-byte aligned rows).
// OUT: 0 = fp32 rows to C; 1 = ELU, then split-bf16 form to C; 2 = fp32 rows to C AND ELU + split form to ext.C2 (the raw
// tensor stays available as a residual operand while the next contraction reads its activated form without any prologue
// work).  Split form: every aligned group of 32 channels (128 bytes as fp32) becomes [32 hi bf16 | 32 lo bf16], so an
// element keeps its 128-byte line and every fp32 stride / offset keeps its meaning.
// OUT: 5 = no C at all: per tile row and 64-column group the maximum of the group's columns and its column index go to ext.C2 as
// (float value, int32 index) pairs, [M][ext.ldc2 = 64-column groups] - the arg-max of a wide projection (the 31 NAR heads: 2048 logits each,
// src/sopro/model.py:338-345) without ever writing the logits; sopro_argmax_partials_i32 finishes the reduction.
// OUT: 6 / 7 / 8 (round 4, the bf16 mode's activation flow; EPI NONE or RES): bf16 rows - 6 = C as bf16 to g.C; 7 = ELU(C) as bf16
// to g.C; 8 = C as bf16 to g.C AND ELU(C) as bf16 to ext.C2.  g.C / ext.C2 / g.R (the skip operand of EPI_RES) point at bf16
// elements and their strides count elements; host guarantees N % 4 == 0 and 8-byte aligned rows.
// rs (optional, LDS): one scale per tile row applied to the accumulator before the bias: the RMSNorm of the operand row
// when its weight vector has been folded into W (out = rs * (x W'^T) + b).  cscale: a constant factor on the accumulators when
// there is no rs (the f16 forms' operand scales; with rs the caller has multiplied it into the row scales).
// (internal epilogue id, not part of the ABI: GELU with gelu_fast - what the three-pass / one-pass launchers instantiate for EPI_GELU)
constexpr int SOPRO_EPI_GELU_FAST = 100;
constexpr int SOPRO_EPI_GLU_FAST = 101;  // GLU with sigmoid_fast (common.h): what the f16 three-pass / one-pass launchers instantiate for EPI_GLU
template <int WM, int WN, int TM, int TN, int EPI, int OUT = 0>
__device__ __forceinline__ void gemm_store_tile(const sopro_gemm_args& g, float* __restrict__ Cs, f32x16 (&acc)[TM][TN],
                                                const float (&biasv)[TN], int m0, int n0,
                                                const sopro_gemm_split_ext* __restrict__ ext = nullptr,
                                                const float* __restrict__ rs = nullptr, float cscale = 1.0f) {
  constexpr int NT = WM * WN * 64;
  constexpr int BM = WM * TM * 32;
  constexpr int BN = WN * TN * 32;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int rps = g.rows_per_seg;
  const bool big = (int64_t)g.M * g.N * (OUT >= 6 ? 2 : 4) >= SOPRO_BIG_BYTES;  // a large output: stored with the non-temporal hint (common.h)
  // ---- epilogue through LDS.  The MFMA accumulator layout gives a lane ONE column of 16 rows, i.e. 4-byte
  // stores/loads (measured: 26k of a workgroup's 119k cycles for a 128x128x256 tile).  The tile buffers are free
  // now, so the accumulators are transposed through LDS and every thread streams whole 16-byte pieces of rows:
  // residual loads and output stores are fully coalesced dwordx4 accesses.  Residual operands are loaded in
  // batches BEFORE the stores of the same batch (R may alias C: in-place residual updates).
  constexpr int CLD = BN + 4;
  // EPI_RES on fp32 rows (round 6): the residual pieces this thread adds - one 16-byte piece per pass, BM / RPP passes - are ALL requested
  // here, in front of the transposition and its barrier.  Requested batch by batch inside the store loop they cost one memory latency
  // per batch with nothing to hide it (shader-clock stamps, profiles/r06_tile_life.txt: 23.5 k cycles of epilogue for the codec's o
  // projection against 6.7 k for the same tile without the residual - 42 % of the tile's life).  The accumulators are about to leave
  // their registers, so the 64 registers are there.  (R may alias C - in-place residual updates: every piece is read before this
  // workgroup's first store, and no other workgroup writes this tile.)
  constexpr bool RES_AHEAD = EPI == SOPRO_EPI_RES && OUT < 5;
  constexpr int RA_TPR = BN / 4, RA_RPP = NT / RA_TPR, RA_NPASS = BM / RA_RPP;
  float4 rahead[RES_AHEAD ? RA_NPASS : 1];
  // A tile that lies wholly inside M x N and inside one segment (all but the last row / column of tiles, and the few tiles that
  // straddle two utterances of a row-window problem): every row pointer is (first row) + q x (rows per pass x leading dimension) -
  // no per-piece bounds tests, segment wraps or null checks, i.e. no divergent branches (round 6: the general walk's compare /
  // exec-mask / branch / 64-bit multiply per piece was a third of a plain epilogue).  Workgroup-uniform.
  const bool whole_tile = m0 + BM <= g.M && n0 + BN <= g.N && (m0 / rps) == ((m0 + BM - 1) / rps);
  if constexpr (RES_AHEAD) {
    const int prow = tid / RA_TPR, ocol = n0 + (tid % RA_TPR) * 4;
    const bool rvec = ((g.N & 3) == 0) && ((g.ldr & 3) == 0) && ((g.r_seg_stride & 3) == 0) && ((reinterpret_cast<uintptr_t>(g.R) & 15u) == 0);
    int m = m0 + prow;
    int seg = m / rps, rr = m - seg * rps;
    if (whole_tile && rvec) {
      const float* rptr = g.R + (int64_t)seg * g.r_seg_stride + (int64_t)rr * g.ldr + ocol;
      const int64_t rstep = (int64_t)RA_RPP * g.ldr;
#pragma unroll
      for (int q = 0; q < RA_NPASS; ++q) rahead[q] = *reinterpret_cast<const float4*>(rptr + q * rstep);
    } else {
#pragma unroll
    for (int q = 0; q < RA_NPASS; ++q) {
      rahead[q] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (ocol < g.N && m < g.M) {
        const float* rptr = g.R + (int64_t)seg * g.r_seg_stride + (int64_t)rr * g.ldr + ocol;
        if (rvec) {
          rahead[q] = *reinterpret_cast<const float4*>(rptr);
        } else {
          rahead[q].x = rptr[0];
          if (ocol + 1 < g.N) rahead[q].y = rptr[1];
          if (ocol + 2 < g.N) rahead[q].z = rptr[2];
          if (ocol + 3 < g.N) rahead[q].w = rptr[3];
        }
      }
      m += RA_RPP; rr += RA_RPP;
      while (rr >= rps) { rr -= rps; ++seg; }
    }
    }
  }
  {
    const int col = lane & 31;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int k = 0; k < 4; ++k) {  // accumulator registers 4 k .. 4 k + 3 are four consecutive tile rows: their row scales are ONE 16-byte LDS read
        const int row0 = (wm * TM + i) * 32 + 8 * k + 4 * (lane >> 5);
        float rs4[4] = {cscale, cscale, cscale, cscale};
        if (rs) {
          const float4 t = *reinterpret_cast<const float4*>(rs + row0);
          rs4[0] = t.x; rs4[1] = t.y; rs4[2] = t.z; rs4[3] = t.w;
        }
        // (the row-scaled product is rounded BEFORE the bias is added, as it always was - it used to sit inside a select, which kept it
        // out of a fused multiply-add; with cscale alone the product is exact and either form gives the same sum)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float p = acc[i][j][4 * k + e] * rs4[e];
            if (rs) asm("" : "+v"(p));  // (an opaque copy: the product cannot be contracted into the addition behind it)
            Cs[(row0 + e) * CLD + (wn * TN + j) * 32 + col] = p + biasv[j];
          }
      }
  }
  __syncthreads();
#ifdef SOPRO_DEV_SWITCHES
  if (g.dbg && tid == 0) g.dbg[(int64_t)blockIdx.x * 8 + 4] = clock64();  // developer probe: behind the transposition
#endif
  if constexpr (OUT == 5) {
    constexpr int TPR5 = NT / BM;   // threads per tile row
    constexpr int CPT = BN / TPR5;  // consecutive columns per thread
    const int row = tid / TPR5, part = tid % TPR5;
    float best = -INFINITY;
    int bi = 0x7fffffff;
#pragma unroll 8
    for (int c = 0; c < CPT; ++c) {  // ascending column: a strict > keeps the first maximum (torch.argmax on exact ties)
      const int n = n0 + part * CPT + c;
      const float v = Cs[row * CLD + part * CPT + c];
      if (n < g.N && v > best) { best = v; bi = n; }
    }
    // one (max, column) pair per 64 columns whatever the tile width (round 6: 128-column tiles for many-row problems - every staged A
    // row serves twice the columns): the GRP threads that share a 64-column group reduce, its first thread writes
    constexpr int GRP = CPT >= 64 ? 1 : 64 / CPT;
    static_assert(BN % 64 == 0 && (CPT >= 64 ? CPT == 64 : 64 % CPT == 0), "arg-max partials are per 64 columns");
#pragma unroll
    for (int o = 1; o < GRP; o <<= 1) {
      const float ov = __shfl_xor(best, o, 64);
      const int oi = __shfl_xor(bi, o, 64);
      if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
    }
    if (part % GRP == 0 && m0 + row < g.M) {
      float* pp = ext->C2 + ((int64_t)(m0 + row) * ext->ldc2 + n0 / 64 + part / GRP) * 2;
      pp[0] = best;
      pp[1] = __int_as_float(bi);
    }
    return;
  }
  if constexpr (OUT >= 6) {
    constexpr bool hres = EPI == SOPRO_EPI_RES;
    static_assert(EPI == SOPRO_EPI_NONE || EPI == SOPRO_EPI_RES, "bf16 rows take EPI_NONE or EPI_RES");
    constexpr int HTPR = BN / 4, HRPP = NT / HTPR, HNPASS = BM / HRPP;
    const int prow = tid / HTPR, pc4 = tid % HTPR;
    const int ocol = n0 + pc4 * 4;
    const bool col_ok = ocol < g.N;
    float4 sc4 = make_float4(1.f, 1.f, 1.f, 1.f);
    if (hres && g.scale && col_ok) sc4 = make_float4(g.scale[ocol], g.scale[ocol + 1], g.scale[ocol + 2], g.scale[ocol + 3]);
    int m = m0 + prow;
    int seg = m / rps, rr = m - seg * rps;
    char* const rawb = reinterpret_cast<char*>(g.C);
    char* const actb = OUT == 8 ? reinterpret_cast<char*>(ext->C2) : rawb;
    const int64_t ldact = OUT == 8 ? ext->ldc2 : g.ldc, segact = OUT == 8 ? ext->c2_seg_stride : g.c_seg_stride;
    const char* const rb = reinterpret_cast<const char*>(g.R);
    auto at = [&](int64_t sg, int64_t r, int64_t ld, int64_t sstride) { return (sg * sstride + r * ld + ocol) * 2; };
    const float* csrc = Cs + prow * CLD + pc4 * 4;
    constexpr int HB = HNPASS < 8 ? HNPASS : 8;
#pragma unroll 1
    for (int p0 = 0; p0 < HNPASS; p0 += HB) {
      int64_t oraw[HB], oact[HB];
      bool okq[HB];
      uint2 rv[HB];
#pragma unroll
      for (int q = 0; q < HB; ++q) {
        okq[q] = col_ok && m < g.M;
        oraw[q] = at(seg, rr, g.ldc, g.c_seg_stride);
        oact[q] = at(seg, rr, ldact, segact);
        rv[q] = make_uint2(0u, 0u);
        if (hres && okq[q]) rv[q] = *reinterpret_cast<const uint2*>(rb + at(seg, rr, g.ldr, g.r_seg_stride));
        m += HRPP; rr += HRPP;
        while (rr >= rps) { rr -= rps; ++seg; }
      }
#pragma unroll
      for (int q = 0; q < HB; ++q) {
        if (!okq[q]) continue;
        float4 v = *reinterpret_cast<const float4*>(csrc + (p0 + q) * HRPP * CLD);
        if (hres) {
          v.x = __uint_as_float(rv[q].x << 16) + sc4.x * v.x; v.y = __uint_as_float(rv[q].x & 0xffff0000u) + sc4.y * v.y;
          v.z = __uint_as_float(rv[q].y << 16) + sc4.z * v.z; v.w = __uint_as_float(rv[q].y & 0xffff0000u) + sc4.w * v.w;
        }
        unsigned lo_;
        if (OUT != 7) {
          uint2 h;
          split2_bf16(v.x, v.y, h.x, lo_);
          split2_bf16(v.z, v.w, h.y, lo_);
          bulk_store_u2(rawb + oraw[q], h, big);
        }
        if (OUT != 6) {
          uint2 h;
          split2_bf16(eluf_(v.x), eluf_(v.y), h.x, lo_);
          split2_bf16(eluf_(v.z), eluf_(v.w), h.y, lo_);
          bulk_store_u2(actb + oact[q], h, big);
        }
      }
    }
    return;
  }
  constexpr bool glu = EPI == SOPRO_EPI_GLU || EPI == SOPRO_EPI_GLU_FAST;
  constexpr bool res = EPI == SOPRO_EPI_RES;
  if constexpr (glu && OUT == 0) {
    // GLU (round 6): EVERY thread of a tile row works - thread t takes the two value columns 64 (t / 16) + 2 (t % 16) .. + 1 of its
    // 64-column group [32 value | 32 gate] and their gates 32 columns further, and writes 8 bytes.  The general path below gives a
    // thread four consecutive tile columns: the threads on gate columns idle, and the waves issue the sigmoids' ~27 instructions per
    // element at half their lanes (stamps: 10.8 k of the GLU launch's 49 k cycles per tile).  Same arithmetic per element.
    constexpr int GT = BN / 4, GRPP = NT / GT, GNP = BM / GRPP;
    const int prow = tid / GT, t = tid % GT;
    const int grp = t >> 4, u = t & 15;
    const int ocol = ((n0 + grp * 64) >> 6) * 32 + 2 * u;
    const bool col_ok = n0 + grp * 64 < g.N;  // (N % 64 == 0: a group lies wholly inside or outside)
    const bool st2 = ((g.ldc & 1) == 0) && ((g.c_seg_stride & 1) == 0) && ((reinterpret_cast<uintptr_t>(g.C) & 7u) == 0);
    const float* csrc = Cs + prow * CLD + grp * 64 + 2 * u;
    auto piece = [&](int q, float* cp) {
      const float2 v = *reinterpret_cast<const float2*>(csrc + q * GRPP * CLD), gt = *reinterpret_cast<const float2*>(csrc + q * GRPP * CLD + 32);
      const float ox = v.x * (EPI == SOPRO_EPI_GLU_FAST ? sigmoid_fast(gt.x) : sigmoidf_(gt.x)), oy = v.y * (EPI == SOPRO_EPI_GLU_FAST ? sigmoid_fast(gt.y) : sigmoidf_(gt.y));
      if (st2) {
        bulk_store_u2(cp, make_uint2(__float_as_uint(ox), __float_as_uint(oy)), big);
      } else {
        cp[0] = ox;
        cp[1] = oy;
      }
    };
    if (!col_ok) return;
    if (m0 + BM <= g.M && (m0 / rps) == ((m0 + BM - 1) / rps)) {  // whole rows inside one segment: straight-line
      const int m = m0 + prow, seg = m / rps, rr = m - seg * rps;
      float* cptr = g.C + (int64_t)seg * g.c_seg_stride + (int64_t)rr * g.ldc + ocol;
      const int64_t cstep = (int64_t)GRPP * g.ldc;
#pragma unroll
      for (int q = 0; q < GNP; ++q) {
        piece(q, cptr + q * cstep);
#ifdef SOPRO_DEV_SWITCHES
        if (g.dbg && tid == 0 && q == GNP / 2 - 1) g.dbg[(int64_t)blockIdx.x * 8 + 5] = clock64();
#endif
      }
    } else {
#pragma unroll 1
      for (int q = 0; q < GNP; ++q) {
        const int m = m0 + prow + q * GRPP;
        if (m >= g.M) break;
        const int seg = m / rps, rr = m - seg * rps;
        piece(q, g.C + (int64_t)seg * g.c_seg_stride + (int64_t)rr * g.ldc + ocol);
      }
    }
    return;
  }
  constexpr int TPR = BN / 4;          // threads per tile row (one float4 each)
  constexpr int RPP = NT / TPR;        // rows per pass
  constexpr int NPASS = BM / RPP;
  const int prow = tid / TPR, pc4 = tid % TPR;
  const int n_out_total = glu ? g.N / 2 : g.N;
  // GLU: columns are packed per 64 as [32 value | 32 gate]; the thread that owns value columns c..c+3 also reads
  // the gate columns c+32..c+35 and writes output columns (c/64)*32 + c%32 ..; threads on gate columns idle.
  const int ncol = n0 + pc4 * 4;                       // first tile column of this thread (pre-activation index)
  const int ocol = glu ? (ncol >> 6) * 32 + (ncol & 31) : ncol;
  const bool col_ok = ocol < n_out_total && (!glu || ((pc4 * 4) & 32) == 0);
  const bool vec_ok = ((n_out_total & 3) == 0) && ((g.ldc & 3) == 0) && ((g.c_seg_stride & 3) == 0) &&
                      ((reinterpret_cast<uintptr_t>(g.C) & 15u) == 0) &&
                      (!res || (((g.ldr & 3) == 0) && ((g.r_seg_stride & 3) == 0) && ((reinterpret_cast<uintptr_t>(g.R) & 15u) == 0)));
  float4 sc4 = make_float4(1.f, 1.f, 1.f, 1.f);
  if (res && g.scale) {
    sc4.x = ncol + 0 < g.N ? g.scale[ncol + 0] : 1.f; sc4.y = ncol + 1 < g.N ? g.scale[ncol + 1] : 1.f;
    sc4.z = ncol + 2 < g.N ? g.scale[ncol + 2] : 1.f; sc4.w = ncol + 3 < g.N ? g.scale[ncol + 3] : 1.f;
  }
  // row walk: one division, then pointer increments; a segment wrap (rare) recomputes the pointers
  int m = m0 + prow;
  int seg = m / rps, rr = m - seg * rps;
  float* cptr = g.C + (int64_t)seg * g.c_seg_stride + (int64_t)rr * g.ldc + ocol;
  // split-form destination: the row walk of cptr, pointing at the 128-byte group of this thread's columns
  constexpr bool second = OUT == 2 || OUT == 4;  // the activated tensor goes to C2, the raw one to C
  constexpr bool split_out = OUT == 1 || OUT == 2;
  float* const dbase = second ? ext->C2 : g.C;
  const int64_t ldd = second ? ext->ldc2 : g.ldc, dseg = second ? ext->c2_seg_stride : g.c_seg_stride;
  const int64_t dcol_bytes = split_out ? (int64_t)(ocol >> 5) * 128 + (ocol & 31) * 2 : (int64_t)ocol * 4;
  char* dptr = OUT != 0 ? reinterpret_cast<char*>(dbase + (int64_t)seg * dseg + (int64_t)rr * ldd) + dcol_bytes : nullptr;
  const int64_t dstep = (int64_t)RPP * ldd * 4;
  const int64_t cstep = (int64_t)RPP * g.ldc;
  const float* csrc = Cs + prow * CLD + pc4 * 4;
  constexpr int BATCH = NPASS < 8 ? NPASS : 8;
  static_assert(!RES_AHEAD || (RA_NPASS == NPASS && RA_RPP == RPP), "the residual pieces requested ahead are the store loop's");
  // one 16-byte piece of a row: pq = its pass (tile row prow + pq RPP), cpq / dpq = its destinations, rvq = its residual piece
  auto emit = [&](int pq, float* cpq, char* dpq, const float4& rvq) {
    {
      float4 v = *reinterpret_cast<const float4*>(csrc + pq * RPP * CLD);
      if (glu) {
        const float4 gt = *reinterpret_cast<const float4*>(csrc + pq * RPP * CLD + 32);
        v.x *= sigmoidf_(gt.x); v.y *= sigmoidf_(gt.y); v.z *= sigmoidf_(gt.z); v.w *= sigmoidf_(gt.w);
      } else if (EPI == SOPRO_EPI_GELU) {
        v.x = gelu_erf(v.x); v.y = gelu_erf(v.y); v.z = gelu_erf(v.z); v.w = gelu_erf(v.w);
      } else if (EPI == SOPRO_EPI_GELU_FAST) {
        v.x = gelu_fast(v.x); v.y = gelu_fast(v.y); v.z = gelu_fast(v.z); v.w = gelu_fast(v.w);
      } else if (EPI == SOPRO_EPI_TANH) {
        v.x = tanhf(v.x); v.y = tanhf(v.y); v.z = tanhf(v.z); v.w = tanhf(v.w);
      } else if (res) {
        v.x = rvq.x + sc4.x * v.x; v.y = rvq.y + sc4.y * v.y; v.z = rvq.z + sc4.z * v.z; v.w = rvq.w + sc4.w * v.w;
        if (OUT == 0 && ext && ext->ln_stats_out) {
          // LayerNorm statistics of the updated stream for the next contraction (sopro_gemm_split_ext.ln_stats): the 16 lanes that hold a
          // 64-column group of the row reduce (mean, squared deviations) - row_stats_kernel's arithmetic (host guarantees N % 64 == 0:
          // a group is entirely inside N or entirely outside, and rows >= M left through the `continue` above group-wise)
          float s1 = (v.x + v.y) + (v.z + v.w);
          s1 += __shfl_xor(s1, 1, 64); s1 += __shfl_xor(s1, 2, 64); s1 += __shfl_xor(s1, 4, 64); s1 += __shfl_xor(s1, 8, 64);
          const float mu = s1 * (1.0f / 64.0f);
          const float dx = v.x - mu, dy = v.y - mu, dz = v.z - mu, dw = v.w - mu;
          float s2 = (dx * dx + dy * dy) + (dz * dz + dw * dw);
          s2 += __shfl_xor(s2, 1, 64); s2 += __shfl_xor(s2, 2, 64); s2 += __shfl_xor(s2, 4, 64); s2 += __shfl_xor(s2, 8, 64);
          if ((pc4 & 15) == 0) {
            const int64_t mq = m0 + prow + pq * RPP;
            *reinterpret_cast<float2*>(ext->ln_stats_out + (mq * (g.N >> 6) + (ncol >> 6)) * 2) = make_float2(mu, s2);
          }
        }
      } else if (EPI == SOPRO_EPI_ROPE) {
        // rotate-half RoPE of the q | k heads in the leading rope_cols columns: the partner column (+- dh / 2) is in the LDS tile, the
        // position is the row's index within its utterance (sopro_rope_f32's arithmetic, without its pass over C)
        if (ncol < ext->rope_cols) {
          const int half = ext->rope_dh >> 1, e = ncol & (ext->rope_dh - 1);
          const bool lowh = e < half;
          const float4 pv = *reinterpret_cast<const float4*>(csrc + pq * RPP * CLD + (lowh ? half : -half));
          const int mq = m0 + prow + pq * RPP;
          const int64_t ti = (int64_t)(ext->rope_pos0 + mq % ext->rope_rows_per_seg) * half + (e & (half - 1));
          const float4 c4 = *reinterpret_cast<const float4*>(ext->rope_cos + ti), s4 = *reinterpret_cast<const float4*>(ext->rope_sin + ti);
          if (lowh) { v.x = v.x * c4.x - pv.x * s4.x; v.y = v.y * c4.y - pv.y * s4.y; v.z = v.z * c4.z - pv.z * s4.z; v.w = v.w * c4.w - pv.w * s4.w; }
          else { v.x = v.x * c4.x + pv.x * s4.x; v.y = v.y * c4.y + pv.y * s4.y; v.z = v.z * c4.z + pv.z * s4.z; v.w = v.w * c4.w + pv.w * s4.w; }
        }
      }
      if (split_out) {  // host guarantees N % 4 == 0 and 128-byte aligned rows
        uint2 h, l;
        split2_bf16(eluf_(v.x), eluf_(v.y), h.x, l.x);
        split2_bf16(eluf_(v.z), eluf_(v.w), h.y, l.y);
        *reinterpret_cast<uint2*>(dpq) = h;
        *reinterpret_cast<uint2*>(dpq + 64) = l;
        if (OUT == 1) return;
      } else if (OUT != 0) {
        bulk_store4(reinterpret_cast<float*>(dpq), make_float4(eluf_(v.x), eluf_(v.y), eluf_(v.z), eluf_(v.w)), big);
        if (OUT == 3) return;
      }
      if (vec_ok) {
        bulk_store4(cpq, v, big);
      } else {
        cpq[0] = v.x;
        if (ocol + 1 < n_out_total) cpq[1] = v.y;
        if (ocol + 2 < n_out_total) cpq[2] = v.z;
        if (ocol + 3 < n_out_total) cpq[3] = v.w;
      }
    }
  };
  if (whole_tile && vec_ok) {  // straight-line: see whole_tile above
    if (col_ok) {  // (GLU: the threads on gate columns have nothing to store; otherwise true for a whole tile)
#pragma unroll
      for (int q = 0; q < NPASS; ++q) {
        emit(q, cptr + q * cstep, OUT != 0 ? dptr + q * dstep : nullptr, rahead[RES_AHEAD ? q : 0]);
#ifdef SOPRO_DEV_SWITCHES
        if (g.dbg && tid == 0 && q == BATCH - 1) g.dbg[(int64_t)blockIdx.x * 8 + 5] = clock64();
#endif
      }
    }
    return;
  }
  auto batch = [&](int p0, const float4* rvb) {
    float* cp[BATCH];
    char* dp[BATCH];
    float4 rv[BATCH];
#pragma unroll
    for (int q = 0; q < BATCH; ++q) {
      const bool ok = col_ok && m < g.M;
      cp[q] = ok ? cptr : nullptr;
      dp[q] = dptr;
      rv[q] = res ? rvb[q] : make_float4(0.f, 0.f, 0.f, 0.f);
      m += RPP; rr += RPP; cptr += cstep;
      if (OUT != 0) dptr += dstep;
      if (rr >= rps) {
        do { rr -= rps; ++seg; } while (rr >= rps);
        cptr = g.C + (int64_t)seg * g.c_seg_stride + (int64_t)rr * g.ldc + ocol;
        if (OUT != 0) dptr = reinterpret_cast<char*>(dbase + (int64_t)seg * dseg + (int64_t)rr * ldd) + dcol_bytes;
      }
    }
#pragma unroll
    for (int q = 0; q < BATCH; ++q) {
      if (!cp[q]) continue;
      emit(p0 + q, cp[q], dp[q], rv[q]);
    }
  };
  if constexpr (RES_AHEAD) {  // (unrolled: the batches index the pieces requested ahead statically)
#pragma unroll
    for (int b = 0; b < NPASS / BATCH; ++b) {
      batch(b * BATCH, rahead + b * BATCH);
#ifdef SOPRO_DEV_SWITCHES
      if (g.dbg && tid == 0 && b == 0) g.dbg[(int64_t)blockIdx.x * 8 + 5] = clock64();  // developer probe: behind the first batch of rows
#endif
    }
  } else {
#pragma unroll 1
    for (int p0 = 0; p0 < NPASS; p0 += BATCH) {
      batch(p0, nullptr);
#ifdef SOPRO_DEV_SWITCHES
      if (g.dbg && tid == 0 && p0 == 0) g.dbg[(int64_t)blockIdx.x * 8 + 5] = clock64();
#endif
    }
  }
}
