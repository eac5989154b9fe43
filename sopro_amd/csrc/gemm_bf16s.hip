// fp32 contraction at bf16 matrix-core rate: every operand is split into NPL bf16 pieces (x = p0 + p1 [+ p2], each the
// round-to-nearest bf16 of what the previous pieces left) and the product is accumulated in fp32 with
// v_mfma_f32_32x32x16_bf16 over the piece pairs that matter:
//   NPL = 1 ("bf16x1"): p0*p0                              the bf16 mode of the engine (BASELINE configs[1]: bf16 weights
//                       and activations, fp32 accumulators): operands rounded to bf16 once, one MFMA pass.
//   NPL = 2 ("bf16x3"): p1*p0 + p0*p1 + p0*p0             16 mantissa bits per operand, ~2^-17 relative per product.
//                       For paths whose contract is a waveform tolerance (Mimi decoder).
//   NPL = 3 ("bf16x6"): + p1*p1 + p2*p0 + p0*p2            24 mantissa bits per operand, dropped terms <= 2^-25: the
//                       accuracy class of an fp32 fma chain, at 16/6 of the fp32-MFMA rate.  For token paths (NAR).
//
// A (activations, fp32 in HBM, same segmented / overlapping-row addressing as gemm_f32) is split while it is staged
// into LDS: row = NPL x [32 k of one piece, 64 B] + 16 B pad (144 / 208 B), so the ds_read_b128 fragment reads of a
// 16-lane group hit 16 distinct 16-byte slots.  W is split ONCE on the device by sopro_pack_w_bf16 into MFMA fragment
// order [n/32][k/16][piece][lane][8 bf16]; each wave streams its B fragments straight from L2 into registers with fully
// coalesced dwordx4 loads (1 KB per instruction, every byte used once), so W never touches LDS.
#include "common.h"
#include "gemm_epilogue.h"

namespace {

constexpr int BK = 32;

// (The ablation builds that priced this loop's parts - no MFMAs / no A path / no W loads / no barriers - are on record in DESIGN.md
// section 4 and profiles/r04_*; the product kernel carries no switches.)

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ bf16x8 as_frag(const uint4& v) { return *reinterpret_cast<const bf16x8*>(&v); }
__device__ __forceinline__ f16x8 as_frag16(const uint4& v) { return *reinterpret_cast<const f16x8*>(&v); }

// F16 ("f16x3", round 3): the two pieces are fp16 (11 mantissa bits each -> 22 bits per operand, dropped lo*lo <= 2^-22) at the
// same three MFMA passes as bf16x3: the token paths' accuracy class (NAR arg-max margins) at half of bf16x6's passes.  fp16
// has 5 exponent bits, so both operands are scaled by powers of two into its range (activations by A16_SCALE while they are
// staged, a weight matrix by its own power of two when it is packed; products are exact in the scale, the accumulator is
// multiplied by 1 / (both) in the epilogue): below 2^-14 a piece goes subnormal and the absolute resolution stays 2^-24 of
// the scaled value, i.e. ~7e-9 of the unscaled activation - small against 2^-22 of an O(1) row (the NAR stream is O(1-10): its
// contraction inputs are RMS-normalised or GELU outputs; rows of energy << 1e-2 lose relative precision to that floor).
constexpr float A16_SCALE = 8.0f;  // |activation| <= 8188 is exact in range; beyond it the operand saturates (finite, wrong)
// Round 5 (the range contract was documented, not enforced):
//  * fused-RMSNorm forms (AMODE 4) stage the UN-NORMALISED residual stream, whose row scale is whatever the checkpoint makes it:
//    there the scale is chosen PER ROW, a power of two that puts the largest of the row's first 32 elements into [2^7, 2^8) - 2^8
//    of headroom before fp16 saturates, subnormal floor at 2^-31 of that element - and is undone, exactly, with the row's RMS scale
//    in the epilogue.  Rows of RMS 1e-3 and 1e+3 therefore keep the same 22 bits.
//  * every f16 form tracks the largest scaled magnitude it staged; a workgroup that had to saturate an element adds one to
//    ext.range_events (device word, optional).  The engine's refinement checks the word after a pass and repeats the pass on the
//    six-pass bf16 operands, which have fp32's exponent range (sopro_nar_refine; tests/test_gpu_range.py).

// two fp32 values -> NPL packed 16-bit pairs (piece p of x in the low half, of y in the high half)
template <int NPL, bool F16 = false>
__device__ __forceinline__ void split_pair(float x, float y, unsigned (&pc)[NPL]) {
#pragma unroll
  for (int p = 0; p < NPL; ++p) {
    const f32x2_t v = {x, y};
    if constexpr (F16) {
      const f16x2_t h = __builtin_convertvector(v, f16x2_t);  // round to nearest even
      pc[p] = *reinterpret_cast<const unsigned*>(&h);
      x -= (float)h[0];  // exact
      y -= (float)h[1];
    } else {
      const bf16x2_t h = __builtin_convertvector(v, bf16x2_t);
      pc[p] = *reinterpret_cast<const unsigned*>(&h);
      x -= __uint_as_float(pc[p] << 16);  // exact
      y -= __uint_as_float(pc[p] & 0xffff0000u);
    }
  }
}

// AMODE: 0 = A is fp32 rows, split while staged; 1 = the same with ELU applied first; 2 = A is already in split form
// (NPL == 2 only: each 32-channel group = [32 hi | 32 lo] bf16, written by a producer's OUT = 1 / 2 epilogue): pure
// 16-byte copies; 3 = fp32 rows + pro_vec[k] (PRO_ADDVEC); 4 = fp32 rows whose RMSNorm is fused: the row's sum of squares is
// accumulated while it is staged and rsqrt(mean + eps) scales the accumulator in the epilogue (the norm's weight vector is
// folded into W by the host; needs K % 32 == 0 and no split-K: every workgroup sees its rows' whole K); 5 (round 4, NPL == 1 only:
// the bf16 mode's activation flow) = A is bf16 rows in memory (lda / a_seg_stride count bf16 elements): a K-step of a row is 64
// bytes, staged by pure 8-byte copies - half the operand bytes of the loop that bounds the one-pass kernel, no conversion work;
// 6 (round 5) = fp32 rows whose LayerNorm is fused: (a - mean) * rstd is staged instead of a, mean / rstd from the (mean, squared
// deviations) pairs the stream's producer left per 64-column group (sopro_gemm_split_ext.ln_stats; weight / bias of the norm folded
// into W' / the bias by the host; K % 64 == 0).
template <int NPL, int WM, int WN, int TM, int TN, int EPI, int AMODE, int OUT, bool SK, bool F16 = false>
__global__ __launch_bounds__(WM* WN * 64) void gemm_bf16s_kernel(const sopro_gemm_args g, const uint4* __restrict__ Wp,
                                                                 int ksubs, const sopro_gemm_split_ext ext) {
  constexpr int AROW = NPL * 64 + 16;  // bytes per LDS row
  constexpr bool FOLD = F16;  // the next step's staging inside this step's MFMA stream: see compute_spread
  constexpr int NT = WM * WN * 64;
  constexpr int BM = WM * TM * 32;
  constexpr int BN = WN * TN * 32;
  constexpr int A_F4 = BM * 8 / NT;
  constexpr int RSTEP = NT / 8;
  extern __shared__ float4 smem4[];
  unsigned char* As = reinterpret_cast<unsigned char*>(smem4);  // [2][BM][AROW]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int ntn = (g.N + BN - 1) / BN;
  const int bid = xcd_contiguous((int)blockIdx.x, (int)gridDim.x);
  int mt = bid / ntn, nt = bid % ntn;
  if (ext.group_m > 1) {  // grouped walk: group_m row tiles at a time, row tile fastest (see sopro_gemm_split_ext.group_m)
    const int ntm_all = (g.M + BM - 1) / BM;
    const int per = ext.group_m * ntn;
    const int grp = bid / per, rem = bid - grp * per;
    const int first = grp * ext.group_m;
    const int gsz = min(ntm_all - first, ext.group_m);
    mt = first + rem % gsz;
    nt = rem / gsz;
  }
  const int m0 = mt * BM, n0 = nt * BN;
  const int lrow = tid >> 3, lc4 = tid & 7;
  const int rps = g.rows_per_seg;
  long long* dbg = g.dbg ? g.dbg + (int64_t)blockIdx.x * 8 : nullptr;  // developer probe: shader-clock stamps
  if (dbg && tid == 0) dbg[0] = clock64();

  // Branch-free operand addressing: rows >= M and column tiles >= N are clamped onto valid ones (their accumulators
  // are never stored) and the K tail re-reads the row's last in-range piece (W is zero-padded there), so the main loop
  // is one basic block and the compiler can count its outstanding loads instead of draining them.
  static_assert(AMODE != 5 || NPL == 1, "bf16 rows are a one-pass (bf16 mode) operand form");
  constexpr int AES = AMODE == 5 ? 2 : 4;  // bytes per stored A element
  const char* ap[A_F4];
#pragma unroll
  for (int i = 0; i < A_F4; ++i) {
    const int m = min(m0 + lrow + i * RSTEP, g.M - 1);
    const int seg = m / rps;
    const int r = m - seg * rps;
    ap[i] = reinterpret_cast<const char*>(g.A) + ((int64_t)seg * g.a_seg_stride + (int64_t)r * g.lda) * AES;
  }
  const int ntiles = (g.N + 31) >> 5;
  const uint4* bp[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int t = min((n0 >> 5) + wn * TN + j, ntiles - 1);
    bp[j] = Wp + ((int64_t)t * ksubs * NPL) * 64 + lane;
  }
  // W fragments as (uniform base of the K-step) + (32-bit lane offset of the column tile) + (immediate of the substep / piece): a
  // request is ONE instruction, no vector address arithmetic (the packed W of one matrix is far below 2^31 fragments)
  int boff[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int t = min((n0 >> 5) + wn * TN + j, ntiles - 1);
    boff[j] = (t * ksubs * NPL) * 64 + lane;
  }
  float biasv[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int n = n0 + (wn * TN + j) * 32 + (lane & 31);
    biasv[j] = (g.bias && n < g.N) ? g.bias[n] : 0.f;
  }

  float4 raA[A_F4], raB[A_F4];  // A rows are requested TWO K-steps ahead (they come from HBM; W, one step ahead, from L2)
  float ssq[A_F4];
  float rsc[A_F4];    // f16 fused-RMSNorm form: the row's power-of-two staging scale (set by the first lstore)
  float a16max = 0.f; // f16 forms: largest scaled magnitude this thread staged (range guard)
#pragma unroll
  for (int i = 0; i < A_F4; ++i) { ssq[i] = 0.f; rsc[i] = A16_SCALE; }
  float sapin[A_F4];  // f16 forms on plain rows: the constant staging scale, one register per row piece (see FOLD)
#pragma unroll
  for (int i = 0; i < A_F4; ++i) sapin[i] = A16_SCALE;
  float lmu[A_F4], lrs[A_F4];  // fused LayerNorm: the staged rows' -mean * rstd and rstd = 1 / sqrt(var + eps)
  float4 pvA = make_float4(0.f, 0.f, 0.f, 0.f), pvB = pvA;
  const int KT = (g.K + BK - 1) / BK;
  const int klast = g.K - 4;

  auto gload = [&](int kt, float4 (&ra)[A_F4], float4& pv) {  // split-form A has the fp32 addresses: a K-step is the same 128 bytes of the row
    const int k = min(kt * BK + lc4 * 4, klast);
    if constexpr (AMODE == 5) {  // four bf16 = 8 bytes per thread and row, carried in .x / .y (bit patterns)
#pragma unroll
      for (int i = 0; i < A_F4; ++i) {
        const uint2 v = *reinterpret_cast<const uint2*>(ap[i] + (int64_t)k * 2);
        ra[i].x = __uint_as_float(v.x);
        ra[i].y = __uint_as_float(v.y);
      }
    } else {
#pragma unroll
      for (int i = 0; i < A_F4; ++i) ra[i] = *reinterpret_cast<const float4*>(ap[i] + (int64_t)k * 4);
    }
    if (AMODE == 3) pv = *reinterpret_cast<const float4*>(g.pro_vec + k);
  };
  auto bload = [&](int kt, uint4 (&rb)[TN][2][NPL]) {
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int p = 0; p < NPL; ++p) rb[j][s][p] = bp[j][((int64_t)(kt * 2 + s) * NPL + p) * 64];
  };
  // fresh: not the clamped re-stage of the last step (RMSNorm sums count once); first: the slice's first K-step (f16 row scales are chosen)
  // one row piece (i) of a K-step: the transform of its A format, the split, the LDS stores
  auto lstore_piece = [&](int buf, float4 (&ra)[A_F4], const float4& pv, int i, bool fresh, bool first) {
    if constexpr (FOLD) {  // the operand every first operation on the piece takes is (re)defined HERE: see FOLD at compute_spread
      if constexpr (AMODE == 4) asm volatile("" : "+v"(rsc[i]), "+v"(ssq[i]));
      else asm volatile("" : "+v"(sapin[i]));
    }
    if constexpr (AMODE == 5) {  // the row piece is already what the MFMA reads: 8-byte copy into the piece-0 plane
      unsigned char* a = As + buf * BM * AROW + lrow * AROW + lc4 * 8;
      *reinterpret_cast<uint2*>(a + i * RSTEP * AROW) = make_uint2(__float_as_uint(ra[i].x), __float_as_uint(ra[i].y));
    } else if constexpr (AMODE == 2) {  // pieces 0-3 of the 128-byte group are the hi halves, 4-7 the lo halves: the LDS row layout
      unsigned char* a = As + buf * BM * AROW + lrow * AROW + lc4 * 16;
      // member-wise: whole-struct copies keep the array in scratch memory (no promotion to registers)
      *reinterpret_cast<f32x4*>(a + i * RSTEP * AROW) = (f32x4){ra[i].x, ra[i].y, ra[i].z, ra[i].w};
    } else {
      if constexpr (AMODE == 1) {
        ra[i].x = eluf_(ra[i].x); ra[i].y = eluf_(ra[i].y); ra[i].z = eluf_(ra[i].z); ra[i].w = eluf_(ra[i].w);
      } else if constexpr (AMODE == 3) {
        ra[i].x += pv.x; ra[i].y += pv.y; ra[i].z += pv.z; ra[i].w += pv.w;
      } else if constexpr (AMODE == 6) {
        // x * rstd - mean * rstd: two packed fused multiply-adds per row piece (lmu holds -mean * rstd)
        const f32x2_t r2 = {lrs[i], lrs[i]}, n2 = {lmu[i], lmu[i]};
        const f32x2_t lo2 = (f32x2_t){ra[i].x, ra[i].y} * r2 + n2, hi2 = (f32x2_t){ra[i].z, ra[i].w} * r2 + n2;
        ra[i].x = lo2[0]; ra[i].y = lo2[1]; ra[i].z = hi2[0]; ra[i].w = hi2[1];
      }
      if constexpr (AMODE == 4) {
        if (fresh) ssq[i] = fmaf(ra[i].x, ra[i].x, fmaf(ra[i].y, ra[i].y, fmaf(ra[i].z, ra[i].z, fmaf(ra[i].w, ra[i].w, ssq[i]))));
      }
      unsigned char* a = As + buf * BM * AROW + lrow * AROW + lc4 * 8;
      {
        unsigned c0[NPL], c1[NPL];
        if constexpr (F16) {  // scaled, and saturated at fp16's largest finite value (no inf / NaN downstream; counted: range guard)
          const float F16MAX = 65504.0f;
          if constexpr (AMODE == 4) {
            if (first) {  // the eight lanes that stage a row's K-step agree on its scale: 2^(7 - exponent of the largest of 32 elements)
              float mx = fmaxf(fmaxf(fabsf(ra[i].x), fabsf(ra[i].y)), fmaxf(fabsf(ra[i].z), fabsf(ra[i].w)));
              mx = fmaxf(mx, __shfl_xor(mx, 1, 64));
              mx = fmaxf(mx, __shfl_xor(mx, 2, 64));
              mx = fmaxf(mx, __shfl_xor(mx, 4, 64));
              const int ex = (int)((__float_as_uint(mx) >> 23) & 255u);  // biased exponent; 0: zero / subnormal lead-in, 255: inf / NaN
              rsc[i] = (ex == 0 || ex == 255) ? A16_SCALE : __uint_as_float((unsigned)min(max(261 - ex, 2), 252) << 23);
            }
          }
          const float sa = AMODE == 4 ? rsc[i] : sapin[i];
          const float vx = ra[i].x * sa, vy = ra[i].y * sa, vz = ra[i].z * sa, vw = ra[i].w * sa;
          a16max = fmaxf(a16max, fmaxf(fmaxf(fabsf(vx), fabsf(vy)), fmaxf(fabsf(vz), fabsf(vw))));
          split_pair<NPL, true>(__builtin_amdgcn_fmed3f(vx, -F16MAX, F16MAX), __builtin_amdgcn_fmed3f(vy, -F16MAX, F16MAX), c0);
          split_pair<NPL, true>(__builtin_amdgcn_fmed3f(vz, -F16MAX, F16MAX), __builtin_amdgcn_fmed3f(vw, -F16MAX, F16MAX), c1);
        } else {
          split_pair<NPL>(ra[i].x, ra[i].y, c0);
          split_pair<NPL>(ra[i].z, ra[i].w, c1);
        }
#pragma unroll
        for (int p = 0; p < NPL; ++p) *reinterpret_cast<uint2*>(a + i * RSTEP * AROW + p * 64) = make_uint2(c0[p], c1[p]);
      }
    }
  };
  // fresh: not the clamped re-stage of the last step (RMSNorm sums count once); first: the slice's first K-step (f16 row scales are chosen)
  auto lstore = [&](int buf, float4 (&ra)[A_F4], const float4& pv, bool fresh = true, bool first = false) {
#pragma unroll
    for (int i = 0; i < A_F4; ++i) lstore_piece(buf, ra, pv, i, fresh, first);
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int frow = lane & 31, fg = lane >> 5;
  // piece pairs (A piece, B piece), smallest terms first
  constexpr int NPAIR = NPL == 1 ? 1 : (NPL == 2 ? 3 : 6);
  constexpr int PA[6] = {NPL == 1 ? 0 : (NPL == 2 ? 1 : 2), 0, NPL == 2 ? 0 : 1, 1, 0, 0};
  constexpr int PB[6] = {0, NPL == 2 ? 1 : 2, NPL == 2 ? 0 : 1, 0, 1, 0};
  auto compute = [&](int buf, const uint4 (&rb)[TN][2][NPL]) {
    const unsigned char* a = As + buf * BM * AROW + (wm * TM * 32 + frow) * AROW + fg * 16;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      uint4 af[TM][NPL];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int p = 0; p < NPL; ++p) af[i][p] = *reinterpret_cast<const uint4*>(a + i * 32 * AROW + p * 64 + s * 32);
      // each pass walks all accumulators so that consecutive MFMAs are independent
#pragma unroll
      for (int q = 0; q < NPAIR; ++q)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            if constexpr (F16)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_frag16(af[i][PA[q]]), as_frag16(rb[j][s][PB[q]]), acc[i][j], 0, 0, 0);
            else
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_frag(af[i][PA[q]]), as_frag(rb[j][s][PB[q]]), acc[i][j], 0, 0, 0);
    }
  };

  // The K-step with the NEXT steps' global requests dealt out between its MFMAs instead of issued in one burst behind the barrier.
  // Shader-clock sums inside the loop (profiles/r04_gemm_kstep_stamps.txt) showed a wave spending 38-40 % of the K loop ISSUING its
  // twelve requests: all eight waves of a CU push them (12 KB per wave) into the CU's one address path at the same moment, issue is
  // in order - no MFMA goes out until the path has taken them - and the path then idles through the MFMA phase.  One request behind
  // every NM / NL MFMAs, W fragments of step kB first (needed next step), then the A rows of step kA; W addresses are (K-step base)
  // + (lane offset) + immediate, so a request is one instruction.  The fences around a request keep it between those MFMAs; for the
  // three-pass forms ALU work and LDS traffic (the next substep's fragment reads) may cross them (mask 0x786: strict fences cost
  // them 3-19 %), for the one-pass form nothing may (strict: +5-14 %, relaxed: +-0).  Same products in the same order: bit-identical.
  // Measured (tools/gemm_loop_probe.py, profiles/r04_gemm_spread_requests_*.txt): three-pass +5-10 %, one-pass +5-14 %.
  // FOLD (round 6, f16 forms): the staging of the NEXT step's rows (rs: requested during the step before this one) is placed INSIDE this
  // step's MFMA stream - row piece i behind MFMA NM/2 + i (NM/2) / A_F4.  As a separate lstore() behind the step its arithmetic is pure
  // register work that the compiler orders by data dependence alone - and it put the f16 forms' scale multiply directly behind the
  // rows' requests: `s_waitcnt vmcnt(0)` on a load just issued, in every unsplit f16 kernel on plain rows (one memory latency per
  // trip; the fused-RMSNorm kernels waited for the rows just requested at the top of every other step).  Fences do not hold pure
  // arithmetic in place (they order the machine scheduler, not instruction selection); a data dependence on a volatile asm statement
  // does: the first operation on a piece takes an operand (its scale; its running sum of squares) that an empty asm statement
  // re-defines where the piece's code stands.  A piece is then a K-step old when it is first touched, and the second half of the
  // step's MFMAs hides its arithmetic.  Measured (192 CUs, shader-clock stamps, profiles/r06_tile_life_*.txt): K-step of the
  // refinement's ff2 2175 -> 1970 cycles, refinement pass 14.73 -> 14.36 ms.  The bf16 kernels (whose staging the compiler happened
  // to leave late in one of the two steps of a trip) lose 1-10 % per K-step with it: they keep the separate lstore().
  auto compute_spread = [&](int buf, const uint4 (&rb)[TN][2][NPL], int kA, float4 (&ra)[A_F4], float4& pv, int kB, uint4 (&rbn)[TN][2][NPL],
                            float4 (&rs)[A_F4], const float4& pvs, bool fresh_s) {
    constexpr int NB = TN * 2 * NPL, NL = NB + A_F4, NM = 2 * NPAIR * TM * TN;
    constexpr int STRIDE = NM / NL > 0 ? NM / NL : 1;
    const int kcol = min(kA * BK + lc4 * 4, klast);
    if (AMODE == 3) pv = *reinterpret_cast<const float4*>(g.pro_vec + kcol);
    auto issue = [&](int n) {
      if (n < NB) {
        const int j = n / (2 * NPL), ss = (n / NPL) % 2, p = n % NPL;
        const uint4* wk = Wp + (int64_t)__builtin_amdgcn_readfirstlane(kB) * (2 * NPL * 64);  // uniform: scalar registers
        rbn[j][ss][p] = wk[boff[j] + (ss * NPL + p) * 64];
      } else if (n < NL) {
        const int i = n - NB;
        if constexpr (AMODE == 5) {
          const uint2 v = *reinterpret_cast<const uint2*>(ap[i] + (int64_t)kcol * 2);
          ra[i].x = __uint_as_float(v.x);
          ra[i].y = __uint_as_float(v.y);
        } else {
          ra[i] = *reinterpret_cast<const float4*>(ap[i] + (int64_t)kcol * 4);
        }
      }
    };
    const unsigned char* a = As + buf * BM * AROW + (wm * TM * 32 + frow) * AROW + fg * 16;
    int m = 0, nl = 0;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      uint4 af[TM][NPL];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int p = 0; p < NPL; ++p) af[i][p] = *reinterpret_cast<const uint4*>(a + i * 32 * AROW + p * 64 + s * 32);
#pragma unroll
      for (int q = 0; q < NPAIR; ++q)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            if constexpr (F16)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_frag16(af[i][PA[q]]), as_frag16(rb[j][s][PB[q]]), acc[i][j], 0, 0, 0);
            else
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_frag(af[i][PA[q]]), as_frag(rb[j][s][PB[q]]), acc[i][j], 0, 0, 0);
            ++m;
            if constexpr (FOLD) {
#pragma unroll
              for (int i = 0; i < A_F4; ++i)
                if (m == NM / 2 + i * (NM / 2) / A_F4) {
                  __builtin_amdgcn_sched_barrier(0x104);  // (LDS reads and scalar work may cross; vector work, requests, MFMAs, LDS writes may not)
                  lstore_piece(buf ^ 1, rs, pvs, i, fresh_s, false);
                }
            }
            if (m % STRIDE == 0 && nl < NL) {
              if constexpr (NPL == 1) {
                __builtin_amdgcn_sched_barrier(0);
                issue(nl++);
                __builtin_amdgcn_sched_barrier(0);
              } else {
                __builtin_amdgcn_sched_barrier(0x786);
                issue(nl++);
                __builtin_amdgcn_sched_barrier(0x786);
              }
            }
          }
    }
#pragma unroll
    for (int n = 0; n < NL; ++n)  // (whatever the MFMA count left over)
      if (n >= nl) issue(n);
  };

  // Split-K (ext.ksplit > 1, for problems with too few tiles to fill the chip: a long K loop on a handful of workgroups
  // is one memory latency per K-step): blockIdx.y picks a contiguous range of K-steps; every slice stores its raw
  // accumulators, the LAST slice to arrive at the tile's ticket adds all of them IN SLICE ORDER (deterministic, whoever
  // is last) and runs the epilogue.
  const int KS = SK ? ext.ksplit : 1;  // a separate instantiation: the reduction's registers must not weigh on the unsplit kernel
  const int per = (KT + KS - 1) / KS;
  const int kt0 = (int)blockIdx.y * per;
  const int nkt = max(0, min(KT, kt0 + per) - kt0);  // K-steps of this slice (0 for a trailing empty slice)
  const int ktl = max(kt0, kt0 + nkt - 1);           // last valid step of the slice (prefetch clamp)

  // Two K-steps per trip (register double buffers for the B fragments and the A rows, LDS double buffer for the split A).
  // Prefetches beyond the last step are clamped onto it: redundant but branch-free.
  uint4 rb0[TN][2][NPL], rb1[TN][2][NPL];
  constexpr bool DEEP = true;  // A rows two K-steps ahead (false: one step ahead, one register set)
  gload(min(kt0, KT - 1), raA, pvA);
  bload(min(kt0, KT - 1), rb0);
  if (DEEP) gload(min(min(kt0 + 1, ktl), KT - 1), raB, pvB);
  if constexpr (AMODE == 6) {
    // behind the first operand requests (their latency covers these): the eight lanes that stage a row share its K / 64 pairs -
    // lane c takes pairs c, c + 8, .. - and combine them with Chan's update: sum of the groups' squared deviations + 64 (group mean - mean)^2
    const int P = g.K >> 6;
#pragma unroll
    for (int i = 0; i < A_F4; ++i) {
      const float2* st = reinterpret_cast<const float2*>(ext.ln_stats) + (int64_t)min(m0 + lrow + i * RSTEP, g.M - 1) * P;
      float sm = 0.f;
      for (int p = lc4; p < P; p += 8) sm += st[p].x;
      sm += __shfl_xor(sm, 1, 64); sm += __shfl_xor(sm, 2, 64); sm += __shfl_xor(sm, 4, 64);
      const float mean = sm / (float)P;
      float m2 = 0.f;
      for (int p = lc4; p < P; p += 8) {
        const float2 v = st[p];
        const float d = v.x - mean;
        m2 += fmaf(64.0f * d, d, v.y);
      }
      m2 += __shfl_xor(m2, 1, 64); m2 += __shfl_xor(m2, 2, 64); m2 += __shfl_xor(m2, 4, 64);
      lrs[i] = rsqrtf(m2 / (float)g.K + ext.rms_eps);
      lmu[i] = -mean * lrs[i];
    }
  }
  lstore(0, raA, pvA, true, true);
  __syncthreads();
  if (dbg && tid == 0) dbg[1] = clock64();
  {
    // The two-step body is ONE basic block: the odd last step is peeled off instead of leaving through a break in the middle
    // (which let the optimiser sink the second step's W requests out of the first step and cost a vmcnt(0) at the loop header).
    // +2-3 % on the large decoder shapes, bit-identical (tools/gemm_ab_probe.py; pinning every step's requests ahead of its
    // MFMAs with sched_barrier measured 0-9 % SLOWER).
    int it = 0;
    for (; it + 1 < nkt; it += 2) {
      const int k1 = min(kt0 + it + 1, ktl), k2 = min(kt0 + it + 2, ktl), k3 = min(kt0 + it + 3, ktl);
      // step it; requests: A rows of step it + 2, W fragments of step it + 1; stages the rows of step it + 1
      compute_spread(0, rb0, k2, raA, pvA, k1, rb1, raB, pvB, true);
      if constexpr (!FOLD) lstore(1, raB, pvB, true);
      __syncthreads();
      compute_spread(1, rb1, k3, raB, pvB, k2, rb0, raA, pvA, it + 2 < nkt);
      if constexpr (!FOLD) lstore(0, raA, pvA, it + 2 < nkt);
      __syncthreads();
    }
    if (it < nkt) {  // (workgroup-uniform)
      compute(0, rb0);
      // the epilogue's tile (and the split-K flag word) alias A buffer 0: every wave must be done reading its fragments
      // before any wave writes there (odd step counts - K = 32, 96, ... - raced without this barrier)
      __syncthreads();
    }
  }
  if (dbg && tid == 0) dbg[2] = clock64();
  if constexpr (SK) {
    constexpr int PER_THREAD = TM * TN * 16;
    const int tile = bid, ntile = gridDim.x;
    float* mine = ext.ws + ((int64_t)((int64_t)blockIdx.y * ntile + tile) * NT + tid) * PER_THREAD;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; r += 4)
          *reinterpret_cast<float4*>(mine + (i * TN + j) * 16 + r) = make_float4(acc[i][j][r], acc[i][j][r + 1], acc[i][j][r + 2], acc[i][j][r + 3]);
    __syncthreads();  // all partial stores of this workgroup are issued before its ticket
    int* flag = reinterpret_cast<int*>(smem4);
    if (tid == 0) {
      __atomic_thread_fence(__ATOMIC_RELEASE);  // agent scope: partials visible before the ticket
      const int old = __hip_atomic_fetch_add(ext.tickets + tile, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
      *flag = old;
      if (old == KS - 1) __hip_atomic_store(ext.tickets + tile, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // ready for the next launch
    }
    __syncthreads();
    if (*flag != KS - 1) return;  // whole workgroup
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
    __syncthreads();  // the flag word is part of the epilogue's LDS tile
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    // all slices of a 4-float piece are requested together (up to 16 loads in flight), then added in slice order
    const int64_t sl_stride = (int64_t)ntile * NT * PER_THREAD;
    const float* part0 = ext.ws + ((int64_t)tile * NT + tid) * PER_THREAD;
    for (int s0 = 0; s0 < KS; s0 += 16) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int r = 0; r < 16; r += 4) {
            f32x4 v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
              const int sl = min(s0 + u, KS - 1);  // clamped: a repeated slice is read but not added
              v[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(part0 + sl * sl_stride + (i * TN + j) * 16 + r));
            }
#pragma unroll
            for (int u = 0; u < 16; ++u) {
              if (s0 + u < KS) {
                acc[i][j][r] += v[u][0]; acc[i][j][r + 1] += v[u][1]; acc[i][j][r + 2] += v[u][2]; acc[i][j][r + 3] += v[u][3];
              }
            }
          }
    }
  }
  // f16 forms: the operand scales (powers of two: exact) are undone where the accumulators leave for the epilogue's tile - one
  // multiply-add per element there (acc x cscale + bias, or acc x (row scale x cscale) + bias: the product is exact either way, so
  // the sum rounds as it did when the accumulators were scaled in place first; round 6: 64 multiplies and their AGPR round trips less)
  float cscale = 1.0f;
  if constexpr (F16) {
    // (fused-RMSNorm form: the activation scale is per row and leaves with the row's RMS scale below; acc_scale holds the constant one)
    cscale = AMODE == 4 ? ext.acc_scale * A16_SCALE : ext.acc_scale;
    if (ext.range_events && !(a16max <= 65504.0f)) atomicAdd(ext.range_events, 1);  // saturated (or NaN) operands: see the header note
  }
  const float* rs = nullptr;
  if constexpr (AMODE == 4) {
    // the 8 threads that staged a row hold its partial sums (consecutive lanes): reduce, publish one scale per tile row
    // behind the epilogue tile
    float* rsl = reinterpret_cast<float*>(smem4) + BM * (BN + 4);
#pragma unroll
    for (int i = 0; i < A_F4; ++i) {
      float v = ssq[i];
      v += __shfl_xor(v, 1, 64);
      v += __shfl_xor(v, 2, 64);
      v += __shfl_xor(v, 4, 64);
      float r = rsqrtf(v / (float)g.K + ext.rms_eps);
      if constexpr (F16) r *= __uint_as_float(0x7f000000u - __float_as_uint(rsc[i])) * cscale;  // 1 / (a power of two), exactly: exponent 254 - e; x the constant scales
      if (lc4 == 0) rsl[lrow + i * RSTEP] = r;
    }
    __syncthreads();
    rs = rsl;
  }
  gemm_store_tile<WM, WN, TM, TN, EPI, OUT>(g, reinterpret_cast<float*>(smem4), acc, biasv, m0, n0, &ext, rs, cscale);
  if (dbg && tid == 0) dbg[3] = clock64();
}

// W [N, ldw] fp32 -> fragment-ordered bf16 pieces; one thread per 16-byte fragment piece
template <int NPL, bool F16 = false>
__global__ __launch_bounds__(256) void pack_w_kernel(const float* __restrict__ W, int64_t ldw, int N, int K, uint4* __restrict__ out,
                                                     int ksubs, int64_t total, float wscale = 1.0f) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int lane = (int)(i & 63);
  const int64_t pk = i >> 6;
  const int p = (int)(pk % NPL);
  const int64_t tk = pk / NPL;
  const int ks = (int)(tk % ksubs);
  const int t = (int)(tk / ksubs);
  const int n = t * 32 + (lane & 31);
  const int k0 = ks * 16 + (lane >> 5) * 8;
  unsigned w[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int k = k0 + 2 * e;
    const float x = (n < N && k < K) ? W[(int64_t)n * ldw + k] : 0.f;
    const float y = (n < N && k + 1 < K) ? W[(int64_t)n * ldw + k + 1] : 0.f;
    unsigned pc[NPL];
    if constexpr (F16) split_pair<NPL, true>(x * wscale, y * wscale, pc);
    else split_pair<NPL>(x, y, pc);
    w[e] = pc[p];
  }
  out[i] = make_uint4(w[0], w[1], w[2], w[3]);
}

template <int NPL, int WM, int WN, int TM, int TN, int EPI, int AMODE, int OUT, bool F16 = false>
int launch_one(const sopro_gemm_args& g, const uint4* wp, int ksubs, const sopro_gemm_split_ext& ext, hipStream_t s) {
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  constexpr size_t lds_main = (size_t)2 * BM * (NPL * 64 + 16), lds_epi = (size_t)BM * (BN + 4 + (AMODE == 4 ? 1 : 0)) * sizeof(float);
  constexpr size_t lds = lds_main > lds_epi ? lds_main : lds_epi;
  const int ks = ext.ksplit > 1 ? ext.ksplit : 1;
  auto kern = ks > 1 ? gemm_bf16s_kernel<NPL, WM, WN, TM, TN, EPI, AMODE, OUT, true, F16> : gemm_bf16s_kernel<NPL, WM, WN, TM, TN, EPI, AMODE, OUT, false, F16>;
  constexpr size_t lds_cap = lds > (size_t)96 * 1024 ? lds : (size_t)96 * 1024;  // room for sopro_set_lds_floor
  if (ks > 1) SOPRO_SET_MAX_LDS_ONCE(kern, lds_cap);
  else SOPRO_SET_MAX_LDS_ONCE(kern, lds_cap);
  const size_t lds_req = lds > (size_t)g_sopro_lds_floor ? lds : (size_t)g_sopro_lds_floor;
  const int ntm = (g.M + BM - 1) / BM, ntn = (g.N + BN - 1) / BN;
  if (ks > 1) {
    const int64_t need = (int64_t)ks * ntm * ntn * BM * BN * (int64_t)sizeof(float);
    if (!ext.ws || !ext.tickets || ext.ws_bytes < need || ext.n_tickets < ntm * ntn) {
      sopro_set_error("split-K needs a %lld-byte workspace and %d tickets (got %lld, %d)", (long long)need, ntm * ntn,
                      (long long)ext.ws_bytes, ext.n_tickets);
      return -2;
    }
  }
  hipLaunchKernelGGL(kern, dim3(ntm * ntn, ks), dim3(WM * WN * 64), lds_req, s, g, wp, ksubs, ext);
  SOPRO_LAUNCH_CHECK();
}

inline int amode_of(const sopro_gemm_args& g, const sopro_gemm_split_ext& ext) {
  if (ext.a_format == 1) return 2;
  if (ext.a_format == 2) return 5;
  if (ext.rms_norm) return 4;
  if (ext.ln_stats) return 6;
  return g.prologue == SOPRO_PRO_ELU ? 1 : (g.prologue == SOPRO_PRO_ADDVEC ? 3 : 0);
}

// The (epilogue, A format, output mode) combinations the engine issues; anything else is refused.
template <int NPL, int WM, int WN, int TM, int TN>
int launch_cfg3(const sopro_gemm_args& g, const uint4* wp, int ksubs, const sopro_gemm_split_ext& ext, hipStream_t s) {
  const int key = g.epilogue * 100 + amode_of(g, ext) * 10 + ext.c_mode;
  // (EPI_GELU runs the fast erf form on the waveform / bf16-mode paths: see gelu_fast in common.h)
#define SOPRO_CASE(E, A, O) \
  case (E) * 100 + (A) * 10 + (O): return launch_one<NPL, WM, WN, TM, TN, ((E) == SOPRO_EPI_GELU ? SOPRO_EPI_GELU_FAST : (E)), A, O>(g, wp, ksubs, ext, s)
  switch (key) {
    SOPRO_CASE(SOPRO_EPI_NONE, 0, 0);  // transformer qkv, RVQ output projections
    SOPRO_CASE(SOPRO_EPI_GELU, 0, 0);  // transformer fc1
    SOPRO_CASE(SOPRO_EPI_RES, 0, 0);   // transformer o / fc2 (+ layer scale)
    SOPRO_CASE(SOPRO_EPI_ROPE, 0, 0);  // transformer qkv with the rotary embedding of q | k in the epilogue (round 5)
    SOPRO_CASE(SOPRO_EPI_NONE, 6, 0);  // the same three behind a fused LayerNorm (round 5: sopro_gemm_split_ext.ln_stats)
    SOPRO_CASE(SOPRO_EPI_GELU, 6, 0);
    SOPRO_CASE(SOPRO_EPI_ROPE, 6, 0);
    SOPRO_CASE(SOPRO_EPI_NONE, 1, 0);  // fp32 activations with an ELU prologue (SEANet convs)
    SOPRO_CASE(SOPRO_EPI_RES, 1, 0);
    SOPRO_CASE(SOPRO_EPI_NONE, 0, 3);  // activated-copy flow of the SEANet decoder: ELU applied once, by the producer
    SOPRO_CASE(SOPRO_EPI_NONE, 0, 4);
    SOPRO_CASE(SOPRO_EPI_RES, 0, 3);
    default: break;
  }
  if constexpr (NPL == 1) {  // bf16 mode with bf16 activations in memory (round 4): the SEANet decoder's flow
    switch (key) {
      SOPRO_CASE(SOPRO_EPI_NONE, 0, 7);  // first conv: fp32 transformer stream in, ELU as bf16 rows out
      SOPRO_CASE(SOPRO_EPI_NONE, 5, 8);  // transposed convs: bf16 in, raw + activated bf16 copies out
      SOPRO_CASE(SOPRO_EPI_NONE, 5, 7);  // residual block k = 3 conv
      SOPRO_CASE(SOPRO_EPI_RES, 5, 7);   // residual block k = 1 conv + bf16 skip operand
      SOPRO_CASE(SOPRO_EPI_NONE, 5, 6);  // transposed conv in front of the fused 128-channel block: raw bf16 out
      SOPRO_CASE(SOPRO_EPI_NONE, 5, 0);  // (bf16 in, fp32 out: tests, mixed flows)
      default: break;
    }
  }
  if constexpr (NPL == 2) {
    switch (key) {
      SOPRO_CASE(SOPRO_EPI_NONE, 0, 1);  // split-form activation flow (SOPRO_MIMI_SPLIT_FORM): fp32 in, ELU + split out
      SOPRO_CASE(SOPRO_EPI_NONE, 0, 2);
      SOPRO_CASE(SOPRO_EPI_NONE, 2, 0);
      SOPRO_CASE(SOPRO_EPI_NONE, 2, 1);
      SOPRO_CASE(SOPRO_EPI_NONE, 2, 2);
      SOPRO_CASE(SOPRO_EPI_NONE, 2, 4);  // (round 6: split-form input, raw + activated fp32 copies out - a level behind a split-form one)
      SOPRO_CASE(SOPRO_EPI_RES, 2, 1);
      default: break;
    }
  }
#undef SOPRO_CASE
  sopro_set_error("sopro_gemm_bf16x%d: (epilogue %d, prologue %d, a_format %d, c_mode %d) is not an available combination",
                  NPL == 2 ? 3 : NPL, g.epilogue, g.prologue, ext.a_format, ext.c_mode);
  return -2;
}

template <int NPL, int WM, int WN, int TM, int TN, bool F16 = false>
int launch_cfg6(const sopro_gemm_args& g, const uint4* wp, int ksubs, const sopro_gemm_split_ext& ext, hipStream_t s) {
  const int key = g.epilogue * 10 + amode_of(g, ext);
  // (EPI_GELU: the fast erf form for the f16 three-pass and one-pass kernels; the six-pass kernel - conditioning, the refinement's
  // fallback: fp32-class paths - keeps erff)
#define SOPRO_CASE(E, A) \
  case (E) * 10 + (A): return launch_one<NPL, WM, WN, TM, TN, (((E) == SOPRO_EPI_GELU && (F16 || NPL != 3)) ? SOPRO_EPI_GELU_FAST : (E)), A, 0, F16>(g, wp, ksubs, ext, s)
  switch (key) {
    SOPRO_CASE(SOPRO_EPI_NONE, 0);  // plain projections
    SOPRO_CASE(SOPRO_EPI_GELU, 0);  // FF1
    SOPRO_CASE(SOPRO_EPI_RES, 0);   // FF2 + skip
    SOPRO_CASE(SOPRO_EPI_NONE, 3);  // NAR heads: z + head-id embedding
    SOPRO_CASE(SOPRO_EPI_GELU, 4);  // RMSNorm -> FF1 -> GELU
    SOPRO_CASE(SOPRO_EPI_NONE, 4);
    default: break;
  }
#undef SOPRO_CASE
  if constexpr (WN * TN * 32 >= 64) {
    // (EPI_GLU: the hardware-exponential sigmoid for the f16 three-pass and one-pass kernels, the precise one for the six-pass kernel - as for GELU)
    constexpr int GLU_ = (F16 || NPL != 3) ? SOPRO_EPI_GLU_FAST : SOPRO_EPI_GLU;
    if (key == SOPRO_EPI_GLU * 10) return launch_one<NPL, WM, WN, TM, TN, GLU_, 0, 0, F16>(g, wp, ksubs, ext, s);
    if (key == SOPRO_EPI_GLU * 10 + 4) return launch_one<NPL, WM, WN, TM, TN, GLU_, 4, 0, F16>(g, wp, ksubs, ext, s);  // RMSNorm -> GLU
  }
  sopro_set_error("sopro_gemm_%s: (epilogue %d, prologue %d) is not an available combination", F16 ? "f16x3" : (NPL == 3 ? "bf16x6" : "bf16x1"), g.epilogue, g.prologue);
  return -2;
}

// c_mode 5: per-row arg-max partials instead of C (64x64 tiles: 32 partials per 2048 columns)
template <int NPL, bool F16 = false>
int launch_argmax(sopro_gemm_args& g, const uint4* wp, int ksubs, const sopro_gemm_split_ext& ext, hipStream_t s) {
  SOPRO_CHECK_ARG(g.epilogue == SOPRO_EPI_NONE && g.prologue == SOPRO_PRO_NONE && !ext.rms_norm && ext.a_format == 0,
                  "arg-max output takes a plain contraction (no prologue / epilogue / fused norm)");
  SOPRO_CHECK_ARG(ext.C2 && ext.ldc2 >= (g.N + 63) / 64, "arg-max output: C2 = [M][ldc2 >= ceil(N / 64)] (value, index) pairs");
  // 128-row tiles for many-row problems (the refinement of a whole batch: 12800 rows): every W fragment serves twice the rows; the
  // (max, column) pairs stay per 64-column tile, the arithmetic per element is unchanged.  SOPRO_ARGMAX_TM=1: 64-row tiles always.
  static const bool tm1 = SOPRO_DEV_ENV("SOPRO_ARGMAX_TM") != nullptr && SOPRO_DEV_ENV("SOPRO_ARGMAX_TM")[0] == '1';
  // round 6: 128 x 128 tiles for many rows AND many columns (the refinement's stage heads: 25600 rows x up to 32768 logits, K = 256): a
  // staged A row serves 128 columns instead of 64 - half the A traffic and half the split work per product; pairs stay per 64 columns.
  // SOPRO_ARGMAX_TN=1 (developer A/B): the round-5 128 x 64 tiles
  static const bool tn1 = SOPRO_DEV_ENV("SOPRO_ARGMAX_TN") != nullptr && SOPRO_DEV_ENV("SOPRO_ARGMAX_TN")[0] == '1';
  if (g.M >= 8192 && g.N >= 2048 && (g.N & 127) == 0 && !tm1 && !tn1) return launch_one<NPL, 2, 2, 2, 2, SOPRO_EPI_NONE, 0, 5, F16>(g, wp, ksubs, ext, s);
  if (g.M >= 8192 && !tm1) return launch_one<NPL, 2, 2, 2, 1, SOPRO_EPI_NONE, 0, 5, F16>(g, wp, ksubs, ext, s);
  return launch_one<NPL, 2, 2, 1, 1, SOPRO_EPI_NONE, 0, 5, F16>(g, wp, ksubs, ext, s);
}

int check_rope(const sopro_gemm_args& g, const sopro_gemm_split_ext& ext) {
  if (g.epilogue != SOPRO_EPI_ROPE) return 0;
  const int dh = ext.rope_dh;
  SOPRO_CHECK_ARG(ext.rope_cos && ext.rope_sin && aligned16(ext.rope_cos) && aligned16(ext.rope_sin), "EPI_ROPE needs 16-byte aligned cos / sin tables");
  SOPRO_CHECK_ARG(dh >= 8 && dh <= 64 && (dh & (dh - 1)) == 0, "EPI_ROPE: rope_dh must be a power of two in 8..64 (heads must not straddle a 64-column tile)");
  SOPRO_CHECK_ARG(ext.rope_cols > 0 && ext.rope_cols <= g.N && ext.rope_cols % dh == 0, "EPI_ROPE: rope_cols must be whole heads within N");
  SOPRO_CHECK_ARG(ext.rope_rows_per_seg > 0 && ext.rope_pos0 >= 0, "EPI_ROPE: rope_rows_per_seg > 0, rope_pos0 >= 0");
  SOPRO_CHECK_ARG(ext.c_mode == 0 && ext.a_format == 0 && g.prologue == SOPRO_PRO_NONE && !ext.rms_norm, "EPI_ROPE is a plain fp32-rows form");
  SOPRO_CHECK_ARG((g.N & 3) == 0 && (g.ldc & 3) == 0 && (g.c_seg_stride & 3) == 0 && aligned16(g.C), "EPI_ROPE: 16-byte aligned output rows");
  return 0;
}

int check_ln(const sopro_gemm_args& g, const sopro_gemm_split_ext& ext) {
  if (ext.ln_stats) {
    SOPRO_CHECK_ARG((g.K & 63) == 0 && (reinterpret_cast<uintptr_t>(ext.ln_stats) & 7u) == 0, "fused LayerNorm: K % 64 == 0, 8-byte aligned ln_stats");
    SOPRO_CHECK_ARG(ext.a_format == 0 && g.prologue == SOPRO_PRO_NONE && !ext.rms_norm && ext.c_mode == 0, "fused LayerNorm takes plain fp32 rows (no prologue, c_mode 0)");
    SOPRO_CHECK_ARG(g.epilogue == SOPRO_EPI_NONE || g.epilogue == SOPRO_EPI_GELU || g.epilogue == SOPRO_EPI_ROPE, "fused LayerNorm: epilogue NONE, GELU or ROPE");
    SOPRO_CHECK_ARG(ext.rms_eps > 0.f, "fused LayerNorm: rms_eps carries the norm's eps (> 0)");
  }
  if (ext.ln_stats_out) {
    SOPRO_CHECK_ARG(g.epilogue == SOPRO_EPI_RES && ext.c_mode == 0 && (g.N & 63) == 0 && (reinterpret_cast<uintptr_t>(ext.ln_stats_out) & 7u) == 0,
                    "ln_stats_out: an EPI_RES contraction with fp32 output rows, N % 64 == 0, 8-byte aligned pairs");
    SOPRO_CHECK_ARG((g.ldc & 3) == 0 && (g.c_seg_stride & 3) == 0 && aligned16(g.C) && (g.ldr & 3) == 0 && (g.r_seg_stride & 3) == 0 && aligned16(g.R),
                    "ln_stats_out: 16-byte aligned output / residual rows");
  }
  return 0;
}

int check_common(sopro_gemm_args& g, sopro_gemm_split_ext& ext, const void* packed_w) {
  if (g.a_seg_stride == 0) g.a_seg_stride = (int64_t)g.rows_per_seg * g.lda;
  if (g.c_seg_stride == 0) g.c_seg_stride = (int64_t)g.rows_per_seg * g.ldc;
  if (g.r_seg_stride == 0) g.r_seg_stride = (int64_t)g.rows_per_seg * g.ldr;
  if (ext.c2_seg_stride == 0) ext.c2_seg_stride = (int64_t)g.rows_per_seg * ext.ldc2;
  SOPRO_CHECK_ARG(g.M > 0 && g.N > 0 && g.K > 0, "M, N, K must be positive");
  SOPRO_CHECK_ARG((g.K & 3) == 0, "K must be a multiple of 4");
  SOPRO_CHECK_ARG(g.rows_per_seg > 0, "rows_per_seg must be positive");
  SOPRO_CHECK_ARG(g.A && packed_w && (g.C || ext.c_mode == 5), "A, packed_w, C must be non-NULL");
  SOPRO_CHECK_ARG(aligned16(g.A) && aligned16(packed_w), "A and packed_w must be 16-byte aligned");
  SOPRO_CHECK_ARG((g.lda & 3) == 0 && (g.a_seg_stride & 3) == 0, "lda, a_seg_stride must be multiples of 4");
  SOPRO_CHECK_ARG(g.epilogue != SOPRO_EPI_RES || g.R != nullptr, "EPI_RES needs R");
  SOPRO_CHECK_ARG(g.prologue != SOPRO_PRO_ADDVEC || (g.pro_vec && aligned16(g.pro_vec)), "PRO_ADDVEC needs an aligned pro_vec");
  SOPRO_CHECK_ARG(g.epilogue != SOPRO_EPI_GLU || (g.N % 64) == 0, "EPI_GLU needs N % 64 == 0 (packed value/gate blocks)");
  SOPRO_CHECK_ARG(ext.ksplit >= 0 && ext.ksplit <= 64, "ksplit must be in 0..64");
  SOPRO_CHECK_ARG(ext.ksplit <= 1 || g.dbg == nullptr, "the clock-stamp probe is for unsplit launches");
  return 0;
}

}  // namespace

static int g_group_m = 0;
extern "C" int sopro_gemm_set_group_m(int g) {
  g_group_m = g > 1 ? g : 0;
  return 0;
}

// 128x128 tiles on EIGHT waves (2 x 4 waves of 64 x 32: 104 registers, so two workgroups = four waves per SIMD share a CU; the
// four-wave form holds 198 and runs two per SIMD): twice the operand requests in flight per CU.
// Measured (profiles/r04_experiments.md section 6): alone, +2-12 % on the decoder's shapes with N >= 512 (-4-9 % for N <= 256);
// in the pipeline NOTHING - the refinement gets 0.2-0.3 ms per step faster and the generation partition next door as much slower.
// Kept as a developer override (tile override 7 / SOPRO_GEMM_W8=1) for the three-pass decoder path; same K loop per output
// element: bit-identical results.
static bool eight_waves(const sopro_gemm_args& g) {
  static const bool on = SOPRO_DEV_ENV("SOPRO_GEMM_W8") != nullptr && SOPRO_DEV_ENV("SOPRO_GEMM_W8")[0] == '1';
  return on && g.N >= 512 && g.M >= 1024;
}

#ifdef SOPRO_DEV_SWITCHES
// activation-stationary form for short K (gemm_astat.hip): measured 0.85-1.0x of the tile kernel - compiled into the developer build only
bool sopro_gemm_astat_takes(const sopro_gemm_args& g, const sopro_gemm_split_ext& ext);
int sopro_gemm_astat_bf16x3(const sopro_gemm_args& g, const void* packed_w, hipStream_t s);
#endif

static int g_tile_override = 0;  // developer probe: 9: the activation-stationary form (gemm_astat.hip) wherever it applies; 1: 128x128, 2: 256x128, 4: 128x64 (x6: 64x128), 5: 64x64, 7 / 8: 128x128 on eight waves (bf16x3)
extern "C" int sopro_gemm_bf16_set_tile_override(int cfg) {
  g_tile_override = cfg;
  return 0;
}

extern "C" int64_t sopro_packed_w_bytes(int32_t N, int32_t K, int32_t pieces) {
  if (N <= 0 || K <= 0 || pieces < 1 || pieces > 3) return 0;
  return (int64_t)((N + 31) / 32) * ((K + 31) / 32 * 2) * pieces * 64 * 16;
}

extern "C" int sopro_pack_w_bf16(const float* W, int64_t ldw, int32_t N, int32_t K, int32_t pieces, void* packed, void* stream) {
  SOPRO_CHECK_ARG(W && packed && N > 0 && K > 0 && ldw >= K, "bad pointers or sizes");
  SOPRO_CHECK_ARG(pieces >= 1 && pieces <= 3, "pieces must be 1 (bf16x1), 2 (bf16x3) or 3 (bf16x6)");
  SOPRO_CHECK_ARG(aligned16(packed), "packed must be 16-byte aligned");
  const int ksubs = (K + 31) / 32 * 2;
  const int64_t total = (int64_t)((N + 31) / 32) * ksubs * pieces * 64;
  const dim3 grid((unsigned)((total + 255) / 256));
  if (pieces == 1)
    hipLaunchKernelGGL(pack_w_kernel<1>, grid, dim3(256), 0, (hipStream_t)stream, W, ldw, N, K, reinterpret_cast<uint4*>(packed), ksubs, total);
  else if (pieces == 2)
    hipLaunchKernelGGL(pack_w_kernel<2>, grid, dim3(256), 0, (hipStream_t)stream, W, ldw, N, K, reinterpret_cast<uint4*>(packed), ksubs, total);
  else
    hipLaunchKernelGGL(pack_w_kernel<3>, grid, dim3(256), 0, (hipStream_t)stream, W, ldw, N, K, reinterpret_cast<uint4*>(packed), ksubs, total);
  SOPRO_LAUNCH_CHECK();
}

extern "C" int sopro_gemm_bf16x3(const sopro_gemm_args* a, const void* packed_w, const sopro_gemm_split_ext* x, void* stream) {
  SOPRO_CHECK_ARG(a != nullptr, "args is NULL");
  sopro_gemm_args g = *a;
  sopro_gemm_split_ext ext;
  memset(&ext, 0, sizeof(ext));
  if (x) ext = *x;
  if (ext.group_m == 0) ext.group_m = g_group_m;
  if (int rc = check_common(g, ext, packed_w)) return rc;
  if (int rc = check_ln(g, ext)) return rc;
  SOPRO_CHECK_ARG(g.prologue == SOPRO_PRO_NONE || g.prologue == SOPRO_PRO_ELU, "prologue must be NONE or ELU");
  SOPRO_CHECK_ARG(g.epilogue == SOPRO_EPI_NONE || g.epilogue == SOPRO_EPI_GELU || g.epilogue == SOPRO_EPI_RES || g.epilogue == SOPRO_EPI_ROPE,
                  "epilogue must be NONE, GELU, RES or ROPE");
  if (int rc = check_rope(g, ext)) return rc;
  SOPRO_CHECK_ARG(ext.a_format == 0 || ext.a_format == 1, "a_format must be 0 (fp32) or 1 (split form)");
  SOPRO_CHECK_ARG(!ext.rms_norm, "fused RMSNorm is a six-pass (sopro_gemm_bf16x6) feature");
  if (ext.a_format == 1) {
    SOPRO_CHECK_ARG(g.prologue == SOPRO_PRO_NONE, "split-form A carries its activation already");
    SOPRO_CHECK_ARG((g.K & 31) == 0 && (g.lda & 31) == 0 && (g.a_seg_stride & 31) == 0 && (reinterpret_cast<uintptr_t>(g.A) & 127u) == 0,
                    "split-form A: K, lda, a_seg_stride multiples of 32 and a 128-byte aligned base");
  }
  SOPRO_CHECK_ARG(ext.c_mode >= 0 && ext.c_mode <= 4, "c_mode must be 0..4");
  if (ext.c_mode != 0) {
    SOPRO_CHECK_ARG((g.N & 3) == 0, "activated output: N % 4 == 0");
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
  const uint4* wp = reinterpret_cast<const uint4*>(packed_w);
  const int ksubs = (g.K + 31) / 32 * 2;
  // The activation-stationary form for short K (gemm_astat.hip; bit-identical results) is a developer override (9): measured at
  // 0.85-1.0x of the tile kernel on every K <= 512 shape of the decoder / refinement (profiles/r05_experiments.md section 4)
#ifdef SOPRO_DEV_SWITCHES
  if (g_tile_override == 9 && sopro_gemm_astat_takes(g, ext)) return sopro_gemm_astat_bf16x3(g, packed_w, s);
#endif
  switch (g_tile_override) {
    case 1: return launch_cfg3<2, 2, 2, 2, 2>(g, wp, ksubs, ext, s);
    case 2: return launch_cfg3<2, 2, 2, 4, 2>(g, wp, ksubs, ext, s);
    case 4: return launch_cfg3<2, 2, 2, 2, 1>(g, wp, ksubs, ext, s);
    case 5: return launch_cfg3<2, 2, 2, 1, 1>(g, wp, ksubs, ext, s);
    case 7: return launch_cfg3<2, 2, 4, 2, 1>(g, wp, ksubs, ext, s);  // 128x128 on eight waves
#ifdef SOPRO_DEV_SWITCHES
    case 3: return launch_cfg3<2, 1, 4, 4, 1>(g, wp, ksubs, ext, s);  // 128x128, four waves of 128 x 32: no W fragment is requested twice
#endif
    default: break;
  }
  // few columns, or few rows (streaming chunks, batch 1: small tiles keep the split-K partial sums small): 64x64
  if (g.N <= 64 || g.M <= 64) return launch_cfg3<2, 2, 2, 1, 1>(g, wp, ksubs, ext, s);
  if (eight_waves(g)) return launch_cfg3<2, 2, 4, 2, 1>(g, wp, ksubs, ext, s);
  // developer A/B (SOPRO_GEMM_NARROW=1): 128x64 tiles for the N <= 512 contractions of a many-row pass (o / fc2 of the decoder transformer at
  // 25600 rows: 800 tiles of 128x128 on 384 slots of the 192-CU partition = a third round that is 8 % full).  Measured in the pipeline
  // (r05 call 22): decode 13.21 / 13.25 -> 13.38 / 13.37 ms per step - the narrower tile loses more than the fuller last round wins: no-go
  static const bool narrow = SOPRO_DEV_ENV("SOPRO_GEMM_NARROW") != nullptr && SOPRO_DEV_ENV("SOPRO_GEMM_NARROW")[0] == '1';
  if (narrow && g.N <= 512 && g.M >= 8192) return launch_cfg3<2, 2, 2, 2, 1>(g, wp, ksubs, ext, s);
  return launch_cfg3<2, 2, 2, 2, 2>(g, wp, ksubs, ext, s);
}

// bf16 mode: one pass.  Takes what either split entry point takes, except split-form operands (a_format 1, c_mode 1 / 2).
extern "C" int sopro_gemm_bf16x1(const sopro_gemm_args* a, const void* packed_w, const sopro_gemm_split_ext* x, void* stream) {
  SOPRO_CHECK_ARG(a != nullptr, "args is NULL");
  sopro_gemm_args g = *a;
  sopro_gemm_split_ext ext;
  memset(&ext, 0, sizeof(ext));
  if (x) ext = *x;
  if (ext.group_m == 0) ext.group_m = g_group_m;
  SOPRO_CHECK_ARG((ext.a_format == 0 || ext.a_format == 2) && (ext.c_mode == 0 || (ext.c_mode >= 3 && ext.c_mode <= 8)),
                  "bf16x1 reads fp32 rows (a_format 0) or bf16 rows (2) and writes fp32 rows (c_mode 0, 3, 4), arg-max partials (5) or bf16 rows (6, 7, 8)");
  if (ext.a_format == 2 || ext.c_mode >= 6) {
    SOPRO_CHECK_ARG(!ext.rms_norm && g.prologue == SOPRO_PRO_NONE && (g.epilogue == SOPRO_EPI_NONE || g.epilogue == SOPRO_EPI_RES),
                    "bf16 rows: no prologue / fused norm, epilogue NONE or RES (the activation is applied by the producer: c_mode 7 / 8)");
    SOPRO_CHECK_ARG(ext.a_format != 2 || ((g.lda & 3) == 0 && (g.a_seg_stride & 3) == 0), "bf16 A rows: lda, a_seg_stride multiples of 4 elements");
    SOPRO_CHECK_ARG(g.epilogue != SOPRO_EPI_RES || ext.c_mode >= 6, "EPI_RES with bf16 A writes bf16 rows (its skip operand R is bf16 then)");
  }
  SOPRO_CHECK_ARG(!ext.rms_norm || ((g.K & 31) == 0 && ext.ksplit <= 1 && g.prologue == SOPRO_PRO_NONE && ext.rms_eps > 0.f && ext.c_mode == 0),
                  "fused RMSNorm needs K % 32 == 0, no split-K, no prologue, eps > 0 and a plain output");
  if (int rc = check_common(g, ext, packed_w)) return rc;
  if (int rc = check_ln(g, ext)) return rc;
  if (ext.c_mode == 5) return launch_argmax<1>(g, reinterpret_cast<const uint4*>(packed_w), (g.K + 31) / 32 * 2, ext, reinterpret_cast<hipStream_t>(stream));
  if (ext.c_mode >= 6) {  // bf16 rows: C (and C2 for c_mode 8; R for EPI_RES) point at bf16 elements, strides count elements
    SOPRO_CHECK_ARG((g.N & 3) == 0 && g.C && (reinterpret_cast<uintptr_t>(g.C) & 7u) == 0 && (g.ldc & 3) == 0 && (g.c_seg_stride & 3) == 0 && g.ldc >= g.N,
                    "bf16 output rows must be 8-byte aligned (N, ld, seg stride multiples of 4)");
    SOPRO_CHECK_ARG(ext.c_mode != 8 || (ext.C2 && (reinterpret_cast<uintptr_t>(ext.C2) & 7u) == 0 && (ext.ldc2 & 3) == 0 && (ext.c2_seg_stride & 3) == 0 && ext.ldc2 >= g.N),
                    "c_mode 8: C2 (the activated bf16 copy) must be given, 8-byte aligned rows");
    SOPRO_CHECK_ARG(g.epilogue != SOPRO_EPI_RES || ((reinterpret_cast<uintptr_t>(g.R) & 7u) == 0 && (g.ldr & 3) == 0 && (g.r_seg_stride & 3) == 0),
                    "bf16 skip operand rows must be 8-byte aligned");
  } else if (ext.c_mode != 0) {
    const bool second = ext.c_mode == 4;
    float* d = second ? ext.C2 : g.C;
    const int64_t ldd = second ? ext.ldc2 : g.ldc, dseg = second ? ext.c2_seg_stride : g.c_seg_stride;
    SOPRO_CHECK_ARG((g.N & 3) == 0 && d && aligned16(d) && (ldd & 3) == 0 && (dseg & 3) == 0 && ldd >= g.N,
                    "activated output rows must be 16-byte aligned (N, ld, seg stride multiples of 4)");
  }
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const uint4* wp = reinterpret_cast<const uint4*>(packed_w);
  const int ksubs = (g.K + 31) / 32 * 2;
  // few output tiles (NAR's few-thousand-row shapes, streaming chunks): small tiles; the decoder's big shapes: 128x128
  const int64_t t128 = (int64_t)((g.M + 127) / 128) * ((g.N + 127) / 128);
  const bool small = g.N <= 64 || g.M <= 64 || t128 < 256;
  const bool nar_like = ext.rms_norm || g.epilogue == SOPRO_EPI_GLU || g.prologue == SOPRO_PRO_ADDVEC;
  if (nar_like) {
    SOPRO_CHECK_ARG(ext.c_mode == 0 && (g.prologue == SOPRO_PRO_NONE || g.prologue == SOPRO_PRO_ADDVEC), "GLU / RMSNorm / ADDVEC forms write plain fp32");
    if (g.epilogue == SOPRO_EPI_GLU) return small ? launch_cfg6<1, 2, 2, 1, 2>(g, wp, ksubs, ext, s) : launch_cfg6<1, 2, 2, 2, 2>(g, wp, ksubs, ext, s);
    return small ? launch_cfg6<1, 2, 2, 1, 1>(g, wp, ksubs, ext, s) : launch_cfg6<1, 2, 2, 2, 2>(g, wp, ksubs, ext, s);
  }
  SOPRO_CHECK_ARG(g.prologue == SOPRO_PRO_NONE || g.prologue == SOPRO_PRO_ELU, "prologue must be NONE, ELU or ADDVEC");
  SOPRO_CHECK_ARG(g.epilogue == SOPRO_EPI_NONE || g.epilogue == SOPRO_EPI_GELU || g.epilogue == SOPRO_EPI_RES || g.epilogue == SOPRO_EPI_ROPE,
                  "epilogue must be NONE, GELU, RES, GLU or ROPE");
  if (int rc = check_rope(g, ext)) return rc;
  return small ? launch_cfg3<1, 2, 2, 1, 1>(g, wp, ksubs, ext, s) : launch_cfg3<1, 2, 2, 2, 2>(g, wp, ksubs, ext, s);
}

extern "C" int sopro_gemm_bf16x6(const sopro_gemm_args* a, const void* packed_w, const sopro_gemm_split_ext* x, void* stream) {
  SOPRO_CHECK_ARG(a != nullptr, "args is NULL");
  sopro_gemm_args g = *a;
  sopro_gemm_split_ext ext;
  memset(&ext, 0, sizeof(ext));
  if (x) ext = *x;
  if (ext.group_m == 0) ext.group_m = g_group_m;
  SOPRO_CHECK_ARG(ext.a_format == 0 && (ext.c_mode == 0 || ext.c_mode == 5),
                  "the six-pass path reads fp32 rows and writes fp32 rows, or arg-max partials (c_mode 5)");
  SOPRO_CHECK_ARG(!ext.rms_norm || ((g.K & 31) == 0 && ext.ksplit <= 1 && g.prologue == SOPRO_PRO_NONE && ext.rms_eps > 0.f),
                  "fused RMSNorm needs K % 32 == 0, no split-K, no prologue and eps > 0");
  if (int rc = check_common(g, ext, packed_w)) return rc;
  if (int rc = check_ln(g, ext)) return rc;
  SOPRO_CHECK_ARG(g.prologue == SOPRO_PRO_NONE || g.prologue == SOPRO_PRO_ADDVEC, "prologue must be NONE or ADDVEC");
  SOPRO_CHECK_ARG(g.epilogue == SOPRO_EPI_NONE || g.epilogue == SOPRO_EPI_GELU || g.epilogue == SOPRO_EPI_RES || g.epilogue == SOPRO_EPI_GLU,
                  "epilogue must be NONE, GELU, RES or GLU");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const uint4* wp = reinterpret_cast<const uint4*>(packed_w);
  const int ksubs = (g.K + 31) / 32 * 2;
  if (ext.c_mode == 5) return launch_argmax<3>(g, wp, ksubs, ext, s);
  switch (g_tile_override) {
    case 1: return launch_cfg6<3, 2, 2, 2, 2>(g, wp, ksubs, ext, s);
    case 4: return launch_cfg6<3, 2, 2, 1, 2>(g, wp, ksubs, ext, s);
    case 5: if (g.epilogue != SOPRO_EPI_GLU) return launch_cfg6<3, 2, 2, 1, 1>(g, wp, ksubs, ext, s); break;
    default: break;
  }
  // measured on the NAR shapes (tools/gemm_x6_probe.py): a few thousand rows, K <= 1536 -> the small tiles win
  if (g.epilogue == SOPRO_EPI_GLU) return launch_cfg6<3, 2, 2, 1, 2>(g, wp, ksubs, ext, s);
  return launch_cfg6<3, 2, 2, 1, 1>(g, wp, ksubs, ext, s);
}

// ---- f16x3: the token paths at three passes (see A16_SCALE above)
extern "C" int sopro_pack_w_f16x2(const float* W, int64_t ldw, int32_t N, int32_t K, float wscale, void* packed, void* stream) {
  SOPRO_CHECK_ARG(W && packed && N > 0 && K > 0 && ldw >= K, "bad pointers or sizes");
  SOPRO_CHECK_ARG(aligned16(packed), "packed must be 16-byte aligned");
  int ex = 0;
  SOPRO_CHECK_ARG(wscale > 0.f && frexpf(wscale, &ex) == 0.5f, "wscale must be a power of two (the scaling has to be exact)");
  const int ksubs = (K + 31) / 32 * 2;
  const int64_t total = (int64_t)((N + 31) / 32) * ksubs * 2 * 64;
  hipLaunchKernelGGL((pack_w_kernel<2, true>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, W, ldw, N, K,
                     reinterpret_cast<uint4*>(packed), ksubs, total, wscale);
  SOPRO_LAUNCH_CHECK();
}

extern "C" float sopro_f16x3_a_scale(void) { return A16_SCALE; }

extern "C" int sopro_gemm_f16x3(const sopro_gemm_args* a, const void* packed_w, const sopro_gemm_split_ext* x, void* stream) {
  SOPRO_CHECK_ARG(a != nullptr && x != nullptr, "args / ext is NULL (ext carries acc_scale)");
  sopro_gemm_args g = *a;
  sopro_gemm_split_ext ext = *x;
  if (ext.group_m == 0) ext.group_m = g_group_m;
  SOPRO_CHECK_ARG(ext.acc_scale > 0.f, "acc_scale = 1 / (sopro_f16x3_a_scale() * the weight's pack scale) must be set");
  SOPRO_CHECK_ARG(ext.a_format == 0 && (ext.c_mode == 0 || ext.c_mode == 5),
                  "the f16 three-pass path reads fp32 rows and writes fp32 rows, or arg-max partials (c_mode 5)");
  SOPRO_CHECK_ARG(!ext.rms_norm || ((g.K & 31) == 0 && ext.ksplit <= 1 && g.prologue == SOPRO_PRO_NONE && ext.rms_eps > 0.f),
                  "fused RMSNorm needs K % 32 == 0, no split-K, no prologue and eps > 0");
  if (int rc = check_common(g, ext, packed_w)) return rc;
  if (int rc = check_ln(g, ext)) return rc;
  SOPRO_CHECK_ARG(g.prologue == SOPRO_PRO_NONE || g.prologue == SOPRO_PRO_ADDVEC, "prologue must be NONE or ADDVEC");
  SOPRO_CHECK_ARG(g.epilogue == SOPRO_EPI_NONE || g.epilogue == SOPRO_EPI_GELU || g.epilogue == SOPRO_EPI_RES || g.epilogue == SOPRO_EPI_GLU,
                  "epilogue must be NONE, GELU, RES or GLU");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const uint4* wp = reinterpret_cast<const uint4*>(packed_w);
  const int ksubs = (g.K + 31) / 32 * 2;
  if (ext.c_mode == 5) return launch_argmax<2, true>(g, wp, ksubs, ext, s);
  static const int env_tile = SOPRO_DEV_ENV("SOPRO_F16X3_TILE") ? atoi(SOPRO_DEV_ENV("SOPRO_F16X3_TILE")) : 0;  // developer A/B of this family alone
  switch (g_tile_override ? g_tile_override : env_tile) {
    case 1: return launch_cfg6<2, 2, 2, 2, 2, true>(g, wp, ksubs, ext, s);
#ifdef SOPRO_DEV_SWITCHES
    case 3: return launch_cfg6<2, 1, 4, 4, 1, true>(g, wp, ksubs, ext, s);
#endif
    case 4: return launch_cfg6<2, 2, 2, 1, 2, true>(g, wp, ksubs, ext, s);
    case 5: return g.epilogue == SOPRO_EPI_GLU ? launch_cfg6<2, 2, 2, 1, 2, true>(g, wp, ksubs, ext, s) : launch_cfg6<2, 2, 2, 1, 1, true>(g, wp, ksubs, ext, s);
    default: break;
  }
  // many rows (64-utterance passes of the pipeline: 12800 rows): 128x128 tiles - NAR phase 5.8 -> 5.5-5.7 ms per step in the
  // pipeline (profiles/r03_experiments.md); a few thousand rows and below: the small tiles (tools/gemm_x6_probe.py).  Results
  // do not depend on the tile shape (every output element runs the same K loop)
  if (g.M >= 8192 && env_tile == 0) return launch_cfg6<2, 2, 2, 2, 2, true>(g, wp, ksubs, ext, s);
  if (g.epilogue == SOPRO_EPI_GLU) return launch_cfg6<2, 2, 2, 1, 2, true>(g, wp, ksubs, ext, s);
  return launch_cfg6<2, 2, 2, 1, 1, true>(g, wp, ksubs, ext, s);
}
