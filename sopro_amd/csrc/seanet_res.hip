// Fused MimiResnetBlock of the SEANet decoder at the 128-channel / 6 kHz level (HF:modeling_mimi.py:408-447):
//     out = ELU( h + Conv1d(64->128, k=1)( ELU( Conv1d(128->64, k=3)( ELU(h) ) ) ) )
// (the trailing ELU belongs to the next layer, the last transposed convolution, which reads nothing else).  h is the raw
// output of the third transposed convolution: 1.57 GB for 32 x 200 frames, the largest activation the generic GEMM flow
// still moved several times (raw + activated copy written by the producer, the activated copy read by the k=3 conv, the
// 64-channel intermediate written and read back, the raw copy re-read as the skip operand, the result written: ~11 GB).
// Fused, h is read once (+ an L2-resident re-read for the skip operand) and the activated result is written once.
//
// WEIGHT-STATIONARY: both weight matrices (128 KB) live in registers as split-bf16 MFMA B fragments for the whole life of
// a workgroup, spread over its four waves, and the workgroup walks `tiles` consecutive 64-row tiles of one utterance:
//   conv k=3 (K = 384): wave w owns output columns 32*(w&1).. and the K half (w>>1) -> 12 substeps x (hi, lo) = 96 registers;
//                       the two K halves of a 32x32 block are added through LDS in a fixed order (deterministic);
//   conv k=1 (K = 64):  wave w owns output columns 32*w..                            ->  4 substeps x (hi, lo) = 32 registers.
// Same arithmetic class as the decoder's other contractions (gemm_bf16s.hip, NPL = 2): operands x = hi + lo in bf16,
// lo*hi + hi*lo + hi*hi on v_mfma_f32_32x32x16_bf16 with fp32 accumulation.  ELU(h) is activated and split once per
// element while the tile is staged (LDS row = [128 hi | 128 lo] + 16 B pad = 528 B: the k=3 window of a row is three
// consecutive LDS rows and the 16-lane ds_read_b128 fragment reads are conflict free); the intermediate stays in LDS.
#include "common.h"

namespace {

constexpr int RC = 128;           // channels
constexpr int RH = 64;            // hidden channels of the block
constexpr int RTO = 64;           // output rows per tile
constexpr int RHR = RTO + 2;      // staged rows: samples s0-2 .. s0+RTO-1
constexpr int REROW = 2 * RC * 2 + 16;   // 528
constexpr int RYROW = 2 * RH * 2 + 16;   // 272

typedef __bf16 rbf16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ rbf16x8 rfrag(const uint4& v) { return *reinterpret_cast<const rbf16x8*>(&v); }

__device__ __forceinline__ void rsplit8(const float* __restrict__ p, uint4& hi, uint4& lo) {
  const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
  split2_bf16(a.x, a.y, hi.x, lo.x);
  split2_bf16(a.z, a.w, hi.y, lo.y);
  split2_bf16(b.x, b.y, hi.z, lo.z);
  split2_bf16(b.z, b.w, hi.w, lo.w);
}

// Round 6: the NEXT tile's raw rows are brought into LDS by LDS-DMA (global_load_lds_dwordx4) while the current tile's second half runs:
// the split-ELU tile `es` is dead from the barrier behind the first convolution until the next tile is staged, so the raw fp32 rows of
// tile k + 1 (66 rows x 512 B, lane-linear) land there - no registers, no extra LDS - and the staging pass reads them from LDS instead of
// waiting out an HBM round trip per tile (two workgroups per CU were all that hid it: 10.4 us per tile and workgroup, of which the
// matrix cores were busy 1.3).  One LDS array (several __shared__ objects make the compiler drain the DMA queue in front of unrelated
// LDS reads).  Same arithmetic, same order: bit-identical results.
typedef __attribute__((address_space(3))) void r_lds_void;
typedef const __attribute__((address_space(1))) void r_glb_void;
// 16 bytes per lane -> LDS (wave-uniform) lds_off + lane * 16.  By inline asm: the builtin form makes the compiler put `s_waitcnt
// vmcnt(0)` in front of the next LDS read (it cannot tell this kernel's LDS regions apart), i.e. wait the prefetch out where it was
// issued; like this the compiler does not know of the transfer and every wait for it is written by hand (M0 = the destination base, saved
// and restored inside the statement: it is compiler-reserved)
__device__ __forceinline__ void rdma16(const void* g, unsigned lds_off) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(g), "s"(lds_off)
               : "memory");
}
constexpr int RES_ES = RHR * REROW, RES_YS = RTO * RYROW, RES_RED = 4 * 16 * 64 * 4;
constexpr int RES_LDS = RES_ES + RES_YS + RES_RED;
static_assert(RES_ES % 16 == 0 && RES_YS % 16 == 0 && RHR * RC * 4 <= RES_ES, "the raw image of a tile fits the split tile's place");

template <int PASSES>
__global__ __launch_bounds__(256, 2) void seanet_res128_kernel(const float* __restrict__ h, int64_t h_seg_stride,
                                                               const float* __restrict__ w1, const float* __restrict__ b1,
                                                               const float* __restrict__ w2, const float* __restrict__ b2,
                                                               float* __restrict__ out, int64_t out_seg_stride, int T, int tiles) {
  // (DYNAMIC LDS: behind a static __shared__ object the compiler puts `s_waitcnt vmcnt(0)` in front of LDS reads that follow an
  // LDS-DMA - it cannot tell the regions apart - which would wait the prefetch out where it was issued)
  extern __shared__ __attribute__((aligned(1024))) unsigned char lds_all[];
  unsigned char* const es = lds_all;                                   // split ELU(h); between tiles: the next tile's raw rows
  unsigned char* const ys = lds_all + RES_ES;                          // split ELU(intermediate)
  float* const red = reinterpret_cast<float*>(lds_all + RES_ES + RES_YS);  // K-half exchange of the first convolution
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = blockIdx.y;
  const bool big = (int64_t)gridDim.y * T * 128 * 4 >= SOPRO_BIG_BYTES;  // a large output: stored with the non-temporal hint (common.h)
  const int frow = lane & 31, fg = lane >> 5;
  const int nt1 = wave & 1, kh = wave >> 1;
  const float* hb = h + (int64_t)b * h_seg_stride;   // 2 zero rows, then T rows of RC floats
  float* ob = out + (int64_t)b * out_seg_stride;     // same layout

  // ---- weights as (hi, lo) B fragments, once per workgroup: n = lane & 31, k = 16 * substep + 8 * (lane >> 5) .. + 7
  uint4 w1h[12], w1l[12];
#pragma unroll
  for (int s = 0; s < 12; ++s) rsplit8(w1 + (int64_t)(nt1 * 32 + frow) * (3 * RC) + (kh * 12 + s) * 16 + fg * 8, w1h[s], w1l[s]);
  uint4 w2h[4], w2l[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) rsplit8(w2 + (int64_t)(wave * 32 + frow) * RH + s * 16 + fg * 8, w2h[s], w2l[s]);
  const float b1v = b1[nt1 * 32 + frow];
  const float b2v = b2[wave * 32 + frow];

  // tile request: rows s0-2 .. s0+RTO-1 (padded rows s0 .. s0+RTO+1) as 33 DMA instructions of two rows each (wave w: 2 rows per
  // instruction, instructions w, w + 4, ..); rows past the end are redirected to padded row 0, a zero row: branch-free
  const unsigned es_off = (unsigned)(uintptr_t)es;
  auto request = [&](int s0) {
#pragma unroll
    for (int q = 0; q < 9; ++q) {
      // instruction of the tile: rows 2 i, 2 i + 1 (waves 1-3 repeat instruction 32 as their ninth: the same bytes to the same place -
      // one straight-line sequence for every wave)
      const int i = min(wave + 4 * q, RHR / 2 - 1);
      const int r = 2 * i + (lane >> 5);
      const int p = s0 + r;
      const int pc = p < T + 2 ? p : 0;
      rdma16(hb + (int64_t)pc * RC + (lane & 31) * 4, es_off + i * 1024);
      __builtin_amdgcn_sched_barrier(0);  // (one source address at a time: nine 64-bit pairs made ahead spill)
    }
  };
  // barrier that does NOT drain the DMA queue (a __syncthreads() with an LDS-DMA in flight waits vmcnt(0) first): LDS traffic of this
  // wave retired, then the bare s_barrier
  auto lds_barrier = [&]() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  };

  const int it0_s0 = (int)blockIdx.x * tiles * RTO;
  if (it0_s0 < T) request(it0_s0);
  for (int it = 0; it < tiles; ++it) {
    const int s0 = ((int)blockIdx.x * tiles + it) * RTO;
    if (s0 >= T) break;              // uniform over the workgroup

    // ---- the tile's raw rows have landed (requested behind the previous tile's first convolution): read, then ELU + split in place
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    float4 v[9];
    {
      const unsigned char* rp = es + tid * 16;  // float4 index tid + 256 q (32 per row): one address, immediate offsets
#pragma unroll
      for (int q = 0; q < 8; ++q) v[q] = *reinterpret_cast<const float4*>(rp + q * 4096);
      v[8] = *reinterpret_cast<const float4*>(es + (RHR * 32 - 64 + lane) * 16);  // rows 64, 65: wave 0's ninth piece (the others read and drop it)
    }
    // row tile 0's skip operand (raw h in the accumulator layout of the second convolution) straight from the raw image: rows the DMA
    // brought in are not asked for a second time (the re-read used to hit the L2; with a tile in flight ahead it mostly did not any more)
    float skip0[16];
#pragma unroll
    for (int r = 0; r < 16; ++r)
      skip0[r] = *reinterpret_cast<const float*>(es + ((r & 3) + 8 * (r >> 2) + 4 * fg + 2) * (RC * 4) + (wave * 32 + frow) * 4);
    __syncthreads();                 // every thread holds its pieces before the split image overwrites the raw one
#pragma unroll
    for (int q = 0; q < 9; ++q) {
      const int idx = tid + q * 256;
      const int r = idx >> 5, c4 = idx & 31;
      if (r < RHR) {
        uint2 hi, lo;
        split2_bf16(eluf_(v[q].x), eluf_(v[q].y), hi.x, lo.x);
        split2_bf16(eluf_(v[q].z), eluf_(v[q].w), hi.y, lo.y);
        *reinterpret_cast<uint2*>(es + r * REROW + c4 * 8) = hi;
        if (PASSES == 3) *reinterpret_cast<uint2*>(es + r * REROW + 2 * RC + c4 * 8) = lo;
      }
    }
    __syncthreads();

    // ---- conv k=3, 128 -> 64: intermediate row m (sample s0+m) reads staged rows m, m+1, m+2.
    // K index = tap * 128 + channel; substep s covers tap s / 8, channels 16 * (s % 8) .. + 15
    f32x16 acc[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mt][r] = 0.f;
    }
    {
      // both row tiles advance together (two independent accumulator chains) and the fragment reads run DEPTH substeps ahead
      // of the MFMAs that consume them (left to itself the compiler emits read -> wait -> MFMA per substep: every LDS latency
      // exposed); the scheduling barriers pin that order
      constexpr int DEPTH = 2;
      const unsigned char* a = es + frow * REROW + fg * 16;
      auto fptr = [&](int s, int mt) { const int sg = kh * 12 + s; return a + mt * 32 * REROW + (sg >> 3) * REROW + (sg & 7) * 32; };
      uint4 ah[DEPTH][2], al[DEPTH][2];
#pragma unroll
      for (int s = 0; s < DEPTH; ++s)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
          ah[s][mt] = *reinterpret_cast<const uint4*>(fptr(s, mt));
          al[s][mt] = PASSES == 3 ? *reinterpret_cast<const uint4*>(fptr(s, mt) + 2 * RC) : ah[s][mt];
        }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int s = 0; s < 12; ++s) {
        uint4 ch[2], cl[2];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) { ch[mt] = ah[s % DEPTH][mt]; cl[mt] = al[s % DEPTH][mt]; }
        if (PASSES == 3) {
#pragma unroll
          for (int mt = 0; mt < 2; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rfrag(cl[mt]), rfrag(w1h[s]), acc[mt], 0, 0, 0);
#pragma unroll
          for (int mt = 0; mt < 2; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rfrag(ch[mt]), rfrag(w1l[s]), acc[mt], 0, 0, 0);
        }
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rfrag(ch[mt]), rfrag(w1h[s]), acc[mt], 0, 0, 0);
        if (s + DEPTH < 12) {
#pragma unroll
          for (int mt = 0; mt < 2; ++mt) {
            ah[s % DEPTH][mt] = *reinterpret_cast<const uint4*>(fptr(s + DEPTH, mt));
            al[s % DEPTH][mt] = PASSES == 3 ? *reinterpret_cast<const uint4*>(fptr(s + DEPTH, mt) + 2 * RC) : ah[s % DEPTH][mt];
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    // the two K halves of a block meet through LDS: wave (nt, kh) keeps row tile kh and hands row tile 1-kh to wave (nt, 1-kh)
#pragma unroll
    for (int r = 0; r < 16; ++r) red[(wave * 16 + r) * 64 + lane] = kh ? acc[0][r] : acc[1][r];
    __syncthreads();
    // Addresses of the tile's output / skip elements: (utterance base, wave-uniform) + a 32-bit byte offset = this lane's place in the
    // tile's first row group + a compile-time row term - no 64-bit arithmetic, and for whole tiles (all but an utterance's last) no
    // per-element bounds test either (the per-store compare + branch + address pair were ~10 instructions per element)
    const bool whole = s0 + RTO <= T;  // uniform
    const unsigned lane_off = (unsigned)(((s0 + 2 + 4 * fg) * RC + wave * 32 + frow) * 4);
    auto row_off = [&](int mt, int r) { return lane_off + (unsigned)((mt * 32 + (r & 3) + 8 * (r >> 2)) * RC * 4); };
    auto row_ok = [&](int mt, int r) { return whole || s0 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * fg < T; };
    const unsigned last_off = (unsigned)(((T + 1) * RC + wave * 32 + frow) * 4);  // rows past the end are never stored: read the last padded row
    // ---- es is dead until the next tile is staged: its raw rows start their way now, under the exchange and the second convolution
    // (also past the last tile - rows past the end read the zero row; the loop top / the kernel's end wait for them)
    request(s0 + RTO);
    {
      const int mt = kh;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float mine = kh ? acc[1][r] : acc[0][r];
        const float other = red[((wave ^ 2) * 16 + r) * 64 + lane];
        const float v = (kh ? other + mine : mine + other) + b1v;   // K half 0 first, whichever wave adds
        const int mr = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * fg;
        unsigned hi, lo;
        split2_bf16(eluf_(v), 0.f, hi, lo);
        *reinterpret_cast<unsigned short*>(ys + mr * RYROW + (nt1 * 32 + frow) * 2) = (unsigned short)(hi & 0xffffu);
        if (PASSES == 3) *reinterpret_cast<unsigned short*>(ys + mr * RYROW + 2 * RH + (nt1 * 32 + frow) * 2) = (unsigned short)(lo & 0xffffu);
      }
    }
    lds_barrier();

    // ---- conv k=1, 64 -> 128 (wave w: output columns 32w ..) + skip, ELU, store
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      f32x16 acc2;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc2[r] = 0.f;
      const unsigned char* a = ys + (mt * 32 + frow) * RYROW + fg * 16;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const uint4 ah = *reinterpret_cast<const uint4*>(a + s * 32);
        if (PASSES == 3) {
          const uint4 al = *reinterpret_cast<const uint4*>(a + 2 * RH + s * 32);
          acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rfrag(al), rfrag(w2h[s]), acc2, 0, 0, 0);
          acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rfrag(ah), rfrag(w2l[s]), acc2, 0, 0, 0);
        }
        acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rfrag(ah), rfrag(w2h[s]), acc2, 0, 0, 0);
      }
      float skip[16];
      if (mt == 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) skip[r] = skip0[r];
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r)
          skip[r] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(hb) + (row_ok(1, r) ? row_off(1, r) : last_off));
      }
      char* const obp = reinterpret_cast<char*>(ob);
      if (whole) {
#pragma unroll
        for (int r = 0; r < 16; ++r) bulk_store1(reinterpret_cast<float*>(obp + row_off(mt, r)), eluf_(skip[r] + acc2[r] + b2v), big);
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (row_ok(mt, r)) bulk_store1(reinterpret_cast<float*>(obp + row_off(mt, r)), eluf_(skip[r] + acc2[r] + b2v), big);
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the DMA instructions of the tile behind the last one: nothing may land in LDS after the workgroup has gone)
}

// HB (round 4: the bf16 mode's activation flow): h and out are bf16 rows (256 bytes per row) - the 1.57 GB level is read and written
// at half the bytes; one MFMA pass on the rounded operands.  ELU is applied to the widened values and the result is rounded again
// while the tile is staged (LDS row = 128 bf16 + 16 B pad = 272 B); the skip operand is the raw bf16 h widened to fp32; the sum is
// accumulated in fp32 and rounded once when it is stored (two neighbouring columns per lane: 4-byte stores).
constexpr int REROW_H = RC * 2 + 16;   // 272
constexpr int RYROW_H = RH * 2 + 16;   // 144
__device__ __forceinline__ unsigned relu2_bf16(unsigned pk) {  // two packed bf16 -> ELU in fp32 -> two packed bf16
  unsigned o, lo_;
  split2_bf16(eluf_(__uint_as_float(pk << 16)), eluf_(__uint_as_float(pk & 0xffff0000u)), o, lo_);
  return o;
}
__global__ __launch_bounds__(256, 2) void seanet_res128_hb_kernel(const unsigned short* __restrict__ h, int64_t h_seg_stride,
                                                                  const float* __restrict__ w1, const float* __restrict__ b1,
                                                                  const float* __restrict__ w2, const float* __restrict__ b2,
                                                                  unsigned short* __restrict__ out, int64_t out_seg_stride, int T, int tiles) {
  __shared__ __attribute__((aligned(16))) unsigned char es[RHR * REROW_H];   // bf16 ELU(h)
  __shared__ __attribute__((aligned(16))) unsigned char ys[RTO * RYROW_H];   // bf16 ELU(intermediate)
  __shared__ __attribute__((aligned(16))) float red[4 * 16 * 64];            // K-half exchange of the first convolution
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.y;
  const bool big = (int64_t)gridDim.y * T * 128 * 2 >= SOPRO_BIG_BYTES;
  const int frow = lane & 31, fg = lane >> 5;
  const int nt1 = wave & 1, kh = wave >> 1;
  const unsigned short* hb = h + (int64_t)b * h_seg_stride;   // 2 zero rows, then T rows of RC bf16
  unsigned short* ob = out + (int64_t)b * out_seg_stride;

  uint4 w1h[12];
#pragma unroll
  for (int s = 0; s < 12; ++s) { uint4 lo; rsplit8(w1 + (int64_t)(nt1 * 32 + frow) * (3 * RC) + (kh * 12 + s) * 16 + fg * 8, w1h[s], lo); }
  uint4 w2h[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) { uint4 lo; rsplit8(w2 + (int64_t)(wave * 32 + frow) * RH + s * 16 + fg * 8, w2h[s], lo); }
  const float b1v = b1[nt1 * 32 + frow];
  const float b2v = b2[wave * 32 + frow];

  uint4 v[5];  // 66 rows x 16 pieces of 16 bytes = 1056 pieces over 256 threads
  auto request = [&](int s0) {
#pragma unroll
    for (int q = 0; q < 5; ++q) {
      const int idx = tid + q * 256;
      const int r = idx >> 4, c8 = idx & 15;
      const int p = s0 + r;
      const int pc = (r < RHR && p < T + 2) ? p : 0;
      v[q] = *reinterpret_cast<const uint4*>(hb + (int64_t)pc * RC + c8 * 8);
    }
  };

  for (int it = 0; it < tiles; ++it) {
    const int s0 = ((int)blockIdx.x * tiles + it) * RTO;
    if (s0 >= T) break;
    request(s0);
#pragma unroll
    for (int q = 0; q < 5; ++q) {
      const int idx = tid + q * 256;
      const int r = idx >> 4, c8 = idx & 15;
      if (r < RHR) *reinterpret_cast<uint4*>(es + r * REROW_H + c8 * 16) = make_uint4(relu2_bf16(v[q].x), relu2_bf16(v[q].y), relu2_bf16(v[q].z), relu2_bf16(v[q].w));
    }
    __syncthreads();

    f32x16 acc[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mt][r] = 0.f;
    }
    {
      constexpr int DEPTH = 2;
      const unsigned char* a = es + frow * REROW_H + fg * 16;
      auto fptr = [&](int s, int mt) { const int sg = kh * 12 + s; return a + mt * 32 * REROW_H + (sg >> 3) * REROW_H + (sg & 7) * 32; };
      uint4 ah[DEPTH][2];
#pragma unroll
      for (int s = 0; s < DEPTH; ++s)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) ah[s][mt] = *reinterpret_cast<const uint4*>(fptr(s, mt));
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int s = 0; s < 12; ++s) {
        uint4 ch[2];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) ch[mt] = ah[s % DEPTH][mt];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rfrag(ch[mt]), rfrag(w1h[s]), acc[mt], 0, 0, 0);
        if (s + DEPTH < 12) {
#pragma unroll
          for (int mt = 0; mt < 2; ++mt) ah[s % DEPTH][mt] = *reinterpret_cast<const uint4*>(fptr(s + DEPTH, mt));
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) red[(wave * 16 + r) * 64 + lane] = kh ? acc[0][r] : acc[1][r];
    __syncthreads();
    {
      const int mt = kh;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float mine = kh ? acc[1][r] : acc[0][r];
        const float other = red[((wave ^ 2) * 16 + r) * 64 + lane];
        const float v1 = (kh ? other + mine : mine + other) + b1v;   // K half 0 first, whichever wave adds
        const int mr = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * fg;
        unsigned hi, lo;
        split2_bf16(eluf_(v1), 0.f, hi, lo);
        *reinterpret_cast<unsigned short*>(ys + mr * RYROW_H + (nt1 * 32 + frow) * 2) = (unsigned short)(hi & 0xffffu);
      }
    }
    __syncthreads();

    // ---- conv k=1, 64 -> 128 (wave w: output columns 32w ..) + skip, ELU, store (column pairs: see seanet_up128_hb_kernel)
    const bool odd = (lane & 1) != 0;
    const int colp = wave * 32 + (frow & ~1);
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      // skip operand: this lane's (row, column pair) of the raw tile, an L2-resident 4-byte re-read per output pair
      // (round 6: 32-bit byte offsets from the utterance's base + compile-time row terms, and no per-element bounds test for whole tiles:
      // the compare + branch + 64-bit address pair per element were a third of this epilogue's instructions)
      const bool whole = s0 + RTO <= T;  // uniform
      const unsigned lane_off = (unsigned)(((s0 + 2 + 4 * fg + (odd ? 1 : 0)) * RC + colp) * 2);
      auto row_off = [&](int r) { return lane_off + (unsigned)((mt * 32 + (r & 3) + 8 * (r >> 2)) * RC * 2); };
      auto row_ok = [&](int r) { return whole || s0 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * fg + (odd ? 1 : 0) < T; };
      const unsigned last_off = (unsigned)(((T + 1) * RC + colp) * 2);  // rows past the end are never stored: read the last padded row
      unsigned skip[8];
#pragma unroll
      for (int r = 0; r < 16; r += 2)
        skip[r >> 1] = *reinterpret_cast<const unsigned*>(reinterpret_cast<const char*>(hb) + (row_ok(r) ? row_off(r) : last_off));
      f32x16 acc2;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc2[r] = 0.f;
      const unsigned char* a = ys + (mt * 32 + frow) * RYROW_H + fg * 16;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const uint4 ah = *reinterpret_cast<const uint4*>(a + s * 32);
        acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rfrag(ah), rfrag(w2h[s]), acc2, 0, 0, 0);
      }
      unsigned pkv[8];
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        const float mine0 = acc2[r] + b2v, mine1 = acc2[r + 1] + b2v;
        // (the neighbour column's bias travels with its value: mine* already hold it)
        const float got = __shfl_xor(odd ? mine0 : mine1, 1, 64);
        const float c0 = odd ? got : mine0, c1 = odd ? mine1 : got;
        const unsigned sk = skip[r >> 1];
        unsigned pk, lo_;
        split2_bf16(eluf_(__uint_as_float(sk << 16) + c0), eluf_(__uint_as_float(sk & 0xffff0000u) + c1), pk, lo_);
        pkv[r >> 1] = pk;
      }
      if (whole) {  // straight-line stores for every tile but an utterance's last
#pragma unroll
        for (int r = 0; r < 16; r += 2) bulk_store_u1(reinterpret_cast<char*>(ob) + row_off(r), pkv[r >> 1], big);
      } else {
#pragma unroll
        for (int r = 0; r < 16; r += 2)
          if (row_ok(r)) bulk_store_u1(reinterpret_cast<char*>(ob) + row_off(r), pkv[r >> 1], big);
      }
    }
  }
}

}  // namespace

static int g_res_tiles = 0;  // developer probe / tests: tiles per workgroup, 0 = heuristic
extern "C" int sopro_seanet_res_set_tiles(int tiles) {
  g_res_tiles = tiles > 0 ? tiles : 0;
  return 0;
}

extern "C" int sopro_seanet_res128_p_f32(const float* h, int64_t h_seg_stride, const float* w1, const float* b1, const float* w2,
                                          const float* b2, float* out, int64_t out_seg_stride, int32_t B, int32_t T, int32_t passes, void* stream) {
  SOPRO_CHECK_ARG(passes == 1 || passes == 3, "passes must be 3 (three-pass split-bf16) or 1 (bf16 mode)");
  SOPRO_CHECK_ARG(h && w1 && b1 && w2 && b2 && out && B > 0 && T > 0, "bad pointers or sizes");
  SOPRO_CHECK_ARG(h != out, "the block is not computed in place (a tile reads two rows of its left neighbour)");
  SOPRO_CHECK_ARG(aligned16(h) && aligned16(w1) && aligned16(w2) && aligned16(out) && (h_seg_stride & 3) == 0 && (out_seg_stride & 3) == 0,
                  "alignment");
  SOPRO_CHECK_ARG(h_seg_stride >= (int64_t)(T + 2) * RC && out_seg_stride >= (int64_t)(T + 2) * RC, "segments hold 2 + T rows of 128 floats");
  const int ntile = (T + RTO - 1) / RTO;
  // the weight fragments cost ~1000 VALU instructions per wave: amortised over several tiles once every CU has work anyway
  const int64_t all = (int64_t)ntile * B;
  const int tiles = g_res_tiles > 0 ? g_res_tiles : (all >= 16 * 2048 ? 16 : (all >= 8 * 1024 ? 8 : (all >= 2048 ? 2 : 1)));
  dim3 grid((ntile + tiles - 1) / tiles, B);
  SOPRO_CHECK_ARG(T < (1 << 21), "T must stay below 2^21 rows per utterance (32-bit byte offsets of the skip operand)");
  if (passes == 3) {
    auto kern = seanet_res128_kernel<3>;
    SOPRO_SET_MAX_LDS_ONCE(kern, RES_LDS);
    hipLaunchKernelGGL(kern, grid, dim3(256), RES_LDS, (hipStream_t)stream, h, h_seg_stride, w1, b1, w2, b2, out, out_seg_stride, T, tiles);
  } else {  // the engine's bf16 mode: hi * hi only
    auto kern = seanet_res128_kernel<1>;
    SOPRO_SET_MAX_LDS_ONCE(kern, RES_LDS);
    hipLaunchKernelGGL(kern, grid, dim3(256), RES_LDS, (hipStream_t)stream, h, h_seg_stride, w1, b1, w2, b2, out, out_seg_stride, T, tiles);
  }
  SOPRO_LAUNCH_CHECK();
}

// bf16 rows in (raw h), bf16 rows out (the activated result): the bf16 mode's activation flow, one pass.  Strides count bf16 elements.
extern "C" int sopro_seanet_res128_bf16(const void* h, int64_t h_seg_stride, const float* w1, const float* b1, const float* w2,
                                         const float* b2, void* out, int64_t out_seg_stride, int32_t B, int32_t T, void* stream) {
  SOPRO_CHECK_ARG(h && w1 && b1 && w2 && b2 && out && B > 0 && T > 0, "bad pointers or sizes");
  SOPRO_CHECK_ARG(h != out, "the block is not computed in place (a tile reads two rows of its left neighbour)");
  SOPRO_CHECK_ARG(aligned16(h) && aligned16(w1) && aligned16(w2) && (reinterpret_cast<uintptr_t>(out) & 3u) == 0 && (h_seg_stride & 7) == 0 && (out_seg_stride & 1) == 0,
                  "alignment (h rows in 16-byte pieces, out in 4-byte pairs)");
  SOPRO_CHECK_ARG(h_seg_stride >= (int64_t)(T + 2) * RC && out_seg_stride >= (int64_t)(T + 2) * RC, "segments hold 2 + T rows of 128 bf16");
  const int ntile = (T + RTO - 1) / RTO;
  const int64_t all = (int64_t)ntile * B;
  const int tiles = g_res_tiles > 0 ? g_res_tiles : (all >= 16 * 2048 ? 16 : (all >= 8 * 1024 ? 8 : (all >= 2048 ? 2 : 1)));
  dim3 grid((ntile + tiles - 1) / tiles, B);
  hipLaunchKernelGGL(seanet_res128_hb_kernel, grid, dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const unsigned short*>(h), h_seg_stride, w1, b1, w2, b2,
                     reinterpret_cast<unsigned short*>(out), out_seg_stride, T, tiles);
  SOPRO_LAUNCH_CHECK();
}

extern "C" int sopro_seanet_res128_f32(const float* h, int64_t h_seg_stride, const float* w1, const float* b1, const float* w2,
                                        const float* b2, float* out, int64_t out_seg_stride, int32_t B, int32_t T, void* stream) {
  return sopro_seanet_res128_p_f32(h, h_seg_stride, w1, b1, w2, b2, out, out_seg_stride, B, T, 3, stream);
}
