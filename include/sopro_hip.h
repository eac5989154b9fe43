/* sopro_hip.h -- C ABI of libsopro_hip.so, the MI355X (gfx950) kernel library behind the
 * Sopro TTS synthesize/stream hot path.
 *
 * The reference (samuel-vitorino/sopro) is pure Python on torch ATen: it has NO native
 * interface of its own.  Every entry point below therefore replaces an
 * ATen op *sequence* of the reference; the sequence is cited per function as
 * `src/sopro/...:lines` (paths relative to the reference checkout) or `HF:modeling_mimi.py`
 * (the third-party transformers Mimi implementation the reference calls into).
 *
 * Conventions
 *   - plain C types only: device pointers, sizes, a hipStream_t passed as void*;
 *   - every function returns 0 on success and a negative code on failure; the message is
 *     available from sopro_last_error() (thread-local); nothing throws across the ABI;
 *   - nothing allocates, frees or synchronises: the caller owns every buffer and every
 *     launch is enqueued on the caller's stream, so any call sequence can be recorded into
 *     a hipGraph with sopro_capture_begin/_end and replayed with sopro_graph_launch;
 *   - all activations are fp32, row-major, channels-last ([rows, channels]); token ids are
 *     int32; "seg" parameters describe a batch of equally long row segments whose start
 *     addresses are `seg_stride` elements apart (this is how per-utterance zero padding in
 *     front of causal convolutions is addressed without copies);
 *   - weights are fp32 [N, K] row-major (torch nn.Linear layout); convolution weights are
 *     repacked by the host into that layout (sopro_amd/pack.py).
 */
#ifndef SOPRO_HIP_H
#define SOPRO_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SOPRO_ABI_VERSION 41

/* ---- error handling / introspection ------------------------------------------------ */
const char* sopro_last_error(void);
int sopro_abi_version(void);
/* bit 0: a DEVELOPER build (make DEV=1: A/B switches read from the environment, measured no-go kernels compiled in) */
int sopro_build_flags(void);
/* Scheduling knob (no reference counterpart): every launch of the long-running split-bf16 contraction kernels requests at
 * least `bytes` of LDS.  Above 80 KiB that caps them at one workgroup per CU, which leaves wave slots, registers and LDS on
 * every CU for the short kernels of an AR frame generated at the same time on another stream - the alternative to carving
 * the chip up with CU masks (sopro_stream_create_cu_range).  0 = off.  Recorded graphs keep the value they were recorded with. */
int sopro_set_lds_floor(int bytes);
/* How this process's host threads wait for the current device (hipSetDeviceFlags): 0 = spin (runtime default), 1 = block on the
 * completion interrupt.  A scheduler with several launch threads per device (the lanes of sopro_amd/pipeline.py: each waits for
 * its AR poll / phase end most of the time) sets 1, which frees a core per waiting thread; a single latency-critical caller
 * (stream(), batch 1) keeps 0.  No reference counterpart. */
int sopro_set_host_wait(int blocking);
/* device facts: out[0]=CU count, out[1]=LDS bytes per block, out[2]=clock kHz, out[3]=gfx arch number */
int sopro_device_info(int device, int* out4);

/* ---- hipGraph helpers (new; the reference has no graph capture) ---------------------- */
int sopro_capture_begin(void* stream);
int sopro_capture_end(void* stream, void** graph_exec_out);
int sopro_graph_launch(void* graph_exec, void* stream);
/* n replays back to back in one call: a host whose launch thread shares an interpreter lock with other threads (the Python
 * host: ctypes drops the lock for the duration of a call) queues a whole chunk of AR frames without taking it n times. */
int sopro_graph_launch_n(void* graph_exec, void* stream, int32_t n);
int sopro_graph_destroy(void* graph_exec);
/* stream restricted to CUs [first_cu, first_cu + n_cus) (hipExtStreamCreateWithCUMask); destroy with sopro_stream_destroy */
int sopro_stream_create_cu_range(int first_cu, int n_cus, void** stream_out);
/* General form: one bit per CU (bit i of word i/32).  Measured on gfx950 (tools/micro/xcc_probe.hip): a mask with a single
 * bit set still runs workgroups on all eight XCDs, i.e. the bits are dealt to the XCDs and an XCD whose bits are all clear is
 * left unrestricted; a partition therefore has to keep some CUs on every XCD (contiguous ranges do: 1/8 of the range per
 * XCD), and whole-XCD partitions cannot be expressed. */
int sopro_stream_create_cu_mask(const uint32_t* mask, int32_t words, void** stream_out);
int sopro_stream_destroy(void* stream);
/* Page-locked host memory for the small device -> host mirrors a host loop polls (stop counters, slot snapshots), and the
 * copy behind the launches queued so far on `stream` (the caller records an event after it and waits for that event).  The
 * engine keeps these outside torch's pinned-memory cache: that cache remembers an event per stream a block was used on and
 * queries it at the next pinned allocation - after the CU-masked stream was destroyed, or while another thread records a
 * launch sequence (both seen as hipErrorCapturedEvent / hipErrorStreamCaptureUnsupported on ROCm 7.2). */
int sopro_host_alloc(int64_t bytes, void** out);   /* zero-filled */
int sopro_host_free(void* p);
int sopro_copy_to_host_async(void* dst_host, const void* src_dev, int64_t bytes, void* stream);

/* ---- dense contraction ------------------------------------------------------------------ */
enum { SOPRO_PRO_NONE = 0, SOPRO_PRO_ELU = 1, SOPRO_PRO_ADDVEC = 2 };
enum { SOPRO_EPI_NONE = 0, SOPRO_EPI_GELU = 1, SOPRO_EPI_GLU = 2, SOPRO_EPI_RES = 3, SOPRO_EPI_TANH = 4,
       SOPRO_EPI_GLU_DW = 5 /* skinny only */,
       SOPRO_EPI_ROPE = 6 /* sopro_gemm_bf16x3 / _bf16x1 only (round 5): rotate-half RoPE of the first rope_cols output columns in the
                           * epilogue - the q | k blocks of a fused qkv projection (HF:modeling_mimi.py:511-566) - see sopro_gemm_split_ext */ };

/* C[m, n] = epi( sum_k pro(A[m, k]) * W[n, k] + bias[n] ),  fp32 in / fp32 accumulate on
 * v_mfma_f32_32x32x2_f32 (bit-equal to an fmaf chain).
 * Row m lives at A + (m / rows_per_seg) * a_seg_stride + (m % rows_per_seg) * lda (a segment stride
 * of 0 means dense: rows_per_seg * ld; same for C and R); rows may
 * overlap (lda < K) which turns a causal Conv1d / ConvTranspose1d over channels-last data
 * into this contraction.  K must be a multiple of 4, all base pointers 16-byte aligned and
 * lda / a_seg_stride / ldw multiples of 4.
 *   EPI_GLU : W rows are packed per 64 as [32 value rows | 32 gate rows]; C has N/2 columns,
 *             C = value * sigmoid(gate)                       (src/sopro/nn/blocks.py:16-23)
 *   EPI_RES : C = R + scale[n] * (acc + bias)  (scale NULL = 1) (residual adds, LayerScale
 *             HF:modeling_mimi.py:495-507, tanh(gate) of src/sopro/nn/text.py:131)
 *   EPI_GELU: erf form (torch nn.GELU default)
 *   (sopro_gemm_f32 and sopro_gemm_bf16x6 evaluate GELU / the GLU gate with the library's erff / expf and an IEEE division; the f16
 *   three-pass and the one- / three-pass bf16 entry points - the throughput phases - use an Abramowitz-Stegun erf (<= 4.7e-7 absolute)
 *   and the hardware exponential / reciprocal for the gate (<= ~3 ulp): csrc/common.h gelu_fast, sigmoid_fast)
 * Replaces: every nn.Linear / F.linear of src/sopro/nn/{blocks,text,ref,nar,speaker}.py on the
 * full-sequence paths, HF MimiTransformer projections (HF:modeling_mimi.py:602-726) and,
 * through overlapping rows, MimiConv1d / MimiConvTranspose1d (HF:modeling_mimi.py:210-405). */
typedef struct sopro_gemm_args {
  const float* A; int64_t lda; int64_t a_seg_stride;
  const float* W; int64_t ldw;
  const float* bias;
  float* C; int64_t ldc; int64_t c_seg_stride;
  const float* R; int64_t ldr; int64_t r_seg_stride;
  const float* scale;
  const float* pro_vec; /* PRO_ADDVEC: a + pro_vec[k] */
  long long* dbg;       /* optional [workgroups][8] shader-clock stamps (profiling aid), NULL in production */
  int32_t M, N, K, rows_per_seg;
  int32_t prologue, epilogue;
} sopro_gemm_args;
int sopro_gemm_f32(const sopro_gemm_args* args, void* stream);
/* developer probe: force a tile shape (0 = heuristic, 1: 128x128, 2: 64x128, 3: 256x64, 4: 256x32, 5: 64x64) */
int sopro_gemm_set_tile_override(int cfg);

/* The same contraction at bf16 matrix-core rate for paths whose contract is a waveform tolerance (Mimi decoder:
 * HF:modeling_mimi.py MimiConv1d / MimiConvTranspose1d / MimiTransformerModel): operands are split x = hi + lo into two
 * bf16 halves (16 mantissa bits, round-to-nearest) and accumulated in fp32 as lo*hi + hi*lo + hi*hi with
 * v_mfma_f32_32x32x16_bf16.  `a->W`/`a->ldw` are ignored: the weight comes pre-split in MFMA fragment order from
 * sopro_pack_w_bf16 with pieces = 2 (device pointers).
 * Epilogues NONE / GELU / RES; prologues NONE / ELU.  `ext` (may be NULL = all zero) selects "split form" tensors:
 * every aligned group of 32 channels (128 bytes as fp32) is stored as [32 hi bf16 | 32 lo bf16], so an element stays in
 * its 128-byte line and every fp32 stride / offset keeps its meaning (rows must start on 128-byte boundaries).  A
 * producer writes ELU(x) in that form (c_mode 1), optionally next to the raw fp32 tensor (c_mode 2), and the consumer
 * (a_format 1) stages it with plain 16-byte copies: no activation or split work is left in its main loop. */
typedef struct sopro_gemm_split_ext {
  int32_t a_format; /* 0: A is fp32 rows; 1: split form (prologue must be NONE); 2 (sopro_gemm_bf16x1 only, round 4): A points at
                     * bf16 rows, lda / a_seg_stride count bf16 elements (multiples of 4) - the bf16 mode's activation flow */
  int32_t c_mode;   /* 0: fp32 to C; 1: ELU + split form to C; 2: fp32 to C and ELU + split form to C2;
                     * 3: ELU(C) as fp32 to C; 4: fp32 to C and ELU(C) as fp32 to C2; 5: arg-max partials (see sopro_argmax_partials_i32);
                     * sopro_gemm_bf16x1 only, epilogue NONE or RES: 6: C as bf16 rows; 7: ELU(C) as bf16 rows to C; 8: C as bf16 rows to C
                     * AND ELU(C) as bf16 rows to C2 - C / C2 / R (the skip operand of EPI_RES) then point at bf16 elements and
                     * ldc / ldc2 / ldr and the segment strides count bf16 elements (multiples of 4, 8-byte aligned rows) */
  float* C2; int64_t ldc2; int64_t c2_seg_stride; /* c_mode 2 / 4; strides in 4-byte units like ldc / c_seg_stride */
  /* Split-K for problems with too few output tiles to fill the chip (a few rows: streaming chunks, batch 1): `ksplit`
   * workgroups share a tile, each stores its raw accumulators into `ws`, the last to take the tile's ticket adds them in
   * slice order (deterministic) and runs the epilogue.  ws: >= ksplit * tiles * tile_elems * 4 bytes (tiles are at most
   * 128x128, at least 64x64); tickets: one zero-initialised int per tile, left zero by every launch.  Both belong to
   * ONE stream at a time. */
  /* Fused RMSNorm of the A rows (six-pass path; src/sopro/nn/blocks.py:26-37): out = rsqrt(mean_k(a^2) + rms_eps) * (a W'^T)
   * + bias, with the norm's weight vector folded into W' by the host.  K % 32 == 0, no split-K, no prologue. */
  int32_t rms_norm; float rms_eps;
  int32_t ksplit;      /* 0 / 1: off */
  int32_t n_tickets;
  float* ws; int64_t ws_bytes;
  int32_t* tickets;
  /* Tile walk: 0 / 1 = row-major (a workgroup index walks the column tiles of one row tile, then the next row tile);
   * g > 1 = grouped: g row tiles at a time are walked row-tile-fastest, so the workgroups that run at the same time on an XCD
   * (each XCD gets one contiguous range of the walk) share g row blocks of A and a few column blocks of W in its 4 MB L2
   * instead of streaming the whole W once per row tile.  Filled from sopro_gemm_set_group_m when 0. */
  int32_t group_m;
  float acc_scale;     /* sopro_gemm_f16x3 only: 1 / (sopro_f16x3_a_scale() * the weight's pack scale), applied to the accumulator */
  int32_t* range_events; /* sopro_gemm_f16x3 only, optional (round 5): a device word; every workgroup that had to SATURATE a staged activation
                          * at fp16's largest finite value (|a * scale| > 65504: the result is finite and wrong) adds 1.  Plain forms scale by
                          * sopro_f16x3_a_scale() (|a| <= 8188 is in range); fused-RMSNorm forms choose a power of two per row from the row's
                          * first 32 elements (256x headroom).  A caller that finds the word changed repeats the work on sopro_gemm_bf16x6. */
  /* SOPRO_EPI_ROPE: columns [0, rope_cols) are heads of rope_dh columns (a power of two <= 128 that divides the column tile); row m sits at
   * position rope_pos0 + m % rope_rows_per_seg; cos / sin tables [positions][rope_dh / 2] as sopro_rope_f32 takes them.  The epilogue holds
   * the whole tile in LDS, so a column's partner (+- rope_dh / 2) is at hand: out[e] = a c - b s, out[e + dh/2] = b c + a s - what
   * sopro_rope_f32 would do to C in a pass of its own (105 MB read + written per decoder layer at 64 x 200 frames). */
  const float* rope_cos; const float* rope_sin;
  int32_t rope_cols, rope_dh, rope_pos0, rope_rows_per_seg;
  /* Fused LayerNorm between two contractions of a residual stream (round 5; the Mimi decoder transformer's pre-norms,
   * HF:modeling_mimi.py:796-833): the PRODUCER of the stream (sopro_gemm_bf16x3 / _bf16x1, SOPRO_EPI_RES, c_mode 0, N % 64 == 0) also writes,
   * per output row m and 64-column group p, (mean, sum of squared deviations from that mean) to ln_stats_out[(m * (N / 64) + p) * 2 ..];
   * the CONSUMER (ln_stats != NULL, fp32 rows, K % 64 == 0, epilogues NONE / GELU / ROPE) combines a row's K / 64 pairs (Chan's update:
   * no E[x^2] - E[x]^2 cancellation) and stages (a - mean) * rsqrt(var + rms_eps) instead of a; the norm's weight is folded into W' and
   * W lnb into the bias by the host.  The normalised tensor is never written or re-read (105 MB each way per norm at 64 x 200 frames).
   * sopro_row_stats_f32 writes the same pairs for a stream no contraction produced. */
  const float* ln_stats;
  float* ln_stats_out;
} sopro_gemm_split_ext;
int sopro_gemm_bf16x3(const sopro_gemm_args* a, const void* packed_w, const sopro_gemm_split_ext* ext, void* stream);
/* LONG-K form of sopro_gemm_bf16x3 (round 6; csrc/gemm_8p.hip): the same three-pass contraction - bit-identical results - on 256 x 256
 * tiles with BOTH operands in split form in memory, staged by LDS-DMA behind counted waits (no split work, no staging registers in the
 * main loop).  For the SEANet decoder's K >= 1024 contractions (HF:modeling_mimi.py:350-405 MimiConvTranspose1d as a row-window
 * contraction; :408-447 the first residual block's k = 3 convolution): 1.3-1.4x the tile kernel there.  A must be split-form rows
 * (a_format 1: what a producer's c_mode 1 / 2 writes), the weight comes from sopro_pack_w_rows_bf16: [N][K / 32][32 hi | 32 lo] bf16,
 * sopro_packed_w_rows_bytes(N, K) bytes, 128-byte aligned.  N % 256 == 0, K % 32 == 0; epilogue NONE (+ bias); c_mode 0 / 1 / 2 / 4; no
 * split-K.  sopro_gemm_8p_takes: 1 when a call with these (args, ext) is one this form takes AND has enough tiles to pay (what the
 * stage sequences ask before they route a contraction here). */
int64_t sopro_packed_w_rows_bytes(int32_t N, int32_t K);
int sopro_pack_w_rows_bf16(const float* W, int64_t ldw, int32_t N, int32_t K, void* packed, void* stream);
int sopro_gemm_8p_takes(const sopro_gemm_args* a, const sopro_gemm_split_ext* ext);
int sopro_gemm_bf16x3_8p(const sopro_gemm_args* a, const void* w_rows, const sopro_gemm_split_ext* ext, void* stream);
/* Six-pass variant for token paths (NAR refinement, conditioning: src/sopro/nn/nar.py, blocks.py): operands split into
 * THREE bf16 pieces (24 mantissa bits), products p2*p0 + p0*p2 + p1*p1 + p1*p0 + p0*p1 + p0*p0; the dropped terms are
 * <= 2^-25 relative, i.e. the accuracy class of an fp32 fma chain, at 16/6 of the fp32-MFMA rate.  fp32 rows in and out;
 * prologues NONE / ADDVEC, epilogues NONE / GELU / RES / GLU.  The weight must have been packed with pieces = 3; of `ext`
 * (may be NULL) only the split-K fields apply. */
int sopro_gemm_bf16x6(const sopro_gemm_args* a, const void* packed_w, const sopro_gemm_split_ext* ext, void* stream);
/* One-pass variant = the engine's bf16 mode (BASELINE.json configs[1] / SURVEY 8d config 2: bf16 weights and activations,
 * fp32 accumulators, norms and softmax in fp32): both operands are rounded to bf16 once (W when it is packed with
 * pieces = 1, A while it is staged), one v_mfma_f32_32x32x16_bf16 pass per product.  Replaces, for the contractions of
 * src/sopro/nn/nar.py:89-116, blocks.py:113-162 and HF:modeling_mimi.py:210-447,602-726, what torch.autocast(bfloat16)
 * would run.  Takes every (prologue, epilogue, fused-RMSNorm, activated-output c_mode 3 / 4, split-K) form of the two
 * entry points above except the split-form operands (a_format 1, c_mode 1 / 2).  Not a parity path: judged by logit error /
 * token agreement / waveform error against the fp32 oracle (tests/test_gpu_bf16_mode.py). */
int sopro_gemm_bf16x1(const sopro_gemm_args* a, const void* packed_w, const sopro_gemm_split_ext* ext, void* stream);
/* W [N, ldw] fp32 (device) -> `pieces` (2: bf16x3, 3: bf16x6) bf16 planes in MFMA fragment order
 * [n/32][k/16][piece][lane][8]; `packed` holds sopro_packed_w_bytes(N, K, pieces) bytes, 16-byte aligned. */
int sopro_pack_w_bf16(const float* W, int64_t ldw, int32_t N, int32_t K, int32_t pieces, void* packed, void* stream);
/* Three-pass variant for token paths ("f16x3", round 3): operands split into TWO fp16 pieces (22 mantissa bits; the dropped
 * lo*lo term is <= 2^-22 relative), products lo*hi + hi*lo + hi*hi on v_mfma_f32_32x32x16_f16 - the forms of
 * sopro_gemm_bf16x6 (fp32 rows in / out or arg-max partials, prologues NONE / ADDVEC, epilogues NONE / GELU / RES / GLU, fused
 * RMSNorm) at half its passes.  fp16 has a narrow exponent: activations are scaled by sopro_f16x3_a_scale() (a power of two)
 * while they are staged, the weight by `wscale` (a power of two chosen by the host so that max|W| * wscale <= ~2^14) when it is
 * packed (sopro_packed_w_bytes(N, K, 2) bytes), and ext->acc_scale = 1 / (a_scale * wscale) undoes both exactly. */
/* INPUT-RANGE CONTRACT of the f16 path: an activation is staged as round_fp16(a * 8) + a second fp16 piece, so
 *   |a| <= 8188      is in range (exact scaling; 22 mantissa bits);
 *   |a| >  8188      SATURATES at +-65504 / 8 - finite, wrong, and not reported: callers keep their streams below it (the NAR
 *                    stream is O(1-10): its contraction inputs are RMS-normalised rows or GELU outputs; with the fused RMSNorm
 *                    the RAW residual row is what is staged, so the residual stream itself must stay below 8188);
 *   |a| <  2^-17     loses relative precision to the absolute resolution 2^-24 / 8 = 7.5e-9 of a scaled fp16 piece (a row of
 *                    RMS << 1e-2 in front of a fused RMSNorm has its round-off amplified by 1 / rms).
 * The fp32 reference has no such limits (src/sopro/nn/blocks.py:26-37, nar.py:89-116); paths whose inputs are not bounded like
 * that use sopro_gemm_bf16x6 (8 exponent bits). */
int sopro_pack_w_f16x2(const float* W, int64_t ldw, int32_t N, int32_t K, float wscale, void* packed, void* stream);
float sopro_f16x3_a_scale(void);
int sopro_gemm_f16x3(const sopro_gemm_args* a, const void* packed_w, const sopro_gemm_split_ext* ext, void* stream);
int64_t sopro_packed_w_bytes(int32_t N, int32_t K, int32_t pieces);
int sopro_gemm_bf16_set_tile_override(int cfg); /* developer probe: 0 = by shape, 1: 128x128, 2: 256x128, 4: 128x64, 5: 64x64, 7: 128x128 on eight waves (bf16x3 only) */
int sopro_gemm_set_group_m(int g);              /* default tile-walk group of the split-bf16 contractions (see group_m) */

/* Batch-of-at-most-a-few-dozen-rows contraction for the autoregressive step
 * (src/sopro/nn/generator.py:98-130): Y[b, n] = epi( rs[b] * sum_k Xin[b, k] * W[n, k] + bias[n] ),
 * Xin = X + sum_{s<np} Xp[s]  (the producer's partial sums, added in a fixed order; np is 0 or 3).
 * One workgroup = 16 output columns x one 384-wide K slice x 16 batch rows; fp32 v_mfma_f32_16x16x4_f32.
 * K % 384 == 0.  rms_norm != 0 (K == 384): rs[b] = rsqrt(mean(Xin[b]^2) + eps), i.e. the RMSNorm of
 * src/sopro/nn/blocks.py:26-37 whose weight vector the host has folded into W (W[n,k] * w_norm[k]); else rs = 1.
 * ksplit = 1 with K > 384: the K/384 slices go to different workgroups and Y receives K/384 partial results
 *   [slice][B][ldy] (y_part_stride elements apart); with EPI_RES slice 0 carries bias + residual, the other
 *   slices are raw; the consumer passes slice 0 as X and slices 1..3 as Xp.
 *   EPI_GLU_DW (N == 2*K == 768, W in the natural torch layout [value rows | gate rows], rms_norm set):
 *     h = value*sigmoid(gate); ring[(t % L)][b] = h; y = dwconv taps over the ring; Y = Xin + y
 *     == SSMLiteBlock.forward_step first half, src/sopro/nn/blocks.py:150-157 and :76-110.
 *     `step` is a device pointer to the current frame index t; ksize <= 13. */
typedef struct sopro_skinny_args {
  const float* X; int64_t ldx;
  const float* W; int64_t ldw;
  const float* bias;
  float* Y; int64_t ldy;
  const float* R; int64_t ldr;
  const float* scale;
  float* ring;            /* EPI_GLU_DW: [L, ring_bcap, D] */
  const float* dw_w;      /* [ksize, D] (tap-major, oldest tap first) */
  const float* dw_b;      /* [D] */
  const int32_t* step;    /* device scalar */
  const float* Xp; int64_t xp_stride;   /* np partial buffers laid out like X */
  int64_t y_part_stride;
  long long* dbg;         /* optional [workgroups][8] shader-clock stamps (profiling aid), NULL in production */
  float eps;
  int32_t B, N, K, epilogue;
  int32_t ring_len, ring_bcap, dil, ksize;
  int32_t np, ksplit, rms_norm;
  int32_t w_layout;       /* 0: W is [N, ldw] row-major; 1: W was laid out by sopro_pack_skinny_w (ldw unused);
                           * 2: bf16 weights from sopro_pack_skinny_w_bf16 (the engine's bf16 mode: activations are rounded to
                           *    bf16 as MFMA operands, v_mfma_f32_16x16x32_bf16 with fp32 accumulation) */
  /* AUX column tiles (round 4, plain form only; all zero = none): `aux_tiles` (even) more 16-column tiles behind the main ones run the
   * same input rows against a second operand set (weights in the layout of w_layout, [16 * aux_tiles, K]) and write to aux_Y - the
   * query projection of the following text cross-attention block rides on the feed-forward launches (see sopro_ar_frame.k_unfold).
   * aux_flags: bit 0 = no RMSNorm row scale, bit 1 = epilogue NONE (else the launch's epilogue with aux_bias / aux_R). */
  const float* aux_W; const float* aux_bias; const float* aux_R; float* aux_Y;
  int64_t aux_ldy, aux_ldr, aux_y_part_stride;
  int32_t aux_tiles, aux_flags;
  int32_t ring_format;    /* EPI_GLU_DW: 0 = `ring` holds fp32; 1 (with w_layout 2, the engine's bf16 mode) = `ring` points at bf16 elements
                           * [L, ring_bcap, D]: h is rounded once when it is written, the older taps are widened when they are read */
  int32_t mt, nt;         /* workgroup shape: mt 16-row groups of the batch x nt column tiles (0 or 1 = one; 2 = two).  1 x 1 has
                           * the most workgroups and the shortest latency; 2 x 2 reads the weights once per 32 rows and halves the
                           * activation re-reads (throughput form for a small CU partition).  Results are bit-identical. */
} sopro_skinny_args;
int sopro_skinny_f32(const sopro_skinny_args* args, void* stream);
/* Fragment order for the AR-step weights: [column tile][K/32][2][64 lanes][4 floats] - lane (i = lane & 15, g = lane >> 4) of
 * tile t holds k = 32*chunk + 8*g + 4*half .. +3 of weight row t*16 + i (glu = 0) or of the value row t*8 + (i & 7) (i < 8) /
 * gate row N/2 + t*8 + (i & 7) (glu = 1: the packed value/gate pairing of EPI_GLU_DW), so that every load instruction of the
 * kernel reads 1 KiB of consecutive memory.  Rows past N are zero.  `out` holds sopro_skinny_packed_floats(N, K, glu) floats. */
int sopro_pack_skinny_w(const float* W, int64_t ldw, int32_t N, int32_t K, int32_t glu, float* out, void* stream);
int64_t sopro_skinny_packed_floats(int32_t N, int32_t K, int32_t glu);
/* The same fragment order with bf16 elements ([column tile][K/32][64 lanes][8 bf16]): `out` holds
 * sopro_skinny_packed_floats(N, K, glu) * 2 bytes.  For w_layout = 2. */
int sopro_pack_skinny_w_bf16(const float* W, int64_t ldw, int32_t N, int32_t K, int32_t glu, void* out, void* stream);

/* ---- normalisation / elementwise ------------------------------------------------------- */
/* rows x width 32-bit words, pitches in words: fills and device-to-device copies as kernels.  The stage sequences use them
 * instead of hipMemset* / hipMemcpy* so that a recorded sequence holds kernel nodes only. */
int sopro_fill2d_u32(void* p, int64_t pitch, int32_t rows, int32_t width, uint32_t value, void* stream);
int sopro_copy2d_u32(void* dst, int64_t dpitch, const void* src, int64_t spitch, int32_t rows, int32_t width, void* stream);
/* (Either side of sopro_copy2d_u32 may be page-locked HOST memory from sopro_host_alloc: small parameter blocks travel to the device and
 * poll words back to the host as kernels of the library - the timed path holds no runtime copy at all, round 5.)
 * tokens[(b T + t) Q] = clamp(cb0[b cb0_bstride + t], 0, vmax): codebook 0 as the AR loop's history holds it (EOS = V in stopped rows)
 * into column 0 of the refinement's token matrix (src/sopro/model.py:385-390). */
int sopro_nar_seed_i32(int32_t* tokens, int32_t Q, const int32_t* cb0, int64_t cb0_bstride, int32_t B, int32_t T, int32_t vmax, void* stream);
/* fp32 <-> bf16 images of a tensor of n elements (n % 4 == 0; round to nearest even).  The engine's bf16 mode keeps the state the AR
 * frame streams every frame (folded text operands, ring buffers) and the SEANet decoder's activations as bf16 in memory; these
 * make / read such images outside the hot kernels (operand preparation, tests).  No reference counterpart (the reference has
 * no dtype argument anywhere: src/sopro/model.py:419-451). */
int sopro_cvt_f32_bf16(const float* src, void* dst, int64_t n, void* stream);
int sopro_cvt_bf16_f32(const void* src, float* dst, int64_t n, void* stream);
enum { SOPRO_NORM_RMS = 0, SOPRO_NORM_LN = 1 };
/* out[r, :] = (norm(x[r, :]) * w (+ b)) * mul[seg(r), :] + add[seg(r), :]   (mul/add optional, one
 * row per segment of rows_per_seg rows).  RMS: src/sopro/nn/blocks.py:26-37 (eps inside rsqrt);
 * LN: torch nn.LayerNorm (src/sopro/nn/speaker.py:72,83; HF:modeling_mimi.py:739-740).
 * mul/add express SpeakerFiLM (speaker.py:76-85) and NARStageAdapter (nar.py:25-32). */
int sopro_norm_f32(const float* x, int64_t ldx, int64_t x_seg_stride /* 0 = dense */, float* out, int64_t ldo, const float* w, const float* b,
                   const float* mul, const float* add, int32_t rows, int32_t rows_per_seg, int32_t C,
                   float eps, int32_t kind, void* stream);
/* stats[(r * (C / 64) + p) * 2 ..] = (mean, sum of squared deviations from it) of columns 64 p .. 64 p + 63 of row r: the pairs a contraction
 * with sopro_gemm_split_ext.ln_stats stages its A rows by (C % 64 == 0, 16-byte aligned rows).  Written by the producing contraction's
 * epilogue where there is one (ln_stats_out); this kernel serves the first norm of a stack (its stream comes from sopro_upsample2_f32). */
int sopro_row_stats_f32(const float* x, int64_t ldx, int64_t x_seg_stride /* 0 = dense */, int32_t rows, int32_t rows_per_seg, int32_t C,
                        float* stats, void* stream);
/* a * clamp(rms(x)/rms(a), 0, 10) per row, rms = sqrt(mean(t^2)+1e-6)   (src/sopro/nn/ref.py:12-13,101-102) */
int sopro_rms_match_f32(const float* a, const float* x, float* out, int32_t rows, int32_t C, void* stream);
/* out = c0 + c1 * tanh(in)     (FiLM / adapter coefficients, gates) */
int sopro_tanh_affine_f32(const float* in, float* out, float c0, float c1, int64_t n, void* stream);
/* out[b, t, :] = rowvec[b, :] + table[pos0 + t, :]   (src/sopro/model.py:200-202) */
int sopro_add_pos_f32(const float* rowvec, const float* table, float* out, int32_t B, int32_t T, int32_t C,
                      int32_t pos0, void* stream);
/* pooled[b, :] = sum_{t < len[b]} x[b, t, :] / (len[b] + 1e-6)   (src/sopro/nn/text.py:40-43) */
int sopro_masked_mean_f32(const float* x, const int32_t* lens, float* out, int32_t B, int32_t T, int32_t C,
                          void* stream);

/* attentive statistics pooling: a = softmax_{t<len}(logit[b,t]); out[b] = [sum a h | sqrt(clamp_min(sum a (h-mu)^2, 1e-6))]
 * (src/sopro/nn/blocks.py:174-188) and F.normalize(x, eps) (src/sopro/nn/speaker.py:60): Token2SV tail. */
int sopro_stats_pool_f32(const float* h, const float* logit, const int32_t* lens, float* out, int32_t B, int32_t T,
                         int32_t C, void* stream);
int sopro_l2norm_f32(const float* x, float* out, int32_t rows, int32_t C, float eps, void* stream);

/* Depthwise Conv1d over time, channels-last [B, T, C]; taps w[j, c] (tap-major), dilation dil,
 * `left` zero-padded positions on the left (causal: (k-1)*dil, symmetric: ((k-1)*dil)/2), positions
 * >= lens[b] read as zero (lens NULL = T).  mode 0: y; 1: res + y; 2: gelu(y).
 * == DepthwiseConv1d.forward, src/sopro/nn/blocks.py:63-74. */
int sopro_dwconv_f32(const float* x, const float* w, const float* bias, const float* res, float* out,
                     const int32_t* lens, int32_t B, int32_t T, int32_t C, int32_t ksize, int32_t dil,
                     int32_t left, int32_t mode, void* stream);

/* out[r, :] = alpha * base[r, :] + beta * sum_i wq[i] * table[off[i] + tok[r, col[i]], :]
 * (base optional).  Embedding gathers: src/sopro/nn/embeddings.py:54-55,77-112, text.py:31,
 * speaker.py:43-48, HF:modeling_mimi.py:1070-1081.  tok is int32 [rows, ldt]; col/off int32[nq]; wq float[nq]. */
int sopro_codebook_sum_f32(const int32_t* tok, int32_t ldt, const int32_t* col, const int32_t* off, const float* wq,
                           int32_t nq, const float* table, int64_t table_rows, const float* base, float alpha,
                           float beta, float* out, int64_t ldo, int64_t o_seg_stride, int32_t rows,
                           int32_t rows_per_seg, int32_t D, void* stream);
/* out[b, t, :] = table[ids[b, t], :] + pe[t, :]  (0 for t >= lens[b])   src/sopro/nn/text.py:31-33 */
int sopro_text_embed_f32(const int32_t* ids, const int32_t* lens, const float* table, int64_t table_rows,
                         const float* pe, float* out, int32_t B, int32_t T, int32_t C, void* stream);
/* argmax over each row of [rows, N] -> int32 written at out[(r / inner) * ldo + r % inner]  (src/sopro/model.py:340-343;
 * inner > 1: the `inner` heads of a NAR stage are consecutive rows of one logits matrix and fill consecutive codebooks) */
int sopro_argmax_rows_f32(const float* x, int64_t ldx, int32_t* out, int64_t ldo, int32_t inner, int32_t rows, int32_t N,
                          void* stream);
/* Second half of the arg-max fused into a projection (sopro_gemm_bf16x6 / x1 with ext.c_mode = 5, which writes per 64-column
 * tile and row one (float max, int32 column) pair to ext.C2 [M][ext.ldc2] instead of the logits - replaces the
 * `logits.argmax(dim=-1)` of src/sopro/model.py:338-345 without the [M, heads * V] logits ever reaching memory):
 * out[r * ldo + h] = (column of the best of the `per_head` pairs of head h) - h * V, first maximum on ties. */
int sopro_argmax_partials_i32(const float* partials, int64_t ldp, int32_t* out, int64_t ldo, int32_t heads, int32_t per_head, int32_t V,
                              int32_t rows, void* stream);

/* ---- Mimi encode side (reference audio -> tokens: src/sopro/codec/mimi.py:42-63) ---------------------- */
/* Single-input-channel FIR bank, channels-last output:
 *   out[b, t, c] = bias[c] + sum_k w[c, k] * x[b, t*stride + k - left]   (x = 0 outside [0, n_in))
 * The first SEANet encoder conv (HF:modeling_mimi.py MimiEncoder layers[0], 1 -> 64, k = 7, left = 6) and the
 * polyphase windowed-sinc resampler (src/sopro/audio.py:113-123 -> torchaudio.functional.resample: C = new_freq/gcd
 * phases, stride = orig_freq/gcd, left = filter half width).  bias may be NULL. */
int sopro_fir1_f32(const float* x, int64_t x_seg_stride, int32_t n_in, const float* w, const float* bias, float* out,
                   int64_t ldo, int64_t o_seg_stride, int32_t B, int32_t n_out, int32_t C, int32_t K, int32_t stride,
                   int32_t left, void* stream);
/* One residual-VQ layer's assignment (HF:modeling_mimi.py MimiEuclideanCodebook.quantize + the residual update of
 * MimiResidualVectorQuantizer.encode): codes[r*ldc] = first argmax of scores[r, :V]
 * (scores = r.e - |e|^2/2, i.e. the nearest code), then res[r, :D] -= table[code, :D]. */
int sopro_rvq_assign_f32(const float* scores, int64_t lds, int32_t V, const float* table, float* res, int64_t ldr,
                         int32_t D, int32_t* codes, int64_t ldc, int32_t rows, void* stream);

/* ---- attention ---------------------------------------------------------------------------- */
/* softmax(q k^T * scale + mask) v, fp32, heads interleaved in the row ([.., H*dh]).
 * klens[b] (NULL = Tk) masks keys >= klens[b]; causal != 0 additionally keeps only keys with
 * q_abs - window < k_abs <= q_abs where q_abs = q_pos0 + tq, k_abs = k_pos0 + tk.
 * dh in {64, 96, 192}.  Replaces F.scaled_dot_product_attention at src/sopro/nn/text.py:118-126,
 * src/sopro/nn/ref.py:88-96 and HF:modeling_mimi.py:657-726 (sliding window :882-888). */
typedef struct sopro_attn_args {
  const float* Q; int64_t ldq; int64_t q_bstride;
  const float* K; int64_t ldk; int64_t k_bstride;
  const float* V; int64_t ldv; int64_t v_bstride;
  float* O; int64_t ldo; int64_t o_bstride;
  const int32_t* klens;
  int32_t B, H, dh, Tq, Tk;
  int32_t causal, q_pos0, k_pos0, window;
  float scale;
  const int32_t* kv_index;  /* NULL, or [B]: batch row b reads K / V block kv_index[b] (x k_bstride / v_bstride) instead of block b:
                             * rows that share a voice share one copy of its keys (sopro_attention_f32 only) */
} sopro_attn_args;
int sopro_attention_f32(const sopro_attn_args* args, void* stream);
/* Decoder-only form of the causal window attention with dh == 64 (HF:modeling_mimi.py:657-726, sliding window :882-888):
 * Q, K, V and the softmax weights as two bf16 pieces, three bf16 MFMA passes per product (passes == 3: 16 mantissa bits,
 * the precision of the decoder's contractions under its waveform tolerance) or one piece (passes == 1, bf16 mode).
 * Any other shape runs sopro_attention_f32. */
int sopro_attention_split_bf16(const sopro_attn_args* args, int32_t passes, void* stream);
/* Tq == 1 form for the AR frame (cached text K/V, src/sopro/nn/text.py:85-132): one workgroup per
 * (batch row, head), all loads issued up front; dh in {64, 96}; no causal mask. */
int sopro_attn_decode_f32(const sopro_attn_args* args, void* stream);
/* The whole cached text cross-attention block of the AR frame in one launch (src/sopro/nn/text.py:85-132) on
 * per-utterance folded operands Kp[b,h] = K_h Wq_h and Vp[b,h] = V_h Wo_h^T (both [S_cap, D], D == 384):
 * Y[h][b] = (h == 0 ? Xin[b] : 0) + gate * sum_k softmax_k(<RMSNorm(Xin[b]), Kp[b,h,k]> * scale) * Vp[b,h,k],
 * Xin = X + sum_{s<np} Xp[s].  The H partial outputs (y_part_stride apart) are summed by the next kernel. */
typedef struct sopro_xattn_args {
  const float* X; int64_t ldx;
  const float* Xp; int64_t xp_stride;
  const float* norm_w;
  const float* Kp; const float* Vp;
  const int32_t* klens;
  float* Y; int64_t y_part_stride;
  float eps, gate, scale;
  int32_t np, B, H, D, S_cap;
  /* Unfolded keys (round 4): k_unfolded = 1: Kp points at K [B, S_cap, D] (the k projection, head h in columns (D / H) h ..; 4x fewer
   * bytes than the folded K') and Qp at nqp (1..4) partial sums [B, D] (qp_stride elements apart) of the RAW query Wq' x (RMSNorm_nq's
   * weight folded into Wq'), which the kernel adds in order and scales by the row scale of x; norm_w must be NULL.  Vp stays folded. */
  const float* Qp; int64_t qp_stride;
  int32_t nqp, k_unfolded;
  int32_t kv_format;      /* 0: Kp / Vp hold fp32; 1 (the engine's bf16 mode): they point at bf16 elements, same [B, H, S_cap, D] layout -
                           * half the bytes of the block's dominant stream; scores, softmax and the weighted sum stay fp32 */
} sopro_xattn_args;
int sopro_xattn_step_f32(const sopro_xattn_args* args, void* stream);

/* rotate-half RoPE in place on [rows, H*dh] with host-made tables cos/sin [npos, dh/2]; position of
 * row r is pos0 + (r % rows_per_seg).  HF:modeling_mimi.py:511-599. */
int sopro_rope_f32(float* x, int64_t ldx, const float* cos_t, const float* sin_t, int32_t rows,
                   int32_t rows_per_seg, int32_t pos0, int32_t H, int32_t dh, void* stream);

/* ---- Mimi decoder specials --------------------------------------------------------------- */
/* depthwise ConvTranspose1d k=4 s=2, causal trim: y[2t+r, c] = x[t,c] w[c,r] + x[t-1,c] w[c,r+2]
 * (HF:modeling_mimi.py:1208-1216, 350-405).  x [B, T, C] dense; y segments [B][2T, C] at y_seg_stride. */
int sopro_upsample2_f32(const float* x, const float* w, float* y, int64_t y_seg_stride, int32_t B, int32_t T,
                        int32_t C, void* stream);
/* last SEANet layer: wav[b, n] = bias + sum_{j<3} sum_c elu(h[b, n-2+j, c]) w[j, c], C = 64; h has two
 * zero rows in front of each segment (HF:modeling_mimi.py:957-960). */
int sopro_final_conv_f32(const float* h, int64_t h_seg_stride, const float* w, float bias, float* wav,
                         int64_t wav_seg_stride, int32_t B, int32_t T, void* stream);

/* Fused 24 kHz tail of the SEANet decoder: last MimiResnetBlock (dim 64: k=3 conv 64->32, k=1 conv 32->64, residual)
 * + last layer (ELU, k=3 conv 64->1), HF:modeling_mimi.py:408-447, 957-960.  h [B][2+T][64] (two zero rows in front of
 * each segment) is read once; only wav [B][T] is written.  w1 [32][3*64] (tap-major K), w2 [64][32], wf [3][64]. */
int sopro_seanet_tail_f32(const float* h, int64_t h_seg_stride, const float* w1, const float* b1, const float* w2,
                          const float* b2, const float* wf, float bf, float* wav, int64_t wav_seg_stride, int32_t B,
                          int32_t T, void* stream);
/* the same with the MFMA passes per product chosen: 3 (above) or 1 (the engine's bf16 mode: operands rounded to bf16 once; long
 * inputs only - short ones run the three-pass kernel either way) */
int sopro_seanet_tail_p_f32(const float* h, int64_t h_seg_stride, const float* w1, const float* b1, const float* w2,
                            const float* b2, const float* wf, float bf, float* wav, int64_t wav_seg_stride, int32_t B,
                            int32_t T, int32_t passes, void* stream);
/* Fused MimiResnetBlock at the 128-channel level of the SEANet decoder (HF:modeling_mimi.py:408-447, dim 128: k=3 conv
 * 128->64, k=1 conv 64->128, residual) followed by the ELU of the next layer: out = ELU(h + c2(ELU(c1(ELU(h))))).
 * h, out: [B][2 + T][128] (two zero rows in front of each segment; out's are not written), h != out.  w1 [64][3*128]
 * (tap-major K), w2 [128][64].  h is read once, out written once; both weight matrices stay in registers. */
int sopro_seanet_res128_f32(const float* h, int64_t h_seg_stride, const float* w1, const float* b1, const float* w2, const float* b2,
                            float* out, int64_t out_seg_stride, int32_t B, int32_t T, void* stream);
int sopro_seanet_res128_p_f32(const float* h, int64_t h_seg_stride, const float* w1, const float* b1, const float* w2, const float* b2,
                              float* out, int64_t out_seg_stride, int32_t B, int32_t T, int32_t passes /* 3, or 1: bf16 mode */, void* stream);
int sopro_seanet_res_set_tiles(int tiles); /* developer probe / tests: 64-row tiles per workgroup, 0 = by size */
/* Last transposed convolution of the SEANet decoder, ConvTranspose1d(128 -> 64, k = 8, s = 4) (HF:modeling_mimi.py:931-961),
 * weight-stationary: out[b][t][0..255] = bias + W . [x[b][t] | x[b][t+1]], t < T, with x [B][>= 1 + T][128] the ACTIVATED input
 * whose row 0 of every segment is the zero row in front of the first sample, W [256][256] = pack_convtr1d's matrix (row =
 * output phase * 64 + channel, column = tap half * 128 + input channel), bias [256], out rows of 256 floats (= 4 output
 * samples x 64 channels).  Same results as sopro_gemm_bf16x3 (passes = 3) / sopro_gemm_bf16x1 (passes = 1) on that shape, bit
 * for bit; the weights are split once per workgroup and stay in registers. */
int sopro_seanet_up128_f32(const float* x, int64_t x_seg_stride, const float* w, const float* bias, float* out,
                           int64_t out_seg_stride, int32_t B, int32_t T, int32_t passes, void* stream);
/* The bf16 mode's activation flow (round 4): the three fused SEANet kernels on bf16 ROWS in memory - half the bytes of the decoder's
 * largest tensors - with one MFMA pass on the rounded operands and fp32 accumulation / bias / ELU / skip additions.  Pointers are
 * to bf16 elements and the segment strides count bf16 elements; layouts as in the fp32 forms above.
 *   sopro_seanet_res128_bf16: h raw [B][2 + T][128] -> out = ELU(h + c2(ELU(c1(ELU(h))))) [B][2 + T][128]
 *   sopro_seanet_up128_bf16 : x activated [B][>= 1 + T][128] -> out rows of 256 (4 samples x 64 channels), raw
 *   sopro_seanet_tail_bf16  : h raw [B][2 + T][64] -> wav fp32 [B][T]  (the sixteen-wave kernel at any size) */
int sopro_seanet_res128_bf16(const void* h, int64_t h_seg_stride, const float* w1, const float* b1, const float* w2, const float* b2,
                             void* out, int64_t out_seg_stride, int32_t B, int32_t T, void* stream);
int sopro_seanet_up128_bf16(const void* x, int64_t x_seg_stride, const float* w, const float* bias, void* out, int64_t out_seg_stride,
                            int32_t B, int32_t T, void* stream);
int sopro_seanet_tail_bf16(const void* h, int64_t h_seg_stride, const float* w1, const float* b1, const float* w2, const float* b2,
                           const float* wf, float bf, float* wav, int64_t wav_seg_stride, int32_t B, int32_t T, void* stream);
int sopro_seanet_up_set_tiles(int tiles); /* developer probe / tests: 64-row tiles per workgroup, 0 = by size */
/* The whole last level in ONE kernel (round 4, csrc/seanet_uptail.hip): the last transposed convolution (128 -> 64, k = 8, s = 4),
 * the last residual block and the last layer - sopro_seanet_up128_* followed by sopro_seanet_tail_* without the round trip of the
 * 64-channel activation through memory (HF:modeling_mimi.py:931-961, 408-447).  x as for sopro_seanet_up128_f32 / _bf16 (activated
 * input, [B][>= 1 + T][128], the pointer at the zero row in front of the first input row; strides in elements), the weights of the
 * two kernels it replaces, wav [B][4 T] fp32.  passes 3 = three-pass split-bf16 (the decoder's class), 1 = one pass (bf16 mode).
 * Same operand rounding and accumulation per contraction as the two kernels; the sums of the last layer are taken in the order of
 * the sixteen-wave tail.  Written for long inputs (a workgroup walks >= 24 tiles of 32 input rows): the engine uses it from
 * 512 Ki input rows per call, the two kernels below that. */
int sopro_seanet_uptail_f32(const float* x, int64_t x_seg_stride, const float* wu, const float* bu, const float* w1, const float* b1,
                            const float* w2, const float* b2, const float* wf, float bf, float* wav, int64_t wav_seg_stride, int32_t B,
                            int32_t T, int32_t passes, void* stream);
int sopro_seanet_uptail_bf16(const void* x, int64_t x_seg_stride, const float* wu, const float* bu, const float* w1, const float* b1,
                             const float* w2, const float* b2, const float* wf, float bf, float* wav, int64_t wav_seg_stride, int32_t B,
                             int32_t T, void* stream);
int sopro_seanet_uptail_set_tiles(int tiles); /* developer probe / tests: 32-row tiles per workgroup, 0 = by size */
int sopro_seanet_tail_set_tiles(int tiles); /* developer probe / tests: 126-sample tiles per workgroup of the four-wave kernel, 0 = by size
                                             * (long inputs: the sixteen-wave kernel), < 0 = the sixteen-wave kernel at any size */

/* ---- launch timing of the stage sequences (csrc/prof.hip) -------------------------------------
 * bench.py's roofline leg (the reference has no counterpart: its profiler is torch's).  While enabled, every heavy launch
 * of sopro_nar_refine / sopro_mimi_decode* / sopro_ar_fold_text is bracketed by two HIP events on its stream (never while the
 * stream is capturing).  sopro_prof_collect waits for the events recorded so far, sums them per kernel family and forgets
 * them.  gpu_bound / ms_bound / flops_bound cover the samples whose stream still had work queued when the first event was
 * recorded (their span is GPU time only). */
typedef struct sopro_prof_row {
  char family[40];
  int64_t launches, gpu_bound;
  double flops, flops_bound, ms_all, ms_bound;
} sopro_prof_row;
int sopro_prof_enable(int on);
int sopro_prof_collect(sopro_prof_row* rows, int32_t cap, int32_t* n_rows);

/* ---- autoregressive driver state ----------------------------------------------------------- */
/* Device-resident state of ar_stream (src/sopro/model.py:218-305) for up to `bcap` rows. All
 * pointers are caller-allocated device buffers. */
typedef struct sopro_ar_state {
  float* x_cur;            /* [bcap, D] input of the current frame */
  const float* cond;       /* [B, Tar, D] */
  const float* emb;        /* cb_embed.emb.weight [Q*V+1, D] */
  int32_t* hist;           /* [bcap, max_steps] sampled tokens */
  int32_t* step;           /* scalar: current frame index t */
  int32_t* row_step;       /* [bcap] per-row copy of the frame index (each sampler workgroup reads and advances its own: no
                            * ticket, no fence; *step follows row 0 and is read by the NEXT frame's kernels only) */
  int32_t* first_eos;      /* [bcap] first t with tok == EOS, -1 if none */
  int32_t* stop_t;         /* [bcap] first t with tok == EOS and t+1 >= min_gen, -1 if none */
  int32_t* n_stopped;      /* scalar: rows with stop_t >= 0 */
  int32_t* recent;         /* [bcap, 64] rolling window: slot j = token sampled j+1 frames ago, -1 = none */
  const float* params;     /* [8] top_p, temperature, anti_loop, rec_top_p, rec_temperature, rep_penalty, top_k, min_gen */
  uint64_t seed;           /* Philox key when `key` is NULL (frozen into a recorded frame graph: prefer `key`) */
  int32_t B, D, Tar, max_steps, V /* 2048, EOS id == V */, bos_row;
  /* Slot mode (continuous batching; all three NULL = every row starts at frame 0 with the shared params): */
  int32_t* start;          /* [bcap] global frame at which the row was admitted, -1 = free slot; row time = step - start */
  const int32_t* row_max;  /* [bcap] frame budget of the row (rows of its cond block, <= Tar); NULL = Tar */
  const float* row_params; /* [bcap, 8] per-row params; NULL = params */
  /* Per-row run nonce mixed into the sampler's Philox counter (t, row, nonce): the host bumps it for every run / slot
   * admission so that two calls with the same text do not reuse one uniform sequence (the reference draws from torch's
   * global generator, which advances between calls).  NULL = 0.  Device memory, so a recorded frame graph sees updates. */
  const uint32_t* nonce;   /* [bcap] */
  const int32_t* row_id;   /* [bcap] the row's identity in the Philox counter (t, row_id, nonce); NULL = the row index.  A scheduler that
                            * coalesces several requests into one batch gives every row the index it has in its OWN request, so that
                            * its draws - and its audio - do not depend on what it was batched with */
  const uint32_t* key;     /* [2] Philox key (seed low, high word) in device memory, so that ONE recorded frame graph serves every
                            * seed; NULL = the by-value `seed` above */
  long long* dbg;          /* optional [bcap][12] shader-clock stamps of the sampler's phases (profiling aid), NULL in production */
} sopro_ar_state;
/* zero-step initialisation: step=0, flags reset, x_cur[b] = cond[b,0] + emb[bos_row]  (model.py:266-272) */
int sopro_ar_init(const sopro_ar_state* st, void* stream);
/* sample_token (src/sopro/sampling.py:24-93) + anti-loop policy (model.py:274-299) + EOS rule
 * (model.py:301-305) for every row, then x_cur[b] = cond[b, t+1] + emb[tok] and step += 1. */
int sopro_ar_sample(const sopro_ar_state* st, const float* logits, int64_t ld_logits, void* stream);
/* Slot mode: (re)start row `row` at the current global frame: x_cur = cond[row, 0] + emb[bos_row], flags and the recent
 * window reset, start[row] = *step.  The caller has already written the row's cond block, its cross-attention operands
 * and zeroed its ring-buffer columns, all on the same stream. */
int sopro_ar_admit(const sopro_ar_state* st, int32_t row, void* stream);

/* ---- one autoregressive frame as ONE call --------------------------------------------------------------
 * The launch sequence of a frame (6 x [GLU tail, FF1, FF2 (+ text cross-attention)] + head + sampler == ARRVQ1Generator.step,
 * src/sopro/nn/generator.py:98-130, followed by sample_token, src/sopro/sampling.py:24-93) over caller-owned buffers.  Both
 * hosts of this library use it (sopro_amd/model.py inside its stream capture; sopro_ar_run_graph below), so the sequence
 * exists once.  Weights are the fragment-ordered images of sopro_pack_skinny_w (w_layout 1) or _bf16 (w_layout 2), RMSNorm
 * weights folded in.  tile_*: workgroup shape of the stage kind, (mt << 4) | nt as in sopro_skinny_args (0 = 1 x 1). */
#define SOPRO_AR_MAX_LAYERS 16
typedef struct sopro_ar_block {
  const void* glu_w; const float* glu_b; const float* dw_w; const float* dw_b;
  const void* ff1_w; const float* ff1_b;
  const void* ff2_w; const float* ff2_b;
  float* ring;             /* [(ksize-1)*dil + 1, B, D]; bf16 elements when the frame's store_format is 1 */
  const float* kp;         /* xattn != 0: folded text operands [B, H, S_cap, D] (src/sopro/nn/text.py:75-83 x q_proj / out_proj);
                            * bf16 elements when the frame's store_format is 1 */
  const float* vp;
  int32_t dil, xattn;
  float gate;              /* tanh(gate) of the cross-attention block (text.py:131) */
  int32_t pad_;
  /* Unfolded keys (sopro_ar_frame.k_unfold, blocks with xattn != 0): kp then holds K [B, S_cap, D] and the block's feed-forward launches
   * emit the query on aux tiles - qa_w = Wq' [D, D] (q_proj x RMSNorm_nq's weight) rides on FF1, qu_w = Wq' W2 [D, 4 D] (folded in
   * float64 by the host) and q_b = Wq' b2 [D] on FF2's K-slices; all three in the fragment order / element type of the frame's w_layout. */
  const void* qa_w; const void* qu_w; const float* q_b;
} sopro_ar_block;
typedef struct sopro_ar_frame {
  sopro_ar_block blk[SOPRO_AR_MAX_LAYERS];
  const void* head_w; const float* head_b;
  float* x0;               /* [B, D] input of the frame == st.x_cur */
  float* xa; float* xb;    /* [B, D] residual stream (alternating) */
  float* part;             /* [4, B, D] K-slice partial sums of FF2 */
  float* u;                /* [B, 4D] */
  float* xp;               /* [H, B, D] per-head outputs of a cross-attention block */
  float* qa;               /* k_unfold: [B, D] the `Wq' out` part of the raw query (FF1's aux tiles) */
  float* qpart;            /* k_unfold: [4, B, D] K-slice partials of the raw query (FF2's aux tiles; slice 0 carries qa + q_b) */
  float* logits;           /* [B, V1] */
  const int32_t* klens;    /* [B] text lengths */
  int32_t n_layers, B, D, S_cap, V1, H, ksize, w_layout;
  int32_t tile_glu, tile_ff1, tile_ff2, tile_head;
  float eps;
  int32_t k_unfold;        /* 1: the text cross-attention blocks read UNFOLDED keys (sopro_xattn_args.k_unfolded) and their query rides on the
                            * feed-forward launches in front of them (sopro_skinny_args aux tiles): 4x fewer key bytes per frame, no extra launch */
  int32_t store_format;    /* 0: ring buffers and folded text operands in fp32; 1 (bf16 mode, needs w_layout 2): both in bf16 - the
                            * frame's state streams at half the bytes; accumulation, norms, softmax, residual stream stay fp32 */
  sopro_ar_state st;
} sopro_ar_frame;
int sopro_ar_issue_frame(const sopro_ar_frame* frame, void* stream);

/* ==========================================================================================================
 * Stage-level entry points: the launch SEQUENCES of the hot path as C functions, so that a host that is not
 * Python drives generation with a handful of calls instead of re-implementing sopro_amd/model.py + codec.py.
 * They are thin sequencers over the operator entry points above (same kernels, same order as the Python host)
 * and cover the three stages of SoproTTS.synthesize after conditioning:
 *   sopro_ar_begin / sopro_ar_run_graph / sopro_ar_tokens  - SoproTTSModel.ar_stream    (src/sopro/model.py:218-305;
 *                                                            per frame ARRVQ1Generator.step, nn/generator.py:98-130)
 *   sopro_nar_refine                                       - SoproTTSModel.nar_refine   (src/sopro/model.py:307-347)
 *   sopro_mimi_decode                                      - MimiCodec.decode_full      (src/sopro/codec/mimi.py:65-72
 *                                                            -> HF MimiModel._decode_frame, HF:modeling_mimi.py:1388-1406)
 * Weights: the engine is handed the repacked tensors of sopro_amd/pack.py by name (device pointers, fp32 unless noted;
 * `python -m sopro_amd.export` writes them to a flat file for hosts without Python) and builds the matrix-core operand
 * forms (sopro_pack_w_bf16, sopro_pack_skinny_w) itself in sopro_engine_finalize.  All activations / tokens are caller-owned
 * device buffers; scratch comes from caller-provided workspaces sized by the *_workspace_bytes functions.  Every call only
 * enqueues on `stream`; nothing synchronises, nothing is allocated after sopro_engine_finalize (so the AR frame sequence is
 * recorded into a hipGraph on the first sopro_ar_run_graph).  One engine per device, not thread-safe.
 * ========================================================================================================== */
typedef struct sopro_engine sopro_engine;
typedef struct sopro_engine_cfg {
  /* Sopro (src/sopro/config.py) */
  int32_t d_model, codebook_size, num_codebooks, nar_head_dim, bos_row;
  int32_t n_layers_ar, ar_kernel, ar_dilations[16], ar_xattn[16] /* 1: block i is followed by a text cross-attention */;
  float ar_gate[16];                 /* tanh(gate) of that cross-attention block (src/sopro/nn/text.py:131) */
  int32_t n_layers_nar, nar_kernel, nar_dilations[16];
  int32_t n_stages, stage_first_cb[8], stage_n_cb[8];   /* stages B, C, D, E: codebooks [first, first + n) */
  float nar_mix[8][2];               /* softmax(nar.mix.<stage>) (src/sopro/nn/nar.py:95-97) */
  float nar_prev_cb_weights[64];     /* raw nar_prev_cb_weights (softmax over the known codebooks is taken per stage) */
  /* Mimi decode side (HF:configuration_mimi.py) */
  int32_t mimi_hidden, mimi_codebook_dim, mimi_heads, mimi_head_dim, mimi_layers, mimi_window, mimi_inter;
  int32_t mimi_n_ratios, mimi_ratios[8], mimi_num_filters, mimi_kernel, mimi_res_kernel, mimi_last_kernel, mimi_compress;
  int32_t mimi_n_semantic, mimi_rope_positions;
  float mimi_norm_eps, mimi_final_bias;
  int32_t precision;                 /* 0: the fp32 parity configuration (AR exact fp32, NAR f16x3, Mimi bf16x3); 1: the bf16 mode
                                      * (bf16 AR weights, one-pass NAR / Mimi contractions, fp32 accumulators / norms / residuals) */
  /* conditioning side (src/sopro/config.py): text encoder, reference encoder, reference cross-attention, Token2SV */
  int32_t n_layers_text, ref_enc_layers, ref_xattn_layers, ref_xattn_heads, sv_student_dim, enc_kernel /* 7: both encoders */;
} sopro_engine_cfg;
int sopro_engine_create(const sopro_engine_cfg* cfg, sopro_engine** out);
/* name: a key of sopro_amd.pack.pack_sopro / pack_mimi ("ar.blocks.0.glu.w", "nar.heads.B.w", "tr.3.qkv.w", "sea.up1.w", ...)
 * plus "rope.cos" / "rope.sin" [positions, head_dim / 2].  The engine keeps the pointer; the caller keeps the memory. */
int sopro_engine_set_tensor(sopro_engine* e, const char* name, const void* dev_ptr, const int64_t* shape, int32_t ndim);
/* The same from HOST memory: the engine allocates device memory of its own for the tensor (freed by sopro_engine_destroy), copies and
 * registers it.  For hosts that hold the weights on the CPU (sopro_engine_from_checkpoint below). */
int sopro_engine_upload_tensor(sopro_engine* e, const char* name, const float* host_ptr, const int64_t* shape, int32_t ndim, void* stream);
/* builds the packed operand forms (allocates device memory).  A stage family takes part when its marker tensor was given -
 * "ar.head.w" (AR), "nar.pre.w" (NAR), "rvq_proj.w" (Mimi decoder), "text_enc.embed" (conditioning + reference preparation,
 * with the position table "pe" [positions, D]) - and must then be complete; a stage call on an engine
 * without its family is refused.  After this call the engine is read-only except for its AR plan: NAR / Mimi calls of several
 * host threads (lanes with their own workspaces and streams) may share it. */
int sopro_engine_finalize(sopro_engine* e, void* stream);
/* The folded text operands of ONE cross-attention layer of the AR loop for B utterances of S text positions
 * (src/sopro/nn/text.py:75-83 with q_proj / out_proj folded in: K'_h = K_h Wq_h, V'_h = V_h Wo_h^T): txt [B*S, D] ->
 * kp, vp [B, H, S_cap, D].  q_wT: [H, D, D/H] (RMSNorm_nq folded in), o_w: [D, D]; nkv [B*S, D], kvd [B*S, 2D] scratch.
 * The one place this preparation is written down: sopro_ar_begin and the Python host (batched and slot admission) call it. */
int sopro_ar_fold_text(const float* txt, const float* nkv_weight, const float* kv_w, const float* q_wT, const float* o_w, float* nkv, float* kvd,
                       float* kp, float* vp, int32_t B, int32_t S, int32_t S_cap, int32_t D, int32_t H, float eps, void* stream);
/* The same for a frame with UNFOLDED keys (sopro_ar_frame.k_unfold): kq [B, S_cap, D] = k_proj(RMSNorm_nkv(txt)) as it is (head h in columns
 * (D / H) h ..), vp [B, H, S_cap, D] folded as above; no q_wT (the query projection rides on the frame's feed-forward launches). */
int sopro_ar_fold_text_uk(const float* txt, const float* nkv_weight, const float* kv_w, const float* o_w, float* nkv, float* kvd, float* kq, float* vp,
                          int32_t B, int32_t S, int32_t S_cap, int32_t D, int32_t H, float eps, void* stream);
int sopro_engine_destroy(sopro_engine* e);
/* workgroup shapes of the AR-step stage kinds, (mt << 4) | nt each (see sopro_skinny_args; 0 = 1 x 1).  Drops a recorded frame graph. */
int sopro_engine_set_ar_tiles(sopro_engine* e, int32_t glu, int32_t ff1, int32_t ff2, int32_t head);

/* ---- checkpoint -> engine without Python (round 4; csrc/checkpoint.hip).  A host starts where the reference starts
 * (src/sopro/model.py:419-451, src/sopro/hub.py:30-52): from model.safetensors - the reference's own state_dict keys
 * ("ar.blocks.0.glu.pro.weight", "ar.x_attns.1.q_proj.weight", "cb_embed.emb.weight", "nar.heads.B.0.weight", ...; config JSON in the
 * header's __metadata__["cfg"], key-intersection load as hub.py:44-48) - and the Mimi codec's model.safetensors (HuggingFace
 * MimiModel.state_dict() keys; src/sopro/codec/mimi.py:28-31).  sopro_checkpoint_open reads both (F32 / F16 / BF16 / F64 -> fp32) and
 * applies on the host the repacking sopro_amd/pack.py applies (GLU interleave, tap-major convolution weights, folded norm vectors /
 * head-id embeddings / unfolded-key query operands in float64, codebooks, softmax / tanh of the scalar parameters, the position and
 * RoPE tables) under the packed names sopro_engine_set_tensor takes; mimi_path may be NULL (no codec family).
 * sopro_engine_from_checkpoint = sopro_engine_create + sopro_engine_upload_tensor for every packed tensor + sopro_engine_finalize.
 * sopro_checkpoint_tensor enumerates the packed tensors (host pointers valid until sopro_checkpoint_close; shape4 padded with 1s). */
typedef struct sopro_checkpoint sopro_checkpoint;
int sopro_checkpoint_open(const char* sopro_path, const char* mimi_path, sopro_checkpoint** out);
int sopro_checkpoint_close(sopro_checkpoint* ck);
int32_t sopro_checkpoint_count(const sopro_checkpoint* ck);
int sopro_checkpoint_tensor(const sopro_checkpoint* ck, int32_t i, const char** name, const float** data, int64_t* shape4, int32_t* ndim);
int sopro_checkpoint_engine_cfg(const sopro_checkpoint* ck, int32_t precision, sopro_engine_cfg* cfg);
int sopro_engine_from_checkpoint(const sopro_checkpoint* ck, int32_t precision, void* stream, sopro_engine** out);

/* ---- conditioning stage (src/sopro/model.py:172-216 for B utterances at once): text encoder (src/sopro/nn/text.py:29-44),
 * pooled text + frame positions (model.py:200-202), SpeakerFiLM with per-row coefficients (src/sopro/nn/speaker.py:76-85; made
 * once per voice by sopro_film_coeffs), the reference cross-attention blocks over the voices' cached K / V
 * (src/sopro/nn/ref.py:54-108), cond_norm (model.py:208).  ids [B, S] int32 (rows padded with anything past lens[b]), lens [B],
 * ragged != 0 when some lens[b] < S; film_mul / film_add [B, D]; ref_k[i] / ref_v[i] (host arrays of ref_xattn_layers device
 * pointers): dense [*, Tr, D] keys / values per layer, block of row b = (kv_index ? kv_index[b] : b) * kv_bstride floats in
 * (kv_bstride 0: one voice for all rows); ref_klens [B] or NULL (= Tr).  Out: txt_seq [B, S, D], txt_pool [B, D],
 * cond_ar [B, Tar, D].  Launches only. */
int64_t sopro_cond_workspace_bytes(const sopro_engine* e, int32_t B, int32_t S, int32_t Tar);
int sopro_cond_prepare(sopro_engine* e, void* workspace, const int32_t* ids, const int32_t* lens, int32_t ragged, const float* film_mul, const float* film_add,
                       const float* const* ref_k, const float* const* ref_v, int64_t kv_bstride, const int32_t* kv_index, const int32_t* ref_klens,
                       int32_t B, int32_t S, int32_t Tar, int32_t Tr, float* txt_seq, float* txt_pool, float* cond_ar, void* stream);
/* SpeakerFiLM coefficients of n voices (speaker.py:76-85): sv [n, sv_student_dim] -> mul = 1 + style * tanh(gamma), add = style *
 * tanh(beta), both [n, D]; scratch: n * 5 * D floats. */
int sopro_film_coeffs(sopro_engine* e, const float* sv, float style, int32_t n, float* scratch, float* mul, float* add, void* stream);
/* Reference preparation of one voice from its codec tokens (src/sopro/model.py:151-170): Token2SV (src/sopro/nn/speaker.py:37-61)
 * -> sv [sv_student_dim]; reference sequence encoder (model.py:133-149) -> ref_seq [T, D]; K | V rows of every reference
 * cross-attention block (src/sopro/nn/ref.py:120-128) -> kv[i] [T, 2 D] (host array of ref_xattn_layers device pointers).
 * tokens [T, Q] int32.  ref_seq == NULL and kv == NULL: the speaker vector alone (SoproTTS.encode_speaker,
 * src/sopro/model.py:457-475 -> token2sv).  Launches only. */
int64_t sopro_ref_workspace_bytes(const sopro_engine* e, int32_t T);
int sopro_ref_prepare(sopro_engine* e, void* workspace, const int32_t* tokens, int32_t T, float* sv, float* ref_seq, float* const* kv, void* stream);

/* ---- autoregressive stage.  S_cap = S rounded up to 64.  The workspace must stay alive (and untouched) until the tokens
 * have been read; one generation at a time per engine. */
int64_t sopro_ar_workspace_bytes(const sopro_engine* e, int32_t B, int32_t S, int32_t Tar);
/* cond_ar [B, Tar, D], txt_seq [B, S, D] (conditioning outputs), text_lens [B] or NULL (= S); params = {top_p, temperature,
 * anti_loop, recovery top_p, recovery temperature, repetition penalty, top_k <= 64, min_gen_frames}; (seed, nonce) key the sampler. */
int sopro_ar_begin(sopro_engine* e, void* workspace, int32_t B, const float* cond_ar, const float* txt_seq, const int32_t* text_lens,
                   int32_t S, int32_t Tar, const float params[8], uint64_t seed, uint32_t nonce, void* stream);
/* n_steps more frames (23 launches each, replayed from a hipGraph recorded on the first call) */
int sopro_ar_run_graph(sopro_engine* e, int32_t n_steps, void* stream);
/* device-to-device copies: hist [B, Tar] int32 (codebook-0 tokens, EOS = codebook_size), first_eos [B] (-1 = none), n_stopped [1] */
int sopro_ar_tokens(sopro_engine* e, int32_t* hist, int32_t* first_eos, int32_t* n_stopped, void* stream);

/* ---- NAR refinement: tokens [B, T, Q] int32 out; column 0 <- rvq1 [B, T], columns 1..Q-1 refined.  lens [B] or NULL (= T). */
int64_t sopro_nar_workspace_bytes(const sopro_engine* e, int32_t B, int32_t T);
int sopro_nar_refine(sopro_engine* e, void* workspace, const float* cond, int64_t cond_bstride, const int32_t* rvq1, const int32_t* lens,
                     int32_t B, int32_t T, int32_t* tokens, void* stream);
/* The same with its operands where the stages in front of it left them (round 5: no copies between the stages of a pass).
 *   cond / cond_bstride  the first T rows of each utterance's conditioning block (the AR plan's buffer: cond_ar [B, Tar, D])
 *   cb0 / cb0_bstride    codebook 0 as rows of the AR loop's history [B, >= T] (EOS = codebook_size in rows that stopped: clamped here)
 *   lens                 [B] or NULL; may live in page-locked host memory (it is copied into the workspace first)
 *   safe                 0: the f16 three-pass operands (precision 0) - 1: the six-pass bf16 operands, which have fp32's exponent range
 *   range_out            optional word (device or page-locked host memory) that receives the number of RANGE EVENTS of this call: workgroups
 *                        of the f16 contractions that had to saturate an activation (sopro_gemm_split_ext.range_events).  Non-zero = this
 *                        pass's tokens are not trustworthy: repeat it with safe = 1 (sopro_amd/model.py does; a checkpoint whose residual
 *                        stream leaves fp16's range pays the slower path instead of returning wrong tokens silently).
 * sopro_nar_refine(...) = this with dense cb0 rows, safe = 0 and no range word. */
typedef struct sopro_nar_io {
  const float* cond; int64_t cond_bstride;
  const int32_t* cb0; int64_t cb0_bstride;
  const int32_t* lens;
  int32_t* tokens;      /* [B T, Q] out */
  int32_t* range_out;
  int32_t safe;
} sopro_nar_io;
int sopro_nar_refine_io(sopro_engine* e, void* workspace, const sopro_nar_io* io, int32_t B, int32_t T, void* stream);

/* ---- Mimi encode (reference audio -> codec tokens; src/sopro/codec/mimi.py:42-63 -> HF MimiModel.encode): SEANet encoder,
 * encoder transformer, stride-2 downsample, split residual VQ, every contraction in exact fp32.  Family marker "enc.conv0.w"
 * (+ the "etr.*" transformer tensors, "rope.cos" / "rope.sin").  wav [B, N] at the codec rate -> codes [B, ceil(N / 1920), Q]
 * int32.  Launches only. */
int64_t sopro_mimi_encode_workspace_bytes(const sopro_engine* e, int32_t B, int32_t N);
int sopro_mimi_encode(sopro_engine* e, void* workspace, const float* wav, int32_t B, int32_t N, int32_t* codes, void* stream);

/* ---- Mimi decode: tokens [B, T, Q] int32 -> wav [B, T * 1920] fp32 */
int64_t sopro_mimi_workspace_bytes(const sopro_engine* e, int32_t B, int32_t T);
int sopro_mimi_decode(sopro_engine* e, void* workspace, const int32_t* tokens, int32_t B, int32_t T, float* wav, void* stream);
/* Large batches are decoded in balanced row chunks of at most ~12800 frames each (SOPRO_MIMI_CHUNK_CELLS; a 64 x 400 decode in one
 * piece measured slower per utterance than two 32 x 400 ones and needs 148 GB of scratch): sopro_mimi_decode does that itself and
 * sopro_mimi_workspace_bytes sizes the workspace for ONE chunk, so every host gets it (round 5; it lived in the Python host).
 * sopro_mimi_chunk_rows: the rows per chunk sopro_mimi_decode uses for (B, T).
 * (the setting is read once per process).
 * sopro_mimi_decode_parts: ONE chunk (B <= sopro_mimi_chunk_rows(B, T) rows - more is refused -, workspace of
 * sopro_mimi_workspace_bytes(e, B, T) bytes or more) in two parts - 1: every
 * launch but the last, 2: the last launch alone, the only one that touches `wav` (the fused last SEANet level or its tail kernel), 3: both.
 * A host that replays part 1 from a recorded graph launches part 2 itself with the destination of THIS call: the decoder writes straight
 * into the caller's buffer and a recorded sequence holds no pointer of the caller's (src/sopro/codec/mimi.py:65-72: what decode_full
 * returns).  `tokens` is only read by part 1. */
int32_t sopro_mimi_chunk_rows(int32_t B, int32_t T);
int sopro_mimi_decode_parts(sopro_engine* e, void* workspace, const int32_t* tokens, int32_t B, int32_t T, float* wav, int32_t parts, void* stream);

/* ---- streaming decode (MimiStreamDecoder.decode_step's MimiModel.decode(..., decoder_past_key_values=...) call,
 * src/sopro/codec/mimi.py:152-156): one utterance, T frames at a time, the decoder transformer attending over the cached
 * keys / values of earlier calls.  The chunking policy around it (2-frame token overlap, cropping: mimi.py:131-181) is
 * pointer arithmetic and stays with the host (sopro_amd/codec.py MimiStreamDecoder shows it).  `kv` is a caller-owned
 * device buffer of sopro_mimi_stream_kv_bytes(e, cap_rows) bytes; the host-side fields are updated by the calls.
 * evict = 1: the cache keeps the last window-1 positions after each call (transformers 5.x sliding-window layers);
 * sopro_mimi_stream_trim drops the last `n` cached positions and switches to evict = 0, positions continuing from the
 * trimmed length: drop_cache_tail's legacy branch (mimi.py:92-103, transformers 4.57.6; quirk Q6). */
typedef struct sopro_mimi_stream_state {
  float* kv;          /* [layers][2 halves][cap_rows][2 * hidden] post-RoPE (k | v) rows */
  int32_t cap_rows;   /* >= window - 1 + 2 * (frames per call) while evict = 1; everything generated while evict = 0 */
  int32_t kv_len, pos, evict, half;
} sopro_mimi_stream_state;
int64_t sopro_mimi_stream_kv_bytes(const sopro_engine* e, int32_t cap_rows);
int sopro_mimi_stream_init(const sopro_engine* e, sopro_mimi_stream_state* st, void* kv, int32_t cap_rows);
int sopro_mimi_stream_trim(sopro_mimi_stream_state* st, int32_t n);
/* workspace: sopro_mimi_workspace_bytes(e, 1, T) */
int sopro_mimi_decode_stream(sopro_engine* e, void* workspace, sopro_mimi_stream_state* st, const int32_t* tokens, int32_t T, float* wav,
                             void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SOPRO_HIP_H */
