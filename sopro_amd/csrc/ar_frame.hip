// The launch sequence of ONE autoregressive frame (reference step: src/sopro/nn/generator.py:98-130 + the sampler,
// src/sopro/sampling.py:24-93, src/sopro/model.py:274-305) as a C function over the operator entry points.  This is the only
// place the sequence is written down: the Python host (sopro_amd/model.py, _ARPlan) and the stage-level C API
// (stages.hip, sopro_ar_run_graph) both describe their buffers in a sopro_ar_frame and call it - inside a stream capture when
// the frame is to be replayed from a hipGraph.  Host code only, no kernels.
#include "common.h"

#define FRM(call)                       \
  do {                                  \
    const int rc_ = (call);             \
    if (rc_ != 0) return rc_;           \
  } while (0)

extern "C" int sopro_ar_issue_frame(const sopro_ar_frame* fp, void* stream) {
  SOPRO_CHECK_ARG(fp != nullptr, "frame descriptor is NULL");
  const sopro_ar_frame& f = *fp;
  SOPRO_CHECK_ARG(f.n_layers >= 1 && f.n_layers <= SOPRO_AR_MAX_LAYERS && f.B > 0 && f.D == 384 && f.H == 4 && f.V1 > 1 && f.ksize >= 1,
                  "1..16 blocks, d_model 384, 4 heads");
  SOPRO_CHECK_ARG(f.x0 && f.xa && f.xb && f.part && f.u && f.xp && f.logits && f.head_w && f.head_b && f.st.step, "NULL buffer");
  SOPRO_CHECK_ARG(f.w_layout == 1 || f.w_layout == 2, "w_layout: 1 (sopro_pack_skinny_w) or 2 (sopro_pack_skinny_w_bf16)");
  SOPRO_CHECK_ARG(f.store_format == 0 || (f.store_format == 1 && f.w_layout == 2), "store_format: 0 (fp32 state), or 1 (bf16 rings / folded operands) in bf16 mode");
  SOPRO_CHECK_ARG(f.st.x_cur == f.x0, "the sampler must write the next frame's input where block 0 reads (st.x_cur == x0)");
  hipStream_t s = (hipStream_t)stream;
  const int B = f.B, D = f.D, H = f.H, KSL = 4 * D / 384;  // FF2 K slices
  const int64_t BD = (int64_t)B * D;
  SOPRO_CHECK_ARG(f.k_unfold == 0 || (f.k_unfold == 1 && f.qa && f.qpart), "k_unfold needs the qa / qpart buffers");
  // Residual stream = a base buffer plus (optionally) three pending partial buffers that the next kernel adds while it stages
  // its input: the K-slices of a feed-forward output (slice 0 carries bias + residual) or the per-head outputs of a
  // cross-attention block (head 0 carries the residual).
  const float* base = f.x0;
  const float* pend = nullptr;
  for (int i = 0; i < f.n_layers; ++i) {
    const sopro_ar_block& b = f.blk[i];
    SOPRO_CHECK_ARG(b.glu_w && b.glu_b && b.dw_w && b.dw_b && b.ff1_w && b.ff1_b && b.ff2_w && b.ff2_b && b.ring && b.dil >= 1, "block has NULL operands");
    float* out = (i % 2 == 0) ? f.xa : f.xb;
    sopro_skinny_args a;
    // RMSNorm -> GLU -> ring write -> dilated taps -> +x   (src/sopro/nn/blocks.py:150-157, 76-110)
    memset(&a, 0, sizeof(a));
    a.X = base; a.ldx = D; a.W = (const float*)b.glu_w; a.ldw = D; a.w_layout = f.w_layout; a.bias = b.glu_b;
    a.Y = out; a.ldy = D; a.ldr = D;
    a.ring = b.ring; a.dw_w = b.dw_w; a.dw_b = b.dw_b; a.step = f.st.step;
    a.Xp = pend; a.xp_stride = BD; a.np = pend ? 3 : 0;
    a.eps = f.eps; a.B = B; a.N = 2 * D; a.K = D; a.epilogue = SOPRO_EPI_GLU_DW;
    a.ring_len = (f.ksize - 1) * b.dil + 1; a.ring_bcap = B; a.dil = b.dil; a.ksize = f.ksize; a.rms_norm = 1;
    a.ring_format = f.store_format;
    a.mt = f.tile_glu >> 4; a.nt = f.tile_glu & 15;
    FRM(sopro_skinny_f32(&a, s));
    // RMSNorm -> Linear -> GELU (blocks.py:158-160)
    memset(&a, 0, sizeof(a));
    a.X = out; a.ldx = D; a.W = (const float*)b.ff1_w; a.ldw = D; a.w_layout = f.w_layout; a.bias = b.ff1_b;
    a.Y = f.u; a.ldy = 4 * D; a.ldr = 4 * D; a.eps = f.eps; a.B = B; a.N = 4 * D; a.K = D; a.epilogue = SOPRO_EPI_GELU; a.rms_norm = 1;
    a.mt = f.tile_ff1 >> 4; a.nt = f.tile_ff1 & 15;
    const bool uk = b.xattn && f.k_unfold;
    if (uk) {  // the `Wq' out` part of the following cross-attention block's raw query: D more columns on the rows FF1 stages anyway
      SOPRO_CHECK_ARG(b.qa_w && b.qu_w && b.q_b, "k_unfold: block without its query operands");
      a.aux_tiles = D / 16; a.aux_W = (const float*)b.qa_w; a.aux_Y = f.qa; a.aux_ldy = D; a.aux_flags = 3;  // no row scale, no GELU, no bias
    }
    FRM(sopro_skinny_f32(&a, s));
    // Linear 4D -> D + residual as K-slices on 4x the workgroups (blocks.py:161-162)
    memset(&a, 0, sizeof(a));
    a.X = f.u; a.ldx = 4 * D; a.W = (const float*)b.ff2_w; a.ldw = 4 * D; a.w_layout = f.w_layout; a.bias = b.ff2_b;
    a.Y = f.part; a.ldy = D; a.R = out; a.ldr = D; a.eps = f.eps; a.B = B; a.N = D; a.K = 4 * D; a.epilogue = SOPRO_EPI_RES;
    a.ksplit = 1; a.y_part_stride = BD;
    a.mt = f.tile_ff2 >> 4; a.nt = f.tile_ff2 & 15;
    if (uk) {  // ... and its `(Wq' W2) u + Wq' b2` part as K-slices, slice 0 carrying qa: q_raw = Wq' (out + b2 + W2 u) = Wq' x
      a.aux_tiles = D / 16; a.aux_W = (const float*)b.qu_w; a.aux_bias = b.q_b; a.aux_R = f.qa; a.aux_ldr = D; a.aux_Y = f.qpart; a.aux_ldy = D;
      a.aux_y_part_stride = BD;
    }
    FRM(sopro_skinny_f32(&a, s));
    SOPRO_CHECK_ARG(KSL == 4, "the partial-sum hand-over is written for four K slices");
    base = f.part; pend = f.part + BD;
    if (b.xattn) {
      // cached text cross-attention, projections folded into the cached operands (src/sopro/nn/text.py:85-132)
      SOPRO_CHECK_ARG(b.kp && b.vp && f.klens && f.S_cap > 0, "cross-attention block without its folded operands");
      sopro_xattn_args x;
      memset(&x, 0, sizeof(x));
      x.X = base; x.ldx = D; x.Xp = pend; x.xp_stride = BD; x.np = 3;
      x.Kp = b.kp; x.Vp = b.vp; x.klens = f.klens; x.Y = f.xp; x.y_part_stride = BD;
      x.eps = f.eps; x.gate = b.gate; x.scale = 1.0f / sqrtf((float)(D / H));
      x.B = B; x.H = H; x.D = D; x.S_cap = f.S_cap; x.kv_format = f.store_format;
      if (uk) { x.k_unfolded = 1; x.Qp = f.qpart; x.qp_stride = BD; x.nqp = KSL; }
      FRM(sopro_xattn_step_f32(&x, s));
      base = f.xp; pend = f.xp + BD;
    }
  }
  sopro_skinny_args a;
  memset(&a, 0, sizeof(a));
  a.X = base; a.ldx = D; a.W = (const float*)f.head_w; a.ldw = D; a.w_layout = f.w_layout; a.bias = f.head_b;
  a.Y = f.logits; a.ldy = f.V1; a.ldr = f.V1; a.Xp = pend; a.xp_stride = BD; a.np = pend ? 3 : 0;
  a.eps = f.eps; a.B = B; a.N = f.V1; a.K = D; a.rms_norm = 1;
  a.mt = f.tile_head >> 4; a.nt = f.tile_head & 15;
  FRM(sopro_skinny_f32(&a, s));
  // the sampler writes the next frame's input into st.x_cur == x0, where block 0 reads
  return sopro_ar_sample(&f.st, f.logits, f.V1, s);
}
