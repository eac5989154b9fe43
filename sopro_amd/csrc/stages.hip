// Stage-level entry points (include/sopro_hip.h, "Stage-level entry points"): the launch sequences of the AR loop, the NAR
// refinement and the Mimi decoder as host C++ over the operator entry points of this library.  No kernels here.  The
// sequences are the ones sopro_amd/model.py (_ARRun / _ARPlan.issue_step / _nar_issue) and sopro_amd/codec.py
// (_decode_issue / _transformer / _seanet_act) issue; tests/test_gpu_stages.py checks the two hosts against each other
// and against the oracle.
#include <stdlib.h>

#include <algorithm>
#include <map>
#include <string>
#include <vector>

#include "common.h"

namespace {

struct Ten {
  const void* p = nullptr;
  int64_t shape[4] = {0, 0, 0, 0};
  int ndim = 0;
  const float* f() const { return reinterpret_cast<const float*>(p); }
};

struct Wt {  // weight operand of a contraction: fp32 [N, K] row-major and / or its packed 16-bit pieces
  const float* f32 = nullptr;
  const void* packed = nullptr;
  const void* rows = nullptr;  // pieces == 2: the matrix as split-form rows too (sopro_pack_w_rows_bf16), for the long-K form (gemm_8p.hip)
  int pieces = 0;
  bool f16 = false;        // two fp16 pieces (sopro_gemm_f16x3) instead of bf16 ones
  float acc_scale = 1.f;   // f16: 1 / (activation scale * pack scale)
};

constexpr float RMS_EPS = 1e-6f;  // src/sopro/nn/blocks.py:27

struct Carver {  // carves a caller-provided workspace into 256-byte aligned buffers
  char* base;
  size_t off = 0;
  explicit Carver(void* b) : base(reinterpret_cast<char*>(b)) {}
  template <class T>
  T* take(size_t n) {
    T* p = base ? reinterpret_cast<T*>(base + off) : nullptr;
    off += (n * sizeof(T) + 255) & ~size_t(255);
    return p;
  }
};

struct ArPlan {
  int B = 0, S = 0, S_cap = 0, Tar = 0;
  void* ws = nullptr;
  float *cond, *x[4], *part, *u, *logits, *kp[16], *vp[16], *xp, *params, *rings[16], *nkv, *kvd;
  float *qa = nullptr, *qpart = nullptr;       // unfolded keys (fp32 frame): the raw query's `Wq' out` part and its four K-slice partials
  bool k_unfold = false;
  float *fold_k = nullptr, *fold_v = nullptr;  // bf16 mode: fp32 scratch the text operands are folded into before they are rounded to bf16
  int32_t *klens, *hist, *ctr, *first_eos, *stop_t, *recent;
  uint32_t* nonce;
  uint32_t* key;
  int32_t* row_step;
  sopro_ar_state st;
  void* graph = nullptr;
  int graph_B = 0, graph_S_cap = 0, graph_Tar = 0;
  void* graph_ws = nullptr;
};

}  // namespace

struct sopro_engine {
  sopro_engine_cfg c;
  std::map<std::string, Ten> t;
  std::map<std::string, Wt> w;       // contraction operands by pack name
  std::map<std::string, const float*> sk;  // AR-step weights in skinny fragment order
  std::vector<void*> owned;
  bool final = false;
  bool has_ar = false, has_nar = false, has_mimi = false, has_cond = false, has_enc = false;  // families whose tensors were given before sopro_engine_finalize
  int32_t *q_col = nullptr, *q_off = nullptr;  // conditioning: codebook columns 0..Q-1 and their table offsets q * V
  // NAR constants
  std::vector<int32_t*> nar_cols, nar_offs;
  std::vector<float*> nar_cw, ad_mul, ad_add;
  std::vector<int> nar_known;
  // Mimi constants
  int32_t *sem_col = nullptr, *sem_off = nullptr, *ac_col = nullptr, *ac_off = nullptr;
  float* ones = nullptr;
  ArPlan ar;
  int ar_tiles[4] = {0, 0, 0, 0};  // workgroup shapes of the AR-step stages (sopro_engine_set_ar_tiles)
};

namespace {

#define STG(call)                       \
  do {                                  \
    const int rc_ = (call);             \
    if (rc_ != 0) return rc_;           \
  } while (0)

int need(const sopro_engine* e, const std::string& name, const Ten** out, int ndim = 0) {
  auto it = e->t.find(name);
  if (it == e->t.end() || !it->second.p) {
    sopro_set_error("stage API: tensor '%s' was not given to sopro_engine_set_tensor", name.c_str());
    return -2;
  }
  if (ndim && it->second.ndim != ndim) {
    sopro_set_error("stage API: tensor '%s' has %d dimensions, expected %d", name.c_str(), it->second.ndim, ndim);
    return -2;
  }
  *out = &it->second;
  return 0;
}

const Wt& WT(const sopro_engine* e, const std::string& key) {  // a packed operand made by sopro_engine_finalize (read-only: lanes share an engine)
  static const Wt none;
  auto it = e->w.find(key);
  return it == e->w.end() ? none : it->second;
}

const float* F(const sopro_engine* e, const std::string& name) {
  auto it = e->t.find(name);
  return it == e->t.end() ? nullptr : it->second.f();
}

template <class T>
int dev_alloc(sopro_engine* e, size_t n, T** out) {
  void* p = nullptr;
  SOPRO_HIP(hipMalloc(&p, n * sizeof(T)));
  e->owned.push_back(p);
  *out = reinterpret_cast<T*>(p);
  return 0;
}

template <class T>
int dev_upload(sopro_engine* e, const std::vector<T>& h, T** out) {
  STG(dev_alloc(e, h.size(), out));
  SOPRO_HIP(hipMemcpy(*out, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
  return 0;
}

constexpr int F16X2 = 22;  // pack_pieces: two fp16 pieces (the NAR contractions, sopro_gemm_f16x3)
const char* const NAR_SAFE = "#x6";  // key suffix of the refinement's six-pass twins (sopro_nar_io.safe)

// [N, K] fp32 device matrix (optionally with its columns scaled by a device vector: an RMSNorm weight folded in) -> bf16 pieces
int pack_pieces(sopro_engine* e, const std::string& key, const std::string& out_key, int pieces, const char* fold_vec, hipStream_t s,
                bool with_rows = false) {
  const Ten* t;
  STG(need(e, key, &t, 2));
  const int N = (int)t->shape[0], K = (int)t->shape[1];
  const float* src = t->f();
  if (fold_vec) {
    const Ten* v;
    STG(need(e, fold_vec, &v));
    std::vector<float> hw((size_t)N * K), hv(K);
    SOPRO_HIP(hipMemcpy(hw.data(), src, hw.size() * 4, hipMemcpyDeviceToHost));
    SOPRO_HIP(hipMemcpy(hv.data(), v->p, (size_t)K * 4, hipMemcpyDeviceToHost));
    for (int n = 0; n < N; ++n)
      for (int k = 0; k < K; ++k) hw[(size_t)n * K + k] *= hv[k];  // (x * rstd * w_norm) W^T == rstd * x (W * w_norm)^T
    float* d;
    STG(dev_upload(e, hw, &d));
    src = d;
  }
  const bool f16 = pieces == F16X2;
  if (f16) pieces = 2;
  const int64_t bytes = sopro_packed_w_bytes(N, K, pieces);
  char* dst;
  STG(dev_alloc(e, (size_t)bytes, &dst));
  Wt w;
  if (f16) {
    // scale the matrix by a power of two so that max |w| lands in [2^13, 2^14) (see sopro_pack_w_f16x2)
    std::vector<float> hw((size_t)N * K);
    SOPRO_HIP(hipMemcpy(hw.data(), src, hw.size() * 4, hipMemcpyDeviceToHost));
    float amax = 0.f;
    for (float v : hw) amax = fmaxf(amax, fabsf(v));
    int ex = 0;
    (void)frexpf(amax, &ex);
    const float wscale = amax > 0.f ? ldexpf(1.0f, 13 - ex + 1) : 1.0f;
    STG(sopro_pack_w_f16x2(src, K, N, K, wscale, dst, s));
    w.f16 = true;
    w.acc_scale = 1.0f / (sopro_f16x3_a_scale() * wscale);
  } else {
    STG(sopro_pack_w_bf16(src, K, N, K, pieces, dst, s));
  }
  if (with_rows && pieces == 2 && !f16 && (K & 31) == 0) {  // the long-K form's weight operand (rows must start on 128-byte boundaries)
    char* rows;
    STG(dev_alloc(e, (size_t)sopro_packed_w_rows_bytes(N, K) + 128, &rows));
    rows += (128 - (reinterpret_cast<uintptr_t>(rows) & 127u)) & 127u;
    STG(sopro_pack_w_rows_bf16(src, K, N, K, rows, s));
    w.rows = rows;
  }
  w.f32 = fold_vec ? nullptr : t->f();
  w.packed = dst;
  w.pieces = pieces;
  e->w[out_key] = w;
  return 0;
}

// out[n] = sum_k W[n, k] v[k] (+ b[n]) as a device vector registered under out_key: the bias a contraction behind a fused LayerNorm
// carries for the norm's bias vector (W (w * xhat + b) = (W * w) xhat + W b)
int fold_bias(sopro_engine* e, const std::string& w_key, const std::string& v_key, const std::string& out_key) {
  const Ten *t, *v;
  STG(need(e, w_key, &t, 2));
  STG(need(e, v_key, &v));
  const int N = (int)t->shape[0], K = (int)t->shape[1];
  std::vector<float> hw((size_t)N * K), hv(K), ho(N);
  SOPRO_HIP(hipMemcpy(hw.data(), t->p, hw.size() * 4, hipMemcpyDeviceToHost));
  SOPRO_HIP(hipMemcpy(hv.data(), v->p, (size_t)K * 4, hipMemcpyDeviceToHost));
  for (int n = 0; n < N; ++n) {
    double acc = 0.0;
    for (int k = 0; k < K; ++k) acc += (double)hw[(size_t)n * K + k] * hv[k];
    ho[n] = (float)acc;
  }
  float* d;
  STG(dev_upload(e, ho, &d));
  Ten o;
  o.p = d; o.shape[0] = N; o.ndim = 1;
  e->t[out_key] = o;
  return 0;
}

int plain(sopro_engine* e, const std::string& key) {
  const Ten* t;
  STG(need(e, key, &t));
  Wt w;
  w.f32 = t->f();
  e->w[key] = w;
  return 0;
}

// Split-K scratch of a stage call (carved from the caller's workspace; belongs to the call's stream): problems with too few output
// tiles to occupy the chip (streaming chunks, batch 1) run their K loop on `ksplit` workgroups per tile (sopro_gemm_split_ext).
constexpr size_t SPLITK_WS_BYTES = (size_t)32 << 20;
constexpr int SPLITK_TICKETS = 1024;
struct SplitK {
  float* ws = nullptr;
  int32_t* tickets = nullptr;
};

// K slices for a problem of M x N x K on tiles of the kernel family `pieces` selects (3: six-pass / f16 three-pass rules,
// 1: one-pass, 2: three-pass).  The split costs ~10 us (device-scope release / acquire around the ticket): it pays from ~32
// K-steps up; a slice costs ~0.9 us per K-step, the reducing workgroup ~0.36 us per slice: ks ~ sqrt(2.5 * K-steps).
int auto_ksplit(int M, int N, int K, int pieces, int epi) {
  int bm, bn;
  if (pieces == 3) {
    bm = 64; bn = epi == SOPRO_EPI_GLU ? 128 : 64;
  } else if (pieces == 1) {
    const bool small = N <= 64 || M <= 64 || (int64_t)((M + 127) / 128) * ((N + 127) / 128) < 256;
    if (small) { bm = 64; bn = epi == SOPRO_EPI_GLU ? 128 : 64; } else { bm = bn = 128; }
  } else {
    if (N <= 64 || M <= 64) bm = bn = 64; else bm = bn = 128;
  }
  const int64_t tiles = (int64_t)((M + bm - 1) / bm) * ((N + bn - 1) / bn);
  const int kt = (K + 31) / 32;
  if (tiles >= 96 || kt < 32 || tiles > SPLITK_TICKETS) return 1;
  int ks = (int)lrint(sqrt(2.5 * kt));
  if (ks > 16) ks = 16;
  if (ks > 512 / (int)tiles) ks = 512 / (int)tiles;
  if (ks < 1) ks = 1;
  while (ks > 1 && (size_t)ks * tiles * bm * bn * 4 > SPLITK_WS_BYTES) --ks;
  return ks;
}

struct G {  // one contraction: mirrors sopro_amd.hip.gemm's keyword arguments
  int M = 0, N = 0, K = 0;
  int64_t lda = -1, ldc = -1, ldr = -1, ldw = -1;
  const float *bias = nullptr, *R = nullptr, *scale = nullptr, *pro_vec = nullptr;
  int epi = SOPRO_EPI_NONE, pro = SOPRO_PRO_NONE, rows_per_seg = -1;
  int64_t a_seg = 0, c_seg = 0, r_seg = 0;
  int c_mode = 0;
  int a_fmt = 0;  // sopro_gemm_split_ext.a_format (2: bf16 rows - the bf16 mode's SEANet flow)
  float* C2 = nullptr;
  int64_t ldc2 = -1, c2_seg = 0;
  float rms_eps = 0.f;
  const SplitK* sk = nullptr;  // non-NULL: few-row problems may run split-K on this scratch
  int32_t* range_events = nullptr;  // f16 operands: the call's range-event word (sopro_gemm_split_ext.range_events)
  const float *rope_cos = nullptr, *rope_sin = nullptr;  // epi = SOPRO_EPI_ROPE (sopro_gemm_split_ext.rope_*)
  int rope_cols = 0, rope_dh = 0, rope_pos0 = 0, rope_rps = 0;
  const float* ln_stats = nullptr;  // fused LayerNorm of the A rows (sopro_gemm_split_ext.ln_stats; ln_eps = the norm's eps)
  float* ln_stats_out = nullptr;    // EPI_RES: the updated stream's statistics for the next contraction
  float ln_eps = 0.f;
};

// attention launch with its timing scope: 4 * dh flops per visible (query, key) pair
int attend(const sopro_attn_args& a, hipStream_t s, int split_passes = 0) {
  double pairs = (double)a.Tq * a.Tk;
  if (a.causal) {
    pairs = 0;
    for (int q = 0; q < a.Tq; ++q) {
      const int hi = std::min(a.Tk - 1, a.q_pos0 + q - a.k_pos0), lo = std::max(0, a.q_pos0 + q - a.window + 1 - a.k_pos0);
      pairs += std::max(0, hi - lo + 1);
    }
  }
  sopro_prof_scope prof(split_passes ? "attention_split_kernel" : "attention_kernel", 4.0 * a.dh * pairs * a.B * a.H, s);
  if (split_passes) return sopro_attention_split_bf16(&a, split_passes, s);
  return sopro_attention_f32(&a, s);
}

int gemm(hipStream_t s, const float* A, const Wt& w, const float* w_f32_override, float* C, const G& o) {
  const int n_out = o.epi == SOPRO_EPI_GLU ? o.N / 2 : o.N;
  sopro_gemm_args g;
  memset(&g, 0, sizeof(g));
  g.A = A;
  g.lda = o.lda < 0 ? o.K : o.lda;
  g.a_seg_stride = o.a_seg;
  g.W = w_f32_override ? w_f32_override : w.f32;
  g.ldw = o.ldw < 0 ? o.K : o.ldw;
  g.bias = o.bias;
  g.C = C;
  g.ldc = o.ldc < 0 ? n_out : o.ldc;
  g.c_seg_stride = o.c_seg;
  g.R = o.R;
  g.ldr = o.ldr < 0 ? n_out : o.ldr;
  g.r_seg_stride = o.r_seg;
  g.scale = o.scale;
  g.pro_vec = o.pro_vec;
  g.M = o.M; g.N = o.N; g.K = o.K;
  g.rows_per_seg = o.rows_per_seg < 0 ? o.M : o.rows_per_seg;
  g.prologue = o.pro; g.epilogue = o.epi;
  const bool split = !w_f32_override && w.packed;
  sopro_prof_scope prof(!split ? "gemm_f32_kernel" : w.f16 ? "gemm_f16x3_kernel" : w.pieces == 3 ? "gemm_bf16x6_kernel"
                        : w.pieces == 1 ? "gemm_bf16x1_kernel" : "gemm_bf16x3_kernel", 2.0 * o.M * o.N * o.K, s);
  if (split) {
    sopro_gemm_split_ext x;
    memset(&x, 0, sizeof(x));
    x.c_mode = o.c_mode;
    x.a_format = o.a_fmt;
    x.C2 = o.C2;
    x.ldc2 = o.ldc2 < 0 ? n_out : o.ldc2;
    x.c2_seg_stride = o.c2_seg;
    if (o.rms_eps > 0.f) { x.rms_norm = 1; x.rms_eps = o.rms_eps; }
    x.rope_cos = o.rope_cos; x.rope_sin = o.rope_sin; x.rope_cols = o.rope_cols; x.rope_dh = o.rope_dh; x.rope_pos0 = o.rope_pos0;
    x.rope_rows_per_seg = o.rope_rps;
    x.ln_stats = o.ln_stats; x.ln_stats_out = o.ln_stats_out;
    if (o.ln_stats) x.rms_eps = o.ln_eps;
    static const bool no_splitk = getenv("SOPRO_NO_SPLITK") && getenv("SOPRO_NO_SPLITK")[0] == '1';  // developer switch
    if (!no_splitk && o.sk && o.sk->ws && o.rms_eps <= 0.f && o.c_mode != 5) {
      const int ks = auto_ksplit(o.M, o.N, o.K, w.f16 ? 3 : w.pieces, o.epi);
      if (ks > 1) {
        x.ksplit = ks; x.n_tickets = SPLITK_TICKETS; x.ws = o.sk->ws; x.ws_bytes = (int64_t)SPLITK_WS_BYTES; x.tickets = o.sk->tickets;
      }
    }
    if (w.f16) {
      x.acc_scale = w.acc_scale;
      x.range_events = o.range_events;
      return sopro_gemm_f16x3(&g, w.packed, &x, s);
    }
    if (w.pieces == 3) return sopro_gemm_bf16x6(&g, w.packed, &x, s);
    if (w.pieces == 1) return sopro_gemm_bf16x1(&g, w.packed, &x, s);
    if (w.rows) {  // long K, many tiles, split-form A: the 256 x 256 LDS-DMA form (same results bit for bit)
      sopro_gemm_split_ext x8 = x;
      x8.ksplit = 0;
      if (sopro_gemm_8p_takes(&g, &x8)) return sopro_gemm_bf16x3_8p(&g, w.rows, &x8, s);
    }
    return sopro_gemm_bf16x3(&g, w.packed, &x, s);
  }
  if (!g.W) {
    sopro_set_error("stage API: a contraction has neither an fp32 nor a packed weight");
    return -2;
  }
  return sopro_gemm_f32(&g, s);
}

int norm(hipStream_t s, const float* x, float* out, const float* w, int rows, int C, float eps, int kind = SOPRO_NORM_RMS,
         const float* b = nullptr, const float* mul = nullptr, const float* add = nullptr, int rows_per_seg = -1, int64_t x_seg = 0) {
  return sopro_norm_f32(x, C, x_seg, out, C, w, b, mul, add, rows, rows_per_seg < 0 ? rows : rows_per_seg, C, eps, kind, s);
}

}  // namespace

extern "C" {

int sopro_engine_create(const sopro_engine_cfg* cfg, sopro_engine** out) {
  SOPRO_CHECK_ARG(cfg && out, "NULL argument");
  SOPRO_CHECK_ARG(cfg->d_model == 384 && cfg->n_layers_ar >= 1 && cfg->n_layers_ar <= 16 && cfg->n_layers_nar >= 1 && cfg->n_layers_nar <= 16,
                  "d_model must be 384 (kernel family), 1..16 layers");
  SOPRO_CHECK_ARG(cfg->n_stages >= 1 && cfg->n_stages <= 8 && cfg->num_codebooks <= 64 && cfg->mimi_n_ratios >= 1 && cfg->mimi_n_ratios <= 8,
                  "1..8 NAR stages, <= 64 codebooks, 1..8 decoder ratios");
  sopro_engine* e = new sopro_engine();
  e->c = *cfg;
  *out = e;
  return 0;
}

int sopro_engine_set_tensor(sopro_engine* e, const char* name, const void* dev_ptr, const int64_t* shape, int32_t ndim) {
  SOPRO_CHECK_ARG(e && name && dev_ptr && shape && ndim >= 1 && ndim <= 4, "bad arguments");
  SOPRO_CHECK_ARG(!e->final, "the engine is finalized");
  Ten t;
  t.p = dev_ptr;
  t.ndim = ndim;
  for (int i = 0; i < ndim; ++i) t.shape[i] = shape[i];
  e->t[name] = t;
  return 0;
}

int sopro_engine_upload_tensor(sopro_engine* e, const char* name, const float* host_ptr, const int64_t* shape, int32_t ndim, void* stream) {
  SOPRO_CHECK_ARG(e && name && host_ptr && shape && ndim >= 1 && ndim <= 4, "bad arguments");
  SOPRO_CHECK_ARG(!e->final, "the engine is finalized");
  size_t n = 1;
  for (int i = 0; i < ndim; ++i) {
    SOPRO_CHECK_ARG(shape[i] > 0, "empty dimension");
    n *= (size_t)shape[i];
  }
  float* d = nullptr;
  STG(dev_alloc(e, n, &d));  // engine-owned: freed by sopro_engine_destroy
  (void)stream;              // (pageable host memory: the copy is synchronous either way)
  SOPRO_HIP(hipMemcpy(d, host_ptr, n * sizeof(float), hipMemcpyHostToDevice));
  return sopro_engine_set_tensor(e, name, d, shape, ndim);
}

int sopro_engine_set_ar_tiles(sopro_engine* e, int32_t glu, int32_t ff1, int32_t ff2, int32_t head) {
  SOPRO_CHECK_ARG(e != nullptr, "engine is NULL");
  e->ar_tiles[0] = glu; e->ar_tiles[1] = ff1; e->ar_tiles[2] = ff2; e->ar_tiles[3] = head;
  if (e->ar.graph) {  // the recorded frame holds the old shapes
    (void)sopro_graph_destroy(e->ar.graph);
    e->ar.graph = nullptr;
  }
  return 0;
}

int sopro_engine_destroy(sopro_engine* e) {
  if (!e) return 0;
  if (e->ar.graph) (void)sopro_graph_destroy(e->ar.graph);
  for (void* p : e->owned) (void)hipFree(p);
  delete e;
  return 0;
}

int sopro_engine_finalize(sopro_engine* e, void* stream) {
  SOPRO_CHECK_ARG(e && !e->final, "NULL or already finalized engine");
  hipStream_t s = (hipStream_t)stream;
  const sopro_engine_cfg& c = e->c;
  const Ten* t;
  const bool bf16 = c.precision == 1;
  // A stage family is present when its marker tensor was given; a family that is present must be complete.
  e->has_ar = e->t.count("ar.head.w") != 0;
  e->has_nar = e->t.count("nar.pre.w") != 0;
  e->has_mimi = e->t.count("rvq_proj.w") != 0;
  e->has_cond = e->t.count("text_enc.embed") != 0;
  e->has_enc = e->t.count("enc.conv0.w") != 0;
  if (!e->has_ar && !e->has_nar && !e->has_mimi && !e->has_cond && !e->has_enc) {
    // report the first tensor of the first stage by name (what a host that forgot sopro_engine_set_tensor wants to read)
    STG(need(e, "ar.blocks.0.glu.w", &t, 2));
  }
  // ---- AR step: skinny fragment order for the four big projections of every block and the head (bf16 mode: bf16 weights)
  if (e->has_ar) {
    auto pack_skinny = [&](const std::string& key, int glu) -> int {
      STG(need(e, key, &t, 2));
      const int N = (int)t->shape[0], K = (int)t->shape[1];
      const int64_t nfl = sopro_skinny_packed_floats(N, K, glu);
      float* d;
      STG(dev_alloc(e, (size_t)(bf16 ? nfl / 2 : nfl), &d));
      if (bf16) STG(sopro_pack_skinny_w_bf16(t->f(), K, N, K, glu, d, s));
      else STG(sopro_pack_skinny_w(t->f(), K, N, K, glu, d, s));
      e->sk[key] = d;
      return 0;
    };
    for (int i = 0; i < c.n_layers_ar; ++i) {
      const std::string p = "ar.blocks." + std::to_string(i);
      STG(pack_skinny(p + ".glu.w", 1));
      STG(pack_skinny(p + ".ff1.w", 0));
      STG(pack_skinny(p + ".ff2.w", 0));
      for (const char* nm : {".glu.b", ".dw.w", ".dw.b", ".ff1.b", ".ff2.b"}) STG(need(e, p + nm, &t));
      if (c.ar_xattn[i]) {
        const std::string pa = "ar.x_attns." + std::to_string(i);
        for (const char* nm : {".nkv.weight", ".kv.w", ".q.wT", ".o.w"}) STG(need(e, pa + nm, &t));
        // unfolded keys (sopro_ar_frame.k_unfold; the fp32 frame): the query operands ride on the feed-forward launches
        if (!bf16 && e->t.count(pa + ".qa.w") && e->t.count(pa + ".qu.w") && e->t.count(pa + ".q.b")) {
          STG(pack_skinny(pa + ".qa.w", 0));
          STG(pack_skinny(pa + ".qu.w", 0));
        }
      }
    }
    STG(pack_skinny("ar.head.w", 0));
    STG(need(e, "ar.head.b", &t));
    STG(need(e, "cb_embed", &t, 2));
  }
  // ---- NAR: two fp16 pieces / three passes (22 mantissa bits), the two RMSNorm weights of a block folded into the
  // projections they feed; bf16 mode: one bf16 piece
  if (e->has_nar) {
    const int np = bf16 ? 1 : F16X2;
    for (int i = 0; i < c.n_layers_nar; ++i) {
      const std::string p = "nar.blocks." + std::to_string(i);
      STG(pack_pieces(e, p + ".glu.w", p + ".glu.wn", np, (p + ".norm.weight").c_str(), s));
      STG(pack_pieces(e, p + ".ff1.w", p + ".ff1.wn", np, (p + ".ff.norm.weight").c_str(), s));
      STG(pack_pieces(e, p + ".ff2.w", p + ".ff2.w", np, nullptr, s));
      if (!bf16) {  // the range guard's fallback operands: three bf16 pieces / six passes (fp32's exponent range)
        STG(pack_pieces(e, p + ".glu.w", p + ".glu.wn" + NAR_SAFE, 3, (p + ".norm.weight").c_str(), s));
        STG(pack_pieces(e, p + ".ff1.w", p + ".ff1.wn" + NAR_SAFE, 3, (p + ".ff.norm.weight").c_str(), s));
        STG(pack_pieces(e, p + ".ff2.w", p + ".ff2.w" + NAR_SAFE, 3, nullptr, s));
      }
      for (const char* nm : {".glu.b", ".dw.w", ".dw.b", ".ff1.b", ".ff2.b"}) STG(need(e, p + nm, &t));
    }
    STG(pack_pieces(e, "nar.pre.w", "nar.pre.w", np, nullptr, s));
    if (!bf16) STG(pack_pieces(e, "nar.pre.w", std::string("nar.pre.w") + NAR_SAFE, 3, nullptr, s));
    const char* stage_names[8] = {"B", "C", "D", "E", "F", "G", "H", "I"};
    std::vector<int> known = {0};
    for (int sgi = 0; sgi < c.n_stages; ++sgi) {
      const std::string hk = std::string("nar.heads.") + stage_names[sgi];
      STG(pack_pieces(e, hk + ".w", hk + ".w", np, nullptr, s));
      if (!bf16) STG(pack_pieces(e, hk + ".w", hk + ".w" + NAR_SAFE, 3, nullptr, s));
      STG(need(e, hk + ".b", &t));
      // prev = sum_j softmax(w[known])_j * E[cb_j * V + tok_j]   (src/sopro/nn/embeddings.py:77-112)
      std::vector<int32_t> cols(known.begin(), known.end()), offs;
      std::vector<float> cw;
      float mx = -1e30f, sum = 0.f;
      for (int k : known) mx = fmaxf(mx, c.nar_prev_cb_weights[k]);
      for (int k : known) { cw.push_back(expf(c.nar_prev_cb_weights[k] - mx)); sum += cw.back(); }
      for (float& v : cw) v /= sum;
      for (int k : known) offs.push_back(k * c.codebook_size);
      int32_t *dc, *dofs;
      float* dw;
      STG(dev_upload(e, cols, &dc));
      STG(dev_upload(e, offs, &dofs));
      STG(dev_upload(e, cw, &dw));
      e->nar_cols.push_back(dc); e->nar_offs.push_back(dofs); e->nar_cw.push_back(dw);
      e->nar_known.push_back((int)known.size());
      for (int j = 0; j < c.stage_n_cb[sgi]; ++j) known.push_back(c.stage_first_cb[sgi] + j);
    }
    for (const char* nm : {"nar.norm.weight", "nar.pre.b", "nar.stage_emb", "nar.adapter.norm.weight", "nar.adapter.mlp.0.w", "nar.adapter.mlp.0.b",
                           "nar.adapter.mlp.2.w", "nar.adapter.mlp.2.b", "cb_embed"})
      STG(need(e, nm, &t));
    {  // per-stage adapter coefficients (1 + tanh g, tanh b): input independent (src/sopro/nn/nar.py:25-32)
      const int ns = c.n_stages, D = c.d_model;
      float *h, *gb;
      STG(dev_alloc(e, (size_t)ns * 256, &h));
      STG(dev_alloc(e, (size_t)ns * 2 * D, &gb));
      G o; o.M = ns; o.N = 256; o.K = D; o.bias = F(e, "nar.adapter.mlp.0.b"); o.epi = SOPRO_EPI_GELU;
      Wt w0; w0.f32 = F(e, "nar.adapter.mlp.0.w");
      STG(gemm(s, F(e, "nar.stage_emb"), w0, nullptr, h, o));
      G o2; o2.M = ns; o2.N = 2 * D; o2.K = 256; o2.bias = F(e, "nar.adapter.mlp.2.b");
      Wt w2; w2.f32 = F(e, "nar.adapter.mlp.2.w");
      STG(gemm(s, h, w2, nullptr, gb, o2));
      for (int sg = 0; sg < ns; ++sg) {
        float *mul, *add;
        STG(dev_alloc(e, (size_t)D, &mul));
        STG(dev_alloc(e, (size_t)D, &add));
        STG(sopro_tanh_affine_f32(gb + (size_t)sg * 2 * D, mul, 1.0f, 1.0f, D, s));
        STG(sopro_tanh_affine_f32(gb + (size_t)sg * 2 * D + D, add, 0.0f, 1.0f, D, s));
        e->ad_mul.push_back(mul); e->ad_add.push_back(add);
      }
    }
  }
  // ---- conditioning + reference preparation: the two encoders' contractions on three bf16 pieces / six passes (24 mantissa
  // bits: they feed the AR loop's conditioning), their RMSNorm weights folded into the projections they feed
  if (e->has_cond) {
    SOPRO_CHECK_ARG(c.n_layers_text >= 1 && c.ref_enc_layers >= 1 && c.ref_xattn_layers >= 1 && c.ref_xattn_layers <= 8 && c.ref_xattn_heads >= 1 &&
                        c.sv_student_dim >= 1 && c.enc_kernel >= 1, "conditioning fields of sopro_engine_cfg are not set");
    auto enc_block = [&](const std::string& p) -> int {
      STG(pack_pieces(e, p + ".glu.w", p + ".glu.wn", 3, (p + ".norm.weight").c_str(), s));
      STG(pack_pieces(e, p + ".ff1.w", p + ".ff1.wn", 3, (p + ".ff.norm.weight").c_str(), s));
      STG(pack_pieces(e, p + ".ff2.w", p + ".ff2.w", 3, nullptr, s));
      for (const char* nm : {".glu.b", ".dw.w", ".dw.b", ".ff1.b", ".ff2.b"}) STG(need(e, p + nm, &t));
      return 0;
    };
    for (int i = 0; i < c.n_layers_text; ++i) STG(enc_block("text_enc.layers." + std::to_string(i)));
    for (int i = 0; i < c.ref_enc_layers; ++i) STG(enc_block("ref_enc_blocks." + std::to_string(i)));
    for (int i = 0; i < c.ref_xattn_layers; ++i) {
      const std::string p = "ref_xattn.blocks." + std::to_string(i);
      for (const char* nm : {".nq.weight", ".q.w", ".o.w", ".gate_scale", ".nkv.weight", ".kv.w"}) STG(need(e, p + nm, &t));
    }
    for (const char* nm : {"text_enc.embed", "text_enc.norm.weight", "pe", "spk_film.mlp.0.w", "spk_film.mlp.0.b", "spk_film.mlp.2.w", "spk_film.mlp.2.b",
                           "spk_film.norm.weight", "spk_film.norm.bias", "cond_norm.weight", "token2sv.emb", "token2sv.cw", "token2sv.enc.0.w",
                           "token2sv.enc.0.b", "token2sv.enc.3.w", "token2sv.enc.3.b", "token2sv.pool.attn.0.w", "token2sv.pool.attn.0.b",
                           "token2sv.pool.attn.2.w", "token2sv.pool.attn.2.b", "token2sv.proj.w", "token2sv.proj.b", "ref_cw", "ref_enc_norm.weight", "cb_embed"})
      STG(need(e, nm, &t));
    std::vector<int32_t> col, off;
    for (int q = 0; q < c.num_codebooks; ++q) { col.push_back(q); off.push_back(q * c.codebook_size); }
    STG(dev_upload(e, col, &e->q_col));
    STG(dev_upload(e, off, &e->q_off));
  }
  // ---- Mimi decoder: three-pass operands (16 mantissa bits: waveform contract; bf16 mode: one piece), raw fp32 for the fused
  // SEANet kernels
  if (e->has_mimi) {
    const int mp = bf16 ? 1 : 2;
    STG(pack_pieces(e, "rvq_proj.w", "rvq_proj.w", mp, nullptr, s));
    for (int l = 0; l < c.mimi_layers; ++l) {
      const std::string p = "tr." + std::to_string(l);
      for (const char* nm : {".qkv.w", ".o.w", ".fc1.w", ".fc2.w"}) STG(pack_pieces(e, p + nm, p + nm, mp, nullptr, s));
      for (const char* nm : {".ln1.w", ".ln1.b", ".ln2.w", ".ln2.b", ".ls1", ".ls2"}) STG(need(e, p + nm, &t));
      if (c.mimi_hidden % 64 == 0) {  // twins behind a fused LayerNorm (transformer_stack): the norm's weight in W', W lnb as the bias
        STG(pack_pieces(e, p + ".qkv.w", p + ".qkv.w#ln", mp, (p + ".ln1.w").c_str(), s));
        STG(fold_bias(e, p + ".qkv.w", p + ".ln1.b", p + ".qkv.lnb"));
        STG(pack_pieces(e, p + ".fc1.w", p + ".fc1.w#ln", mp, (p + ".ln2.w").c_str(), s));
        STG(fold_bias(e, p + ".fc1.w", p + ".ln2.b", p + ".fc1.lnb"));
      }
    }
    STG(pack_pieces(e, "sea.conv0.w", "sea.conv0.w", mp, nullptr, s));
    STG(need(e, "sea.conv0.b", &t));
    for (int si = 0; si < c.mimi_n_ratios; ++si) {
      const std::string u = "sea.up" + std::to_string(si), r = "sea.res" + std::to_string(si);
      // (K >= 1024 transposed convolutions and the first residual block's k = 3 convolution also as split-form rows: gemm_8p.hip)
      STG(need(e, u + ".w", &t, 2));
      STG(pack_pieces(e, u + ".w", u + ".w", mp, nullptr, s, (int)t->shape[1] >= 1024 && (int)t->shape[0] % 256 == 0));
      STG(need(e, u + ".b", &t));
      for (const char* nm : {".c1.w", ".c1.b", ".c2.w", ".c2.b"}) STG(need(e, r + nm, &t));
      STG(need(e, r + ".c1.w", &t, 2));
      if ((int)t->shape[0] >= 64 && (int)t->shape[0] != 64) {  // hidden > 64: generic contractions (64 = the fused 128-channel block)
        STG(pack_pieces(e, r + ".c1.w", r + ".c1.w", mp, nullptr, s, (int)t->shape[1] >= 1024 && (int)t->shape[0] % 256 == 0));
        STG(pack_pieces(e, r + ".c2.w", r + ".c2.w", mp, nullptr, s));
      }
    }
    for (const char* nm : {"codebooks", "upsample.w", "sea.final.w", "rope.cos", "rope.sin"}) STG(need(e, nm, &t));
    {
      const int Q = c.num_codebooks, V = c.codebook_size, ns = c.mimi_n_semantic;
      std::vector<int32_t> sc, so, ac, ao;
      for (int q = 0; q < ns; ++q) { sc.push_back(q); so.push_back(q * V); }
      for (int q = ns; q < Q; ++q) { ac.push_back(q); ao.push_back(q * V); }
      STG(dev_upload(e, sc, &e->sem_col)); STG(dev_upload(e, so, &e->sem_off));
      STG(dev_upload(e, ac, &e->ac_col)); STG(dev_upload(e, ao, &e->ac_off));
      std::vector<float> one((size_t)Q, 1.0f);
      STG(dev_upload(e, one, &e->ones));
    }
  }
  // ---- Mimi encoder (reference audio -> codes): every contraction in exact fp32 (the codes must equal the reference's)
  if (e->has_enc) {
    for (int si = 0; si < c.mimi_n_ratios; ++si) {
      const std::string r = "enc.res" + std::to_string(si), d = "enc.down" + std::to_string(si);
      for (const char* nm : {".c1.w", ".c2.w"}) STG(plain(e, r + nm));
      STG(plain(e, d + ".w"));
      for (const char* nm : {".c1.b", ".c2.b"}) STG(need(e, r + nm, &t));
      STG(need(e, d + ".b", &t));
    }
    for (int l = 0; l < c.mimi_layers; ++l) {
      const std::string p = "etr." + std::to_string(l);
      for (const char* nm : {".qkv.w", ".o.w", ".fc1.w", ".fc2.w"}) STG(plain(e, p + nm));
      for (const char* nm : {".ln1.w", ".ln1.b", ".ln2.w", ".ln2.b", ".ls1", ".ls2"}) STG(need(e, p + nm, &t));
    }
    for (const char* nm : {"enc.final.w", "enc.ds.w", "enc.inproj.sem.w", "enc.inproj.ac.w"}) STG(plain(e, nm));
    for (const char* nm : {"enc.conv0.w", "enc.conv0.b", "enc.final.b", "enc.cb_bias", "codebooks", "rope.cos", "rope.sin"}) STG(need(e, nm, &t));
  }
  e->final = true;
  return 0;
}

// The per-utterance operands of one text cross-attention layer of the AR loop (src/sopro/nn/text.py:75-83: k / v projections of
// the normalised text), with the layer's query and output projections folded in: K'_h = K_h Wq_h, V'_h = V_h Wo_h^T, written
// as [B, H, S_cap, D] (rows past S of a block are left untouched).  Launches only; nkv [B*S, D] and kvd [B*S, 2D] are scratch.
int sopro_ar_fold_text(const float* txt, const float* nkv_weight, const float* kv_w, const float* q_wT, const float* o_w, float* nkv, float* kvd,
                       float* kp, float* vp, int32_t B, int32_t S, int32_t S_cap, int32_t D, int32_t H, float eps, void* stream) {
  SOPRO_CHECK_ARG(txt && nkv_weight && kv_w && q_wT && o_w && nkv && kvd && kp && vp, "NULL pointer");
  SOPRO_CHECK_ARG(B > 0 && S > 0 && S_cap >= S && D > 0 && H > 0 && D % H == 0, "bad sizes");
  hipStream_t s = (hipStream_t)stream;
  const int dh = D / H;
  STG(norm(s, txt, nkv, nkv_weight, B * S, D, eps));
  G o; o.M = B * S; o.N = 2 * D; o.K = D;
  Wt kv; kv.f32 = kv_w;
  STG(gemm(s, nkv, kv, nullptr, kvd, o));
  for (int h = 0; h < H; ++h) {
    G f; f.M = B * S; f.N = D; f.K = dh; f.lda = 2 * D; f.rows_per_seg = S; f.ldc = D; f.c_seg = (int64_t)H * S_cap * D;
    Wt none;
    STG(gemm(s, kvd + h * dh, none, q_wT + (size_t)h * D * dh, kp + (size_t)h * S_cap * D, f));
    f.ldw = D;
    STG(gemm(s, kvd + D + h * dh, none, o_w + h * dh, vp + (size_t)h * S_cap * D, f));
  }
  return 0;
}

// The same for UNFOLDED keys (sopro_ar_frame.k_unfold): kq [B, S_cap, D] = the k projection itself (head h in columns (D / H) h ..),
// vp [B, H, S_cap, D] folded as above.  The query projection then rides on the frame's feed-forward launches.
int sopro_ar_fold_text_uk(const float* txt, const float* nkv_weight, const float* kv_w, const float* o_w, float* nkv, float* kvd, float* kq, float* vp,
                          int32_t B, int32_t S, int32_t S_cap, int32_t D, int32_t H, float eps, void* stream) {
  SOPRO_CHECK_ARG(txt && nkv_weight && kv_w && o_w && nkv && kvd && kq && vp, "NULL pointer");
  SOPRO_CHECK_ARG(B > 0 && S > 0 && S_cap >= S && D > 0 && H > 0 && D % H == 0, "bad sizes");
  hipStream_t s = (hipStream_t)stream;
  const int dh = D / H;
  STG(norm(s, txt, nkv, nkv_weight, B * S, D, eps));
  G o; o.M = B * S; o.N = 2 * D; o.K = D;
  Wt kv; kv.f32 = kv_w;
  STG(gemm(s, nkv, kv, nullptr, kvd, o));
  if (S == S_cap) {
    STG(sopro_copy2d_u32(kq, D, kvd, 2 * D, B * S, D, s));
  } else {
    for (int b = 0; b < B; ++b) STG(sopro_copy2d_u32(kq + (size_t)b * S_cap * D, D, kvd + (size_t)b * S * 2 * D, 2 * D, S, D, s));
  }
  for (int h = 0; h < H; ++h) {
    G f; f.M = B * S; f.N = D; f.K = dh; f.lda = 2 * D; f.rows_per_seg = S; f.ldc = D; f.c_seg = (int64_t)H * S_cap * D; f.ldw = D;
    Wt none;
    STG(gemm(s, kvd + D + h * dh, none, o_w + h * dh, vp + (size_t)h * S_cap * D, f));
  }
  return 0;
}

// ------------------------------------------------------------------------------------------------ conditioning stage
static int ssm_block_bufs(sopro_engine* e, hipStream_t s, float* h, float* x1, float* u, const SplitK* sk, const float* x, float* out,
                          const std::string& p, int B, int T, int ksize, int dil, const int32_t* lens, const char* sfx = "",
                          int32_t* range = nullptr);  // (with the NAR stage below)
struct CondWs { float *xa, *xb, *h, *x1, *u, *base, *cond, *nq, *q, *a, *am; SplitK sk; };
static size_t cond_carve(const sopro_engine* e, CondWs& w, void* ws, int B, int S, int Tar) {
  const size_t M = (size_t)B * S, R = (size_t)B * Tar, D = e->c.d_model;
  Carver cv(ws);
  w.xa = cv.take<float>(M * D); w.xb = cv.take<float>(M * D); w.h = cv.take<float>(M * D); w.x1 = cv.take<float>(M * D);
  w.u = cv.take<float>(M * 4 * D);
  w.base = cv.take<float>(R * D); w.cond = cv.take<float>(R * D); w.nq = cv.take<float>(R * D); w.q = cv.take<float>(R * D);
  w.a = cv.take<float>(R * D); w.am = cv.take<float>(R * D);
  w.sk.tickets = cv.take<int32_t>(SPLITK_TICKETS);
  w.sk.ws = M <= 1024 ? cv.take<float>(SPLITK_WS_BYTES / 4) : nullptr;
  return cv.off;
}

int64_t sopro_cond_workspace_bytes(const sopro_engine* e, int32_t B, int32_t S, int32_t Tar) {
  if (!e || B <= 0 || S <= 0 || Tar <= 0) return 0;
  CondWs w;
  return (int64_t)cond_carve(e, w, nullptr, B, S, Tar);
}

int sopro_cond_prepare(sopro_engine* e, void* workspace, const int32_t* ids, const int32_t* lens, int32_t ragged, const float* film_mul,
                       const float* film_add, const float* const* ref_k, const float* const* ref_v, int64_t kv_bstride, const int32_t* kv_index,
                       const int32_t* ref_klens, int32_t B, int32_t S, int32_t Tar, int32_t Tr, float* txt_seq, float* txt_pool, float* cond_ar,
                       void* stream) {
  SOPRO_CHECK_ARG(e && e->final && e->has_cond && workspace && ids && lens && film_mul && film_add && ref_k && ref_v && txt_seq && txt_pool && cond_ar,
                  "bad arguments (finalize the engine with the conditioning tensors first)");
  SOPRO_CHECK_ARG(B > 0 && S > 0 && Tar > 0 && Tr > 0, "bad sizes");
  hipStream_t s = (hipStream_t)stream;
  const sopro_engine_cfg& c = e->c;
  const int D = c.d_model, M = B * S, R = B * Tar, H = c.ref_xattn_heads, dh = D / H;
  SOPRO_CHECK_ARG(Tar <= e->t["pe"].shape[0] && S <= e->t["pe"].shape[0], "more positions than the position table holds");
  CondWs w;
  cond_carve(e, w, workspace, B, S, Tar);
  if (w.sk.ws) STG(sopro_fill2d_u32(w.sk.tickets, SPLITK_TICKETS, 1, SPLITK_TICKETS, 0u, s));
  // ---- text encoder (src/sopro/nn/text.py:29-44)
  float *xa = w.xa, *xb = w.xb;
  STG(sopro_text_embed_f32(ids, lens, F(e, "text_enc.embed"), e->t["text_enc.embed"].shape[0], F(e, "pe"), xa, B, S, D, s));
  for (int i = 0; i < c.n_layers_text; ++i) {
    STG(ssm_block_bufs(e, s, w.h, w.x1, w.u, &w.sk, xa, xb, "text_enc.layers." + std::to_string(i), B, S, c.enc_kernel, 1, ragged ? lens : nullptr));
    std::swap(xa, xb);
  }
  STG(norm(s, xa, txt_seq, F(e, "text_enc.norm.weight"), M, D, RMS_EPS));
  STG(sopro_masked_mean_f32(txt_seq, lens, txt_pool, B, S, D, s));
  // ---- base = pooled text + frame positions (model.py:200-202), SpeakerFiLM on its LayerNorm (speaker.py:76-85)
  STG(sopro_add_pos_f32(txt_pool, F(e, "pe"), w.base, B, Tar, D, 0, s));
  STG(norm(s, w.base, w.cond, F(e, "spk_film.norm.weight"), R, D, 1e-5f, SOPRO_NORM_LN, F(e, "spk_film.norm.bias"), film_mul, film_add, Tar));
  // ---- reference cross-attention stack (src/sopro/nn/ref.py:54-108)
  for (int i = 0; i < c.ref_xattn_layers; ++i) {
    const std::string p = "ref_xattn.blocks." + std::to_string(i);
    STG(norm(s, w.cond, w.nq, F(e, p + ".nq.weight"), R, D, RMS_EPS));
    G qg; qg.M = R; qg.N = D; qg.K = D;
    Wt wq; wq.f32 = F(e, p + ".q.w");
    STG(gemm(s, w.nq, wq, nullptr, w.q, qg));
    sopro_attn_args a;
    memset(&a, 0, sizeof(a));
    a.Q = w.q; a.ldq = D; a.q_bstride = (int64_t)Tar * D;
    a.K = ref_k[i]; a.ldk = D; a.k_bstride = kv_bstride;
    a.V = ref_v[i]; a.ldv = D; a.v_bstride = kv_bstride;
    a.O = w.a; a.ldo = D; a.o_bstride = (int64_t)Tar * D;
    a.klens = ref_klens; a.kv_index = kv_index;
    a.B = B; a.H = H; a.dh = dh; a.Tq = Tar; a.Tk = Tr; a.scale = 1.0f / sqrtf((float)dh);
    STG(attend(a, s));
    STG(sopro_rms_match_f32(w.a, w.cond, w.am, R, D, s));
    G og; og.M = R; og.N = D; og.K = D; og.epi = SOPRO_EPI_RES; og.R = w.cond; og.scale = F(e, p + ".gate_scale");
    Wt wo; wo.f32 = F(e, p + ".o.w");
    STG(gemm(s, w.am, wo, nullptr, w.cond, og));
  }
  return norm(s, w.cond, cond_ar, F(e, "cond_norm.weight"), R, D, RMS_EPS);
}

int sopro_film_coeffs(sopro_engine* e, const float* sv, float style, int32_t n, float* scratch, float* mul, float* add, void* stream) {
  SOPRO_CHECK_ARG(e && e->final && e->has_cond && sv && scratch && mul && add && n > 0, "bad arguments (finalize the engine with the conditioning tensors first)");
  hipStream_t s = (hipStream_t)stream;
  const int D = e->c.d_model, svd = e->c.sv_student_dim;
  float *f1 = scratch, *film = f1 + (size_t)n * D, *gam = film + (size_t)n * 2 * D, *bet = gam + (size_t)n * D;
  G g1; g1.M = n; g1.N = D; g1.K = svd; g1.bias = F(e, "spk_film.mlp.0.b"); g1.epi = SOPRO_EPI_GELU;
  Wt w0; w0.f32 = F(e, "spk_film.mlp.0.w");
  STG(gemm(s, sv, w0, nullptr, f1, g1));
  G g2; g2.M = n; g2.N = 2 * D; g2.K = D; g2.bias = F(e, "spk_film.mlp.2.b");
  Wt w2; w2.f32 = F(e, "spk_film.mlp.2.w");
  STG(gemm(s, f1, w2, nullptr, film, g2));
  STG(sopro_copy2d_u32(gam, D, film, 2 * D, n, D, s));
  STG(sopro_copy2d_u32(bet, D, film + D, 2 * D, n, D, s));
  STG(sopro_tanh_affine_f32(gam, mul, 1.0f, style, (int64_t)n * D, s));
  return sopro_tanh_affine_f32(bet, add, 0.0f, style, (int64_t)n * D, s);
}

// ------------------------------------------------------------------------------------------------ reference preparation
struct RefWs { float *x, *h1, *h2, *a1, *lg, *st, *ev, *xa, *xb, *h, *x1, *u, *nkv; SplitK sk; };
static size_t ref_carve(const sopro_engine* e, RefWs& w, void* ws, int T) {
  const size_t D = e->c.d_model, SD = e->t.count("token2sv.emb") ? (size_t)e->t.at("token2sv.emb").shape[1] : 0;
  Carver cv(ws);
  w.x = cv.take<float>(T * SD); w.h1 = cv.take<float>(T * SD); w.h2 = cv.take<float>(T * SD); w.a1 = cv.take<float>(T * SD);
  w.lg = cv.take<float>(T); w.st = cv.take<float>(2 * SD); w.ev = cv.take<float>(e->c.sv_student_dim);
  w.xa = cv.take<float>(T * D); w.xb = cv.take<float>(T * D); w.h = cv.take<float>(T * D); w.x1 = cv.take<float>(T * D);
  w.u = cv.take<float>(T * 4 * D); w.nkv = cv.take<float>(T * D);
  w.sk.tickets = cv.take<int32_t>(SPLITK_TICKETS);
  w.sk.ws = (size_t)T <= 1024 ? cv.take<float>(SPLITK_WS_BYTES / 4) : nullptr;
  return cv.off;
}

int64_t sopro_ref_workspace_bytes(const sopro_engine* e, int32_t T) {
  if (!e || T <= 0) return 0;
  RefWs w;
  return (int64_t)ref_carve(e, w, nullptr, T);
}

int sopro_ref_prepare(sopro_engine* e, void* workspace, const int32_t* tokens, int32_t T, float* sv, float* ref_seq, float* const* kv, void* stream) {
  SOPRO_CHECK_ARG(e && e->final && e->has_cond && workspace && tokens && sv && T > 0 && ((ref_seq && kv) || (!ref_seq && !kv)),
                  "bad arguments (finalize the engine with the conditioning tensors first; ref_seq and kv are given or omitted together)");
  hipStream_t s = (hipStream_t)stream;
  const sopro_engine_cfg& c = e->c;
  const int D = c.d_model, Q = c.num_codebooks, SD = (int)e->t["token2sv.emb"].shape[1], svd = c.sv_student_dim;
  RefWs w;
  ref_carve(e, w, workspace, T);
  if (w.sk.ws) STG(sopro_fill2d_u32(w.sk.tickets, SPLITK_TICKETS, 1, SPLITK_TICKETS, 0u, s));
  // ---- Token2SV (src/sopro/nn/speaker.py:37-61)
  STG(sopro_codebook_sum_f32(tokens, Q, e->q_col, e->q_off, F(e, "token2sv.cw"), Q, F(e, "token2sv.emb"), e->t["token2sv.emb"].shape[0], nullptr, 0.f,
                             1.f, w.x, SD, 0, T, T, SD, s));
  STG(sopro_dwconv_f32(w.x, F(e, "token2sv.enc.0.w"), F(e, "token2sv.enc.0.b"), nullptr, w.h1, nullptr, 1, T, SD, 7, 1, 3, 2, s));
  STG(sopro_dwconv_f32(w.h1, F(e, "token2sv.enc.3.w"), F(e, "token2sv.enc.3.b"), nullptr, w.h2, nullptr, 1, T, SD, 7, 1, 3, 2, s));
  G ga; ga.M = T; ga.N = SD; ga.K = SD; ga.bias = F(e, "token2sv.pool.attn.0.b"); ga.epi = SOPRO_EPI_TANH;
  Wt wa; wa.f32 = F(e, "token2sv.pool.attn.0.w");
  STG(gemm(s, w.h2, wa, nullptr, w.a1, ga));
  G gl; gl.M = T; gl.N = 1; gl.K = SD; gl.bias = F(e, "token2sv.pool.attn.2.b");
  Wt wl; wl.f32 = F(e, "token2sv.pool.attn.2.w");
  STG(gemm(s, w.a1, wl, nullptr, w.lg, gl));
  STG(sopro_stats_pool_f32(w.h2, w.lg, nullptr, w.st, 1, T, SD, s));
  G gp; gp.M = 1; gp.N = svd; gp.K = 2 * SD; gp.bias = F(e, "token2sv.proj.b");
  Wt wp; wp.f32 = F(e, "token2sv.proj.w");
  STG(gemm(s, w.st, wp, nullptr, w.ev, gp));
  STG(sopro_l2norm_f32(w.ev, sv, 1, svd, 1e-6f, s));
  if (!ref_seq) return 0;  // speaker vector only: SoproTTS.encode_speaker (src/sopro/model.py:457-475)
  // ---- reference sequence encoder (model.py:133-149)
  float *xa = w.xa, *xb = w.xb;
  STG(sopro_codebook_sum_f32(tokens, Q, e->q_col, e->q_off, F(e, "ref_cw"), Q, F(e, "cb_embed"), e->t["cb_embed"].shape[0], nullptr, 0.f, 1.f, xa, D, 0,
                             T, T, D, s));
  for (int i = 0; i < c.ref_enc_layers; ++i) {
    STG(ssm_block_bufs(e, s, w.h, w.x1, w.u, &w.sk, xa, xb, "ref_enc_blocks." + std::to_string(i), 1, T, c.enc_kernel, 1, nullptr));
    std::swap(xa, xb);
  }
  STG(norm(s, xa, ref_seq, F(e, "ref_enc_norm.weight"), T, D, RMS_EPS));
  // ---- K | V rows of the reference cross-attention blocks (src/sopro/nn/ref.py:120-128)
  for (int i = 0; i < c.ref_xattn_layers; ++i) {
    const std::string p = "ref_xattn.blocks." + std::to_string(i);
    STG(norm(s, ref_seq, w.nkv, F(e, p + ".nkv.weight"), T, D, RMS_EPS));
    G gk; gk.M = T; gk.N = 2 * D; gk.K = D;
    Wt wk; wk.f32 = F(e, p + ".kv.w");
    STG(gemm(s, w.nkv, wk, nullptr, kv[i], gk));
  }
  return 0;
}

// ------------------------------------------------------------------------------------------------ autoregressive stage
static size_t ar_carve(const sopro_engine* e, ArPlan& p, void* ws, int B, int S, int Tar) {
  const sopro_engine_cfg& c = e->c;
  const int D = c.d_model, V1 = c.codebook_size + 1, S_cap = (S + 63) / 64 * 64, H = 4;
  Carver cv(ws);
  p.cond = cv.take<float>((size_t)B * Tar * D);
  for (int i = 0; i < 4; ++i) p.x[i] = cv.take<float>((size_t)B * D);
  p.part = cv.take<float>((size_t)4 * B * D);
  p.u = cv.take<float>((size_t)B * 4 * D);
  p.logits = cv.take<float>((size_t)B * V1);
  for (int i = 0; i < c.n_layers_ar; ++i) {
    p.kp[i] = p.vp[i] = nullptr;
    if (c.ar_xattn[i]) {
      p.kp[i] = cv.take<float>((size_t)B * H * S_cap * D);
      p.vp[i] = cv.take<float>((size_t)B * H * S_cap * D);
    }
  }
  p.xp = cv.take<float>((size_t)H * B * D);
  p.params = cv.take<float>(8);
  for (int i = 0; i < c.n_layers_ar; ++i) p.rings[i] = cv.take<float>((size_t)((c.ar_kernel - 1) * c.ar_dilations[i] + 1) * B * D);
  p.nkv = cv.take<float>((size_t)B * S * D);
  p.kvd = cv.take<float>((size_t)B * S * 2 * D);
  p.qa = cv.take<float>((size_t)B * D);
  p.qpart = cv.take<float>((size_t)4 * B * D);
  p.fold_k = p.fold_v = nullptr;
  if (c.precision == 1) {  // bf16 mode: kp / vp / rings hold bf16 elements (the first half of their fp32-sized buffers)
    p.fold_k = cv.take<float>((size_t)B * H * S_cap * D);
    p.fold_v = cv.take<float>((size_t)B * H * S_cap * D);
  }
  p.klens = cv.take<int32_t>(B);
  p.hist = cv.take<int32_t>((size_t)B * Tar);
  p.ctr = cv.take<int32_t>(8);
  p.first_eos = cv.take<int32_t>(B);
  p.stop_t = cv.take<int32_t>(B);
  p.recent = cv.take<int32_t>((size_t)B * 64);
  p.nonce = cv.take<uint32_t>(B);
  p.key = cv.take<uint32_t>(2);
  p.row_step = cv.take<int32_t>(B);
  return cv.off;
}

int64_t sopro_ar_workspace_bytes(const sopro_engine* e, int32_t B, int32_t S, int32_t Tar) {
  if (!e || B <= 0 || S <= 0 || Tar <= 0) return 0;
  ArPlan tmp;
  return (int64_t)ar_carve(e, tmp, nullptr, B, S, Tar);
}

// one frame: the buffers of this plan described to sopro_ar_issue_frame (ar_frame.hip), where the launch sequence lives
static int ar_issue_step(sopro_engine* e, hipStream_t s) {
  const sopro_engine_cfg& c = e->c;
  ArPlan& p = e->ar;
  sopro_ar_frame f;
  memset(&f, 0, sizeof(f));
  for (int i = 0; i < c.n_layers_ar; ++i) {
    const std::string pr = "ar.blocks." + std::to_string(i);
    sopro_ar_block& b = f.blk[i];
    b.glu_w = e->sk[pr + ".glu.w"]; b.glu_b = F(e, pr + ".glu.b"); b.dw_w = F(e, pr + ".dw.w"); b.dw_b = F(e, pr + ".dw.b");
    b.ff1_w = e->sk[pr + ".ff1.w"]; b.ff1_b = F(e, pr + ".ff1.b"); b.ff2_w = e->sk[pr + ".ff2.w"]; b.ff2_b = F(e, pr + ".ff2.b");
    b.ring = p.rings[i]; b.dil = c.ar_dilations[i]; b.xattn = c.ar_xattn[i] ? 1 : 0; b.gate = c.ar_gate[i];
    b.kp = p.kp[i]; b.vp = p.vp[i];
    if (p.k_unfold && c.ar_xattn[i]) {
      const std::string pa = "ar.x_attns." + std::to_string(i);
      b.qa_w = e->sk[pa + ".qa.w"]; b.qu_w = e->sk[pa + ".qu.w"]; b.q_b = F(e, pa + ".q.b");
    }
  }
  f.head_w = e->sk["ar.head.w"]; f.head_b = F(e, "ar.head.b");
  f.x0 = p.x[0]; f.xa = p.x[1]; f.xb = p.x[2]; f.part = p.part; f.u = p.u; f.xp = p.xp; f.logits = p.logits; f.klens = p.klens;
  f.n_layers = c.n_layers_ar; f.B = p.B; f.D = c.d_model; f.S_cap = p.S_cap; f.V1 = c.codebook_size + 1; f.H = 4; f.ksize = c.ar_kernel;
  f.w_layout = c.precision == 1 ? 2 : 1;
  f.store_format = c.precision == 1 ? 1 : 0;  // bf16 mode: ring buffers and folded text operands as bf16 in memory
  if (p.k_unfold) { f.k_unfold = 1; f.qa = p.qa; f.qpart = p.qpart; }
  f.tile_glu = e->ar_tiles[0]; f.tile_ff1 = e->ar_tiles[1]; f.tile_ff2 = e->ar_tiles[2]; f.tile_head = e->ar_tiles[3];
  f.eps = RMS_EPS;
  f.st = p.st;
  return sopro_ar_issue_frame(&f, s);
}

int sopro_ar_begin(sopro_engine* e, void* workspace, int32_t B, const float* cond_ar, const float* txt_seq, const int32_t* text_lens,
                   int32_t S, int32_t Tar, const float params[8], uint64_t seed, uint32_t nonce, void* stream) {
  SOPRO_CHECK_ARG(e && e->final && e->has_ar && workspace && cond_ar && txt_seq && params && B > 0 && S > 0 && Tar > 0,
                  "bad arguments (finalize the engine with the AR tensors first)");
  SOPRO_CHECK_ARG(params[6] >= 1.f && params[6] <= 64.f, "top_k must be in [1, 64] (the reference uses 50)");
  hipStream_t s = (hipStream_t)stream;
  const sopro_engine_cfg& c = e->c;
  ArPlan& p = e->ar;
  const int D = c.d_model, H = 4;
  ar_carve(e, p, workspace, B, S, Tar);
  p.B = B; p.S = S; p.S_cap = (S + 63) / 64 * 64; p.Tar = Tar; p.ws = workspace;
  // (kernels of the library for every copy and clear, as in the recorded sequences: one kind of node on a C host's timeline too)
  STG(sopro_copy2d_u32(p.cond, (int64_t)Tar * D, cond_ar, (int64_t)Tar * D, B, Tar * D, s));
  if (text_lens) {
    STG(sopro_copy2d_u32(p.klens, B, text_lens, B, 1, B, s));
  } else {
    STG(sopro_fill2d_u32(p.klens, B, 1, B, (uint32_t)S, s));
  }
  // unfolded keys whenever the engine was given the query operands (fp32 frame; the Python host takes the same decision)
  {
    static const bool off = SOPRO_DEV_ENV("SOPRO_AR_KUNFOLD") != nullptr && SOPRO_DEV_ENV("SOPRO_AR_KUNFOLD")[0] == '0';  // developer A/B
    bool all = c.precision == 0 && !off;
    for (int i = 0; i < c.n_layers_ar; ++i)
      if (c.ar_xattn[i] && !e->sk.count("ar.x_attns." + std::to_string(i) + ".qa.w")) all = false;
    if (p.graph && p.k_unfold != all) { STG(sopro_graph_destroy(p.graph)); p.graph = nullptr; }
    p.k_unfold = all;
  }
  // K/V of the text for the cross-attention layers (src/sopro/nn/text.py:75-83), query / output projections folded in
  for (int i = 0; i < c.n_layers_ar; ++i) {
    if (!c.ar_xattn[i]) continue;
    const std::string pa = "ar.x_attns." + std::to_string(i);
    if (p.k_unfold) {  // K as it is ([B, S_cap, D] in the first quarter of the layer's kp buffer), V' folded
      STG(sopro_ar_fold_text_uk(txt_seq, F(e, pa + ".nkv.weight"), F(e, pa + ".kv.w"), F(e, pa + ".o.w"), p.nkv, p.kvd, p.kp[i], p.vp[i], B, S, p.S_cap, D,
                                H, RMS_EPS, s));
      continue;
    }
    float* kdst = c.precision == 1 ? p.fold_k : p.kp[i];
    float* vdst = c.precision == 1 ? p.fold_v : p.vp[i];
    STG(sopro_ar_fold_text(txt_seq, F(e, pa + ".nkv.weight"), F(e, pa + ".kv.w"), F(e, pa + ".q.wT"), F(e, pa + ".o.w"), p.nkv, p.kvd, kdst,
                           vdst, B, S, p.S_cap, D, H, RMS_EPS, s));
    if (c.precision == 1) {  // rounded once; the frame streams half the bytes (sopro_ar_frame.store_format)
      STG(sopro_cvt_f32_bf16(kdst, p.kp[i], (int64_t)B * H * p.S_cap * D, s));
      STG(sopro_cvt_f32_bf16(vdst, p.vp[i], (int64_t)B * H * p.S_cap * D, s));
    }
  }
  for (int i = 0; i < c.n_layers_ar; ++i) {
    const int64_t n = (int64_t)((c.ar_kernel - 1) * c.ar_dilations[i] + 1) * B * D;  // the whole buffer (bf16 state uses its first half)
    STG(sopro_fill2d_u32(p.rings[i], n, 1, (int32_t)n, 0u, s));
  }
  STG(sopro_fill2d_u32(p.hist, (int64_t)B * Tar, 1, B * Tar, 0u, s));
  for (int k = 0; k < 8; ++k) {  // eight scalars from the caller's (pageable) host array: by value
    uint32_t u;
    memcpy(&u, &params[k], 4);
    STG(sopro_fill2d_u32(p.params + k, 1, 1, 1, u, s));
  }
  STG(sopro_fill2d_u32(p.nonce, B, 1, B, nonce, s));
  // the Philox key lives in device memory: the recorded frame graph serves every seed
  STG(sopro_fill2d_u32(p.key, 1, 1, 1, (uint32_t)seed, s));
  STG(sopro_fill2d_u32(p.key + 1, 1, 1, 1, (uint32_t)(seed >> 32), s));
  sopro_ar_state& st = p.st;
  memset(&st, 0, sizeof(st));
  st.x_cur = p.x[0]; st.cond = p.cond; st.emb = F(e, "cb_embed"); st.hist = p.hist;
  st.step = p.ctr; st.row_step = p.row_step; st.n_stopped = p.ctr + 2;
  st.first_eos = p.first_eos; st.stop_t = p.stop_t; st.recent = p.recent; st.params = p.params; st.nonce = p.nonce;
  st.key = p.key; st.seed = 0; st.B = B; st.D = D; st.Tar = Tar; st.max_steps = Tar; st.V = c.codebook_size; st.bos_row = c.bos_row;
  STG(sopro_ar_init(&st, s));
  // a recorded frame graph holds raw pointers into the workspace it was recorded on
  if (p.graph && (p.graph_B != B || p.graph_S_cap != p.S_cap || p.graph_Tar != Tar || p.graph_ws != workspace)) {
    STG(sopro_graph_destroy(p.graph));
    p.graph = nullptr;
  }
  return 0;
}

int sopro_ar_run_graph(sopro_engine* e, int32_t n_steps, void* stream) {
  SOPRO_CHECK_ARG(e && e->ar.ws && n_steps >= 0 && stream, "sopro_ar_begin first; a non-default stream is needed for graph capture");
  ArPlan& p = e->ar;
  if (!p.graph) {
    STG(sopro_capture_begin(stream));
    const int rc = ar_issue_step(e, (hipStream_t)stream);
    void* g = nullptr;
    const int rc2 = sopro_capture_end(stream, &g);
    if (rc != 0) return rc;
    if (rc2 != 0) return rc2;
    p.graph = g; p.graph_B = p.B; p.graph_S_cap = p.S_cap; p.graph_Tar = p.Tar; p.graph_ws = p.ws;
  }
  for (int i = 0; i < n_steps; ++i) STG(sopro_graph_launch(p.graph, stream));
  return 0;
}

int sopro_ar_tokens(sopro_engine* e, int32_t* hist, int32_t* first_eos, int32_t* n_stopped, void* stream) {
  SOPRO_CHECK_ARG(e && e->ar.ws, "sopro_ar_begin first");
  hipStream_t s = (hipStream_t)stream;
  const ArPlan& p = e->ar;
  // (destinations may be device or page-locked host memory: the copies are kernels)
  if (hist) STG(sopro_copy2d_u32(hist, (int64_t)p.B * p.Tar, p.hist, (int64_t)p.B * p.Tar, 1, p.B * p.Tar, s));
  if (first_eos) STG(sopro_copy2d_u32(first_eos, p.B, p.first_eos, p.B, 1, p.B, s));
  if (n_stopped) STG(sopro_copy2d_u32(n_stopped, 1, p.ctr + 2, 1, 1, 1, s));
  return 0;
}

// ------------------------------------------------------------------------------------------------ NAR refinement
struct NarWs { float *xa, *xb, *h, *x1, *u, *z, *part, *cond; int32_t *lens, *range; SplitK sk; };
static size_t nar_carve(const sopro_engine* e, NarWs& w, void* ws, int B, int T) {
  const sopro_engine_cfg& c = e->c;
  const size_t M = (size_t)B * T, D = c.d_model;
  int nh_max = 0;
  for (int i = 0; i < c.n_stages; ++i) nh_max = c.stage_n_cb[i] > nh_max ? c.stage_n_cb[i] : nh_max;
  Carver cv(ws);
  w.xa = cv.take<float>(M * D); w.xb = cv.take<float>(M * D); w.h = cv.take<float>(M * D); w.x1 = cv.take<float>(M * D);
  w.u = cv.take<float>(M * 4 * D); w.z = cv.take<float>(M * c.nar_head_dim);
  w.part = cv.take<float>(M * nh_max * (c.codebook_size / 64) * 2);
  w.cond = cv.take<float>(M * D);
  w.lens = cv.take<int32_t>(B);
  w.range = cv.take<int32_t>(4);
  w.sk.tickets = cv.take<int32_t>(SPLITK_TICKETS);
  w.sk.ws = M <= 1024 ? cv.take<float>(SPLITK_WS_BYTES / 4) : nullptr;  // few rows (streaming windows, batch 1): K loops may be split
  return cv.off;
}

int64_t sopro_nar_workspace_bytes(const sopro_engine* e, int32_t B, int32_t T) {
  if (!e || B <= 0 || T <= 0) return 0;
  NarWs w;
  return (int64_t)nar_carve(e, w, nullptr, B, T);
}

// full-sequence SSMLiteBlock over dense [B*T, D] rows (reference: src/sopro/nn/blocks.py:143-148), norms fused into the contractions;
// h, x1 [B*T, D] and u [B*T, 4D] are scratch, sk the split-K scratch of the last contraction (few-row problems)
static int ssm_block_bufs(sopro_engine* e, hipStream_t s, float* h, float* x1, float* u, const SplitK* sk, const float* x, float* out,
                          const std::string& p, int B, int T, int ksize, int dil, const int32_t* lens, const char* sfx, int32_t* range) {
  const int D = e->c.d_model, M = B * T;
  const int total = (ksize - 1) * dil, left = total / 2;  // non-causal: symmetric zero padding (blocks.py:68-72)
  G g; g.M = M; g.N = 2 * D; g.K = D; g.bias = F(e, p + ".glu.b"); g.epi = SOPRO_EPI_GLU; g.rms_eps = RMS_EPS; g.range_events = range;
  STG(gemm(s, x, WT(e, p + ".glu.wn" + sfx), nullptr, h, g));
  STG(sopro_dwconv_f32(h, F(e, p + ".dw.w"), F(e, p + ".dw.b"), x, x1, lens, B, T, D, ksize, dil, left, 1, s));
  G f1; f1.M = M; f1.N = 4 * D; f1.K = D; f1.bias = F(e, p + ".ff1.b"); f1.epi = SOPRO_EPI_GELU; f1.rms_eps = RMS_EPS; f1.range_events = range;
  STG(gemm(s, x1, WT(e, p + ".ff1.wn" + sfx), nullptr, u, f1));
  G f2; f2.M = M; f2.N = D; f2.K = 4 * D; f2.bias = F(e, p + ".ff2.b"); f2.epi = SOPRO_EPI_RES; f2.R = x1; f2.sk = sk; f2.range_events = range;
  return gemm(s, u, WT(e, p + ".ff2.w" + sfx), nullptr, out, f2);
}
static int ssm_block_seq(sopro_engine* e, hipStream_t s, const NarWs& w, const float* x, float* out, const std::string& p, int B, int T,
                         int ksize, int dil, const int32_t* lens, const char* sfx) {
  return ssm_block_bufs(e, s, w.h, w.x1, w.u, &w.sk, x, out, p, B, T, ksize, dil, lens, sfx, w.range);
}

int sopro_nar_refine_io(sopro_engine* e, void* workspace, const sopro_nar_io* io, int32_t B, int32_t T, void* stream) {
  SOPRO_CHECK_ARG(e && e->final && e->has_nar && workspace && io && io->cond && io->cb0 && io->tokens && B > 0 && T > 0,
                  "bad arguments (finalize the engine with the NAR tensors first)");
  hipStream_t s = (hipStream_t)stream;
  const sopro_engine_cfg& c = e->c;
  const int D = c.d_model, V = c.codebook_size, Q = c.num_codebooks, HD = c.nar_head_dim, M = B * T;
  SOPRO_CHECK_ARG(V % 64 == 0, "codebook_size must be a multiple of 64 (arg-max partials per 64 columns)");
  SOPRO_CHECK_ARG(io->cond_bstride >= (int64_t)T * D && io->cb0_bstride >= T, "cond / cb0 blocks shorter than T rows");
  const bool safe = io->safe != 0 && c.precision == 0;  // (bf16 mode: one operand set, bf16 has fp32's exponent range already)
  const char* sfx = safe ? NAR_SAFE : "";
  int32_t* tokens = io->tokens;
  NarWs w;
  nar_carve(e, w, workspace, B, T);
  if (w.sk.ws) STG(sopro_fill2d_u32(w.sk.tickets, SPLITK_TICKETS, 1, SPLITK_TICKETS, 0u, s));  // split-K tickets start (and are left) at zero
  STG(sopro_fill2d_u32(w.range, 4, 1, 4, 0u, s));
  int nh_max = 0;
  for (int i = 0; i < c.n_stages; ++i) nh_max = c.stage_n_cb[i] > nh_max ? c.stage_n_cb[i] : nh_max;
  const float* cnd = io->cond;
  if (io->cond_bstride != (int64_t)T * D) {  // the first T rows of longer conditioning blocks (cond_ar has max_frames + 1 rows): densify once
    STG(sopro_copy2d_u32(w.cond, (int64_t)T * D, io->cond, io->cond_bstride, B, T * D, s));
    cnd = w.cond;
  }
  STG(sopro_nar_seed_i32(tokens, Q, io->cb0, io->cb0_bstride, B, T, V - 1, s));  // column 0 <- codebook 0
  const int32_t* lens_d = nullptr;
  if (io->lens) {  // device or page-locked host memory of the caller: read once, here
    STG(sopro_copy2d_u32(w.lens, B, io->lens, B, 1, B, s));
    lens_d = w.lens;
  }
  const char* stage_names[8] = {"B", "C", "D", "E", "F", "G", "H", "I"};
  for (int sid = 0; sid < c.n_stages; ++sid) {
    float *xa = w.xa, *xb = w.xb;
    // prev = sum_j cw_j * E[cb_j * V + tok_j]; x = mix0 * cond + mix1 * prev   (embeddings.py:77-112, nar.py:95-97)
    STG(sopro_codebook_sum_f32(tokens, Q, e->nar_cols[sid], e->nar_offs[sid], e->nar_cw[sid], e->nar_known[sid], F(e, "cb_embed"),
                               e->t["cb_embed"].shape[0], cnd, c.nar_mix[sid][0], c.nar_mix[sid][1], xa, D, 0, M, M, D, s));
    STG(norm(s, xa, xb, F(e, "nar.adapter.norm.weight"), M, D, RMS_EPS, SOPRO_NORM_RMS, nullptr, e->ad_mul[sid], e->ad_add[sid], M));
    std::swap(xa, xb);
    for (int i = 0; i < c.n_layers_nar; ++i) {
      STG(ssm_block_seq(e, s, w, xa, xb, "nar.blocks." + std::to_string(i), B, T, c.nar_kernel, c.nar_dilations[i], lens_d, sfx));
      std::swap(xa, xb);
    }
    STG(norm(s, xa, xb, F(e, "nar.norm.weight"), M, D, RMS_EPS));
    G pz; pz.M = M; pz.N = HD; pz.K = D; pz.bias = F(e, "nar.pre.b"); pz.range_events = w.range;
    STG(gemm(s, xb, WT(e, std::string("nar.pre.w") + sfx), nullptr, w.z, pz));
    // all heads of the stage in one contraction, arg-max in its epilogue (head-id embeddings live in the bias)
    const int nh = c.stage_n_cb[sid];
    const std::string hk = std::string("nar.heads.") + stage_names[sid];
    G hg; hg.M = M; hg.N = nh * V; hg.K = HD; hg.bias = F(e, hk + ".b"); hg.c_mode = 5; hg.C2 = w.part; hg.ldc2 = (int64_t)nh_max * (V / 64);
    hg.range_events = w.range;
    STG(gemm(s, w.z, WT(e, hk + ".w" + sfx), nullptr, nullptr, hg));
    STG(sopro_argmax_partials_i32(w.part, (int64_t)nh_max * (V / 64), tokens + c.stage_first_cb[sid], Q, nh, V / 64, V, M, s));
  }
  if (io->range_out) STG(sopro_copy2d_u32(io->range_out, 1, w.range, 1, 1, 1, s));
  return 0;
}

int sopro_nar_refine(sopro_engine* e, void* workspace, const float* cond, int64_t cond_bstride, const int32_t* rvq1, const int32_t* lens,
                     int32_t B, int32_t T, int32_t* tokens, void* stream) {
  sopro_nar_io io;
  memset(&io, 0, sizeof(io));
  io.cond = cond; io.cond_bstride = cond_bstride; io.cb0 = rvq1; io.cb0_bstride = T; io.lens = lens; io.tokens = tokens;
  return sopro_nar_refine_io(e, workspace, &io, B, T, stream);
}

// ------------------------------------------------------------------------------------------------ Mimi decode
struct MimiWs {
  int32_t* tok;
  float *emb, *q, *X, *y, *qkv, *ao, *hd, *e0, *hraw[8], *hact[8], *y1[8];
  float* lnst;  // (mean, squared deviations) per transformer row and 64-column group (fused LayerNorm)
  SplitK sk;
};
// bf16 mode (round 4): the SEANet decoder's activations - everything from the first convolution's output on - live in memory as
// bf16 rows (SOPRO_MIMI_BF16=0: fp32 rows with operands rounded in flight, the round-3 form); the transformer stream stays fp32.
static bool mimi_half(const sopro_engine* e) {
  static const bool off = getenv("SOPRO_MIMI_BF16") != nullptr && getenv("SOPRO_MIMI_BF16")[0] == '0';  // (run-time: tests/test_gpu_bf16_mode.py compares the two forms)
  return e->c.precision == 1 && !off;
}

// The last level (128 -> 64 channels, x 4) as ONE kernel (csrc/seanet_uptail.hip) for long inputs: its workgroups walk >= 24 tiles
// of 32 input rows each, so below ~512 Ki input rows per call the two kernels - which spread a short input over the chip - stay
// (streaming chunks, single short utterances).  SOPRO_SEANET_FUSE=0: always the two kernels (the round-3 sequence).
static bool seanet_fused(int B, int rows) {
  static const bool off = SOPRO_DEV_ENV("SOPRO_SEANET_FUSE") != nullptr && SOPRO_DEV_ENV("SOPRO_SEANET_FUSE")[0] == '0';
  return !off && (int64_t)B * rows >= 512 * 1024;
}

static size_t mimi_carve(const sopro_engine* e, MimiWs& w, void* ws, int B, int T) {
  const sopro_engine_cfg& c = e->c;
  const bool half = mimi_half(e);
  auto act = [&](Carver& cv_, size_t n) { return half ? reinterpret_cast<float*>(cv_.take<uint16_t>(n)) : cv_.take<float>(n); };
  const size_t HS = c.mimi_hidden, CD = c.mimi_codebook_dim, N2 = 2 * (size_t)T, PADX = c.mimi_kernel - 1;
  Carver cv(ws);
  w.tok = cv.take<int32_t>((size_t)B * T * c.num_codebooks);
  w.sk.tickets = cv.take<int32_t>(SPLITK_TICKETS);
  w.sk.ws = (size_t)B * N2 <= 512 ? cv.take<float>(SPLITK_WS_BYTES / 4) : nullptr;  // few rows (streaming chunks): K loops may be split
  w.emb = cv.take<float>((size_t)B * T * 2 * CD);
  w.q = cv.take<float>((size_t)B * T * HS);
  w.X = cv.take<float>((size_t)B * (PADX + N2) * HS);
  w.y = cv.take<float>((size_t)B * N2 * HS);
  w.qkv = cv.take<float>((size_t)B * N2 * 3 * HS);
  w.ao = cv.take<float>((size_t)B * N2 * HS);
  w.hd = cv.take<float>((size_t)B * N2 * c.mimi_inter);
  w.lnst = cv.take<float>((size_t)B * N2 * ((HS + 63) / 64) * 2);
  size_t ch = (size_t)c.mimi_num_filters << c.mimi_n_ratios, rows = N2;
  w.e0 = act(cv, (size_t)B * (1 + rows) * ch);
  for (int si = 0; si < c.mimi_n_ratios; ++si) {
    const size_t co = ch / 2, orow = rows * c.mimi_ratios[si];
    // (the last level's 64-channel activation stays on the CU when the level runs as one kernel - seanet_uptail: no buffer for it;
    // it was 6.3 GB of a 64 x 200-frame call's 19 GB)
    const bool on_cu = si + 1 == c.mimi_n_ratios && ch == 128 && c.mimi_ratios[si] == 4 && seanet_fused(B, (int)rows);
    w.hraw[si] = on_cu ? nullptr : act(cv, (size_t)B * (2 + orow) * co);
    w.hact[si] = si + 1 < c.mimi_n_ratios ? act(cv, (size_t)B * (2 + orow) * co) : nullptr;
    w.y1[si] = si + 1 < c.mimi_n_ratios ? act(cv, (size_t)B * orow * (co / c.mimi_compress)) : nullptr;
    ch = co; rows = orow;
  }
  return cv.off;
}

int64_t sopro_mimi_workspace_bytes(const sopro_engine* e, int32_t B, int32_t T) {
  if (!e || B <= 0 || T <= 0) return 0;
  MimiWs w;
  return (int64_t)mimi_carve(e, w, nullptr, sopro_mimi_chunk_rows(B, T), T);  // one chunk's worth (see sopro_mimi_decode)
}

// Pre-norm causal sliding-window RoPE transformer over the zero-padded residual stream X [B, pad + n (+ tail), HS], in place
// (HF:modeling_mimi.py MimiTransformerModel, 729-928): the decoder's ("tr", packed operands, optional streaming cache) and the
// encoder's ("etr", fp32 operands).  y [B n, HS], qkv [B n, 3 HS], ao [B n, HS], hd [B n, inter] are scratch.
static int transformer_stack(sopro_engine* e, hipStream_t s, const char* pre, float* X, int PADX, int64_t xs, float* y, float* qkv, float* ao,
                             float* hd, const SplitK* sk, int B, int n, int past, sopro_mimi_stream_state* sst, int attn_split = 0,
                             float* lnst = nullptr) {
  const sopro_engine_cfg& c = e->c;
  const int HS = c.mimi_hidden, H = c.mimi_heads, dh = c.mimi_head_dim;
  struct { float *X, *y, *qkv, *ao, *hd; SplitK sk; } w{X, y, qkv, ao, hd, sk ? *sk : SplitK()};
  // The two pre-norms of a layer ride on the contractions either side of them (round 5; sopro_gemm_split_ext.ln_stats): the o / fc2
  // contraction that updates the stream leaves each row's (mean, squared deviations) per 64 columns, the qkv / fc1 contraction stages
  // (x - mean) * rstd with the norm's weight folded into its W' and W lnb as its bias.  The normalised tensor is never written: 16 norm
  // passes of 105 MB read + 105 MB written at 64 x 200 frames (35 us each) become one statistics pass over the first layer's input.
  // SOPRO_LN_FUSE=0 (developer A/B): the separate norm kernels.
  static const bool ln_off = SOPRO_DEV_ENV("SOPRO_LN_FUSE") != nullptr && SOPRO_DEV_ENV("SOPRO_LN_FUSE")[0] == '0';
  const bool fused_ln = !ln_off && lnst != nullptr && WT(e, std::string(pre) + ".0.qkv.w#ln").packed != nullptr;
  if (fused_ln) STG(sopro_row_stats_f32(w.X + (size_t)PADX * HS, HS, xs, B * n, n, HS, lnst, s));
  for (int l = 0; l < c.mimi_layers; ++l) {
    const std::string p = std::string(pre) + "." + std::to_string(l);
    if (!fused_ln)
      STG(norm(s, w.X + (size_t)PADX * HS, w.y, F(e, p + ".ln1.w"), B * n, HS, c.mimi_norm_eps, SOPRO_NORM_LN, F(e, p + ".ln1.b"), nullptr, nullptr, n, xs));
    G qg; qg.sk = &w.sk; qg.M = B * n; qg.N = 3 * HS; qg.K = HS;
    if (fused_ln) {  // A = the stream itself (rows of utterance b start at b * xs)
      qg.ln_stats = lnst; qg.ln_eps = c.mimi_norm_eps; qg.bias = F(e, p + ".qkv.lnb"); qg.a_seg = xs; qg.rows_per_seg = n;
      qg.c_seg = (int64_t)n * 3 * HS;
    }
    // queries and keys are 2 H consecutive heads of dh columns.  The decoder's packed (split-bf16) contraction rotates them in its
    // epilogue (round 5: SOPRO_EPI_ROPE - the tile is in LDS, a column's partner is at hand; sopro_rope_f32 re-read and re-wrote
    // 105 MB per layer at 64 x 200 frames: 67 us x 8 per pass); the encoder's exact-fp32 contraction keeps the pass of its own
    static const bool rope_off = SOPRO_DEV_ENV("SOPRO_ROPE_FUSE") != nullptr && SOPRO_DEV_ENV("SOPRO_ROPE_FUSE")[0] == '0';  // developer A/B
    const bool fused_rope = !rope_off && WT(e, p + ".qkv.w").packed != nullptr && dh <= 64 && (dh & (dh - 1)) == 0;
    if (fused_rope) {
      qg.epi = SOPRO_EPI_ROPE; qg.rope_cos = F(e, "rope.cos"); qg.rope_sin = F(e, "rope.sin"); qg.rope_cols = 2 * H * dh; qg.rope_dh = dh;
      qg.rope_pos0 = past; qg.rope_rps = n;
    }
    STG(gemm(s, fused_ln ? w.X + (size_t)PADX * HS : w.y, WT(e, p + (fused_ln ? ".qkv.w#ln" : ".qkv.w")), nullptr, w.qkv, qg));
    if (!fused_rope) STG(sopro_rope_f32(w.qkv, 3 * HS, F(e, "rope.cos"), F(e, "rope.sin"), B * n, n, past, 2 * H, dh, s));
    sopro_attn_args a;
    memset(&a, 0, sizeof(a));
    a.Q = w.qkv; a.ldq = 3 * HS; a.q_bstride = (int64_t)n * 3 * HS;
    a.O = w.ao; a.ldo = HS; a.o_bstride = (int64_t)n * HS;
    a.B = B; a.H = H; a.dh = dh; a.Tq = n; a.causal = 1; a.window = c.mimi_window; a.scale = 1.0f / sqrtf((float)dh);
    if (!sst) {
      a.K = w.qkv + HS; a.ldk = 3 * HS; a.k_bstride = (int64_t)n * 3 * HS;
      a.V = w.qkv + 2 * HS; a.ldv = 3 * HS; a.v_bstride = (int64_t)n * 3 * HS;
      a.Tk = n;
    } else {
      // keys / values of earlier calls (post-RoPE) followed by this call's: append the (k | v) rows to the layer's cache
      float* cache = sst->kv + ((size_t)(l * 2 + sst->half) * sst->cap_rows) * 2 * HS;
      STG(sopro_copy2d_u32(cache + (size_t)sst->kv_len * 2 * HS, 2 * HS, w.qkv + HS, 3 * HS, n, 2 * HS, s));
      const int Tk = sst->kv_len + n;
      a.K = cache; a.ldk = 2 * HS; a.V = cache + HS; a.ldv = 2 * HS; a.Tk = Tk;
      a.q_pos0 = past; a.k_pos0 = past + n - Tk;
      a.q_bstride = a.o_bstride = 0;
      // what the next call sees: sliding-window layers keep the last window-1 positions (moved to the other half of the
      // layer's buffer), plain layers everything
      if (sst->evict && Tk > c.mimi_window - 1) {
        const int keep = c.mimi_window - 1;
        float* other = sst->kv + ((size_t)(l * 2 + (sst->half ^ 1)) * sst->cap_rows) * 2 * HS;
        STG(attend(a, s, attn_split));
        STG(sopro_copy2d_u32(other, 2 * HS, cache + (size_t)(Tk - keep) * 2 * HS, 2 * HS, keep, 2 * HS, s));
        goto attended;
      }
    }
    STG(attend(a, s, attn_split));
  attended:;
    G og; og.sk = &w.sk; og.M = B * n; og.N = HS; og.K = HS; og.epi = SOPRO_EPI_RES; og.R = w.X + (size_t)PADX * HS; og.scale = F(e, p + ".ls1");
    og.c_seg = xs; og.r_seg = xs; og.rows_per_seg = n;
    if (fused_ln) { og.ln_stats_out = lnst; og.a_seg = (int64_t)n * HS; }
    STG(gemm(s, w.ao, WT(e, p + ".o.w"), nullptr, w.X + (size_t)PADX * HS, og));
    if (!fused_ln)
      STG(norm(s, w.X + (size_t)PADX * HS, w.y, F(e, p + ".ln2.w"), B * n, HS, c.mimi_norm_eps, SOPRO_NORM_LN, F(e, p + ".ln2.b"), nullptr, nullptr, n, xs));
    G f1; f1.sk = &w.sk; f1.M = B * n; f1.N = c.mimi_inter; f1.K = HS; f1.epi = SOPRO_EPI_GELU;
    if (fused_ln) {
      f1.ln_stats = lnst; f1.ln_eps = c.mimi_norm_eps; f1.bias = F(e, p + ".fc1.lnb"); f1.a_seg = xs; f1.rows_per_seg = n;
      f1.c_seg = (int64_t)n * c.mimi_inter;
    }
    STG(gemm(s, fused_ln ? w.X + (size_t)PADX * HS : w.y, WT(e, p + (fused_ln ? ".fc1.w#ln" : ".fc1.w")), nullptr, w.hd, f1));
    G f2; f2.sk = &w.sk; f2.M = B * n; f2.N = HS; f2.K = c.mimi_inter; f2.epi = SOPRO_EPI_RES; f2.R = w.X + (size_t)PADX * HS; f2.scale = F(e, p + ".ls2");
    f2.c_seg = xs; f2.r_seg = xs; f2.rows_per_seg = n;
    if (fused_ln) { f2.ln_stats_out = lnst; f2.a_seg = (int64_t)n * c.mimi_inter; }
    STG(gemm(s, w.hd, WT(e, p + ".fc2.w"), nullptr, w.X + (size_t)PADX * HS, f2));
  }
  return 0;
}

// parts: 1 = every launch but the last, 2 = the last launch alone (the only one that writes `wav`), 3 = both (sopro_mimi_decode_parts)
#define BODY(call)                 \
  do {                             \
    if (parts & 1) STG(call);      \
  } while (0)
static int mimi_decode_core(sopro_engine* e, void* workspace, const int32_t* tokens, int32_t B, int32_t T, float* wav, void* stream,
                            sopro_mimi_stream_state* sst, int parts = 3) {
  SOPRO_CHECK_ARG(e && e->final && e->has_mimi && workspace && tokens && wav && B > 0 && T > 0,
                  "bad arguments (finalize the engine with the Mimi tensors first)");
  hipStream_t s = (hipStream_t)stream;
  const sopro_engine_cfg& c = e->c;
  const int Q = c.num_codebooks, HS = c.mimi_hidden, CD = c.mimi_codebook_dim, N2 = 2 * T, PADX = c.mimi_kernel - 1;
  const int ns = c.mimi_n_semantic;
  const int past = sst ? sst->pos : 0;
  SOPRO_CHECK_ARG(past + N2 <= c.mimi_rope_positions, "more positions than the RoPE tables hold");
  SOPRO_CHECK_ARG(!sst || (B == 1 && sst->kv && sst->kv_len + N2 <= sst->cap_rows), "streaming state: one utterance, kv_len + 2T <= cap_rows");
  SOPRO_CHECK_ARG(c.mimi_res_kernel == 3 && c.mimi_last_kernel == 3 && c.mimi_compress == 2, "the SEANet sequence is written for k = 3 residual / last convs, compress 2");
  MimiWs w;
  mimi_carve(e, w, workspace, B, T);
  if (w.sk.ws) BODY(sopro_fill2d_u32(w.sk.tickets, SPLITK_TICKETS, 1, SPLITK_TICKETS, 0u, s));
  // the zero rows in front of every segment of a convolution input are never written by the kernels: clear just those
  {
    BODY(sopro_fill2d_u32(w.X, (int64_t)(PADX + N2) * HS, B, PADX * HS, 0u, s));
    size_t chz = (size_t)c.mimi_num_filters << c.mimi_n_ratios, rowz = (size_t)N2;
    const int wd = mimi_half(e) ? 2 : 1;  // activation elements per 32-bit word
    BODY(sopro_fill2d_u32(w.e0, (int64_t)((1 + rowz) * chz / wd), B, (int)(chz / wd), 0u, s));
    for (int si = 0; si < c.mimi_n_ratios; ++si) {
      const size_t co = chz / 2, orow = rowz * c.mimi_ratios[si];
      if (w.hraw[si]) BODY(sopro_fill2d_u32(w.hraw[si], (int64_t)((2 + orow) * co / wd), B, (int)(2 * co / wd), 0u, s));
      if (w.hact[si]) BODY(sopro_fill2d_u32(w.hact[si], (int64_t)((2 + orow) * co / wd), B, (int)(2 * co / wd), 0u, s));
      chz = co; rowz = orow;
    }
  }
  // ---- RVQ decode + output projections (HF:modeling_mimi.py:1128-1137)
  const int64_t cb_rows = e->t["codebooks"].shape[0];
  BODY(sopro_codebook_sum_f32(tokens, Q, e->sem_col, e->sem_off, e->ones, ns, F(e, "codebooks"), cb_rows, nullptr, 0.f, 1.f, w.emb, 2 * CD, 0,
                             B * T, B * T, CD, s));
  BODY(sopro_codebook_sum_f32(tokens, Q, e->ac_col, e->ac_off, e->ones, Q - ns, F(e, "codebooks"), cb_rows, nullptr, 0.f, 1.f, w.emb + CD, 2 * CD,
                             0, B * T, B * T, CD, s));
  G pj; pj.sk = &w.sk; pj.M = B * T; pj.N = HS; pj.K = 2 * CD;
  BODY(gemm(s, w.emb, WT(e, "rvq_proj.w"), nullptr, w.q, pj));
  // ---- upsample into the zero-padded transformer stream (HF:1208-1216)
  const int64_t xs = (int64_t)(PADX + N2) * HS;
  BODY(sopro_upsample2_f32(w.q, F(e, "upsample.w"), w.X + (size_t)PADX * HS, xs, B, T, HS, s));
  // ---- transformer: 8 pre-LN layers, RoPE, causal window (HF:729-928)
  // the decoder's attention on the waveform path's operand precision (two bf16 pieces, three passes; one in bf16 mode when
  // SOPRO_ATTN_PASSES=1); SOPRO_ATTN_SPLIT=0 keeps the exact-fp32 kernel the encoder uses
  static const bool attn_exact = SOPRO_DEV_ENV("SOPRO_ATTN_SPLIT") != nullptr && SOPRO_DEV_ENV("SOPRO_ATTN_SPLIT")[0] == '0';
  static const bool attn_one = SOPRO_DEV_ENV("SOPRO_ATTN_PASSES") != nullptr && SOPRO_DEV_ENV("SOPRO_ATTN_PASSES")[0] == '1';
  const int attn_split = attn_exact ? 0 : ((c.precision == 1 && attn_one) ? 1 : 3);
  BODY(transformer_stack(e, s, "tr", w.X, PADX, xs, w.y, w.qkv, w.ao, w.hd, &w.sk, B, N2, past, sst, attn_split, w.lnst));
  static const bool three = SOPRO_DEV_ENV("SOPRO_SEANET_PASSES3") != nullptr;  // developer A/B: the fused kernels' three-pass form in bf16 mode too
  const int sea_passes = (c.precision == 1 && !three) ? 1 : 3;
  // ---- SEANet decoder (HF:931-961), activated-copy flow of sopro_amd.codec.MimiCodec._seanet_act
  int ch = c.mimi_num_filters << c.mimi_n_ratios, rows = N2, pad_in = 1;
  if (mimi_half(e)) {
    // bf16 mode: the same flow with every activation of the decoder as bf16 rows in memory (strides below count bf16 elements):
    // the first convolution reads the fp32 transformer stream and writes ELU(.) as bf16; every later contraction reads bf16
    // rows (a_format 2: staged by plain copies) and writes bf16 rows (c_mode 6 raw / 7 activated / 8 both); the fused
    // kernels of the two 24 kHz-side levels have bf16-row forms.  Accumulation, bias, ELU and the skip additions are fp32.
    auto HP = [](float* p, size_t n) { return reinterpret_cast<float*>(reinterpret_cast<uint16_t*>(p) + n); };
    {
      G g; g.sk = &w.sk; g.M = B * rows; g.N = ch; g.K = c.mimi_kernel * HS; g.lda = HS; g.bias = F(e, "sea.conv0.b"); g.rows_per_seg = rows; g.a_seg = xs;
      g.c_seg = (int64_t)(1 + rows) * ch; g.ldc = ch; g.c_mode = 7;
      BODY(gemm(s, w.X, WT(e, "sea.conv0.w"), nullptr, HP(w.e0, ch), g));
    }
    float* He16 = w.e0;
    for (int si = 0; si < c.mimi_n_ratios; ++si) {
      const int r = c.mimi_ratios[si], co = ch / 2, orow = rows * r, hid = co / c.mimi_compress;
      const bool last = si == c.mimi_n_ratios - 1;
      const std::string u = "sea.up" + std::to_string(si), rs = "sea.res" + std::to_string(si);
      float* Ho = w.hraw[si];
      G up; up.sk = &w.sk; up.M = B * rows; up.N = r * co; up.K = 2 * ch; up.lda = ch; up.bias = F(e, u + ".b"); up.rows_per_seg = rows; up.a_fmt = 2;
      up.a_seg = (int64_t)(pad_in + rows) * ch; up.c_seg = (int64_t)(2 + orow) * co; up.ldc = (int64_t)r * co;
      float* A = HP(He16, (size_t)(pad_in - 1) * ch);
      if (last) {
        SOPRO_CHECK_ARG(co == 64 && hid == 32, "the fused tail is written for a 64-channel last stage");
        if (ch == 128 && r == 4 && seanet_fused(B, rows)) {  // the whole level in one kernel: h never reaches memory
          if (!(parts & 2)) return 0;
          sopro_prof_scope prof("seanet_uptail_kernel", 2.0 * B * rows * 256 * 256 + 2.0 * B * orow * (3 * 64 * 32 + 32 * 64 + 3 * 64), s);
          return sopro_seanet_uptail_bf16(A, up.a_seg, F(e, u + ".w"), F(e, u + ".b"), F(e, rs + ".c1.w"), F(e, rs + ".c1.b"), F(e, rs + ".c2.w"),
                                          F(e, rs + ".c2.b"), F(e, "sea.final.w"), c.mimi_final_bias, wav, orow, B, rows, s);
        }
        if (ch == 128 && r == 4) {
          sopro_prof_scope prof((parts & 1) ? "seanet_up128_kernel" : nullptr, 2.0 * B * rows * 256 * 256, s);
          BODY(sopro_seanet_up128_bf16(A, up.a_seg, F(e, u + ".w"), F(e, u + ".b"), HP(Ho, 2 * co), up.c_seg, B, rows, s));
        } else {
          up.c_mode = 6;
          BODY(gemm(s, A, WT(e, u + ".w"), nullptr, HP(Ho, 2 * co), up));
        }
        if (!(parts & 2)) return 0;
        sopro_prof_scope prof("seanet_tail_kernel", 2.0 * B * orow * (3 * 64 * 32 + 32 * 64 + 3 * 64), s);
        return sopro_seanet_tail_bf16(Ho, (int64_t)(2 + orow) * co, F(e, rs + ".c1.w"), F(e, rs + ".c1.b"), F(e, rs + ".c2.w"), F(e, rs + ".c2.b"),
                                      F(e, "sea.final.w"), c.mimi_final_bias, wav, orow, B, orow, s);
      }
      float* Hn = w.hact[si];
      if (co == 128 && hid == 64) {
        up.c_mode = 6;
        BODY(gemm(s, A, WT(e, u + ".w"), nullptr, HP(Ho, 2 * co), up));
        sopro_prof_scope prof((parts & 1) ? "seanet_res128_kernel" : nullptr, 2.0 * B * orow * (3 * 128 * 64 + 64 * 128), s);
        BODY(sopro_seanet_res128_bf16(Ho, (int64_t)(2 + orow) * co, F(e, rs + ".c1.w"), F(e, rs + ".c1.b"), F(e, rs + ".c2.w"), F(e, rs + ".c2.b"), Hn,
                                     (int64_t)(2 + orow) * co, B, orow, s));
      } else {
        up.c_mode = 8; up.C2 = HP(Hn, 2 * co); up.ldc2 = (int64_t)r * co; up.c2_seg = (int64_t)(2 + orow) * co;
        BODY(gemm(s, A, WT(e, u + ".w"), nullptr, HP(Ho, 2 * co), up));
        G c1; c1.sk = &w.sk; c1.M = B * orow; c1.N = hid; c1.K = 3 * co; c1.lda = co; c1.bias = F(e, rs + ".c1.b"); c1.rows_per_seg = orow; c1.a_fmt = 2;
        c1.a_seg = (int64_t)(2 + orow) * co; c1.c_mode = 7;
        BODY(gemm(s, Hn, WT(e, rs + ".c1.w"), nullptr, w.y1[si], c1));
        G c2; c2.sk = &w.sk; c2.M = B * orow; c2.N = co; c2.K = hid; c2.bias = F(e, rs + ".c2.b"); c2.epi = SOPRO_EPI_RES; c2.R = HP(Ho, 2 * co); c2.rows_per_seg = orow;
        c2.a_fmt = 2; c2.c_seg = (int64_t)(2 + orow) * co; c2.r_seg = (int64_t)(2 + orow) * co; c2.ldc = co; c2.ldr = co; c2.c_mode = 7;
        BODY(gemm(s, w.y1[si], WT(e, rs + ".c2.w"), nullptr, HP(Hn, 2 * co), c2));
      }
      He16 = Hn; ch = co; rows = orow; pad_in = 2;
    }
    sopro_set_error("sopro_mimi_decode: the decoder has no last stage");
    return -2;
  }
  // Round 6: the levels whose contractions have K >= 1024 hand their ACTIVATED tensors over in split form (every 32 channels =
  // [32 hi | 32 lo] bf16: same bytes, lines and strides as fp32) - the producer's epilogue splits once (c_mode 1 / 2), and the long-K
  // form of the contraction (gemm_8p.hip: both operands by LDS-DMA) takes them where there are tiles enough; the tile kernel reads the
  // same tensors (a_format 1) when there are not (streaming chunks).  `in_split`: this level's input is in split form.
  auto rows_of = [&](const std::string& k) { return WT(e, k).rows != nullptr; };
  static const bool no_8p = SOPRO_DEV_ENV("SOPRO_GEMM_8P") != nullptr && SOPRO_DEV_ENV("SOPRO_GEMM_8P")[0] == '0';  // developer A/B: the round-5 flow
  bool in_split = !no_8p && rows_of("sea.up0.w");
  {  // first conv k = 7 -> ELU; one zero row in front = x[t-1] of the transposed conv
    G g; g.sk = &w.sk; g.M = B * rows; g.N = ch; g.K = c.mimi_kernel * HS; g.lda = HS; g.bias = F(e, "sea.conv0.b"); g.rows_per_seg = rows; g.a_seg = xs;
    g.c_seg = (int64_t)(1 + rows) * ch; g.ldc = ch; g.c_mode = in_split ? 1 : 3;
    BODY(gemm(s, w.X, WT(e, "sea.conv0.w"), nullptr, w.e0 + ch, g));
  }
  const float* He = w.e0;
  for (int si = 0; si < c.mimi_n_ratios; ++si) {
    const int r = c.mimi_ratios[si], co = ch / 2, orow = rows * r, hid = co / c.mimi_compress;
    const bool last = si == c.mimi_n_ratios - 1;
    const std::string u = "sea.up" + std::to_string(si), rs = "sea.res" + std::to_string(si);
    float* Ho = w.hraw[si];
    G up; up.sk = &w.sk; up.M = B * rows; up.N = r * co; up.K = 2 * ch; up.lda = ch; up.bias = F(e, u + ".b"); up.rows_per_seg = rows;
    up.a_seg = (int64_t)(pad_in + rows) * ch; up.c_seg = (int64_t)(2 + orow) * co; up.ldc = (int64_t)r * co;
    up.a_fmt = in_split ? 1 : 0;
    const float* A = He + (size_t)(pad_in - 1) * ch;
    if (last) {
      SOPRO_CHECK_ARG(co == 64 && hid == 32, "the fused tail is written for a 64-channel last stage");
      if (ch == 128 && r == 4 && seanet_fused(B, rows)) {  // the whole level in one kernel: h never reaches memory
        if (!(parts & 2)) return 0;
        sopro_prof_scope prof("seanet_uptail_kernel", 2.0 * B * rows * 256 * 256 + 2.0 * B * orow * (3 * 64 * 32 + 32 * 64 + 3 * 64), s);
        return sopro_seanet_uptail_f32(A, up.a_seg, F(e, u + ".w"), F(e, u + ".b"), F(e, rs + ".c1.w"), F(e, rs + ".c1.b"), F(e, rs + ".c2.w"),
                                       F(e, rs + ".c2.b"), F(e, "sea.final.w"), c.mimi_final_bias, wav, orow, B, rows, sea_passes, s);
      }
      if (ch == 128 && r == 4) {  // weight-stationary form of the K = 256, N = 256 contraction (same results)
        sopro_prof_scope prof((parts & 1) ? "seanet_up128_kernel" : nullptr, 2.0 * B * rows * 256 * 256, s);
        BODY(sopro_seanet_up128_f32(A, up.a_seg, F(e, u + ".w"), F(e, u + ".b"), Ho + 2 * co, up.c_seg, B, rows, c.precision == 1 ? 1 : 3, s));
      } else {
        BODY(gemm(s, A, WT(e, u + ".w"), nullptr, Ho + 2 * co, up));
      }
      // last residual block (k=3 conv 64->32, k=1 conv 32->64) + final k=3 conv 64->1 per output sample
      if (!(parts & 2)) return 0;
      sopro_prof_scope prof("seanet_tail_kernel", 2.0 * B * orow * (3 * 64 * 32 + 32 * 64 + 3 * 64), s);
      return sopro_seanet_tail_p_f32(Ho, (int64_t)(2 + orow) * co, F(e, rs + ".c1.w"), F(e, rs + ".c1.b"), F(e, rs + ".c2.w"), F(e, rs + ".c2.b"),
                                     F(e, "sea.final.w"), c.mimi_final_bias, wav, orow, B, orow, sea_passes, s);
    }
    float* Hn = w.hact[si];
    if (co == 128 && hid == 64) {
      BODY(gemm(s, A, WT(e, u + ".w"), nullptr, Ho + 2 * co, up));
      sopro_prof_scope prof((parts & 1) ? "seanet_res128_kernel" : nullptr, 2.0 * B * orow * (3 * 128 * 64 + 64 * 128), s);
      BODY(sopro_seanet_res128_p_f32(Ho, (int64_t)(2 + orow) * co, F(e, rs + ".c1.w"), F(e, rs + ".c1.b"), F(e, rs + ".c2.w"), F(e, rs + ".c2.b"), Hn,
                                    (int64_t)(2 + orow) * co, B, orow, sea_passes, s));
      in_split = false;
    } else {
      // the block's tensors in split form when its k = 3 convolution has the long-K operand; its output (only ever read through ELU,
      // by the next level) goes on in split form with them
      // (never into the last level: its fused kernels read fp32 rows)
      const bool mid_split = !no_8p && rows_of(rs + ".c1.w") && (co & 31) == 0 && (hid & 31) == 0 && si + 1 < c.mimi_n_ratios - 1;
      up.c_mode = mid_split ? 2 : 4; up.C2 = Hn + 2 * co; up.ldc2 = (int64_t)r * co; up.c2_seg = (int64_t)(2 + orow) * co;
      BODY(gemm(s, A, WT(e, u + ".w"), nullptr, Ho + 2 * co, up));
      // residual block: x + Conv1d(k=1)(ELU(Conv1d(k=3)(ELU(x)))); its output is only ever read through ELU
      G c1; c1.sk = &w.sk; c1.M = B * orow; c1.N = hid; c1.K = 3 * co; c1.lda = co; c1.bias = F(e, rs + ".c1.b"); c1.rows_per_seg = orow;
      c1.a_seg = (int64_t)(2 + orow) * co; c1.c_mode = mid_split ? 1 : 3; c1.a_fmt = mid_split ? 1 : 0;
      BODY(gemm(s, Hn, WT(e, rs + ".c1.w"), nullptr, w.y1[si], c1));
      G c2; c2.sk = &w.sk; c2.M = B * orow; c2.N = co; c2.K = hid; c2.bias = F(e, rs + ".c2.b"); c2.epi = SOPRO_EPI_RES; c2.R = Ho + 2 * co; c2.rows_per_seg = orow;
      c2.c_seg = (int64_t)(2 + orow) * co; c2.r_seg = (int64_t)(2 + orow) * co; c2.ldc = co; c2.ldr = co; c2.c_mode = mid_split ? 1 : 3;
      c2.a_fmt = mid_split ? 1 : 0;
      BODY(gemm(s, w.y1[si], WT(e, rs + ".c2.w"), nullptr, Hn + 2 * co, c2));
      in_split = mid_split;
    }
    He = Hn; ch = co; rows = orow; pad_in = 2;
  }
  sopro_set_error("sopro_mimi_decode: the decoder has no last stage");
  return -2;
}

#undef BODY

// ------------------------------------------------------------------------------------------------ Mimi encode
// HF:modeling_mimi.py MimiModel._encode_frame: SEANet encoder (strided convs = overlapping-row windows of the padded level
// buffers) -> transformer -> stride-2 downsample -> split residual VQ.  Exact-fp32 contractions throughout.
struct EncPlan {
  float* H[9]; int rows[9], pad[9], len[9], ch[9];  // level buffers [B][rows][ch]: pad zero rows, len rows in use, a zero tail
  float *Y1, *X, *y, *qkv, *ao, *hd, *Dn, *res, *scores;
  int T25, T, pad_x, tail_x;
};
static size_t enc_carve(const sopro_engine* e, EncPlan& p, void* ws, int B, int N) {
  const sopro_engine_cfg& c = e->c;
  const int nr = c.mimi_n_ratios, rk = c.mimi_res_kernel, lk = c.mimi_last_kernel, HS = c.mimi_hidden;
  Carver cv(ws);
  int ch = c.mimi_num_filters, L = N;
  size_t y1max = 0;
  auto ratio = [&](int si) { return c.mimi_ratios[nr - 1 - si]; };  // the encoder walks the decoder's ratios backwards
  p.pad[0] = std::max(ratio(0), rk - 1);
  p.len[0] = L; p.ch[0] = ch;
  p.rows[0] = p.pad[0] + (L + ratio(0) - 1) / ratio(0) * ratio(0);
  p.H[0] = cv.take<float>((size_t)B * p.rows[0] * ch);
  for (int si = 0; si < nr; ++si) {
    const int r = ratio(si), Lo = (L + r - 1) / r;
    y1max = std::max(y1max, (size_t)B * L * (ch / c.mimi_compress));
    const bool last = si == nr - 1;
    const int npad = last ? lk - 1 : std::max(ratio(si + 1), rk - 1);
    const int ntail = last ? 0 : (Lo + ratio(si + 1) - 1) / ratio(si + 1) * ratio(si + 1) - Lo;
    p.pad[si + 1] = npad; p.len[si + 1] = Lo; p.ch[si + 1] = 2 * ch; p.rows[si + 1] = npad + Lo + ntail;
    p.H[si + 1] = cv.take<float>((size_t)B * p.rows[si + 1] * 2 * ch);
    ch *= 2; L = Lo;
  }
  p.Y1 = cv.take<float>(y1max);
  p.T25 = L; p.T = (L + 1) / 2; p.pad_x = 2; p.tail_x = p.T * 2 - L;
  const size_t xr = (size_t)p.pad_x + L + p.tail_x;
  p.X = cv.take<float>((size_t)B * xr * HS);
  p.y = cv.take<float>((size_t)B * L * HS); p.qkv = cv.take<float>((size_t)B * L * 3 * HS); p.ao = cv.take<float>((size_t)B * L * HS);
  p.hd = cv.take<float>((size_t)B * L * c.mimi_inter);
  p.Dn = cv.take<float>((size_t)B * p.T * HS); p.res = cv.take<float>((size_t)B * p.T * c.mimi_codebook_dim);
  p.scores = cv.take<float>((size_t)B * p.T * c.codebook_size);
  return cv.off;
}

int64_t sopro_mimi_encode_workspace_bytes(const sopro_engine* e, int32_t B, int32_t N) {
  if (!e || B <= 0 || N <= 0) return 0;
  EncPlan p;
  return (int64_t)enc_carve(e, p, nullptr, B, N);
}

int sopro_mimi_encode(sopro_engine* e, void* workspace, const float* wav, int32_t B, int32_t N, int32_t* codes, void* stream) {
  SOPRO_CHECK_ARG(e && e->final && e->has_enc && workspace && wav && codes && B > 0 && N > 0,
                  "bad arguments (finalize the engine with the Mimi encoder tensors first)");
  hipStream_t s = (hipStream_t)stream;
  const sopro_engine_cfg& c = e->c;
  const int nr = c.mimi_n_ratios, rk = c.mimi_res_kernel, lk = c.mimi_last_kernel, k0 = c.mimi_kernel, HS = c.mimi_hidden, CD = c.mimi_codebook_dim;
  const int V = c.codebook_size, Q = c.num_codebooks, ns = c.mimi_n_semantic;
  SOPRO_CHECK_ARG(c.mimi_compress == 2, "the encoder sequence is written for compress 2");
  EncPlan p;
  enc_carve(e, p, workspace, B, N);
  SOPRO_CHECK_ARG(p.T25 <= c.mimi_rope_positions, "more positions than the RoPE tables hold");
  for (int i = 0; i <= nr; ++i) STG(sopro_fill2d_u32(p.H[i], (int64_t)p.rows[i] * p.ch[i], B, p.rows[i] * p.ch[i], 0u, s));  // pads and tails read as zeros
  // first conv 1 -> 64, k = 7, causal
  STG(sopro_fir1_f32(wav, N, N, F(e, "enc.conv0.w"), F(e, "enc.conv0.b"), p.H[0] + (size_t)p.pad[0] * p.ch[0], p.ch[0], (int64_t)p.rows[0] * p.ch[0], B, N,
                     p.ch[0], k0, 1, k0 - 1, s));
  for (int si = 0; si < nr; ++si) {
    const int r = c.mimi_ratios[nr - 1 - si], ch = p.ch[si], L = p.len[si], pad = p.pad[si], hid = ch / c.mimi_compress, Lo = p.len[si + 1];
    const int64_t seg = (int64_t)p.rows[si] * ch;
    const std::string rs = "enc.res" + std::to_string(si), dn = "enc.down" + std::to_string(si);
    float* Hc = p.H[si];
    // residual block: x + Conv1d(k=1)(ELU(Conv1d(k=3)(ELU(x))))
    G c1; c1.M = B * L; c1.N = hid; c1.K = rk * ch; c1.lda = ch; c1.bias = F(e, rs + ".c1.b"); c1.pro = SOPRO_PRO_ELU; c1.rows_per_seg = L; c1.a_seg = seg;
    STG(gemm(s, Hc + (size_t)(pad - (rk - 1)) * ch, WT(e, rs + ".c1.w"), nullptr, p.Y1, c1));
    G c2; c2.M = B * L; c2.N = ch; c2.K = hid; c2.bias = F(e, rs + ".c2.b"); c2.pro = SOPRO_PRO_ELU; c2.epi = SOPRO_EPI_RES; c2.R = Hc + (size_t)pad * ch;
    c2.rows_per_seg = L; c2.c_seg = seg; c2.r_seg = seg; c2.ldc = ch; c2.ldr = ch;
    STG(gemm(s, p.Y1, WT(e, rs + ".c2.w"), nullptr, Hc + (size_t)pad * ch, c2));
    // ELU -> Conv1d(ch -> 2 ch, k = 2 r, stride r): frame j contracts rows j r - r .. j r + r - 1
    G d; d.M = B * Lo; d.N = 2 * ch; d.K = 2 * r * ch; d.lda = (int64_t)r * ch; d.bias = F(e, dn + ".b"); d.pro = SOPRO_PRO_ELU; d.rows_per_seg = Lo; d.a_seg = seg;
    d.c_seg = (int64_t)p.rows[si + 1] * 2 * ch; d.ldc = 2 * ch;
    STG(gemm(s, Hc + (size_t)(pad - r) * ch, WT(e, dn + ".w"), nullptr, p.H[si + 1] + (size_t)p.pad[si + 1] * 2 * ch, d));
  }
  // ELU -> last conv (k = 3) into the transformer stream, which carries the downsample conv's replicate pads
  const int T25 = p.T25, T = p.T, chl = p.ch[nr];
  const int64_t xs = (int64_t)(p.pad_x + T25 + p.tail_x) * HS;
  G fc; fc.M = B * T25; fc.N = HS; fc.K = lk * chl; fc.lda = chl; fc.bias = F(e, "enc.final.b"); fc.pro = SOPRO_PRO_ELU; fc.rows_per_seg = T25;
  fc.a_seg = (int64_t)p.rows[nr] * chl; fc.c_seg = xs; fc.ldc = HS;
  STG(gemm(s, p.H[nr], WT(e, "enc.final.w"), nullptr, p.X + (size_t)p.pad_x * HS, fc));
  STG(transformer_stack(e, s, "etr", p.X, p.pad_x, xs, p.y, p.qkv, p.ao, p.hd, nullptr, B, T25, 0, nullptr));
  // downsample: Conv1d(k = 4, stride 2, no bias), replicate padding on both sides
  for (int r = 0; r < p.pad_x; ++r) STG(sopro_copy2d_u32(p.X + (size_t)r * HS, xs, p.X + (size_t)p.pad_x * HS, xs, B, HS, s));
  for (int r = 0; r < p.tail_x; ++r)
    STG(sopro_copy2d_u32(p.X + (size_t)(p.pad_x + T25 + r) * HS, xs, p.X + (size_t)(p.pad_x + T25 - 1) * HS, xs, B, HS, s));
  G ds; ds.M = B * T; ds.N = HS; ds.K = 4 * HS; ds.lda = 2 * HS; ds.rows_per_seg = T; ds.a_seg = xs;
  STG(gemm(s, p.X, WT(e, "enc.ds.w"), nullptr, p.Dn, ds));
  // split residual VQ: semantic group (input_proj + ns layers), acoustic group (input_proj + the rest)
  for (int grp = 0; grp < 2; ++grp) {
    G ip; ip.M = B * T; ip.N = CD; ip.K = HS;
    STG(gemm(s, p.Dn, WT(e, grp == 0 ? "enc.inproj.sem.w" : "enc.inproj.ac.w"), nullptr, p.res, ip));
    for (int q = grp == 0 ? 0 : ns; q < (grp == 0 ? ns : Q); ++q) {
      const float* table = F(e, "codebooks") + (size_t)q * V * CD;
      G sc; sc.M = B * T; sc.N = V; sc.K = CD; sc.bias = F(e, "enc.cb_bias") + (size_t)q * V;
      Wt wq; wq.f32 = table;
      STG(gemm(s, p.res, wq, nullptr, p.scores, sc));
      STG(sopro_rvq_assign_f32(p.scores, V, V, table, p.res, CD, CD, codes + q, Q, B * T, s));
    }
  }
  return 0;
}

// Balanced row chunks of at most ~`cells` frames each: 64 x 400 -> 2 x 32 rows, 65 x 200 -> 33 + 32 (never a one-row remainder with a
// workspace and a recorded sequence of its own: ADVICE r4)
int32_t sopro_mimi_chunk_rows(int32_t B, int32_t T) {
  if (B <= 0 || T <= 0) return 0;
  // read ONCE per process: the sizing call and the decode call must agree on the chunk (a value that changed in between
  // made sopro_mimi_decode run chunks larger than the workspace it was given)
  static const int64_t cells = [] {
    const char* env = getenv("SOPRO_MIMI_CHUNK_CELLS");
    return env ? std::max<int64_t>(1, atoll(env)) : (int64_t)12800;
  }();
  const int64_t rows_max = std::max<int64_t>(1, cells / T);
  if (B <= rows_max) return B;
  const int64_t nchunks = (B + rows_max - 1) / rows_max;
  return (int32_t)((B + nchunks - 1) / nchunks);
}

int sopro_mimi_decode_parts(sopro_engine* e, void* workspace, const int32_t* tokens, int32_t B, int32_t T, float* wav, int32_t parts, void* stream) {
  SOPRO_CHECK_ARG(parts >= 1 && parts <= 3, "parts: 1 = all launches but the last, 2 = the last, 3 = both");
  // ONE chunk: sopro_mimi_workspace_bytes(e, B, T) sizes the workspace for sopro_mimi_chunk_rows(B, T) rows, so more rows than that
  // would run past it - a host with a larger batch calls sopro_mimi_decode, or this function once per chunk
  SOPRO_CHECK_ARG(e && B > 0 && T > 0 && B <= sopro_mimi_chunk_rows(B, T),
                  "sopro_mimi_decode_parts decodes one chunk: B must not exceed sopro_mimi_chunk_rows(B, T)");
  return mimi_decode_core(e, workspace, tokens, B, T, wav, stream, nullptr, parts);
}

int sopro_mimi_decode(sopro_engine* e, void* workspace, const int32_t* tokens, int32_t B, int32_t T, float* wav, void* stream) {
  SOPRO_CHECK_ARG(e && B > 0 && T > 0, "bad arguments");
  const int rows = sopro_mimi_chunk_rows(B, T);
  int64_t hop = 2;  // samples per frame: the x2 upsample times the SEANet ratios (1920)
  for (int i = 0; i < e->c.mimi_n_ratios; ++i) hop *= e->c.mimi_ratios[i];
  for (int b0 = 0; b0 < B; b0 += rows) {  // stream order: a chunk's launches follow the previous chunk's last reader of the shared workspace
    const int bc = std::min(rows, B - b0);
    STG(mimi_decode_core(e, workspace, tokens + (int64_t)b0 * T * e->c.num_codebooks, bc, T, wav + (int64_t)b0 * T * hop, stream, nullptr));
  }
  return 0;
}

int64_t sopro_mimi_stream_kv_bytes(const sopro_engine* e, int32_t cap_rows) {
  if (!e || cap_rows <= 0) return 0;
  return (int64_t)e->c.mimi_layers * 2 * cap_rows * 2 * e->c.mimi_hidden * 4;
}

int sopro_mimi_stream_init(const sopro_engine* e, sopro_mimi_stream_state* st, void* kv, int32_t cap_rows) {
  SOPRO_CHECK_ARG(e && st && kv && cap_rows >= e->c.mimi_window, "NULL argument or cap_rows below the attention window");
  st->kv = reinterpret_cast<float*>(kv);
  st->cap_rows = cap_rows;
  st->kv_len = 0; st->pos = 0; st->evict = 1; st->half = 0;
  return 0;
}

int sopro_mimi_stream_trim(sopro_mimi_stream_state* st, int32_t n) {
  SOPRO_CHECK_ARG(st && n >= 0, "bad arguments");
  if (st->kv_len == 0 || n == 0) return 0;
  st->kv_len = st->kv_len > n ? st->kv_len - n : 0;
  st->pos = st->kv_len;  // positions continue from the trimmed length: the rebuilt cache reports its own length
  st->evict = 0;
  return 0;
}

int sopro_mimi_decode_stream(sopro_engine* e, void* workspace, sopro_mimi_stream_state* st, const int32_t* tokens, int32_t T, float* wav,
                             void* stream) {
  SOPRO_CHECK_ARG(st != nullptr, "state is NULL");
  const int n = 2 * T, Tk = st->kv_len + n;
  const int rc = mimi_decode_core(e, workspace, tokens, 1, T, wav, stream, st);
  if (rc != 0) return rc;
  if (st->evict && Tk > e->c.mimi_window - 1) {
    st->kv_len = e->c.mimi_window - 1;
    st->half ^= 1;
  } else {
    st->kv_len = Tk;
  }
  st->pos += n;
  return 0;
}

}  // extern "C"
