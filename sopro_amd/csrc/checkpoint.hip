// Checkpoint -> engine WITHOUT Python (round 4; SURVEY.md 8b: "names = reference state_dict keys").
//
// A host that is not Python starts where the reference starts (src/sopro/hub.py:30-52, src/sopro/model.py:419-451): from
//   * model.safetensors          - SoproTTSModel.state_dict() names ("ar.blocks.0.glu.pro.weight", "ar.x_attns.1.q_proj.weight",
//                                   "cb_embed.emb.weight", "nar.heads.B.0.weight", ...), the config JSON in the header's
//                                   __metadata__["cfg"] (hub.py:30-48, key-intersection load: hub.py:44-48)
//   * the Mimi model.safetensors - HuggingFace MimiModel.state_dict() names (src/sopro/codec/mimi.py:28-31)
// sopro_checkpoint_open parses both files (safetensors: 8-byte header length, JSON table, raw little-endian tensors; F32 / F16 /
// BF16 / F64 are read as fp32) and applies, on the host, exactly the repacking sopro_amd/pack.py applies for the Python host: GLU
// value / gate interleave, tap-major convolution weights, [s Cout, 2 Cin] transposed-convolution matrices, RMSNorm weights folded
// into the projections they feed, head-id embeddings folded into the head biases and the unfolded-key query operands in float64,
// codebooks = embed_sum / clamp(cluster_usage), softmax / tanh of the small scalar parameters - under the SAME packed names the
// engine already takes (sopro_engine_set_tensor), plus the position table "pe" (src/sopro/nn/embeddings.py:11-25) and the RoPE
// tables (HF:modeling_mimi.py:511-566).  sopro_engine_from_checkpoint uploads them into engine-owned device memory and
// finalizes.  tests/test_checkpoint_c.py compares every packed tensor with pack.py's on the CPU (no GPU needed) and
// tests/test_gpu_stages.py runs a whole utterance on an engine built this way.  Host code only, no kernels.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <exception>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "common.h"

namespace {

// ------------------------------------------------------------------------------------------------ a small JSON reader
struct Json {
  enum Kind { Null, Bool, Num, Str, Arr, Obj } kind = Null;
  double num = 0;
  bool b = false;
  std::string str;
  std::vector<Json> arr;
  std::vector<std::pair<std::string, Json>> obj;
  const Json* get(const std::string& k) const {
    for (const auto& kv : obj)
      if (kv.first == k) return &kv.second;
    return nullptr;
  }
};

// The buffer [p, end) MUST be followed by a NUL byte (the callers parse std::string copies): strtod needs a terminator.
struct JsonParser {
  const char* p;
  const char* end;
  bool ok = true;
  int depth = 0;
  static const int kMaxDepth = 64;  // a header is two levels deep; "[[[[..." must not overflow the stack
  static void utf8(std::string* out, uint32_t cp) {
    if (cp < 0x80) *out += (char)cp;
    else if (cp < 0x800) { *out += (char)(0xC0 | (cp >> 6)); *out += (char)(0x80 | (cp & 63)); }
    else if (cp < 0x10000) { *out += (char)(0xE0 | (cp >> 12)); *out += (char)(0x80 | ((cp >> 6) & 63)); *out += (char)(0x80 | (cp & 63)); }
    else { *out += (char)(0xF0 | (cp >> 18)); *out += (char)(0x80 | ((cp >> 12) & 63)); *out += (char)(0x80 | ((cp >> 6) & 63)); *out += (char)(0x80 | (cp & 63)); }
  }
  bool hex4(const char* q, uint32_t* v) const {
    if (end - q < 4) return false;
    uint32_t x = 0;
    for (int i = 0; i < 4; ++i) {
      const char c = q[i];
      x <<= 4;
      if (c >= '0' && c <= '9') x |= (uint32_t)(c - '0');
      else if (c >= 'a' && c <= 'f') x |= (uint32_t)(c - 'a' + 10);
      else if (c >= 'A' && c <= 'F') x |= (uint32_t)(c - 'A' + 10);
      else return false;
    }
    *v = x;
    return true;
  }
  void ws() { while (p < end && (*p == ' ' || *p == '\n' || *p == '\t' || *p == '\r')) ++p; }
  bool lit(const char* s) {
    const size_t n = strlen(s);
    if ((size_t)(end - p) >= n && memcmp(p, s, n) == 0) { p += n; return true; }
    return false;
  }
  std::string string() {
    std::string out;
    if (p >= end || *p != '"') { ok = false; return out; }
    ++p;
    while (p < end && *p != '"') {
      if (*p == '\\' && p + 1 < end) {
        ++p;
        switch (*p) {
          case 'n': out += '\n'; break;
          case 't': out += '\t'; break;
          case 'r': out += '\r'; break;
          case 'b': out += '\b'; break;
          case 'f': out += '\f'; break;
          case 'u': {  // \uXXXX (+ a surrogate pair) -> UTF-8
            uint32_t cp = 0, lo = 0;
            if (!hex4(p + 1, &cp)) { ok = false; return out; }
            p += 4;
            if (cp >= 0xD800 && cp < 0xDC00 && end - p >= 7 && p[1] == '\\' && p[2] == 'u' && hex4(p + 3, &lo) && lo >= 0xDC00 && lo < 0xE000) {
              cp = 0x10000 + ((cp - 0xD800) << 10) + (lo - 0xDC00);
              p += 6;
            } else if (cp >= 0xD800 && cp < 0xE000) cp = 0xFFFD;  // a lone surrogate
            utf8(&out, cp);
            break;
          }
          default: out += *p;
        }
        ++p;
      } else {
        out += *p++;
      }
    }
    if (p >= end) { ok = false; return out; }
    ++p;
    return out;
  }
  Json value() {
    Json j;
    if (++depth > kMaxDepth) { ok = false; --depth; return j; }
    j = value_at_depth();
    --depth;
    return j;
  }
  Json value_at_depth() {
    Json j;
    ws();
    if (p >= end) { ok = false; return j; }
    if (*p == '{') {
      j.kind = Json::Obj;
      ++p; ws();
      if (p < end && *p == '}') { ++p; return j; }
      while (ok) {
        ws();
        std::string k = string();
        ws();
        if (p >= end || *p != ':') { ok = false; break; }
        ++p;
        j.obj.emplace_back(std::move(k), value());
        ws();
        if (p < end && *p == ',') { ++p; continue; }
        if (p < end && *p == '}') { ++p; break; }
        ok = false;
      }
    } else if (*p == '[') {
      j.kind = Json::Arr;
      ++p; ws();
      if (p < end && *p == ']') { ++p; return j; }
      while (ok) {
        j.arr.push_back(value());
        ws();
        if (p < end && *p == ',') { ++p; continue; }
        if (p < end && *p == ']') { ++p; break; }
        ok = false;
      }
    } else if (*p == '"') {
      j.kind = Json::Str;
      j.str = string();
    } else if (lit("true")) { j.kind = Json::Bool; j.b = true;
    } else if (lit("false")) { j.kind = Json::Bool; j.b = false;
    } else if (lit("null")) { j.kind = Json::Null;
    } else {
      char* e = nullptr;
      j.kind = Json::Num;
      j.num = strtod(p, &e);  // (NUL-terminated buffer: see the struct's note)
      if (e == p || e > end) { ok = false; return j; }
      p = e;
    }
    return j;
  }
};

// ------------------------------------------------------------------------------------------------ safetensors
struct RawTensor {
  std::string dtype;
  std::vector<int64_t> shape;
  size_t begin = 0, end = 0;
  int64_t numel = 0;
};

// a JSON number as a non-negative integer that a double holds exactly (offsets, dims): no (size_t)double casts of negative, NaN or huge values
bool json_index(const Json& j, uint64_t* out) {
  if (j.kind != Json::Num || !(j.num >= 0.0) || j.num > 9007199254740992.0 || j.num != floor(j.num)) return false;
  *out = (uint64_t)j.num;
  return true;
}
size_t dtype_size(const std::string& d) { return d == "F32" ? 4 : (d == "F16" || d == "BF16") ? 2 : d == "F64" ? 8 : 0; }

struct SafeFile {
  std::vector<unsigned char> bytes;
  size_t data0 = 0;
  std::map<std::string, RawTensor> t;
  std::string cfg_json;

  int open(const char* path) {
    FILE* f = fopen(path, "rb");
    if (!f) { sopro_set_error("sopro_checkpoint_open: cannot open %s", path); return -3; }
    fseek(f, 0, SEEK_END);
    const long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    if (n < 8) { fclose(f); sopro_set_error("sopro_checkpoint_open: %s is not a safetensors file", path); return -3; }
    bytes.resize((size_t)n);
    const size_t got = fread(bytes.data(), 1, (size_t)n, f);
    fclose(f);
    if (got != (size_t)n) { sopro_set_error("sopro_checkpoint_open: short read of %s", path); return -3; }
    uint64_t hl = 0;
    memcpy(&hl, bytes.data(), 8);
    if (hl > (uint64_t)n - 8) { sopro_set_error("sopro_checkpoint_open: %s: header length %llu exceeds the file", path, (unsigned long long)hl); return -3; }
    data0 = 8 + (size_t)hl;
    const size_t data_bytes = bytes.size() - data0;
    // the header as a NUL-terminated copy: strtod must not run into tensor data or past the file
    const std::string header((const char*)bytes.data() + 8, (size_t)hl);
    JsonParser jp{header.c_str(), header.c_str() + header.size()};
    Json root = jp.value();
    if (!jp.ok || root.kind != Json::Obj) { sopro_set_error("sopro_checkpoint_open: %s: malformed header", path); return -3; }
    for (const auto& kv : root.obj) {
      if (kv.first == "__metadata__") {
        if (const Json* c = kv.second.get("cfg")) if (c->kind == Json::Str) cfg_json = c->str;
        continue;
      }
      const Json *dt = kv.second.get("dtype"), *sh = kv.second.get("shape"), *off = kv.second.get("data_offsets");
      if (!dt || dt->kind != Json::Str || !sh || sh->kind != Json::Arr || !off || off->kind != Json::Arr || off->arr.size() != 2 || sh->arr.size() > 8) {
        sopro_set_error("sopro_checkpoint_open: %s: malformed entry %s", path, kv.first.c_str());
        return -3;
      }
      RawTensor r;
      r.dtype = dt->str;
      uint64_t numel = 1;
      for (const Json& d : sh->arr) {
        uint64_t v = 0;
        const uint64_t lim = (uint64_t)1 << 40;  // elements; far above any checkpoint, far below where a product could wrap
        if (!json_index(d, &v) || v > lim) {
          sopro_set_error("sopro_checkpoint_open: %s: %s has a dimension that is not a small non-negative integer", path, kv.first.c_str());
          return -3;
        }
        if (v != 0 && numel > lim / v) { sopro_set_error("sopro_checkpoint_open: %s: %s is too large", path, kv.first.c_str()); return -3; }
        numel *= v;
        r.shape.push_back((int64_t)v);
      }
      uint64_t b = 0, e = 0;
      if (!json_index(off->arr[0], &b) || !json_index(off->arr[1], &e) || b > e || e > (uint64_t)data_bytes) {
        sopro_set_error("sopro_checkpoint_open: %s: %s lies outside the file", path, kv.first.c_str());
        return -3;
      }
      const size_t es = dtype_size(r.dtype);  // other dtypes (integers, bool) may sit in the file; they are refused when something asks for them
      if (es != 0 && e - b != numel * es) {
        sopro_set_error("sopro_checkpoint_open: %s: %s holds %llu bytes for %llu %s elements", path, kv.first.c_str(), (unsigned long long)(e - b),
                        (unsigned long long)numel, r.dtype.c_str());
        return -3;
      }
      r.begin = (size_t)b; r.end = (size_t)e; r.numel = (int64_t)numel;
      t[kv.first] = std::move(r);
    }
    return 0;
  }
  bool has(const std::string& k) const { return t.count(k) != 0; }
};

struct T {  // a host tensor, fp32, row-major
  std::vector<int64_t> shape;
  std::vector<float> v;
  int64_t numel() const { int64_t n = 1; for (int64_t d : shape) n *= d; return n; }
  int64_t dim(int i) const { return shape.at((size_t)i); }  // (every tensor's rank is checked by rd() before a pack_* routine sees it)
};

float half_to_float(uint16_t h) {
  const uint32_t s = (uint32_t)(h >> 15) << 31, e = (h >> 10) & 31, m = h & 1023;
  uint32_t u;
  if (e == 0) {
    if (m == 0) u = s;
    else {  // subnormal
      int sh = 0;
      uint32_t mm = m;
      while (!(mm & 1024)) { mm <<= 1; ++sh; }
      u = s | ((uint32_t)(127 - 15 - sh + 1) << 23) | ((mm & 1023) << 13);
    }
  } else if (e == 31) u = s | 0x7f800000u | (m << 13);
  else u = s | ((e + 112) << 23) | (m << 13);
  float f;
  memcpy(&f, &u, 4);
  return f;
}

// want: the shape the config implies, one entry per dimension; -1 = any extent >= 1 (vocabulary sizes, kernel widths that the
// packed form carries through).  A tensor of another rank or extent is refused BEFORE any pack_* / fold loop indexes it.
typedef std::vector<int64_t> Want;
int read_f32(const SafeFile& f, const std::string& name, T* out, const Want& want) {
  auto it = f.t.find(name);
  if (it == f.t.end()) { sopro_set_error("checkpoint: tensor %s is missing", name.c_str()); return -3; }
  const RawTensor& r = it->second;
  bool shape_ok = r.shape.size() == want.size();
  for (size_t d = 0; shape_ok && d < want.size(); ++d) shape_ok = want[d] < 0 ? r.shape[d] >= 1 : r.shape[d] == want[d];
  if (!shape_ok) {
    std::string have = "[", exp = "[";
    for (int64_t d : r.shape) have += std::to_string(d) + ",";
    for (int64_t d : want) exp += (d < 0 ? std::string("*") : std::to_string(d)) + ",";
    sopro_set_error("checkpoint: tensor %s has shape %s] where the config implies %s]", name.c_str(), have.c_str(), exp.c_str());
    return -3;
  }
  const int64_t n = r.numel;
  const size_t es = dtype_size(r.dtype);
  if (es == 0 || r.end - r.begin != (size_t)n * es) {  // (the byte count of a known dtype was checked when the file was opened)
    sopro_set_error("checkpoint: tensor %s has dtype %s (F32, F16, BF16, F64 are read)", name.c_str(), r.dtype.c_str());
    return -3;
  }
  out->shape = r.shape;
  const unsigned char* src = f.bytes.data() + f.data0 + r.begin;
  out->v.resize((size_t)n);
  if (r.dtype == "F32") { if (n) memcpy(out->v.data(), src, (size_t)n * 4); }
  else if (r.dtype == "BF16") {
    for (int64_t i = 0; i < n; ++i) { uint16_t h; memcpy(&h, src + 2 * i, 2); const uint32_t u = (uint32_t)h << 16; memcpy(&out->v[(size_t)i], &u, 4); }
  } else if (r.dtype == "F16") {
    for (int64_t i = 0; i < n; ++i) { uint16_t h; memcpy(&h, src + 2 * i, 2); out->v[(size_t)i] = half_to_float(h); }
  } else {
    for (int64_t i = 0; i < n; ++i) { double d; memcpy(&d, src + 8 * i, 8); out->v[(size_t)i] = (float)d; }
  }
  return 0;
}

// ------------------------------------------------------------------------------------------------ configs (sopro_amd/config.py)
struct SoproCfg {  // reference schema: src/sopro/config.py:7-43 (defaults), hub.py:44-48 (key intersection)
  int num_codebooks = 32, codebook_size = 2048, d_model = 384, n_layers_text = 2, pos_emb_max = 4096, n_layers_ar = 6, ar_kernel = 13;
  std::vector<int> ar_dilation_cycle{1, 2, 4, 1};
  int ar_text_attn_freq = 2, n_layers_nar = 6, nar_head_dim = 256, nar_kernel_size = 11;
  std::vector<int> nar_dilation_cycle{1, 2, 4, 8};
  int stage[4][2] = {{2, 4}, {5, 8}, {9, 16}, {17, 32}};
  int sv_student_dim = 192, ref_enc_layers = 2, ref_xattn_heads = 2, ref_xattn_layers = 3;
  double ref_xattn_gmax = 0.35;
  std::vector<int> cycle(const std::vector<int>& c, int n) const {
    std::vector<int> cyc = c.empty() ? std::vector<int>{1} : c, out;
    while ((int)out.size() < n) out.insert(out.end(), cyc.begin(), cyc.end());
    out.resize((size_t)n);
    return out;
  }
  std::vector<int> stage_cbs(int s) const {  // 0-based codebook columns of stage s (src/sopro/model.py:39-42,85-94)
    std::vector<int> out;
    for (int i = stage[s][0] - 1; i < stage[s][1]; ++i)
      if (i >= 1 && i < num_codebooks) out.push_back(i);
    return out;
  }
};

struct MimiCfg {  // HF:configuration_mimi.py:86-133 defaults, num_quantizers from the Sopro checkpoint (src/sopro/codec/mimi.py:28-31)
  int num_quantizers = 32, num_semantic = 1, codebook_size = 2048, codebook_dim = 256, hidden = 512, num_filters = 64, kernel = 7, last_kernel = 3,
      res_kernel = 3, compress = 2, layers = 8, heads = 8, head_dim = 64, inter = 2048, window = 250, upsample_stride = 2;
  std::vector<int> ratios{8, 6, 5, 4};
  double norm_eps = 1e-5, rope_theta = 10000.0;
};

// Config numbers are converted only when they are integers of a sane size ((int)double of NaN / 1e300 is undefined); anything else
// sets *bad and the open fails.  Ranges are checked once all keys are in (check_cfg).
bool cfg_num(const Json& v, int* dst) {
  if (v.kind != Json::Num || !(v.num >= -1048576.0 && v.num <= 1073741824.0) || v.num != floor(v.num)) return false;
  *dst = (int)v.num;
  return true;
}
void cfg_int(const Json& o, const char* k, int* dst, bool* bad) { if (const Json* v = o.get(k)) if (!cfg_num(*v, dst)) *bad = true; }
void cfg_ints(const Json& o, const char* k, std::vector<int>* dst, bool* bad) {
  if (const Json* v = o.get(k)) {
    if (v->kind != Json::Arr || v->arr.size() > 64) { *bad = true; return; }
    dst->clear();
    for (const Json& e : v->arr) { int x = 0; if (!cfg_num(e, &x)) { *bad = true; return; } dst->push_back(x); }
  }
}
void cfg_pair(const Json& o, const char* k, int (&dst)[2], bool* bad) {
  if (const Json* v = o.get(k)) {
    if (v->kind != Json::Arr || v->arr.size() != 2 || !cfg_num(v->arr[0], &dst[0]) || !cfg_num(v->arr[1], &dst[1])) *bad = true;
  }
}
// what the engine (sopro_engine_cfg's array sizes, the kernels' shape rules) and the loops below can take
const char* check_cfg(const SoproCfg& c) {
  if (c.num_codebooks < 2 || c.num_codebooks > 64) return "num_codebooks outside 2..64";
  if (c.codebook_size < 2 || c.codebook_size > 65536) return "codebook_size outside 2..65536";
  if (c.d_model < 64 || c.d_model > 8192 || c.d_model % 32 != 0) return "d_model must be a multiple of 32 in 64..8192";
  if (c.n_layers_text < 0 || c.n_layers_text > 16 || c.ref_enc_layers < 0 || c.ref_enc_layers > 16 || c.ref_xattn_layers < 0 || c.ref_xattn_layers > 16)
    return "n_layers_text / ref_enc_layers / ref_xattn_layers outside 0..16";
  if (c.n_layers_ar < 1 || c.n_layers_ar > 16 || c.n_layers_nar < 1 || c.n_layers_nar > 16) return "n_layers_ar / n_layers_nar outside 1..16";
  if (c.pos_emb_max < 1 || c.pos_emb_max > (1 << 20)) return "pos_emb_max outside 1..2^20";
  if (c.ar_kernel < 1 || c.ar_kernel > 64 || c.nar_kernel_size < 1 || c.nar_kernel_size > 63) return "ar_kernel / nar_kernel_size outside 1..64";
  if (c.ar_text_attn_freq < 1) return "ar_text_attn_freq < 1";
  if (c.nar_head_dim < 8 || c.nar_head_dim > 4096) return "nar_head_dim outside 8..4096";
  if (c.sv_student_dim < 1 || c.sv_student_dim > 4096) return "sv_student_dim outside 1..4096";
  if (c.ref_xattn_heads < 1 || c.ref_xattn_heads > 64 || c.d_model % c.ref_xattn_heads != 0) return "ref_xattn_heads must divide d_model";
  if (!(c.ref_xattn_gmax == c.ref_xattn_gmax) || c.ref_xattn_gmax < -1e6 || c.ref_xattn_gmax > 1e6) return "ref_xattn_gmax is not a finite number";
  for (int x : c.ar_dilation_cycle) if (x < 1 || x > 4096) return "ar_dilation_cycle entry outside 1..4096";
  for (int x : c.nar_dilation_cycle) if (x < 1 || x > 4096) return "nar_dilation_cycle entry outside 1..4096";
  for (int s = 0; s < 4; ++s)
    // (last < first is an EMPTY, disabled stage in the reference - _stage_range_to_indices filters to 1 <= i < Q and stage_order skips
    // stages without codebooks, src/sopro/model.py:39-42,92-94 -: accepted; stage_cbs applies the same filter)
    if (c.stage[s][0] < -4096 || c.stage[s][0] > 4096 || c.stage[s][1] < -4096 || c.stage[s][1] > 4096) return "a stage pair is outside -4096..4096";
  return nullptr;
}

// ------------------------------------------------------------------------------------------------ pack.py, restated
typedef std::map<std::string, T> Pack;

T view(const T& a, std::vector<int64_t> shape) { T o = a; o.shape = std::move(shape); return o; }

T pack_glu_w(const T& w) {  // [2D, K] (value rows then gate rows) -> per-64 blocks [32 value | 32 gate]   (pack.py pack_glu)
  const int64_t d = w.dim(0) / 2, K = w.dim(1);
  T o; o.shape = w.shape; o.v.resize(w.v.size());
  for (int64_t blk = 0; blk < d / 32; ++blk)
    for (int64_t r = 0; r < 32; ++r) {
      memcpy(&o.v[(size_t)((blk * 64 + r) * K)], &w.v[(size_t)((blk * 32 + r) * K)], (size_t)K * 4);
      memcpy(&o.v[(size_t)((blk * 64 + 32 + r) * K)], &w.v[(size_t)((d + blk * 32 + r) * K)], (size_t)K * 4);
    }
  return o;
}
T pack_glu_b(const T& b) {
  const int64_t d = b.dim(0) / 2;
  T o; o.shape = b.shape; o.v.resize(b.v.size());
  for (int64_t blk = 0; blk < d / 32; ++blk)
    for (int64_t r = 0; r < 32; ++r) { o.v[(size_t)(blk * 64 + r)] = b.v[(size_t)(blk * 32 + r)]; o.v[(size_t)(blk * 64 + 32 + r)] = b.v[(size_t)(d + blk * 32 + r)]; }
  return o;
}
T pack_dw(const T& w) {  // [C, 1, k] -> [k, C]
  const int64_t C = w.dim(0), k = w.dim(2);
  T o; o.shape = {k, C}; o.v.resize((size_t)(k * C));
  for (int64_t c = 0; c < C; ++c)
    for (int64_t j = 0; j < k; ++j) o.v[(size_t)(j * C + c)] = w.v[(size_t)(c * k + j)];
  return o;
}
T pack_conv1d(const T& w) {  // [Cout, Cin, k] -> [Cout, k*Cin], K index = tap*Cin + ci
  const int64_t co = w.dim(0), ci = w.dim(1), k = w.dim(2);
  T o; o.shape = {co, k * ci}; o.v.resize((size_t)(co * k * ci));
  for (int64_t a = 0; a < co; ++a)
    for (int64_t c = 0; c < ci; ++c)
      for (int64_t j = 0; j < k; ++j) o.v[(size_t)(a * k * ci + j * ci + c)] = w.v[(size_t)((a * ci + c) * k + j)];
  return o;
}
void pack_convtr1d(const T& w, const T& b, int s, T* wo, T* bo) {  // [Cin, Cout, 2s] -> ([s*Cout, 2*Cin], [s*Cout]); row r*Cout+co, col half*Cin+ci = w[ci, co, (1-half)*s + r]
  const int64_t ci = w.dim(0), co = w.dim(1);
  wo->shape = {s * co, 2 * ci}; wo->v.resize((size_t)(s * co * 2 * ci));
  for (int64_t r = 0; r < s; ++r)
    for (int64_t o = 0; o < co; ++o)
      for (int half = 0; half < 2; ++half)
        for (int64_t c = 0; c < ci; ++c) wo->v[(size_t)((r * co + o) * 2 * ci + half * ci + c)] = w.v[(size_t)((c * co + o) * 2 * s + (1 - half) * s + r)];
  bo->shape = {s * co}; bo->v.resize((size_t)(s * co));
  for (int64_t r = 0; r < s; ++r)
    for (int64_t o = 0; o < co; ++o) bo->v[(size_t)(r * co + o)] = b.v[(size_t)o];
}
T scale_cols(const T& w, const T& v) {  // w[n, k] * v[k]   (fp32 products, as torch's broadcasting multiply)
  const int64_t N = w.dim(0), K = w.dim(1);
  T o = w;
  for (int64_t n = 0; n < N; ++n)
    for (int64_t k = 0; k < K; ++k) o.v[(size_t)(n * K + k)] = w.v[(size_t)(n * K + k)] * v.v[(size_t)k];
  return o;
}
T softmax1(const T& x) {  // torch.softmax(x.float(), dim=0)
  T o = x;
  double mx = -1e300, sum = 0;
  for (float f : x.v) mx = f > mx ? f : mx;
  std::vector<double> e(x.v.size());
  for (size_t i = 0; i < x.v.size(); ++i) { e[i] = exp((double)x.v[i] - mx); sum += e[i]; }
  for (size_t i = 0; i < x.v.size(); ++i) o.v[i] = (float)(e[i] / sum);
  return o;
}
T cat0(const std::vector<const T*>& parts) {
  T o;
  o.shape = parts[0]->shape;
  int64_t rows = 0;
  for (const T* p : parts) { rows += p->dim(0); o.v.insert(o.v.end(), p->v.begin(), p->v.end()); }
  o.shape[0] = rows;
  return o;
}

struct Ck {
  SafeFile fs, fm;
  bool have_mimi = false;
  SoproCfg c;
  MimiCfg m;
  Pack p;                         // packed name -> tensor (what sopro_engine_set_tensor takes)
  std::vector<std::string> names; // stable order for enumeration
  float gate[16] = {0};           // tanh(gate) of the AR cross-attention blocks
  float mix[8][2] = {{0}};
  float final_bias = 0.f;

  int rd(const SafeFile& f, const std::string& k, T* t, const Want& want) { return read_f32(f, k, t, want); }
  void put(const std::string& k, T t) {
    if (!p.count(k)) names.push_back(k);
    p[k] = std::move(t);
  }

#define CKR(call) do { const int rc_ = (call); if (rc_ != 0) return rc_; } while (0)

  int ssm_block(const std::string& pre, bool packed_glu, int kernel) {  // pack.py _ssm_block; shapes: src/sopro/nn/blocks.py:113-134
    T nw, gw, gb, dw, db, fn, w1, b1, w2, b2;
    const int64_t d = c.d_model;
    CKR(rd(fs, pre + ".norm.weight", &nw, {d})); CKR(rd(fs, pre + ".glu.pro.weight", &gw, {2 * d, d})); CKR(rd(fs, pre + ".glu.pro.bias", &gb, {2 * d}));
    CKR(rd(fs, pre + ".dw.dw.weight", &dw, {d, 1, kernel})); CKR(rd(fs, pre + ".dw.dw.bias", &db, {d})); CKR(rd(fs, pre + ".ff.0.weight", &fn, {d}));
    CKR(rd(fs, pre + ".ff.1.weight", &w1, {4 * d, d})); CKR(rd(fs, pre + ".ff.1.bias", &b1, {4 * d}));
    CKR(rd(fs, pre + ".ff.3.weight", &w2, {d, 4 * d})); CKR(rd(fs, pre + ".ff.3.bias", &b2, {d}));
    put(pre + ".norm.weight", nw);
    if (packed_glu) { put(pre + ".glu.w", pack_glu_w(gw)); put(pre + ".glu.b", pack_glu_b(gb)); }
    else { put(pre + ".glu.w", gw); put(pre + ".glu.b", gb); }
    put(pre + ".dw.w", pack_dw(dw)); put(pre + ".dw.b", db); put(pre + ".ff.norm.weight", fn);
    put(pre + ".ff1.w", w1); put(pre + ".ff1.b", b1); put(pre + ".ff2.w", w2); put(pre + ".ff2.b", b2);
    return 0;
  }

  int xattn(const std::string& pre, double gate_mul, int heads, float* gate_out) {  // pack.py _xattn
    T g, nq, nkv, q, k, v, o;
    const int64_t d = c.d_model;
    CKR(rd(fs, pre + ".gate", &g, {})); CKR(rd(fs, pre + ".nq.weight", &nq, {d})); CKR(rd(fs, pre + ".nkv.weight", &nkv, {d}));
    CKR(rd(fs, pre + ".q_proj.weight", &q, {d, d})); CKR(rd(fs, pre + ".k_proj.weight", &k, {d, d})); CKR(rd(fs, pre + ".v_proj.weight", &v, {d, d}));
    CKR(rd(fs, pre + ".out_proj.weight", &o, {d, d}));
    put(pre + ".nq.weight", nq); put(pre + ".nkv.weight", nkv); put(pre + ".q.w", q);
    put(pre + ".kv.w", cat0({&k, &v})); put(pre + ".o.w", o);
    if (heads) {  // [H, d, dh]: q.wT[h][c][j] = q[h*dh + j][c]  (x RMSNorm_nq's weight, folded by the caller)
      const int64_t dh = d / heads;
      T wt; wt.shape = {heads, d, dh}; wt.v.resize((size_t)(d * d));
      for (int64_t h = 0; h < heads; ++h)
        for (int64_t c = 0; c < d; ++c)
          for (int64_t j = 0; j < dh; ++j) wt.v[(size_t)((h * d + c) * dh + j)] = q.v[(size_t)((h * dh + j) * d + c)];
      put(pre + ".q.wT", wt);
    }
    const float gs = (float)gate_mul * tanhf(g.v[0]);
    T gsv; gsv.shape = {d}; gsv.v.assign((size_t)d, gs);
    put(pre + ".gate_scale", gsv);
    if (gate_out) *gate_out = gs;
    return 0;
  }

  int pack_sopro() {
    const int64_t d = c.d_model, Q = c.num_codebooks, V = c.codebook_size, HD = c.nar_head_dim, SV = c.sv_student_dim;
    // text encoder (src/sopro/nn/text.py:16-27: kernel 7)
    for (int i = 0; i < c.n_layers_text; ++i) CKR(ssm_block("text_enc.layers." + std::to_string(i), true, 7));
    T t;
    CKR(rd(fs, "text_enc.embed.emb.weight", &t, {-1, d})); put("text_enc.embed", t);
    CKR(rd(fs, "text_enc.norm.weight", &t, {d})); put("text_enc.norm.weight", t);
    CKR(rd(fs, "cb_embed.emb.weight", &t, {Q * V + 1, d})); put("cb_embed", t);
    CKR(rd(fs, "nar_prev_cb_weights", &t, {Q})); put("nar_prev_cb_weights", t);
    // Token2SV (src/sopro/nn/speaker.py:12-61): width sd and kernel from the file, every other tensor tied to them
    CKR(rd(fs, "token2sv.emb.weight", &t, {Q * V, -1})); put("token2sv.emb", t);
    const int64_t sd = t.dim(1);
    CKR(rd(fs, "token2sv.cb_weights", &t, {Q})); put("token2sv.cw", softmax1(t));
    for (int i : {0, 3}) {
      T w, b;
      CKR(rd(fs, "token2sv.enc." + std::to_string(i) + ".dw.weight", &w, {sd, 1, -1})); CKR(rd(fs, "token2sv.enc." + std::to_string(i) + ".dw.bias", &b, {sd}));
      put("token2sv.enc." + std::to_string(i) + ".w", pack_dw(w)); put("token2sv.enc." + std::to_string(i) + ".b", b);
    }
    {
      T w, b;
      CKR(rd(fs, "token2sv.pool.attn.0.weight", &w, {sd, sd})); CKR(rd(fs, "token2sv.pool.attn.0.bias", &b, {sd}));
      put("token2sv.pool.attn.0.w", w); put("token2sv.pool.attn.0.b", b);
      CKR(rd(fs, "token2sv.pool.attn.2.weight", &w, {1, sd})); CKR(rd(fs, "token2sv.pool.attn.2.bias", &b, {1}));
      put("token2sv.pool.attn.2.w", w); put("token2sv.pool.attn.2.b", b);
      CKR(rd(fs, "token2sv.proj.weight", &w, {SV, 2 * sd})); CKR(rd(fs, "token2sv.proj.bias", &b, {SV}));
      put("token2sv.proj.w", w); put("token2sv.proj.b", b);
      CKR(rd(fs, "spk_film.mlp.0.weight", &w, {d, SV})); CKR(rd(fs, "spk_film.mlp.0.bias", &b, {d}));
      put("spk_film.mlp.0.w", w); put("spk_film.mlp.0.b", b);
      CKR(rd(fs, "spk_film.mlp.2.weight", &w, {2 * d, d})); CKR(rd(fs, "spk_film.mlp.2.bias", &b, {2 * d}));
      put("spk_film.mlp.2.w", w); put("spk_film.mlp.2.b", b);
    }
    CKR(rd(fs, "spk_film.norm.weight", &t, {d})); put("spk_film.norm.weight", t);
    CKR(rd(fs, "spk_film.norm.bias", &t, {d})); put("spk_film.norm.bias", t);
    // AR generator: natural GLU layout, RMSNorm weights folded into the projections they feed (blocks.py:26-37)
    for (int i = 0; i < c.n_layers_ar; ++i) {
      const std::string pre = "ar.blocks." + std::to_string(i);
      CKR(ssm_block(pre, false, c.ar_kernel));
      p[pre + ".glu.w"] = scale_cols(p[pre + ".glu.w"], p[pre + ".norm.weight"]);
      p[pre + ".ff1.w"] = scale_cols(p[pre + ".ff1.w"], p[pre + ".ff.norm.weight"]);
    }
    for (int i = 0; i < c.n_layers_ar; ++i) {
      if ((i + 1) % c.ar_text_attn_freq != 0) continue;
      const std::string pre = "ar.x_attns." + std::to_string(i);
      CKR(xattn(pre, 1.0, 4, &gate[i]));  // 4 heads: src/sopro/nn/generator.py:36
      T& wt = p[pre + ".q.wT"];
      const T& nq = p[pre + ".nq.weight"];
      const int64_t dh = d / 4;
      for (int64_t h = 0; h < 4; ++h)
        for (int64_t cc = 0; cc < d; ++cc)
          for (int64_t j = 0; j < dh; ++j) wt.v[(size_t)((h * d + cc) * dh + j)] *= nq.v[(size_t)cc];
      // unfolded keys (sopro_ar_frame.k_unfold): qa.w = Wq' = q_proj * w_nq, qu.w = Wq' W2, q.b = Wq' b2, products in float64
      const T& q = p[pre + ".q.w"];
      const T& w2 = p["ar.blocks." + std::to_string(i) + ".ff2.w"];
      const T& b2 = p["ar.blocks." + std::to_string(i) + ".ff2.b"];
      std::vector<double> wq((size_t)(d * d));
      for (int64_t n = 0; n < d; ++n)
        for (int64_t k = 0; k < d; ++k) wq[(size_t)(n * d + k)] = (double)q.v[(size_t)(n * d + k)] * (double)nq.v[(size_t)k];
      T qa; qa.shape = {d, d}; qa.v.resize((size_t)(d * d));
      for (size_t e = 0; e < wq.size(); ++e) qa.v[e] = (float)wq[e];
      const int64_t F4 = w2.dim(1);
      T qu; qu.shape = {d, F4}; qu.v.resize((size_t)(d * F4));
      T qb; qb.shape = {d}; qb.v.resize((size_t)d);
      std::vector<double> acc((size_t)F4);
      for (int64_t n = 0; n < d; ++n) {
        std::fill(acc.begin(), acc.end(), 0.0);
        double bb = 0;
        for (int64_t k = 0; k < d; ++k) {
          const double a = wq[(size_t)(n * d + k)];
          const float* row = &w2.v[(size_t)(k * F4)];
          for (int64_t f = 0; f < F4; ++f) acc[(size_t)f] += a * (double)row[f];
          bb += a * (double)b2.v[(size_t)k];
        }
        for (int64_t f = 0; f < F4; ++f) qu.v[(size_t)(n * F4 + f)] = (float)acc[(size_t)f];
        qb.v[(size_t)n] = (float)bb;
      }
      put(pre + ".qa.w", qa); put(pre + ".qu.w", qu); put(pre + ".q.b", qb);
    }
    T an, hw, hb;
    CKR(rd(fs, "ar.norm.weight", &an, {d})); CKR(rd(fs, "ar.head.weight", &hw, {V + 1, d})); CKR(rd(fs, "ar.head.bias", &hb, {V + 1}));
    put("ar.norm.weight", an); put("ar.head.w", scale_cols(hw, an)); put("ar.head.b", hb);
    // NAR refiner
    for (int i = 0; i < c.n_layers_nar; ++i) CKR(ssm_block("nar.blocks." + std::to_string(i), true, c.nar_kernel_size));
    CKR(rd(fs, "nar.norm.weight", &t, {d})); put("nar.norm.weight", t);
    CKR(rd(fs, "nar.pre.weight", &t, {HD, d})); put("nar.pre.w", t);
    CKR(rd(fs, "nar.pre.bias", &t, {HD})); put("nar.pre.b", t);
    CKR(rd(fs, "nar.stage_emb.weight", &t, {-1, d})); put("nar.stage_emb", t);
    CKR(rd(fs, "nar.adapter.norm.weight", &t, {d})); put("nar.adapter.norm.weight", t);
    {
      T w, b;
      CKR(rd(fs, "nar.adapter.mlp.0.weight", &w, {-1, d}));
      const int64_t ah = w.dim(0);
      CKR(rd(fs, "nar.adapter.mlp.0.bias", &b, {ah}));
      put("nar.adapter.mlp.0.w", w); put("nar.adapter.mlp.0.b", b);
      CKR(rd(fs, "nar.adapter.mlp.2.weight", &w, {2 * d, ah})); CKR(rd(fs, "nar.adapter.mlp.2.bias", &b, {2 * d}));
      put("nar.adapter.mlp.2.w", w); put("nar.adapter.mlp.2.b", b);
    }
    const char* stage_names[4] = {"B", "C", "D", "E"};
    const char* pos_names = "BCDEFGHI";
    int pos = 0;
    for (int s = 0; s < 4; ++s) {
      const std::vector<int> cbs = c.stage_cbs(s);
      if (cbs.empty()) continue;
      const std::string sn = stage_names[s];
      // logits_j = (z + e_j) W_j^T + b_j = z W_j^T + (b_j + W_j e_j)   (src/sopro/nn/nar.py:100-116), folded in float64
      T hid;
      CKR(rd(fs, "nar.head_id_emb." + sn + ".weight", &hid, {(int64_t)cbs.size(), HD}));
      std::vector<T> ws(cbs.size()), bs(cbs.size());
      std::vector<const T*> wp, bp;
      for (size_t j = 0; j < cbs.size(); ++j) {
        CKR(rd(fs, "nar.heads." + sn + "." + std::to_string(j) + ".weight", &ws[j], {V, HD}));
        CKR(rd(fs, "nar.heads." + sn + "." + std::to_string(j) + ".bias", &bs[j], {V}));
        for (int64_t r = 0; r < V; ++r) {
          double a = (double)bs[j].v[(size_t)r];
          double dot = 0;
          for (int64_t k = 0; k < HD; ++k) dot += (double)ws[j].v[(size_t)(r * HD + k)] * (double)hid.v[(size_t)(j * HD + k)];
          bs[j].v[(size_t)r] = (float)(a + dot);
        }
        wp.push_back(&ws[j]); bp.push_back(&bs[j]);
      }
      T W = cat0(wp), Bv = cat0(bp);
      put("nar.heads." + sn + ".w", W); put("nar.heads." + sn + ".b", Bv);
      put(std::string("nar.heads.") + pos_names[pos] + ".w", W); put(std::string("nar.heads.") + pos_names[pos] + ".b", Bv);  // the engine names the stages by position
      T mx;
      CKR(rd(fs, "nar.mix." + sn, &mx, {2}));
      T sm = softmax1(mx);
      mix[pos][0] = sm.v[0]; mix[pos][1] = sm.v[1];
      put("nar.mix." + sn, sm);
      ++pos;
    }
    CKR(rd(fs, "cond_norm.weight", &t, {d})); put("cond_norm.weight", t);
    for (int i = 0; i < c.ref_enc_layers; ++i) CKR(ssm_block("ref_enc_blocks." + std::to_string(i), true, 7));  // src/sopro/model.py:100-104
    CKR(rd(fs, "ref_enc_norm.weight", &t, {d})); put("ref_enc_norm.weight", t);
    CKR(rd(fs, "ref_cb_weights", &t, {Q})); put("ref_cw", softmax1(t));
    for (int i = 0; i < c.ref_xattn_layers; ++i) CKR(xattn("ref_xattn.blocks." + std::to_string(i), c.ref_xattn_gmax, 0, nullptr));
    // position table of the conditioning (src/sopro/nn/embeddings.py:11-25; src/sopro/model.py:62-64: pos_emb_max + 8 rows)
    {
      const int n = c.pos_emb_max + 8;
      T pe; pe.shape = {n, d}; pe.v.assign((size_t)n * d, 0.f);
      const float k = (float)(-log(10000.0) / d);
      for (int j = 0; j < d; j += 2) {
        const float div = expf((float)j * k);
        for (int ps = 0; ps < n; ++ps) {
          const float a = (float)ps * div;
          pe.v[(size_t)ps * d + j] = (float)sin((double)a);
          if (j + 1 < d) pe.v[(size_t)ps * d + j + 1] = (float)cos((double)a);
        }
      }
      put("pe", pe);
    }
    return 0;
  }

  int transformer(const std::string& pre, const std::string& name) {  // pack.py _pack_transformer; shapes: HF:modeling_mimi.py:568-700
    const int64_t hs = m.hidden, inter = m.inter;
    for (int li = 0; li < m.layers; ++li) {
      const std::string q = name + ".layers." + std::to_string(li), o = pre + "." + std::to_string(li);
      T a, b, cc, t;
      CKR(rd(fm, q + ".self_attn.q_proj.weight", &a, {hs, hs})); CKR(rd(fm, q + ".self_attn.k_proj.weight", &b, {hs, hs}));
      CKR(rd(fm, q + ".self_attn.v_proj.weight", &cc, {hs, hs}));
      put(o + ".qkv.w", cat0({&a, &b, &cc}));
      CKR(rd(fm, q + ".self_attn.o_proj.weight", &t, {hs, hs})); put(o + ".o.w", t);
      CKR(rd(fm, q + ".mlp.fc1.weight", &t, {inter, hs})); put(o + ".fc1.w", t);
      CKR(rd(fm, q + ".mlp.fc2.weight", &t, {hs, inter})); put(o + ".fc2.w", t);
      CKR(rd(fm, q + ".input_layernorm.weight", &t, {hs})); put(o + ".ln1.w", t);
      CKR(rd(fm, q + ".input_layernorm.bias", &t, {hs})); put(o + ".ln1.b", t);
      CKR(rd(fm, q + ".post_attention_layernorm.weight", &t, {hs})); put(o + ".ln2.w", t);
      CKR(rd(fm, q + ".post_attention_layernorm.bias", &t, {hs})); put(o + ".ln2.b", t);
      CKR(rd(fm, q + ".self_attn_layer_scale.scale", &t, {hs})); put(o + ".ls1", t);
      CKR(rd(fm, q + ".mlp_layer_scale.scale", &t, {hs})); put(o + ".ls2", t);
    }
    return 0;
  }

  int pack_mimi(int rope_positions) {
    const int ns = m.num_semantic;
    const int64_t CS = m.codebook_size, CD = m.codebook_dim, H = m.hidden;
    T cb; cb.shape = {(int64_t)m.num_quantizers * CS, CD};
    cb.v.reserve((size_t)(m.num_quantizers * CS * CD));
    for (int q = 0; q < m.num_quantizers; ++q) {
      const std::string grp = q < ns ? "semantic" : "acoustic";
      const std::string pp = "quantizer." + grp + "_residual_vector_quantizer.layers." + std::to_string(q < ns ? q : q - ns) + ".codebook";
      T es, cu;
      CKR(rd(fm, pp + ".embed_sum", &es, {CS, CD})); CKR(rd(fm, pp + ".cluster_usage", &cu, {CS}));
      for (int64_t r = 0; r < CS; ++r) {  // HF:modeling_mimi.py:979-983
        const float den = cu.v[(size_t)r] < 1e-5f ? 1e-5f : cu.v[(size_t)r];
        for (int64_t k = 0; k < CD; ++k) cb.v.push_back(es.v[(size_t)(r * CD + k)] / den);
      }
    }
    put("codebooks", cb);
    T psem, pac;
    CKR(rd(fm, "quantizer.semantic_residual_vector_quantizer.output_proj.weight", &psem, {H, CD, 1}));
    CKR(rd(fm, "quantizer.acoustic_residual_vector_quantizer.output_proj.weight", &pac, {H, CD, 1}));
    {
      T pj; pj.shape = {H, 2 * CD}; pj.v.resize((size_t)(H * 2 * CD));
      for (int64_t h = 0; h < H; ++h) {
        memcpy(&pj.v[(size_t)(h * 2 * CD)], &psem.v[(size_t)(h * CD)], (size_t)CD * 4);
        memcpy(&pj.v[(size_t)(h * 2 * CD + CD)], &pac.v[(size_t)(h * CD)], (size_t)CD * 4);
      }
      put("rvq_proj.w", pj);
    }
    T t;
    CKR(rd(fm, "upsample.conv.weight", &t, {H, 1, 2 * m.upsample_stride})); put("upsample.w", view(t, {t.dim(0), t.dim(2)}));
    if (fm.has("decoder_transformer.layers.0.mlp.fc1.weight")) CKR(transformer("tr", "decoder_transformer"));
    if (fm.has("encoder_transformer.layers.0.mlp.fc1.weight")) CKR(transformer("etr", "encoder_transformer"));
    if (fm.has("encoder.layers.0.conv.weight")) CKR(mimi_encoder());
    // SEANet decoder (HF:modeling_mimi.py:931-961): channels halve at every ratio
    T w, b;
    int64_t ch = (int64_t)m.num_filters << m.ratios.size();
    CKR(rd(fm, "decoder.layers.0.conv.weight", &w, {ch, H, m.kernel})); CKR(rd(fm, "decoder.layers.0.conv.bias", &b, {ch}));
    put("sea.conv0.w", pack_conv1d(w)); put("sea.conv0.b", b);
    int li = 1;
    for (size_t si = 0; si < m.ratios.size(); ++si) {
      li += 1;
      T wo, bo;
      const int64_t co = ch / 2, hd = co / m.compress;
      CKR(rd(fm, "decoder.layers." + std::to_string(li) + ".conv.weight", &w, {ch, co, 2 * m.ratios[si]}));
      CKR(rd(fm, "decoder.layers." + std::to_string(li) + ".conv.bias", &b, {co}));
      pack_convtr1d(w, b, m.ratios[si], &wo, &bo);
      put("sea.up" + std::to_string(si) + ".w", wo); put("sea.up" + std::to_string(si) + ".b", bo);
      li += 1;
      const std::string blk = "decoder.layers." + std::to_string(li) + ".block";
      CKR(rd(fm, blk + ".1.conv.weight", &w, {hd, co, m.res_kernel})); put("sea.res" + std::to_string(si) + ".c1.w", pack_conv1d(w));
      CKR(rd(fm, blk + ".1.conv.bias", &b, {hd})); put("sea.res" + std::to_string(si) + ".c1.b", b);
      CKR(rd(fm, blk + ".3.conv.weight", &w, {co, hd, 1})); put("sea.res" + std::to_string(si) + ".c2.w", pack_conv1d(w));
      CKR(rd(fm, blk + ".3.conv.bias", &b, {co})); put("sea.res" + std::to_string(si) + ".c2.b", b);
      li += 1;
      ch = co;
    }
    li += 1;
    CKR(rd(fm, "decoder.layers." + std::to_string(li) + ".conv.weight", &w, {1, m.num_filters, m.last_kernel}));  // [1, 64, 3]
    {
      const int64_t C = w.dim(1), k = w.dim(2);
      T fw; fw.shape = {k, C}; fw.v.resize((size_t)(k * C));
      for (int64_t c2 = 0; c2 < C; ++c2)
        for (int64_t j = 0; j < k; ++j) fw.v[(size_t)(j * C + c2)] = w.v[(size_t)(c2 * k + j)];
      put("sea.final.w", fw);
    }
    CKR(rd(fm, "decoder.layers." + std::to_string(li) + ".conv.bias", &b, {1}));
    final_bias = b.v[0];
    put("sea.final.b", view(b, {1}));
    // RoPE tables (HF:modeling_mimi.py:511-566; sopro_amd/pack.py rope_tables): cos / sin [npos, dh / 2]
    {
      const int dh = m.head_dim, half = dh / 2;
      T cs, sn;
      cs.shape = sn.shape = {rope_positions, half};
      cs.v.resize((size_t)rope_positions * half); sn.v.resize((size_t)rope_positions * half);
      for (int j = 0; j < half; ++j) {
        const float inv = 1.0f / powf((float)m.rope_theta, (float)(2 * j) / (float)dh);
        for (int ps = 0; ps < rope_positions; ++ps) {
          const float a = (float)ps * inv;
          cs.v[(size_t)ps * half + j] = (float)cos((double)a);
          sn.v[(size_t)ps * half + j] = (float)sin((double)a);
        }
      }
      put("rope.cos", cs); put("rope.sin", sn);
    }
    return 0;
  }

  int mimi_encoder() {  // pack.py _pack_mimi_encoder; shapes: HF:modeling_mimi.py MimiEncoder (channels double at every ratio, reversed)
    T w, b;
    const int64_t H = m.hidden, CD = m.codebook_dim;
    int64_t ch = m.num_filters;
    CKR(rd(fm, "encoder.layers.0.conv.weight", &w, {ch, 1, m.kernel})); put("enc.conv0.w", view(w, {w.dim(0), w.dim(2)}));
    CKR(rd(fm, "encoder.layers.0.conv.bias", &b, {ch})); put("enc.conv0.b", b);
    int li = 1;
    for (size_t si = 0; si < m.ratios.size(); ++si) {
      const std::string blk = "encoder.layers." + std::to_string(li) + ".block", s = std::to_string(si);
      const int64_t hd = ch / m.compress, r = m.ratios[m.ratios.size() - 1 - si];
      CKR(rd(fm, blk + ".1.conv.weight", &w, {hd, ch, m.res_kernel})); put("enc.res" + s + ".c1.w", pack_conv1d(w));
      CKR(rd(fm, blk + ".1.conv.bias", &b, {hd})); put("enc.res" + s + ".c1.b", b);
      CKR(rd(fm, blk + ".3.conv.weight", &w, {ch, hd, 1})); put("enc.res" + s + ".c2.w", pack_conv1d(w));
      CKR(rd(fm, blk + ".3.conv.bias", &b, {ch})); put("enc.res" + s + ".c2.b", b);
      li += 2;
      CKR(rd(fm, "encoder.layers." + std::to_string(li) + ".conv.weight", &w, {2 * ch, ch, 2 * r})); put("enc.down" + s + ".w", pack_conv1d(w));
      CKR(rd(fm, "encoder.layers." + std::to_string(li) + ".conv.bias", &b, {2 * ch})); put("enc.down" + s + ".b", b);
      li += 1;
      ch *= 2;
    }
    li += 1;
    CKR(rd(fm, "encoder.layers." + std::to_string(li) + ".conv.weight", &w, {H, ch, m.last_kernel})); put("enc.final.w", pack_conv1d(w));
    CKR(rd(fm, "encoder.layers." + std::to_string(li) + ".conv.bias", &b, {H})); put("enc.final.b", b);
    CKR(rd(fm, "downsample.conv.weight", &w, {H, H, 2 * m.upsample_stride})); put("enc.ds.w", pack_conv1d(w));
    CKR(rd(fm, "quantizer.semantic_residual_vector_quantizer.input_proj.weight", &w, {CD, H, 1})); put("enc.inproj.sem.w", view(w, {w.dim(0), w.dim(1)}));
    CKR(rd(fm, "quantizer.acoustic_residual_vector_quantizer.input_proj.weight", &w, {CD, H, 1})); put("enc.inproj.ac.w", view(w, {w.dim(0), w.dim(1)}));
    const T& cb = p["codebooks"];  // nearest code = argmax_e (r.e - |e|^2 / 2)
    T bias; bias.shape = {cb.dim(0)}; bias.v.resize((size_t)cb.dim(0));
    for (int64_t r = 0; r < cb.dim(0); ++r) {
      double s2 = 0;
      for (int64_t k = 0; k < cb.dim(1); ++k) { const double e = cb.v[(size_t)(r * cb.dim(1) + k)]; s2 += e * e; }
      bias.v[(size_t)r] = (float)(-0.5 * s2);
    }
    put("enc.cb_bias", bias);
    return 0;
  }
};

}  // namespace

struct sopro_checkpoint {
  Ck k;
  int rope_positions = 8192;
};

extern "C" {

// Exceptions (std::bad_alloc from a hostile size, std::out_of_range from T::dim / map::at) must not cross the C boundary.
#define SOPRO_C_GUARD_BEGIN try {
#define SOPRO_C_GUARD_END(fn_)                                                                   \
  } catch (const std::exception& ex) {                                                           \
    sopro_set_error("%s: %s", fn_, ex.what());                                                   \
    return -3;                                                                                   \
  } catch (...) {                                                                                \
    sopro_set_error("%s: unknown exception", fn_);                                              \
    return -3;                                                                                   \
  }

int sopro_checkpoint_open(const char* sopro_path, const char* mimi_path, sopro_checkpoint** out) {
  SOPRO_CHECK_ARG(sopro_path && out, "sopro_path / out is NULL");
  SOPRO_C_GUARD_BEGIN
  std::unique_ptr<sopro_checkpoint> ck(new sopro_checkpoint());
  Ck& k = ck->k;
  if (int rc = k.fs.open(sopro_path)) return rc;
  if (k.fs.cfg_json.empty()) {  // src/sopro/hub.py:37-39
    sopro_set_error("sopro_checkpoint_open: no 'cfg' metadata found in %s", sopro_path);
    return -3;
  }
  {
    JsonParser jp{k.fs.cfg_json.c_str(), k.fs.cfg_json.c_str() + k.fs.cfg_json.size()};
    Json o = jp.value();
    if (!jp.ok || o.kind != Json::Obj) { sopro_set_error("sopro_checkpoint_open: malformed cfg JSON in %s", sopro_path); return -3; }
    SoproCfg& c = k.c;  // key-intersection load: unknown keys are ignored, missing ones keep their defaults (hub.py:44-48)
    bool bad = false;
    cfg_int(o, "num_codebooks", &c.num_codebooks, &bad); cfg_int(o, "codebook_size", &c.codebook_size, &bad); cfg_int(o, "d_model", &c.d_model, &bad);
    cfg_int(o, "n_layers_text", &c.n_layers_text, &bad); cfg_int(o, "pos_emb_max", &c.pos_emb_max, &bad); cfg_int(o, "n_layers_ar", &c.n_layers_ar, &bad);
    cfg_int(o, "ar_kernel", &c.ar_kernel, &bad); cfg_ints(o, "ar_dilation_cycle", &c.ar_dilation_cycle, &bad);
    cfg_int(o, "ar_text_attn_freq", &c.ar_text_attn_freq, &bad);
    cfg_int(o, "n_layers_nar", &c.n_layers_nar, &bad); cfg_int(o, "nar_head_dim", &c.nar_head_dim, &bad); cfg_int(o, "nar_kernel_size", &c.nar_kernel_size, &bad);
    cfg_ints(o, "nar_dilation_cycle", &c.nar_dilation_cycle, &bad);
    cfg_pair(o, "stage_B", c.stage[0], &bad); cfg_pair(o, "stage_C", c.stage[1], &bad); cfg_pair(o, "stage_D", c.stage[2], &bad);
    cfg_pair(o, "stage_E", c.stage[3], &bad);
    cfg_int(o, "sv_student_dim", &c.sv_student_dim, &bad); cfg_int(o, "ref_enc_layers", &c.ref_enc_layers, &bad);
    cfg_int(o, "ref_xattn_heads", &c.ref_xattn_heads, &bad); cfg_int(o, "ref_xattn_layers", &c.ref_xattn_layers, &bad);
    if (const Json* v = o.get("ref_xattn_gmax")) { if (v->kind == Json::Num) c.ref_xattn_gmax = v->num; else bad = true; }
    if (bad) { sopro_set_error("sopro_checkpoint_open: a cfg entry of %s is not an integer (list) of a usable size", sopro_path); return -3; }
    if (const char* why = check_cfg(c)) { sopro_set_error("sopro_checkpoint_open: checkpoint config outside what the engine takes: %s", why); return -3; }
  }
  k.m.num_quantizers = k.c.num_codebooks;
  if (int rc = k.pack_sopro()) return rc;
  if (mimi_path) {
    if (int rc = k.fm.open(mimi_path)) return rc;
    k.have_mimi = true;
    if (int rc = k.pack_mimi(ck->rope_positions)) return rc;
  }
  *out = ck.release();
  return 0;
  SOPRO_C_GUARD_END("sopro_checkpoint_open")
}

int sopro_checkpoint_close(sopro_checkpoint* ck) {
  delete ck;
  return 0;
}

int32_t sopro_checkpoint_count(const sopro_checkpoint* ck) { return ck ? (int32_t)ck->k.names.size() : 0; }

int sopro_checkpoint_tensor(const sopro_checkpoint* ck, int32_t i, const char** name, const float** data, int64_t* shape4, int32_t* ndim) {
  SOPRO_CHECK_ARG(ck && i >= 0 && i < (int32_t)ck->k.names.size() && name && data && shape4 && ndim, "bad index or NULL output");
  SOPRO_C_GUARD_BEGIN
  const std::string& n = ck->k.names[(size_t)i];
  const T& t = ck->k.p.at(n);
  *name = n.c_str();
  *data = t.v.data();
  *ndim = (int32_t)t.shape.size();
  for (int d = 0; d < 4; ++d) shape4[d] = d < (int)t.shape.size() ? t.shape[(size_t)d] : 1;
  return 0;
  SOPRO_C_GUARD_END("sopro_checkpoint_tensor")
}

int sopro_checkpoint_engine_cfg(const sopro_checkpoint* ck, int32_t precision, sopro_engine_cfg* cfg) {
  SOPRO_CHECK_ARG(ck && cfg && (precision == 0 || precision == 1), "NULL argument, or precision not 0 (fp32 parity) / 1 (bf16 mode)");
  SOPRO_C_GUARD_BEGIN
  const SoproCfg& c = ck->k.c;
  const MimiCfg& m = ck->k.m;
  memset(cfg, 0, sizeof(*cfg));
  cfg->d_model = c.d_model; cfg->codebook_size = c.codebook_size; cfg->num_codebooks = c.num_codebooks; cfg->nar_head_dim = c.nar_head_dim;
  cfg->bos_row = c.num_codebooks * c.codebook_size;
  cfg->n_layers_ar = c.n_layers_ar; cfg->ar_kernel = c.ar_kernel;
  const std::vector<int> ad = c.cycle(c.ar_dilation_cycle, c.n_layers_ar), nd = c.cycle(c.nar_dilation_cycle, c.n_layers_nar);
  for (int i = 0; i < c.n_layers_ar; ++i) {
    cfg->ar_dilations[i] = ad[(size_t)i];
    if ((i + 1) % c.ar_text_attn_freq == 0) { cfg->ar_xattn[i] = 1; cfg->ar_gate[i] = ck->k.gate[i]; }
  }
  cfg->n_layers_nar = c.n_layers_nar; cfg->nar_kernel = c.nar_kernel_size;
  for (int i = 0; i < c.n_layers_nar; ++i) cfg->nar_dilations[i] = nd[(size_t)i];
  int pos = 0;
  for (int s = 0; s < 4; ++s) {
    const std::vector<int> cbs = c.stage_cbs(s);
    if (cbs.empty()) continue;
    cfg->stage_first_cb[pos] = cbs[0]; cfg->stage_n_cb[pos] = (int32_t)cbs.size();
    cfg->nar_mix[pos][0] = ck->k.mix[pos][0]; cfg->nar_mix[pos][1] = ck->k.mix[pos][1];
    ++pos;
  }
  cfg->n_stages = pos;
  const T& pw = ck->k.p.at("nar_prev_cb_weights");
  for (size_t i = 0; i < pw.v.size() && i < 64; ++i) cfg->nar_prev_cb_weights[i] = pw.v[i];
  cfg->mimi_hidden = m.hidden; cfg->mimi_codebook_dim = m.codebook_dim; cfg->mimi_heads = m.heads; cfg->mimi_head_dim = m.head_dim;
  cfg->mimi_layers = m.layers; cfg->mimi_window = m.window; cfg->mimi_inter = m.inter;
  cfg->mimi_n_ratios = (int32_t)m.ratios.size();
  for (size_t i = 0; i < m.ratios.size(); ++i) cfg->mimi_ratios[i] = m.ratios[i];
  cfg->mimi_num_filters = m.num_filters; cfg->mimi_kernel = m.kernel; cfg->mimi_res_kernel = m.res_kernel; cfg->mimi_last_kernel = m.last_kernel;
  cfg->mimi_compress = m.compress; cfg->mimi_n_semantic = m.num_semantic; cfg->mimi_rope_positions = ck->rope_positions;
  cfg->mimi_norm_eps = (float)m.norm_eps; cfg->mimi_final_bias = ck->k.final_bias;
  cfg->precision = precision;
  cfg->n_layers_text = c.n_layers_text; cfg->ref_enc_layers = c.ref_enc_layers; cfg->ref_xattn_layers = c.ref_xattn_layers;
  cfg->ref_xattn_heads = c.ref_xattn_heads; cfg->sv_student_dim = c.sv_student_dim; cfg->enc_kernel = 7;
  return 0;
  SOPRO_C_GUARD_END("sopro_checkpoint_engine_cfg")
}

int sopro_engine_from_checkpoint(const sopro_checkpoint* ck, int32_t precision, void* stream, sopro_engine** out) {
  SOPRO_CHECK_ARG(ck && out, "NULL argument");
  SOPRO_C_GUARD_BEGIN
  sopro_engine_cfg cfg;
  if (int rc = sopro_checkpoint_engine_cfg(ck, precision, &cfg)) return rc;
  sopro_engine* e = nullptr;
  if (int rc = sopro_engine_create(&cfg, &e)) return rc;
  for (const std::string& n : ck->k.names) {
    const T& t = ck->k.p.at(n);
    if (t.shape.empty() || t.shape.size() > 4) continue;  // (scalars travel in the config)
    int64_t shape[4] = {1, 1, 1, 1};
    for (size_t d = 0; d < t.shape.size(); ++d) shape[d] = t.shape[d];
    if (int rc = sopro_engine_upload_tensor(e, n.c_str(), t.v.data(), shape, (int32_t)t.shape.size(), stream)) { sopro_engine_destroy(e); return rc; }
  }
  if (int rc = sopro_engine_finalize(e, stream)) { sopro_engine_destroy(e); return rc; }
  *out = e;
  return 0;
  SOPRO_C_GUARD_END("sopro_engine_from_checkpoint")
}

}  // extern "C"
