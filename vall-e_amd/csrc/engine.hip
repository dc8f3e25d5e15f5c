// libvallex.so — host side of the engine and the C ABI declared in include/vallex.h.
// One engine = one model replica on one GPU: weights, KV cache, activation arena, a private
// stream and the captured hipGraph of one AR decode step.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <array>
#include <cstring>
#include <string>
#include <type_traits>
#include <unordered_map>
#include <vector>

#include "../../include/vallex.h"
#ifdef VX_STAMPS
__device__ unsigned long long g_vx_stamps[32];
__device__ unsigned long long* g_vx_kstamps = nullptr;
__device__ unsigned long long* g_fq_stamps = nullptr;
#endif
#include "ar_kernels.hpp"
#include "ar_tp.hpp"
#include "rows_kernels.hpp"
#include "mfma_kernels.hpp"
#include "batch_kernels.hpp"
#include "mx_kernels.hpp"

using namespace vx;

// ------------------------------------------------------------------------------ errors
static thread_local std::string g_err;
static int fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}
#define HIPC(expr)                                                                                  \
  do {                                                                                              \
    hipError_t e_ = (expr);                                                                         \
    if (e_ != hipSuccess) return fail(VX_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                                      __FILE__, __LINE__);                                          \
  } while (0)
#define VXC(expr)               \
  do {                          \
    int r_ = (expr);            \
    if (r_ != VX_OK) return r_; \
  } while (0)

extern "C" const char* vx_last_error(void) { return g_err.c_str(); }

// Every entry point runs on the engine's device and leaves the caller's current device as it found it.
struct DevGuard {
  int prev = -1;
  hipError_t err = hipSuccess;
  explicit DevGuard(int dev) {
    err = hipGetDevice(&prev);
    if (err == hipSuccess && prev != dev) err = hipSetDevice(dev);
    else if (err == hipSuccess) prev = -1;  // already current: nothing to restore
  }
  ~DevGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};
#define ON_DEVICE(dev)   \
  DevGuard dev_guard_(dev); \
  HIPC(dev_guard_.err)

// ------------------------------------------------------------------------------ engine state
struct Tensor {
  void* p = nullptr;
  std::vector<int64_t> shape;
  size_t numel = 0;
  bool low = false;  // stored in the precision's matrix type (bf16 in VX_PREC_BF16)
  bool set = false;
  // VX_PREC_FP8_NAR: MXFP8 copy of the NAR stack's in_proj / linear1 / linear2 (mx_kernels.hpp): e4m3 bytes (N, K) and E8M0
  // block scales (K/32, N); the bf16 copy stays (row counts below the 256-tile path run on it)
  uint8_t *q8 = nullptr, *s8 = nullptr;
};

struct LayerW {
  const uint8_t *in_q8 = nullptr, *in_s8 = nullptr, *w1_q8 = nullptr, *w1_s8 = nullptr, *w2_q8 = nullptr, *w2_s8 = nullptr;
  const void *in_w, *out_w, *w1, *w2;
  const float *in_b, *out_b, *b1, *b2;
  const float *n1_g, *n1_b, *n2_g, *n2_b;
  // VALL-F (TransformerDecoderLayer): cross-attention over the text memory and the third norm
  const void *cin_w = nullptr, *cout_w = nullptr;
  const float *cin_b = nullptr, *cout_b = nullptr, *n3_g = nullptr, *n3_b = nullptr;
};

constexpr int POLL_CHUNK = 32;

struct vx_engine {
  vx_config cfg{};
  bool bf16 = false;
  bool fp8nar = false;  // VX_PREC_FP8_NAR: bf16 everywhere + MXFP8 QKV / FFN GEMMs in the NAR stages at >= 4096 rows
  uint8_t *Hn8 = nullptr, *SHn = nullptr, *FF8 = nullptr, *SFF = nullptr;  // MXFP8 row operands and their scales (ld = mx_ld)
  int mx_ld = 0;
  bool vallf = false;  // VX_FLAG_VALLF: decoder layers with cross-attention over the text (valle.py:49-719)
  int npl = 2;         // norms per layer: 2 (encoder layers) / 3 (decoder layers)
  void *xkv_ar = nullptr, *xkv_nar = nullptr;  // VALL-F: per-layer K / V of the text memory, [layer][K|V][head][max_text][hd]
  int mem_len = 0;     // text rows of the current utterance's AR memory
  bool hd64 = true;  // head_dim 64 in both stacks: the MFMA row kernels and the batched decode apply
  size_t esz = 4;  // bytes per matrix / KV / GEMM-operand element
  int num_cu = 256;
  hipStream_t es = nullptr;
  hipEvent_t ev_in = nullptr, ev_out = nullptr, ev_t[6] = {};
  hipEvent_t ev_poll[2] = {};
  std::unordered_map<std::string, Tensor> w;
  std::vector<std::string> keys;
  bool finalized = false;
  std::vector<LayerW> ar_l, nar_l;
  // sine tables
  float *pe_ar = nullptr, *pe_nar = nullptr;
  int pe_rows = 0;
  bool pe_ar_set = false, pe_nar_set = false;
  // AR buffers
  int ctx_max = 0;
  float *ar_x = nullptr, *ar_xn = nullptr, *ar_q = nullptr, *ar_part = nullptr, *ar_f = nullptr, *ar_logits = nullptr;
  void* kv = nullptr;  // [L][2][H][ctx_max][hd]
  // in-launch hand-overs of the sharded decode step (ar_granules.hpp): per-layer granule scratch, the step counter that tags the
  // granules ([0]) and the spin-timeout word ([1])
  // XCD-sharded decode step (ar_tp.hpp): two launches per layer.  Re-laid-out out-projection / linear2 weights per layer, the
  // partial-sum vectors of the two sharded GEMVs, the second residual buffer, the hidden-unit granules
  bool tp = false;
  std::vector<void*> tp_wo, tp_w2;
  long long* tp_xacc = nullptr;  // (3, d) int64 fixed-point residual accumulators in rotation (ar_tp.hpp TpAccArgs)
  float* tp_gbb = nullptr;  // (2 L + 1, 3, d): {gamma, beta, arriving bias} per norm site of the step
  fq_gran* tp_gh = nullptr;
  fq_gran *fq_gq = nullptr, *fq_gp = nullptr;
  unsigned* d_epoch = nullptr;
  ArState* d_st = nullptr;
  ArState* h_st = nullptr;  // pinned: [0] staging, [1..2] poll slots
  int *d_tokens = nullptr, *d_sampled = nullptr, *d_argmax = nullptr;
  float* d_noise = nullptr;
  size_t noise_cap = 0;
  long long* d_forced = nullptr;
  size_t forced_cap = 0;
  // row buffers
  int n_max = 0;
  float* X = nullptr;
  void *Hn = nullptr, *QKV = nullptr, *ATT = nullptr, *FF = nullptr, *VT = nullptr;
  int vt_ld = 0;
  float *yemb = nullptr, *nar_logits = nullptr, *ada = nullptr;
  long long *ids_text = nullptr, *ids_audio = nullptr, *ids_prompts = nullptr, *ids_samples = nullptr, *d_codes = nullptr;
  long long* d_fcodes = nullptr;  // teacher-forced NAR stages (vx_nar_ex): the caller's (T, Q) codes
  // batched decode (slots)
  int bmax = 0;
  float *bx = nullptr, *bq = nullptr, *bpart = nullptr, *blogits = nullptr, *btrace = nullptr;
  vx::bf16 *bh = nullptr, *batt = nullptr, *bff = nullptr, *bkv = nullptr;
  size_t bkv_slot = 0;  // elements per slot
  ArState* bst = nullptr;    // device, BMAX
  ArState* h_bst = nullptr;  // pinned: [0..BMAX) staging, [BMAX..3*BMAX) two poll slots
  int *btok = nullptr, *bsamp = nullptr, *bargm = nullptr;
  int btok_stride = 0;
  std::unordered_map<int, hipGraphExec_t> bgraphs;
  int *d_seg_start = nullptr, *d_seg_len = nullptr;  // segments of the concatenated row buffer (batched NAR / prefill)
  int* d_seg_text = nullptr;                          // per-segment text length (prefix mask of a batched prefill)
  // prenets (VX_FLAG_PRENET): scratch rows, conv weights re-laid out as [k][ci][co], decode-step vectors
  float *pn_a = nullptr, *pn_b = nullptr, *pn_h1 = nullptr, *pn_h2 = nullptr, *pn_text = nullptr, *d_zero = nullptr;
  float *ar_e = nullptr, *ar_h1 = nullptr, *ar_h2 = nullptr;
  float* convT[2][3] = {{nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr}};
  float* slab = nullptr;                              // split-K slabs of the N = d row GEMMs (4 x slab_rows x d fp32)
  int slab_rows = 0;
  bool seg_text_on = false;                           // true only while a batched prefill runs its stack
  long long *bp_text = nullptr, *bp_audio = nullptr;  // id staging of the batched prefill
  size_t cap_audio = 0, cap_text = 0;                 // rows the id / yemb / logits staging buffers hold
  int nseg = 0, max_seg_len = 0;
  int bS[BMAX] = {}, bP[BMAX] = {}, bbos[BMAX] = {}, bngen[BMAX] = {}, breason[BMAX] = {};
  bool bprefilled[BMAX] = {};
  double t_bdecode = 0, n_blaunch = 0;
  // graph
  hipGraph_t graph = nullptr;
  hipGraphExec_t gexec = nullptr;
  // per-utterance state
  int S = 0, P = 0, bos = 0;
  bool prefilled = false, decoded = false;
  int n_gen = 0, stop_reason = 0, n_pass = 0, last_T = 0, last_N = 0;
  double t_prefill = 0, t_decode = 0, t_nar = 0, n_launch = 0;
  // VX_TIME_GEMMS=1 (bench.py): HIP-event pairs around every QKV / out-projection / FFN GEMM of the NAR stages
  std::vector<hipEvent_t> gemm_ev;
  size_t gemm_ev_used = 0;
  double gemm_flops = 0, t_gemm = 0, gemm_flops_done = 0;
  std::vector<void*> allocs;
  char* arena = nullptr;
  size_t arena_used = 256, arena_cap = (size_t)4 << 20;  // no sub-block equals the base pointer (which `allocs` owns)
};

// VX_POISON=1 (tests): every fresh device allocation is filled with 0xFF bytes (NaN as bf16 / fp32, -1 as integers)
// before the engine's own initialisation runs, so a read of memory nothing wrote shows up as NaN output instead of
// depending on what the allocator happened to hand back.
static bool poison_on() {
  static const bool on = [] { const char* v = getenv("VX_POISON"); return v && atoi(v) != 0; }();
  return on;
}
// Every fill below goes to the engine's stream: `es` is a non-blocking stream, a null-stream hipMemset is not ordered with the
// kernels enqueued on it right afterwards (a fill landing late would wipe rows the first kernels had already written).
static int dalloc(vx_engine* e, void** p, size_t bytes) {
  // small blocks (the decode step's vectors, states, per-site norm parameters) share ONE 4 MB block: one translation entry
  // serves them all (each is touched by every workgroup of every launch of the step)
  static const bool arena_on = !(getenv("VX_ARENA") && atoi(getenv("VX_ARENA")) == 0);
  if (arena_on && bytes <= (64u << 10)) {
    const size_t need = (bytes ? bytes : 16) + 255 & ~(size_t)255;
    if (e->arena == nullptr) {
      HIPC(hipMalloc((void**)&e->arena, e->arena_cap));
      e->allocs.push_back(e->arena);
    }
    if (e->arena_used + need <= e->arena_cap) {
      *p = e->arena + e->arena_used;
      e->arena_used += need;
      if (poison_on()) {
        HIPC(hipMemsetAsync(*p, 0xFF, need, e->es));
        HIPC(hipStreamSynchronize(e->es));
      }
      return VX_OK;
    }
  }
  HIPC(hipMalloc(p, bytes ? bytes : 16));
  if (poison_on()) {  // debug mode: also finished before anything on another stream (weight uploads) can touch the block
    HIPC(hipMemsetAsync(*p, 0xFF, bytes ? bytes : 16, e->es));
    HIPC(hipStreamSynchronize(e->es));
  }
  e->allocs.push_back(*p);
  return VX_OK;
}
template <typename T> static int dalloc_t(vx_engine* e, T** p, size_t n) { return dalloc(e, (void**)p, n * sizeof(T)); }

constexpr int PRENET_H = 256;  // hidden width of the audio prenets (valle.py:116-122)

static bool is_matrix_key(const std::string& k) {
  auto ends = [&](const char* s) { size_t n = strlen(s); return k.size() >= n && k.compare(k.size() - n, n, s) == 0; };
  if (k.find("project_layer") != std::string::npos) return false;
  return ends("in_proj_weight") || ends("out_proj.weight") || ends("linear1.weight") || ends("linear2.weight") ||
         k.rfind("ar_predict_layer", 0) == 0 || k.rfind("nar_predict_layers", 0) == 0;
}

static bool is_mx_key(const std::string& k) {  // the NAR stack's QKV / FFN matrices (BASELINE configs[4])
  auto ends = [&](const char* s) { size_t n = strlen(s); return k.size() >= n && k.compare(k.size() - n, n, s) == 0; };
  return k.rfind("nar_decoder.layers.", 0) == 0 && (ends("in_proj_weight") || ends("linear1.weight") || ends("linear2.weight"));
}

// The reference's state_dict layout (valle.py:85-259); mirrored by valle_amd/weights.py.
static void add_encoder_keys(vx_engine* e, const std::string& pre, int d, int L, bool adaptive) {
  const bool post = e->cfg.flags & VX_FLAG_POST_NORM;  // norm=... if norm_first else None (valle.py:151, 242-246)
  const bool cross = e->cfg.flags & VX_FLAG_VALLF;     // TransformerDecoderLayer (modules/transformer.py:412-500)
  auto add = [&](const std::string& k, std::vector<int64_t> s) {
    Tensor t; t.shape = s; t.numel = 1; for (auto v : s) t.numel *= (size_t)v;
    t.low = e->bf16 && is_matrix_key(k);
    e->w[k] = t; e->keys.push_back(k);
  };
  auto norm = [&](const std::string& p) {
    if (adaptive) {
      add(p + ".project_layer.weight", {2 * d, d}); add(p + ".project_layer.bias", {2 * d});
      add(p + ".norm.weight", {d}); add(p + ".norm.bias", {d});
    } else {
      add(p + ".weight", {d}); add(p + ".bias", {d});
    }
  };
  for (int i = 0; i < L; ++i) {
    const std::string p = pre + ".layers." + std::to_string(i);
    add(p + ".self_attn.in_proj_weight", {3 * d, d}); add(p + ".self_attn.in_proj_bias", {3 * d});
    add(p + ".self_attn.out_proj.weight", {d, d}); add(p + ".self_attn.out_proj.bias", {d});
    if (cross) {
      add(p + ".multihead_attn.in_proj_weight", {3 * d, d}); add(p + ".multihead_attn.in_proj_bias", {3 * d});
      add(p + ".multihead_attn.out_proj.weight", {d, d}); add(p + ".multihead_attn.out_proj.bias", {d});
    }
    add(p + ".linear1.weight", {4 * d, d}); add(p + ".linear1.bias", {4 * d});
    add(p + ".linear2.weight", {d, 4 * d}); add(p + ".linear2.bias", {d});
    norm(p + ".norm1"); norm(p + ".norm2");
    if (cross) norm(p + ".norm3");
  }
  if (!post) norm(pre + ".norm");
}

static void build_key_table(vx_engine* e) {
  const vx_config& c = e->cfg;
  const int d = c.d_model, dn = c.nar_d_model, Q = c.num_quantizers;
  auto add = [&](const std::string& k, std::vector<int64_t> s) {
    Tensor t; t.shape = s; t.numel = 1; for (auto v : s) t.numel *= (size_t)v;
    t.low = e->bf16 && is_matrix_key(k);
    e->w[k] = t; e->keys.push_back(k);
  };
  add("ar_text_embedding.word_embeddings.weight", {512, d});
  add("nar_text_embedding.word_embeddings.weight", {512, dn});
  add("ar_audio_embedding.word_embeddings.weight", {1025 + (c.prepend_bos ? 1 : 0), d});
  // prenets (valle.py:96-123, 181-213): Sequential indices as in the reference; BatchNorm's num_batches_tracked is an
  // integer counter the forward pass never reads and is not passed through the C ABI
  auto prenet = [&](const std::string& pre, int dd) {
    for (int conv : {1, 5, 9}) {
      const std::string cv = pre + "_text_prenet." + std::to_string(conv), bn = pre + "_text_prenet." + std::to_string(conv + 1);
      add(cv + ".weight", {dd, dd, 5}); add(cv + ".bias", {dd});
      add(bn + ".weight", {dd}); add(bn + ".bias", {dd}); add(bn + ".running_mean", {dd}); add(bn + ".running_var", {dd});
    }
    add(pre + "_text_prenet.14.weight", {dd, dd}); add(pre + "_text_prenet.14.bias", {dd});
    add(pre + "_audio_prenet.0.weight", {PRENET_H, dd}); add(pre + "_audio_prenet.0.bias", {PRENET_H});
    add(pre + "_audio_prenet.3.weight", {PRENET_H, PRENET_H}); add(pre + "_audio_prenet.3.bias", {PRENET_H});
    add(pre + "_audio_prenet.6.weight", {dd, PRENET_H}); add(pre + "_audio_prenet.6.bias", {dd});
  };
  const bool pn = c.flags & VX_FLAG_PRENET;
  if (pn) prenet("ar", d);
  add("ar_text_position.alpha", {1});
  add("ar_audio_position.alpha", {1});
  add_encoder_keys(e, "ar_decoder", d, c.num_layers, false);
  add("ar_predict_layer.weight", {1025, d});
  if (Q > 1) {
    add("nar_audio_embeddings.0.word_embeddings.weight", {1025, dn});
    for (int j = 1; j < Q; ++j) add("nar_audio_embeddings." + std::to_string(j) + ".word_embeddings.weight", {1024, dn});
    if (pn) prenet("nar", dn);
    add("nar_text_position.alpha", {1});
    add("nar_audio_position.alpha", {1});
    add_encoder_keys(e, "nar_decoder", dn, c.nar_num_layers, true);
    for (int j = 0; j < Q - 1; ++j) add("nar_predict_layers." + std::to_string(j) + ".weight", {1024, dn});
    for (int j = 0; j < Q - 1; ++j) add("nar_stage_embeddings." + std::to_string(j) + ".word_embeddings.weight", {1, dn});
  }
}

static void host_sine_table(std::vector<float>& t, int rows, int d) {
  t.assign((size_t)rows * d, 0.f);
  for (int i = 0; i < d; i += 2) {
    const float w = expf((float)i * -(logf(10000.0f) / (float)d));
    for (int p = 0; p < rows; ++p) {
      t[(size_t)p * d + i] = sinf((float)p * w);
      if (i + 1 < d) t[(size_t)p * d + i + 1] = cosf((float)p * w);
    }
  }
}

// ------------------------------------------------------------------------------ create/destroy
static int create_body(vx_engine* e);
static int tp_setup(vx_engine* e);
extern "C" void vx_destroy(vx_engine* e);
extern "C" int vx_create(const vx_config* cfg, vx_engine** out) {
  if (!cfg || !out) return fail(VX_ERR_ARG, "null argument");
  if (cfg->struct_size != (int32_t)sizeof(vx_config)) return fail(VX_ERR_ARG, "vx_config.struct_size mismatch");
  const vx_config& c = *cfg;
  if (c.d_model <= 0 || c.nhead <= 0 || c.num_layers <= 0 || c.d_model % c.nhead)
    return fail(VX_ERR_ARG, "bad d_model/nhead/num_layers");
  if (c.num_quantizers < 1 || c.num_quantizers > 8) return fail(VX_ERR_ARG, "num_quantizers must be 1..8");
  if (c.prefix_mode != 0 && c.prefix_mode != 1 && c.prefix_mode != 2 && c.prefix_mode != 4)
    return fail(VX_ERR_ARG, "prefix_mode must be 0/1/2/4");
  // head_dim 64 is the built geometry (MFMA attention, batched decode); 4 / 8 / 16 / 32 run on the plain kernels, batch-1
  // only: the reference's own tests use decoder_dim 64 / nhead 16 and half of that for the NAR stack (valle_test.py:93-95)
  auto hd_ok = [](int hd) { return hd == 4 || hd == 8 || hd == 16 || hd == 32 || hd == 64; };
  if (c.num_quantizers > 1 && (c.nar_nhead <= 0 || c.nar_d_model % c.nar_nhead)) return fail(VX_ERR_ARG, "bad nar_d_model/nar_nhead");
  if (!hd_ok(c.d_model / c.nhead) || (c.num_quantizers > 1 && !hd_ok(c.nar_d_model / c.nar_nhead)))
    return fail(VX_ERR_UNSUPPORTED, "head_dim (d_model/nhead) must be 4, 8, 16, 32 or 64");
  const bool hd64 = c.d_model / c.nhead == 64 && (c.num_quantizers == 1 || c.nar_d_model / c.nar_nhead == 64);
  if (c.d_model % 8 || c.d_model > 1024 || c.nar_d_model > 1024 || (c.num_quantizers > 1 && c.nar_d_model % 8))
    return fail(VX_ERR_UNSUPPORTED, "d_model must be a multiple of 8 and <= 1024");
  if (hd64 && (c.d_model % 64 || (c.num_quantizers > 1 && c.nar_d_model % 64)))
    return fail(VX_ERR_UNSUPPORTED, "d_model must be a multiple of 64 at head_dim 64");
  if (!hd64 && c.max_batch > 1) return fail(VX_ERR_UNSUPPORTED, "batched decode needs head_dim 64");
  if (c.max_text <= 0 || c.max_audio <= 0) return fail(VX_ERR_ARG, "capacities must be positive");
  if (c.precision != VX_PREC_F32 && c.precision != VX_PREC_BF16 && c.precision != VX_PREC_FP8_NAR) return fail(VX_ERR_ARG, "bad precision");
  if (c.precision == VX_PREC_FP8_NAR && (c.num_quantizers < 2 || c.nar_d_model % 256 || c.nar_d_model / c.nar_nhead != 64 ||
                                         (c.flags & (VX_FLAG_POST_NORM | VX_FLAG_PRENET | VX_FLAG_SIMPLE_ROWS))))
    return fail(VX_ERR_UNSUPPORTED, "VX_PREC_FP8_NAR needs a pre-norm NAR stack without prenets, head_dim 64 and nar_d_model % 256 == 0");
  if ((c.flags & (VX_FLAG_POST_NORM | VX_FLAG_PRENET | VX_FLAG_VALLF)) && c.max_batch > 1)
    return fail(VX_ERR_UNSUPPORTED, "post-norm / prenet / VALL-F models run on the batch-1 path only");
  if ((c.flags & VX_FLAG_VALLF) && c.precision == VX_PREC_FP8_NAR) return fail(VX_ERR_UNSUPPORTED, "VX_PREC_FP8_NAR is built for VALL-E only");
  if (c.max_batch < 0 || c.max_batch > BMAX) return fail(VX_ERR_ARG, "max_batch must be 0..%d", BMAX);
  if (c.max_batch > 1 && (c.precision == VX_PREC_F32 || c.d_model % 128))
    return fail(VX_ERR_UNSUPPORTED, "batched decode needs bf16 precision and d_model % 128 == 0");

  ON_DEVICE(c.device);
  vx_engine* e = new vx_engine();
  e->cfg = c;
  const int rc = create_body(e);
  if (rc != VX_OK) {  // g_err holds the failing call; release whatever was allocated up to it
    const std::string keep = g_err;
    vx_destroy(e);
    g_err = keep;
    return rc;
  }
  *out = e;
  return VX_OK;
}

static int create_body(vx_engine* e) {
  const vx_config& c = e->cfg;
  const bool hd64 = c.d_model / c.nhead == 64 && (c.num_quantizers == 1 || c.nar_d_model / c.nar_nhead == 64);
  e->bf16 = c.precision != VX_PREC_F32;
  e->fp8nar = c.precision == VX_PREC_FP8_NAR;
  e->hd64 = hd64;
  e->vallf = c.flags & VX_FLAG_VALLF;
  e->npl = e->vallf ? 3 : 2;
  e->esz = e->bf16 ? 2 : 4;
  hipDeviceProp_t prop;
  HIPC(hipGetDeviceProperties(&prop, c.device));
  e->num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  HIPC(hipStreamCreateWithFlags(&e->es, hipStreamNonBlocking));
  HIPC(hipEventCreateWithFlags(&e->ev_in, hipEventDisableTiming));
  HIPC(hipEventCreateWithFlags(&e->ev_out, hipEventDisableTiming));
  for (auto& ev : e->ev_t) HIPC(hipEventCreate(&ev));
  for (auto& ev : e->ev_poll) HIPC(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
  build_key_table(e);

  const int d = c.d_model, dn = c.num_quantizers > 1 ? c.nar_d_model : c.d_model;
  const int dmax = d > dn ? d : dn;
  e->ctx_max = c.max_text + c.max_audio;
  e->n_max = e->ctx_max;
  const int H = c.nhead, hd = d / H;
  // sine tables (embedding.py:64 starts at 4000 rows and extends on demand)
  e->pe_rows = e->ctx_max > 4000 ? e->ctx_max : 4000;
  VXC(dalloc_t(e, &e->pe_ar, (size_t)e->pe_rows * d));
  VXC(dalloc_t(e, &e->pe_nar, (size_t)e->pe_rows * dn));
  // AR
  VXC(dalloc_t(e, &e->ar_x, d));
  VXC(dalloc_t(e, &e->ar_q, d));
  VXC(dalloc_t(e, &e->ar_xn, d));  // post-norm decode: the normalised residual stream
  VXC(dalloc_t(e, &e->ar_part, (size_t)H * ATT_NSPLIT * ATT_PSTRIDE));
  VXC(dalloc_t(e, &e->ar_f, 4 * (size_t)d));
  VXC(dalloc_t(e, &e->d_epoch, 2));
  {
    const unsigned init[2] = {1u, 0u};  // tags start at 1: zero-filled granules never match
    HIPC(hipMemcpy(e->d_epoch, init, sizeof init, hipMemcpyHostToDevice));
  }
  VXC(tp_setup(e));
  const size_t nlog = (c.flags & VX_FLAG_TRACE_LOGITS) ? (size_t)c.max_audio + 2 : 1;
  VXC(dalloc_t(e, &e->ar_logits, LOGITS_CUR + nlog * AR_VOCAB));
  VXC(dalloc(e, &e->kv, (size_t)c.num_layers * 2 * H * e->ctx_max * hd * e->esz));
  VXC(dalloc_t(e, &e->d_st, 1));
  HIPC(hipHostMalloc((void**)&e->h_st, 3 * sizeof(ArState) + 16));  // + {epoch, spin-timeout word} read back per decode
  VXC(dalloc_t(e, &e->d_tokens, (size_t)c.max_audio + 2));
  VXC(dalloc_t(e, &e->d_sampled, (size_t)c.max_audio + 2));
  VXC(dalloc_t(e, &e->d_argmax, (size_t)c.max_audio + 2));
  // rows
  const size_t n = e->n_max;
  VXC(dalloc_t(e, &e->X, n * dmax));
  VXC(dalloc(e, &e->Hn, n * dmax * e->esz));
  VXC(dalloc(e, &e->QKV, n * 3 * dmax * e->esz));
  VXC(dalloc(e, &e->ATT, n * dmax * e->esz));
  // rows between the segments of a concatenated batch are never written by attention but are read by the
  // out-projection: they must hold finite values (a NaN row would reach valid rows as 0 x NaN through its V^T column)
  HIPC(hipMemsetAsync(e->ATT, 0, n * dmax * e->esz, e->es));
  e->vt_ld = ((e->n_max + 63) / 64) * 64 + 64;  // key-padded row length of V^T (16-byte aligned tiles)
  VXC(dalloc(e, &e->VT, (size_t)dmax * e->vt_ld * 2));
  HIPC(hipMemsetAsync(e->VT, 0, (size_t)dmax * e->vt_ld * 2, e->es));  // padding keys must stay finite (they meet P = 0)
  VXC(dalloc(e, &e->FF, n * 4 * dmax * e->esz));
  VXC(dalloc_t(e, &e->yemb, (size_t)c.max_audio * dn));
  VXC(dalloc_t(e, &e->nar_logits, (size_t)c.max_audio * 1024));
  VXC(dalloc_t(e, &e->ids_text, (size_t)c.max_text));
  VXC(dalloc_t(e, &e->ids_audio, (size_t)c.max_audio + 1));
  VXC(dalloc_t(e, &e->ids_prompts, (size_t)c.max_audio * 8));
  VXC(dalloc_t(e, &e->ids_samples, (size_t)c.max_audio));
  VXC(dalloc_t(e, &e->d_codes, (size_t)c.max_audio * 8));
  VXC(dalloc_t(e, &e->d_fcodes, (size_t)c.max_audio * 8));
  e->cap_audio = c.max_audio; e->cap_text = c.max_text;
  if (e->fp8nar) {
    e->mx_ld = (e->n_max + 255) / 256 * 256;
    VXC(dalloc_t(e, &e->Hn8, n * dmax));
    VXC(dalloc_t(e, &e->FF8, n * 4 * dmax));
    VXC(dalloc_t(e, &e->SHn, (size_t)(dmax / 32) * e->mx_ld));
    VXC(dalloc_t(e, &e->SFF, (size_t)(4 * dmax / 32) * e->mx_ld));
    HIPC(hipMemsetAsync(e->SHn, 0, (size_t)(dmax / 32) * e->mx_ld, e->es));  // the pad rows' scales must stay finite
    HIPC(hipMemsetAsync(e->SFF, 0, (size_t)(4 * dmax / 32) * e->mx_ld, e->es));
  }
  if (e->bf16 && !(c.flags & VX_FLAG_SIMPLE_ROWS)) {
    e->slab_rows = e->n_max < 4095 ? e->n_max : 4095;
    VXC(dalloc_t(e, &e->slab, (size_t)4 * e->slab_rows * dmax));
  }
  if (c.flags & VX_FLAG_PRENET) {
    VXC(dalloc_t(e, &e->pn_a, n * dmax));
    VXC(dalloc_t(e, &e->pn_b, n * dmax));
    VXC(dalloc_t(e, &e->pn_h1, n * PRENET_H));
    VXC(dalloc_t(e, &e->pn_h2, n * PRENET_H));
    VXC(dalloc_t(e, &e->pn_text, (size_t)c.max_text * dmax));
    VXC(dalloc_t(e, &e->d_zero, 4));
    HIPC(hipMemsetAsync(e->d_zero, 0, 16, e->es));
    VXC(dalloc_t(e, &e->ar_e, d));
    VXC(dalloc_t(e, &e->ar_h1, PRENET_H));
    VXC(dalloc_t(e, &e->ar_h2, PRENET_H));
    for (int i = 0; i < 3; ++i) {
      VXC(dalloc_t(e, &e->convT[0][i], (size_t)5 * d * d));
      if (c.num_quantizers > 1) VXC(dalloc_t(e, &e->convT[1][i], (size_t)5 * dn * dn));
    }
  }
  if (c.max_batch > 1) {
    e->bmax = c.max_batch;
    e->btok_stride = c.max_audio + 2;
    e->bkv_slot = (size_t)c.num_layers * 2 * H * e->ctx_max * hd;
    VXC(dalloc_t(e, &e->bx, (size_t)BMAX * d));
    VXC(dalloc_t(e, &e->bq, (size_t)BMAX * d));
    VXC(dalloc_t(e, &e->bpart, (size_t)4 * BMAX * d));
    VXC(dalloc_t(e, &e->blogits, (size_t)BMAX * LOGITS_CUR));
    if (c.flags & VX_FLAG_TRACE_LOGITS)  // parity tests: every pass's logits row of every slot
      VXC(dalloc_t(e, &e->btrace, (size_t)e->bmax * (c.max_audio + 2) * AR_VOCAB));
    VXC(dalloc_t(e, &e->bh, (size_t)BMAX * d));
    VXC(dalloc_t(e, &e->batt, (size_t)BMAX * d));
    VXC(dalloc_t(e, &e->bff, (size_t)BMAX * 4 * d));
    VXC(dalloc_t(e, &e->bkv, (size_t)e->bmax * e->bkv_slot));
    VXC(dalloc_t(e, &e->bst, (size_t)BMAX));
    VXC(dalloc_t(e, &e->btok, (size_t)BMAX * e->btok_stride));
    VXC(dalloc_t(e, &e->bsamp, (size_t)BMAX * e->btok_stride));
    VXC(dalloc_t(e, &e->bargm, (size_t)BMAX * e->btok_stride));
    HIPC(hipHostMalloc((void**)&e->h_bst, 3 * BMAX * sizeof(ArState)));
    VXC(dalloc_t(e, &e->bp_text, (size_t)BMAX * c.max_text));
    VXC(dalloc_t(e, &e->bp_audio, (size_t)BMAX * (c.max_audio + 1)));
    VXC(dalloc_t(e, &e->d_seg_start, (size_t)BMAX));
    VXC(dalloc_t(e, &e->d_seg_len, (size_t)BMAX));
    VXC(dalloc_t(e, &e->d_seg_text, (size_t)BMAX));
    // MFMA A operands always read 32 rows: rows of unused slots must hold finite values
    HIPC(hipMemsetAsync(e->bx, 0, (size_t)BMAX * d * 4, e->es));
    HIPC(hipMemsetAsync(e->bh, 0, (size_t)BMAX * d * 2, e->es));
    HIPC(hipMemsetAsync(e->batt, 0, (size_t)BMAX * d * 2, e->es));
    HIPC(hipMemsetAsync(e->bff, 0, (size_t)BMAX * 4 * d * 2, e->es));
    HIPC(hipMemsetAsync(e->bst, 0, (size_t)BMAX * sizeof(ArState), e->es));
  }
  if (c.num_quantizers > 1)
    VXC(dalloc_t(e, &e->ada, (size_t)(c.num_quantizers - 1) * (e->npl * c.nar_num_layers + 1) * 2 * dn));
  if (e->vallf) {  // K / V of the text memory per layer, in the decode-cache layout with max_text rows per head
    VXC(dalloc(e, &e->xkv_ar, (size_t)c.num_layers * 2 * d * c.max_text * e->esz));
    if (c.num_quantizers > 1) VXC(dalloc(e, &e->xkv_nar, (size_t)c.nar_num_layers * 2 * dn * c.max_text * e->esz));
  }
  // weights
  for (auto& k : e->keys) {
    Tensor& t = e->w[k];
    VXC(dalloc(e, &t.p, t.numel * (t.low ? 2 : 4)));
    if (e->fp8nar && is_mx_key(k)) {
      VXC(dalloc_t(e, &t.q8, t.numel));
      VXC(dalloc_t(e, &t.s8, t.numel / 32));
    }
  }
  HIPC(hipStreamSynchronize(e->es));  // the fills above are done before the caller's uploads (other streams) begin
  return VX_OK;
}

extern "C" void vx_destroy(vx_engine* e) {
  if (!e) return;
  DevGuard dev_guard_(e->cfg.device);
  if (e->es) (void)hipStreamSynchronize(e->es);
  for (auto& kvp : e->bgraphs) (void)hipGraphExecDestroy(kvp.second);
  if (e->h_bst) (void)hipHostFree(e->h_bst);
  if (e->gexec) (void)hipGraphExecDestroy(e->gexec);
  if (e->graph) (void)hipGraphDestroy(e->graph);
  for (void* p : e->allocs) (void)hipFree(p);
  if (e->d_noise) (void)hipFree(e->d_noise);
  if (e->d_forced) (void)hipFree(e->d_forced);
  if (e->h_st) (void)hipHostFree(e->h_st);
  for (auto& ev : e->ev_t) if (ev) (void)hipEventDestroy(ev);
  for (auto& ev : e->gemm_ev) (void)hipEventDestroy(ev);
  for (auto& ev : e->ev_poll) if (ev) (void)hipEventDestroy(ev);
  if (e->ev_in) (void)hipEventDestroy(e->ev_in);
  if (e->ev_out) (void)hipEventDestroy(e->ev_out);
  if (e->es) (void)hipStreamDestroy(e->es);
  delete e;
}

// ------------------------------------------------------------------------------ weights
extern "C" int vx_set_weight(vx_engine* e, const char* key, const float* data, const int64_t* shape, int32_t ndim) {
  if (!e || !key || !data || !shape) return fail(VX_ERR_ARG, "null argument");
  ON_DEVICE(e->cfg.device);
  auto it = e->w.find(key);
  if (it == e->w.end()) return fail(VX_ERR_WEIGHTS, "unexpected key '%s'", key);
  Tensor& t = it->second;
  bool ok = (size_t)ndim == t.shape.size();
  for (int i = 0; ok && i < ndim; ++i) ok = shape[i] == t.shape[i];
  if (!ok) return fail(VX_ERR_WEIGHTS, "shape mismatch for '%s'", key);
  if (!t.low) {
    HIPC(hipMemcpyAsync(t.p, data, t.numel * 4, hipMemcpyDefault, e->es));
  } else {
    float* stage = nullptr;
    HIPC(hipMalloc((void**)&stage, t.numel * 4));
    HIPC(hipMemcpyAsync(stage, data, t.numel * 4, hipMemcpyDefault, e->es));
    convert_kernel<bf16><<<1024, 256, 0, e->es>>>(stage, (bf16*)t.p, t.numel);
    if (t.q8 != nullptr) {  // (N, K) -> e4m3 bytes + (K/32, N) block scales, from the fp32 values
      const int N = (int)t.shape[0], K = (int)t.shape[1];
      mx_quant_rows_kernel<<<(N + 3) / 4, 256, 0, e->es>>>(stage, t.q8, t.s8, N, K, N);
    }
    HIPC(hipGetLastError());
    HIPC(hipStreamSynchronize(e->es));
    HIPC(hipFree(stage));
  }
  HIPC(hipStreamSynchronize(e->es));  // `data` may be freed by the caller on return
  t.set = true;
  e->finalized = false;
  return VX_OK;
}

extern "C" int vx_set_sine_table(vx_engine* e, int32_t which, const float* data, int64_t rows, int64_t dim) {
  if (!e || !data) return fail(VX_ERR_ARG, "null argument");
  ON_DEVICE(e->cfg.device);
  const int d = which == 0 ? e->cfg.d_model : e->cfg.nar_d_model;
  if (dim != d || rows < e->pe_rows) return fail(VX_ERR_ARG, "sine table must be (>= %d, %d)", e->pe_rows, d);
  const int64_t r = rows < e->pe_rows ? rows : e->pe_rows;
  HIPC(hipMemcpyAsync(which == 0 ? e->pe_ar : e->pe_nar, data, (size_t)r * d * 4, hipMemcpyDefault, e->es));
  HIPC(hipStreamSynchronize(e->es));
  (which == 0 ? e->pe_ar_set : e->pe_nar_set) = true;
  return VX_OK;
}

template <typename T> static const T* W(vx_engine* e, const std::string& k) { return (const T*)e->w.at(k).p; }

static void fill_layers(vx_engine* e, const std::string& pre, int L, bool adaptive, std::vector<LayerW>& out) {
  out.resize(L);
  for (int i = 0; i < L; ++i) {
    const std::string p = pre + ".layers." + std::to_string(i);
    LayerW& l = out[i];
    l.in_w = W<void>(e, p + ".self_attn.in_proj_weight"); l.in_b = W<float>(e, p + ".self_attn.in_proj_bias");
    l.out_w = W<void>(e, p + ".self_attn.out_proj.weight"); l.out_b = W<float>(e, p + ".self_attn.out_proj.bias");
    l.w1 = W<void>(e, p + ".linear1.weight"); l.b1 = W<float>(e, p + ".linear1.bias");
    l.w2 = W<void>(e, p + ".linear2.weight"); l.b2 = W<float>(e, p + ".linear2.bias");
    { const Tensor &a = e->w.at(p + ".self_attn.in_proj_weight"), &b = e->w.at(p + ".linear1.weight"), &c2 = e->w.at(p + ".linear2.weight");
      l.in_q8 = a.q8; l.in_s8 = a.s8; l.w1_q8 = b.q8; l.w1_s8 = b.s8; l.w2_q8 = c2.q8; l.w2_s8 = c2.s8; }
    const std::string s = adaptive ? ".norm" : "";
    l.n1_g = W<float>(e, p + ".norm1" + s + ".weight"); l.n1_b = W<float>(e, p + ".norm1" + s + ".bias");
    l.n2_g = W<float>(e, p + ".norm2" + s + ".weight"); l.n2_b = W<float>(e, p + ".norm2" + s + ".bias");
    if (e->vallf) {
      l.cin_w = W<void>(e, p + ".multihead_attn.in_proj_weight"); l.cin_b = W<float>(e, p + ".multihead_attn.in_proj_bias");
      l.cout_w = W<void>(e, p + ".multihead_attn.out_proj.weight"); l.cout_b = W<float>(e, p + ".multihead_attn.out_proj.bias");
      l.n3_g = W<float>(e, p + ".norm3" + s + ".weight"); l.n3_b = W<float>(e, p + ".norm3" + s + ".bias");
    }
  }
}

// AdaLN vectors: index (stage, site) -> 2*dn floats [w | b]; site = npl*layer + {0 .. npl-1} (npl = 2 norms per encoder layer,
// 3 per VALL-F decoder layer), last = final norm
static float* ada_vec(vx_engine* e, int stage, int site) {
  const int sites = e->npl * e->cfg.nar_num_layers + 1;
  return e->ada + ((size_t)stage * sites + site) * 2 * e->cfg.nar_d_model;
}

extern "C" int vx_finalize_weights(vx_engine* e) {
  if (!e) return fail(VX_ERR_ARG, "null engine");
  ON_DEVICE(e->cfg.device);
  for (auto& k : e->keys)
    if (!e->w[k].set) return fail(VX_ERR_WEIGHTS, "missing key '%s' (strict load)", k.c_str());
  const vx_config& c = e->cfg;
  fill_layers(e, "ar_decoder", c.num_layers, false, e->ar_l);
  if (e->tp) {  // the sharded word order of the out-projection and linear2 (ar_tp.hpp)
    for (int li = 0; li < c.num_layers; ++li) {
      const LayerW& l = e->ar_l[li];
      const unsigned g1 = (unsigned)((size_t)TP_D * TP_D * e->esz / 16 / 256), g2 = (unsigned)((size_t)TP_D * TP_FF * e->esz / 16 / 256);
      if (e->bf16) {
        tp_repack_kernel<bf16><<<g1, 256, 0, e->es>>>((const bf16*)l.out_w, (uint4*)e->tp_wo[li], TP_D);
        tp_repack_kernel<bf16><<<g2, 256, 0, e->es>>>((const bf16*)l.w2, (uint4*)e->tp_w2[li], TP_FF);
      } else {
        tp_repack_kernel<float><<<g1, 256, 0, e->es>>>((const float*)l.out_w, (uint4*)e->tp_wo[li], TP_D);
        tp_repack_kernel<float><<<g2, 256, 0, e->es>>>((const float*)l.w2, (uint4*)e->tp_w2[li], TP_FF);
      }
    }
    // {gamma, beta, bias arriving with the partials} of every norm site: 2 li = LN1 (bias = previous linear2's; none at layer 0),
    // 2 li + 1 = LN2 (bias = the out-projection's), 2 L = the final norm (bias = the last linear2's)
    auto site = [&](int idx, const float* g, const float* b, const float* bias) -> int {
      float* dst = e->tp_gbb + (size_t)idx * 3 * TP_D;
      HIPC(hipMemcpyAsync(dst, g, TP_D * 4, hipMemcpyDeviceToDevice, e->es));
      HIPC(hipMemcpyAsync(dst + TP_D, b, TP_D * 4, hipMemcpyDeviceToDevice, e->es));
      if (bias) HIPC(hipMemcpyAsync(dst + 2 * TP_D, bias, TP_D * 4, hipMemcpyDeviceToDevice, e->es));
      return VX_OK;
    };
    for (int li = 0; li < c.num_layers; ++li) {
      const LayerW& l = e->ar_l[li];
      VXC(site(2 * li, l.n1_g, l.n1_b, li ? e->ar_l[li - 1].b2 : nullptr));
      VXC(site(2 * li + 1, l.n2_g, l.n2_b, l.out_b));
    }
    VXC(site(2 * c.num_layers, W<float>(e, "ar_decoder.norm.weight"), W<float>(e, "ar_decoder.norm.bias"), e->ar_l.back().b2));
    HIPC(hipGetLastError());
  }
  std::vector<float> t;
  if (!e->pe_ar_set) {
    host_sine_table(t, e->pe_rows, c.d_model);
    HIPC(hipMemcpy(e->pe_ar, t.data(), t.size() * 4, hipMemcpyHostToDevice));
  }
  if (c.num_quantizers > 1) {
    if (!e->pe_nar_set) {
      host_sine_table(t, e->pe_rows, c.nar_d_model);
      HIPC(hipMemcpy(e->pe_nar, t.data(), t.size() * 4, hipMemcpyHostToDevice));
    }
    fill_layers(e, "nar_decoder", c.nar_num_layers, true, e->nar_l);
    const int dn = c.nar_d_model, Ln = c.nar_num_layers;
    for (int s = 0; s < c.num_quantizers - 1; ++s) {
      const float* emb = W<float>(e, "nar_stage_embeddings." + std::to_string(s) + ".word_embeddings.weight");
      const int npl = e->npl;
      const int nsite = npl * Ln + ((c.flags & VX_FLAG_POST_NORM) ? 0 : 1);  // post-norm: no final AdaLN
      for (int site = 0; site < nsite; ++site) {
        std::string p = site == npl * Ln ? std::string("nar_decoder.norm")
                                         : "nar_decoder.layers." + std::to_string(site / npl) + ".norm" + std::to_string(site % npl + 1);
        project_vec_kernel<<<(2 * dn + 3) / 4, 256, 0, e->es>>>(W<float>(e, p + ".project_layer.weight"),
                                                                W<float>(e, p + ".project_layer.bias"), emb,
                                                                ada_vec(e, s, site), 2 * dn, dn);
      }
    }
    HIPC(hipGetLastError());
  }
  if (c.flags & VX_FLAG_PRENET) {
    for (int which = 0; which < (c.num_quantizers > 1 ? 2 : 1); ++which) {
      const int dd = which ? c.nar_d_model : c.d_model;
      const size_t nel = (size_t)5 * dd * dd;
      for (int i = 0; i < 3; ++i)
        conv_weight_relayout_kernel<<<(unsigned)((nel + 255) / 256), 256, 0, e->es>>>(
            W<float>(e, std::string(which ? "nar" : "ar") + "_text_prenet." + std::to_string(1 + 4 * i) + ".weight"), e->convT[which][i], dd);
    }
    HIPC(hipGetLastError());
  }
  HIPC(hipStreamSynchronize(e->es));
  e->finalized = true;
  return VX_OK;
}

// ------------------------------------------------------------------------------ launch helpers
template <typename WT, int KCH, int RPW, int PRO>
static void launch_gemv_inst(const GemvArgs& a, int grid, hipStream_t s) {
  // leading arguments = what the kernel loads from first (kernarg preload, ar_kernels.hpp)
  const float* xin = PRO == PRO_ATTN ? a.part : a.x;
  const unsigned nk = ((unsigned)a.N << 16) | (unsigned)a.K;  // N, K < 65536 (checked by the caller: K <= 4096, N <= 4 d)
  if (a.kid >= 0) grid += VX_KSTAMP_EXTRA;  // probe builds: one extra workgroup that only records the time (common.hpp)
  if constexpr (std::is_same<WT, bf16>::value) {
    if (a.nt && a.pf != nullptr) { gemv_kernel<WT, KCH, RPW, PRO, 8, true><<<grid, 256, 0, s>>>(a.W, xin, a.gamma, a.beta, nk, a); return; }
  }
  if (a.pf != nullptr) gemv_kernel<WT, KCH, RPW, PRO, 8><<<grid, 256, 0, s>>>(a.W, xin, a.gamma, a.beta, nk, a);
  else gemv_kernel<WT, KCH, RPW, PRO, 0><<<grid, 256, 0, s>>>(a.W, xin, a.gamma, a.beta, nk, a);
}

// The instance for (N, K): KCH = 16-byte chunks per lane per row, RPW = rows per wave so that one
// wave-iteration covers N over 4*grid waves, with at most 16 weight registers-quads in flight.
struct GemvPlan { int grid, kch, rpw; };
static GemvPlan gemv_plan(int N, int K, int vec, int num_cu) {
  const int need_kch = (K + 64 * vec - 1) / (64 * vec);
  // A/B switches (tests/probes/ar_ab.sh): VX_AR_GRID_MULT = workgroups per CU of the decode GEMVs (default 1),
  // VX_AR_HEAD_WGS = workgroups of the 1025-row head GEMV (default: one per CU)
  static const int mult = [] { const char* v = getenv("VX_AR_GRID_MULT"); const int m = v ? atoi(v) : 1; return m >= 1 && m <= 4 ? m : 1; }();
  static const int head_wgs = [] { const char* v = getenv("VX_AR_HEAD_WGS"); return v ? atoi(v) : 65; }();
  // VX_AR_WGS="NxK:wgs,NxK:wgs,...": workgroups of the GEMV with that shape (A/B runs), e.g. "1024x1024:64,1024x4096:128"
  static const std::vector<std::array<int, 3>> wgs_tab = [] {
    std::vector<std::array<int, 3>> t;
    const char* v = getenv("VX_AR_WGS");
    while (v && *v) {
      int n = 0, k = 0, w = 0, used = 0;
      if (sscanf(v, "%dx%d:%d%n", &n, &k, &w, &used) == 3 && w > 0) t.push_back({n, k, w});
      else break;
      v += used;
      if (*v == ',') ++v;
    }
    return t;
  }();
  GemvPlan p{num_cu * mult, 1, 1};
  // the 1025-row head streams 2 MB: 65 workgroups of 4 waves x 4 rows (one pass) finish sooner than one workgroup per CU with one or
  // two rows per wave (A/B on one box, alternating processes: 230.6 vs 236.7 us per step; 129: 234; profiles/r02_notes.md)
  if (N == AR_VOCAB && head_wgs > 0) p.grid = head_wgs;
  for (const auto& t : wgs_tab)
    if (t[0] == N && t[1] == K) p.grid = t[2];
  while (p.kch < need_kch) p.kch <<= 1;
  if ((N + 3) / 4 < p.grid) p.grid = (N + 3) / 4;
  const int need_rpw = (N + p.grid * 4 - 1) / (p.grid * 4);
  p.rpw = need_rpw > 3 ? 4 : need_rpw;  // 3 rows per wave for N = 3 x 4 x grid (the QKV projection): every wave busy
  while (p.rpw * p.kch > 16 && p.rpw > 1) p.rpw >>= 1;
  return p;
}
// Points `a` at the weights the NEXT GEMV of the decode step streams (GemvArgs.pf): 32 KB per workgroup at most.
static void gemv_prefetch(GemvArgs& a, const void* Wn, int Nn, int Kn, bool bf, int num_cu) {
  const GemvPlan p = gemv_plan(Nn, Kn, bf ? 8 : 4, num_cu);
  const size_t slice = (size_t)4 * p.rpw * Kn * (bf ? 2 : 4), total = (size_t)Nn * Kn * (bf ? 2 : 4);
  if (Wn == nullptr || slice > 32768 || total >= (1ull << 32) || total < 16) return;
  a.pf = Wn; a.pf_slice = (unsigned)slice; a.pf_total = (unsigned)total;
}

template <typename WT, int PRO> static int launch_gemv_p(const GemvArgs& a, int num_cu, hipStream_t s) {
  constexpr int VEC = Vec16<WT>::N;
  if (a.K % VEC || a.K > 4096 || a.K % 4) return fail(VX_ERR_UNSUPPORTED, "gemv: K=%d unsupported", a.K);
  if (a.N <= 0 || a.N > 65535) return fail(VX_ERR_UNSUPPORTED, "gemv: N=%d unsupported", a.N);  // (N << 16) | K travels as one argument
  const GemvPlan pl = gemv_plan(a.N, a.K, VEC, num_cu);
  const int kch = pl.kch, rpw = pl.rpw, grid = pl.grid;
  if (kch > 16 || (PRO != PRO_COPY && (kch > 4 || a.K > 1024))) return fail(VX_ERR_UNSUPPORTED, "gemv: K=%d too large for prologue %d", a.K, PRO);
#define GV(KC, RP) if (kch == KC && rpw == RP) { launch_gemv_inst<WT, KC, RP, PRO>(a, grid, s); return VX_OK; }
  GV(1, 1) GV(1, 2) GV(1, 3) GV(1, 4) GV(2, 1) GV(2, 2) GV(2, 3) GV(2, 4) GV(4, 1) GV(4, 2) GV(4, 3) GV(4, 4)
  if constexpr (PRO == PRO_COPY) { GV(8, 1) GV(8, 2) GV(16, 1) }
#undef GV
  return fail(VX_ERR_UNSUPPORTED, "gemv: no instance for kch=%d rpw=%d", kch, rpw);
}

template <typename WT> static int launch_gemv_t(const GemvArgs& a, int num_cu, hipStream_t s) {
  if (a.pro == PRO_LN) return launch_gemv_p<WT, PRO_LN>(a, num_cu, s);
  if (a.pro == PRO_ATTN) return launch_gemv_p<WT, PRO_ATTN>(a, num_cu, s);
  return launch_gemv_p<WT, PRO_COPY>(a, num_cu, s);
}

static int launch_gemv(bool bf, const GemvArgs& a, int num_cu, hipStream_t s) {
  return bf ? launch_gemv_t<bf16>(a, num_cu, s) : launch_gemv_t<float>(a, num_cu, s);
}

// ---- the XCD-sharded decode step (ar_tp.hpp): chosen at vx_create when the geometry is BASELINE's (d = 1024, 16 heads, pre-norm,
// no prenets, VALL-E) AND all 256 workgroups of its launches are resident at once - their hand-overs spin.  VX_AR_TP=0 keeps the
// five-launch step (A/B runs, probe builds).
static int tp_setup(vx_engine* e) {
  const vx_config& c = e->cfg;
  const int d = c.d_model, H = c.nhead;
  e->tp = false;
  const char* tv = getenv("VX_AR_TP");
  if ((tv && atoi(tv) == 0) || e->vallf || d != TP_D || H != TP_H || (c.flags & (VX_FLAG_POST_NORM | VX_FLAG_PRENET))) return VX_OK;
  int n1 = 0, n2 = 0;
  if (e->bf16) {
    HIPC(hipOccupancyMaxActiveBlocksPerMultiprocessor(&n1, (const void*)tp_attn_kernel<bf16, true>, 256, 0));
    HIPC(hipOccupancyMaxActiveBlocksPerMultiprocessor(&n2, (const void*)tp_ffn_kernel<bf16>, 256, 0));
  } else {
    HIPC(hipOccupancyMaxActiveBlocksPerMultiprocessor(&n1, (const void*)tp_attn_kernel<float, true>, 256, 0));
    HIPC(hipOccupancyMaxActiveBlocksPerMultiprocessor(&n2, (const void*)tp_ffn_kernel<float>, 256, 0));
  }
  if ((long long)(n1 < n2 ? n1 : n2) * e->num_cu < TP_X * TP_WG) return VX_OK;
  const size_t L = (size_t)c.num_layers;
  VXC(dalloc_t(e, &e->tp_xacc, (size_t)3 * TP_D));
  VXC(dalloc_t(e, &e->tp_gh, L * TP_X * TP_HID));
  VXC(dalloc_t(e, &e->tp_gbb, (2 * L + 1) * 3 * TP_D));
  VXC(dalloc_t(e, &e->fq_gq, L * H * FQ_QKV));
  VXC(dalloc_t(e, &e->fq_gp, L * H * FQ_G * FQ_PART));
  HIPC(hipMemset(e->tp_gbb, 0, (2 * L + 1) * 3 * TP_D * 4));
  HIPC(hipMemset(e->tp_xacc, 0, (size_t)3 * TP_D * sizeof(long long)));
  HIPC(hipMemset(e->tp_gh, 0, L * TP_X * TP_HID * sizeof(fq_gran)));  // zero tags: never equal to a step counter (starts at 1)
  HIPC(hipMemset(e->fq_gq, 0, L * H * FQ_QKV * sizeof(fq_gran)));
  HIPC(hipMemset(e->fq_gp, 0, L * H * FQ_G * FQ_PART * sizeof(fq_gran)));
  e->tp_wo.assign(L, nullptr); e->tp_w2.assign(L, nullptr);
  for (size_t li = 0; li < L; ++li) {
    VXC(dalloc(e, &e->tp_wo[li], (size_t)TP_D * TP_D * e->esz));
    VXC(dalloc(e, &e->tp_w2[li], (size_t)TP_D * TP_FF * e->esz));
  }
  e->tp = true;
  return VX_OK;
}

template <typename T>
static int gemm_rows_t(bool mfma, const T* A, const T* Wt, const float* bias, void* C, int M, int N, int K, int epi,
                       bool out_f32, hipStream_t s, void* vt = nullptr, int vt_n0 = 0, int vt_ld = 0) {
  if constexpr (std::is_same<T, bf16>::value) {
    if (mfma) return mfma_gemm_dispatch(A, Wt, bias, C, M, N, K, epi, out_f32, s, (bf16*)vt, vt_n0, vt_ld);
  }
  dim3 grid((N + 63) / 64, (M + 63) / 64);
  if (epi == GE_RESID) gemm_simple_kernel<T, float, GE_RESID><<<grid, 256, 0, s>>>(A, Wt, bias, (float*)C, M, N, K);
  else if (epi == GE_PLAIN) gemm_simple_kernel<T, float, GE_PLAIN><<<grid, 256, 0, s>>>(A, Wt, bias, (float*)C, M, N, K);
  else if (epi == GE_BIAS && out_f32) gemm_simple_kernel<T, float, GE_BIAS><<<grid, 256, 0, s>>>(A, Wt, bias, (float*)C, M, N, K);
  else if (epi == GE_RELU && out_f32) gemm_simple_kernel<T, float, GE_RELU><<<grid, 256, 0, s>>>(A, Wt, bias, (float*)C, M, N, K);
  else if (epi == GE_BIAS) gemm_simple_kernel<T, T, GE_BIAS><<<grid, 256, 0, s>>>(A, Wt, bias, (T*)C, M, N, K);
  else gemm_simple_kernel<T, T, GE_RELU><<<grid, 256, 0, s>>>(A, Wt, bias, (T*)C, M, N, K);
  return VX_OK;
}

static bool use_mfma(const vx_engine* e) { return e->bf16 && e->hd64 && !(e->cfg.flags & VX_FLAG_SIMPLE_ROWS); }

static int gemm_rows(vx_engine* e, const void* A, const void* Wt, const float* bias, void* C, int M, int N, int K,
                     int epi, bool out_f32, bool emit_vt = false) {
  if (e->bf16)
    return gemm_rows_t<bf16>(use_mfma(e), (const bf16*)A, (const bf16*)Wt, bias, C, M, N, K, epi, out_f32, e->es,
                             emit_vt ? e->VT : nullptr, 2 * (N / 3), e->vt_ld);
  return gemm_rows_t<float>(false, (const float*)A, (const float*)Wt, bias, C, M, N, K, epi, out_f32, e->es);
}

// fold: the split-K slabs (and bias) of the GEMM in front of this norm, added to x first (rows_kernels.hpp)
struct Fold { const float* part = nullptr; int nsplit = 0; size_t stride = 0; const float* bias = nullptr; };
static int ln_rows(vx_engine* e, const float* x, const float* g, const float* b, const float* aw, const float* ab,
                   void* out, int rows, int d, float* xout = nullptr, Fold f = Fold()) {
  // d <= 1024 in the engine (vx_create): 4 float4 per lane
  if (e->bf16) layernorm_rows_kernel<bf16, 4><<<(rows + 3) / 4, 256, 0, e->es>>>(x, g, b, aw, ab, (bf16*)out, rows, d, xout, f.part, f.nsplit, f.stride, f.bias);
  else layernorm_rows_kernel<float, 4><<<(rows + 3) / 4, 256, 0, e->es>>>(x, g, b, aw, ab, (float*)out, rows, d, xout, f.part, f.nsplit, f.stride, f.bias);
  return VX_OK;
}
// fp32 linear layer on rows (prenets keep fp32 weights in every precision mode): C = [relu](A W^T + b)
static int linear_rows_f32(vx_engine* e, const float* A, const float* Wt, const float* bias, float* Cm, int M, int N, int K, bool relu) {
  return gemm_rows_t<float>(false, A, Wt, bias, Cm, M, N, K, relu ? GE_RELU : GE_BIAS, true, e->es);
}
// text prenet (valle.py:97-113): `in` (S, dd) raw embeddings -> `out` (S, dd); uses pn_a / pn_b as scratch (in/out may be them)
static int text_prenet_rows(vx_engine* e, int which, const float* in, float* out, int S, int dd) {
  const std::string pre = std::string(which ? "nar" : "ar") + "_text_prenet.";
  const float* src = in;
  float* bufs[2] = {e->pn_b, e->pn_a};
  for (int i = 0; i < 3; ++i) {
    const std::string cv = pre + std::to_string(1 + 4 * i), bn = pre + std::to_string(2 + 4 * i);
    float* dst = bufs[i & 1];
    conv5_bn_relu_kernel<<<(S + CONV_TT - 1) / CONV_TT, 256, (size_t)(CONV_TT + 4) * dd * 4, e->es>>>(
        src, e->convT[which][i], W<float>(e, cv + ".bias"), W<float>(e, bn + ".weight"), W<float>(e, bn + ".bias"),
        W<float>(e, bn + ".running_mean"), W<float>(e, bn + ".running_var"), dst, S, dd);
    src = dst;
  }
  // src == pn_b after three convolutions
  return linear_rows_f32(e, src, W<float>(e, pre + "14.weight"), W<float>(e, pre + "14.bias"), out, S, dd, dd, false);
}
// audio prenet (valle.py:115-123): `in` (rows, dd) -> `out` (rows, dd), hidden rows in pn_h1 / pn_h2
static int audio_prenet_rows(vx_engine* e, int which, const float* in, float* out, int rows, int dd) {
  const std::string pre = std::string(which ? "nar" : "ar") + "_audio_prenet.";
  VXC(linear_rows_f32(e, in, W<float>(e, pre + "0.weight"), W<float>(e, pre + "0.bias"), e->pn_h1, rows, PRENET_H, dd, true));
  VXC(linear_rows_f32(e, e->pn_h1, W<float>(e, pre + "3.weight"), W<float>(e, pre + "3.bias"), e->pn_h2, rows, PRENET_H, PRENET_H, true));
  return linear_rows_f32(e, e->pn_h2, W<float>(e, pre + "6.weight"), W<float>(e, pre + "6.bias"), out, rows, dd, PRENET_H, false);
}

// rows of the fp32 residual stream -> the GEMM operand type, no normalisation (input of a post-norm stack)
static int cast_rows(vx_engine* e, const float* x, void* out, size_t n) {
  if (e->bf16) convert_kernel<bf16><<<1024, 256, 0, e->es>>>(x, (bf16*)out, n);
  else convert_kernel<float><<<1024, 256, 0, e->es>>>(x, (float*)out, n);
  return VX_OK;
}

static int attn_rows(vx_engine* e, const void* qkv, void* out, int M, int d, int H, int text_len) {
  const int hd = d / H;
  const float scale = 1.0f / sqrtf((float)hd);
  if (e->nseg > 0 || use_mfma(e)) {
    const int rc = e->nseg > 0  // batched NAR: one launch over all segments of the concatenated rows
        ? mfma_attn_dispatch((const bf16*)qkv, (const bf16*)e->VT, e->vt_ld, (bf16*)out, M, d, H, text_len, e->es,
                             e->d_seg_start, e->d_seg_len, e->nseg, e->max_seg_len, e->seg_text_on ? e->d_seg_text : nullptr)
        : mfma_attn_dispatch((const bf16*)qkv, (const bf16*)e->VT, e->vt_ld, (bf16*)out, M, d, H, text_len, e->es);
    return rc == 0 ? VX_OK : fail(VX_ERR_UNSUPPORTED, "attention: a sequence's q/k/v rows or the V^T buffer exceed 4 GB (rows %d, d %d)", M, d);
  }
  dim3 grid((M + 63) / 64, H);
#define AR(HDV)                                                                                                                   \
  if (hd == HDV) {                                                                                                                \
    if (e->bf16) attn_rows_simple_kernel<bf16, HDV><<<grid, 256, 0, e->es>>>((const bf16*)qkv, (bf16*)out, M, d, text_len, scale); \
    else attn_rows_simple_kernel<float, HDV><<<grid, 256, 0, e->es>>>((const float*)qkv, (float*)out, M, d, text_len, scale);      \
    return VX_OK;                                                                                                                 \
  }
  AR(64) AR(32) AR(16) AR(8) AR(4)
#undef AR
  return fail(VX_ERR_UNSUPPORTED, "attention: head_dim %d", hd);
}

// One encoder stack over M rows held in e->X (valle.py:1035-1038 / 1125-1127).  `ada_stage` < 0:
// plain LayerNorm (AR); otherwise the stage's AdaLN vectors.  If kv_layer0 is non-null the K/V
// rows are also scattered into the decode cache.
static int split_for(int K) {  // K slices of the split-K GEMMs: a multiple of the 64-wide K tile each
  for (int sp = 4; sp > 1; sp >>= 1)
    if (K % (64 * sp) == 0) return sp;
  return 1;
}

static bool use_mfma(const vx_engine* e);
static bool time_gemms() {
  const char* v = getenv("VX_TIME_GEMMS");  // read per call: bench.py turns it on for ONE extra, untimed pass
  const bool on = v && atoi(v) != 0;
  return on;
}
// event before / after a GEMM launch on the engine stream (no-op unless VX_TIME_GEMMS=1)
static void gemm_mark(vx_engine* e, double flops) {
  if (!time_gemms()) return;
  if (e->gemm_ev_used == e->gemm_ev.size()) { hipEvent_t ev; if (hipEventCreate(&ev) != hipSuccess) return; e->gemm_ev.push_back(ev); }
  (void)hipEventRecord(e->gemm_ev[e->gemm_ev_used++], e->es);
  e->gemm_flops += flops;  // counted at the closing mark (flops > 0)
}
// after the stream has been synchronised: sum the pairs' elapsed times
static void gemm_collect(vx_engine* e) {
  e->t_gemm = 0;
  for (size_t i = 0; i + 1 < e->gemm_ev_used; i += 2) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, e->gemm_ev[i], e->gemm_ev[i + 1]) == hipSuccess) e->t_gemm += ms;
  }
  e->gemm_flops_done = e->gemm_flops;
  e->gemm_ev_used = 0; e->gemm_flops = 0;
}
// VX_PREC_FP8_NAR: do this stage's QKV / FFN GEMMs run on MXFP8?  (NAR stages only, pre-norm, at or above the row threshold;
// VX_MX_MIN_ROWS lowers it: the parity tests run one utterance's 1025 rows through the MXFP8 kernels)
static bool mx_on(const vx_engine* e, int ada_stage, int M, int d) {
  const char* v = getenv("VX_MX_MIN_ROWS");
  return e->fp8nar && ada_stage >= 0 && M >= (v ? atoi(v) : 4096) && !(e->cfg.flags & VX_FLAG_POST_NORM) && use_mfma(e) && d % 256 == 0;
}

static int run_stack(vx_engine* e, const std::vector<LayerW>& layers, int M, int d, int H, int text_len, int ada_stage,
                     bool fill_cache, char* kv_base = nullptr) {
  if (kv_base == nullptr) kv_base = (char*)e->kv;
  const bool post = e->cfg.flags & VX_FLAG_POST_NORM;
  const int hd = d / H;
  const size_t kv_layer = (size_t)2 * H * e->ctx_max * hd * e->esz;
  // The two N = d GEMMs of a layer (out-projection, FFN2) at M ~ 1k rows: 128^2 tiles alone are 72 workgroups, so K is
  // split over 2-4 workgroups per tile, every slice writes an fp32 slab, and the LayerNorm that follows the GEMM anyway
  // adds bias + slabs to x in a fixed order.  Larger M (batched rows) has enough tiles and adds in the GEMM epilogue.
  const bool splitk = use_mfma(e) && M < 4096 && M <= e->slab_rows && d % 128 == 0 && d >= 128 && !mx_on(e, ada_stage, M, d);
  const bool mx = mx_on(e, ada_stage, M, d);
  const bool tg = time_gemms() && ada_stage >= 0;  // NAR stages only
  const size_t sstride = (size_t)M * d;
  // VX_SPLIT_D / VX_SPLIT_FF (A/B runs): K slices of the out-projection / FFN2 (1 = no slabs, residual add in the GEMM epilogue)
  static const int env_sp_d = getenv("VX_SPLIT_D") ? atoi(getenv("VX_SPLIT_D")) : 0;
  static const int env_sp_ff = getenv("VX_SPLIT_FF") ? atoi(getenv("VX_SPLIT_FF")) : 0;
  auto pick = [](int env, int K, int dflt) { return (env == 1 || env == 2 || env == 4) && K % (64 * env) == 0 ? env : dflt; };
  // default: the largest of 4 / 2 / 1 slices that keeps (128^2 tiles) x slices within one round of the chip's CUs - at 1025 rows
  // 72 tiles x 4 slices were 288 workgroups, i.e. two rounds, and four slabs for the next LayerNorm to fold (A/B on one box:
  // NAR 7 stages 9.38 ms with 4 / 4 slices, 8.97 ms with 2 / 2; the 272-row prefill is fastest with 4 / 4: 0.91 vs 0.99 ms)
  auto fit = [&](int K) {
    const long long tiles = (long long)((M + 127) / 128) * (d / 128);
    for (int sp = split_for(K); sp > 1; sp >>= 1)
      if (tiles * sp <= e->num_cu) return sp;
    return 1;
  };
  const int sp_d = pick(env_sp_d, d, fit(d)), sp_ff = pick(env_sp_ff, 4 * d, fit(4 * d));
  Fold pend;  // FFN2 slabs of the previous layer, folded by the next norm (pre-norm) or by the trailing fold pass
  for (size_t li = 0; li < layers.size(); ++li) {
    const LayerW& l = layers[li];
    const float *aw1 = nullptr, *ab1 = nullptr, *aw2 = nullptr, *ab2 = nullptr;
    if (ada_stage >= 0) {
      aw1 = ada_vec(e, ada_stage, 2 * (int)li); ab1 = aw1 + d;
      aw2 = ada_vec(e, ada_stage, 2 * (int)li + 1); ab2 = aw2 + d;
    }
    // pre-norm (transformer.py:296-302): Hn = norm1(x).  post-norm (303-308): the block reads x itself; Hn already
    // holds it in operand precision from the previous layer's norm2 (layer 0: cast here)
    if (mx) {  // MXFP8 QKV: the LayerNorm quantises its own row, the GEMM writes bf16 q/k/v (+ V^T) for the bf16 attention
      layernorm_rows_mx_kernel<4><<<(M + 3) / 4, 256, 0, e->es>>>(e->X, l.n1_g, l.n1_b, aw1, ab1, e->Hn8, e->SHn, M, d, e->mx_ld);
      if (tg) gemm_mark(e, 0);
      if (mx_gemm_dispatch(e->Hn8, e->SHn, e->mx_ld, l.in_q8, l.in_s8, 3 * d, l.in_b, e->QKV, nullptr, 0, M, 3 * d, d, GE_BIAS,
                           MX_OUT_BF16, e->es, (bf16*)e->VT, 2 * d, e->vt_ld))
        return fail(VX_ERR_UNSUPPORTED, "mx gemm: shape %d x %d x %d", M, 3 * d, d);
      if (tg) gemm_mark(e, 2.0 * M * 3 * d * d);
    } else {
    if (!post) { VXC(ln_rows(e, e->X, l.n1_g, l.n1_b, aw1, ab1, e->Hn, M, d, nullptr, pend)); pend = Fold(); }
    else if (li == 0) VXC(cast_rows(e, e->X, e->Hn, (size_t)M * d));
    if (tg) gemm_mark(e, 0);
    VXC(gemm_rows(e, e->Hn, l.in_w, l.in_b, e->QKV, M, 3 * d, d, GE_BIAS, false, use_mfma(e)));
    if (tg) gemm_mark(e, 2.0 * M * 3 * d * d);
    }
    if (fill_cache && e->nseg > 0) {  // batched prefill: segment z -> slot z
      const size_t kvl = (size_t)2 * H * e->ctx_max * 64;  // elements per layer
      kv_scatter_seg_kernel<bf16><<<dim3(e->max_seg_len, e->nseg), 256, 0, e->es>>>(
          (const bf16*)e->QKV, e->bkv + li * kvl, e->bkv_slot, kvl / 2, e->d_seg_start, e->d_seg_len, d, 64, e->ctx_max);
    } else if (fill_cache) {
      char* kc = kv_base + li * kv_layer;
      char* vc = kc + kv_layer / 2;
      if (e->bf16) kv_scatter_kernel<bf16><<<M, 256, 0, e->es>>>((const bf16*)e->QKV, (bf16*)kc, (bf16*)vc, M, d, hd, e->ctx_max);
      else kv_scatter_kernel<float><<<M, 256, 0, e->es>>>((const float*)e->QKV, (float*)kc, (float*)vc, M, d, hd, e->ctx_max);
    }
    VXC(attn_rows(e, e->QKV, e->ATT, M, d, H, text_len));
    Fold fo;  // out-projection: x += out_proj(attn), folded into the norm that follows when split
    if (tg && !mx) gemm_mark(e, 0);  // (MXFP8 stages: only the fp8 GEMMs are timed; the out-projection stays bf16)
    if (splitk && sp_d > 1) {
      VXC(mfma_gemm_partial((const bf16*)e->ATT, (const bf16*)l.out_w, e->slab, M, d, d, sp_d, e->es));
      fo.part = e->slab; fo.nsplit = sp_d; fo.stride = sstride; fo.bias = l.out_b;
    } else {
      VXC(gemm_rows(e, e->ATT, l.out_w, l.out_b, e->X, M, d, d, GE_RESID, true));
    }
    if (tg && !mx) gemm_mark(e, 2.0 * M * d * d);
    if (mx) {  // MXFP8 FFN: LN2 -> fp8, FFN1's epilogue quantises its ReLU output per 32-wide block, FFN2 adds into x
      layernorm_rows_mx_kernel<4><<<(M + 3) / 4, 256, 0, e->es>>>(e->X, l.n2_g, l.n2_b, aw2, ab2, e->Hn8, e->SHn, M, d, e->mx_ld);
      if (tg) gemm_mark(e, 0);
      if (mx_gemm_dispatch(e->Hn8, e->SHn, e->mx_ld, l.w1_q8, l.w1_s8, 4 * d, l.b1, e->FF8, e->SFF, e->mx_ld, M, 4 * d, d, GE_RELU,
                           MX_OUT_MX, e->es) ||
          mx_gemm_dispatch(e->FF8, e->SFF, e->mx_ld, l.w2_q8, l.w2_s8, d, l.b2, e->X, nullptr, 0, M, d, 4 * d, GE_RESID, MX_OUT_F32, e->es))
        return fail(VX_ERR_UNSUPPORTED, "mx gemm: FFN shapes at M=%d d=%d", M, d);
      if (tg) gemm_mark(e, 2.0 * 2.0 * M * 4 * d * d);
      continue;
    }
    if (!post) VXC(ln_rows(e, e->X, l.n2_g, l.n2_b, aw2, ab2, e->Hn, M, d, nullptr, fo));
    else VXC(ln_rows(e, e->X, l.n1_g, l.n1_b, aw1, ab1, e->Hn, M, d, e->X, fo));  // x = norm1(x + sa(x))
    if (tg) gemm_mark(e, 0);
    VXC(gemm_rows(e, e->Hn, l.w1, l.b1, e->FF, M, 4 * d, d, GE_RELU, false));
    Fold ff;
    if (splitk && sp_ff > 1) {
      VXC(mfma_gemm_partial((const bf16*)e->FF, (const bf16*)l.w2, e->slab, M, d, 4 * d, sp_ff, e->es));
      ff.part = e->slab; ff.nsplit = sp_ff; ff.stride = sstride; ff.bias = l.b2;
    } else {
      VXC(gemm_rows(e, e->FF, l.w2, l.b2, e->X, M, d, 4 * d, GE_RESID, true));
    }
    if (tg) gemm_mark(e, 2.0 * 2.0 * M * 4 * d * d);
    if (post) VXC(ln_rows(e, e->X, l.n2_g, l.n2_b, aw2, ab2, e->Hn, M, d, e->X, ff));  // x = norm2(x + ff(x))
    else pend = ff;
  }
  if (pend.part != nullptr)  // the last layer's FFN2 slabs: x += b2 + slabs (no norm: the caller applies the final one)
    VXC(ln_rows(e, e->X, nullptr, nullptr, nullptr, nullptr, nullptr, M, d, nullptr, pend));
  HIPC(hipGetLastError());
  return VX_OK;
}

// ---- VALL-F (valle.py:49-719): stacks of TransformerDecoderLayers (modules/transformer.py:409-601) -------------------------
// Text memory -> per-layer K / V in the decode-cache layout (nhead, max_text, hd).  `mem` = S rows of the embedded text in
// operand precision (e->Hn).  The packed cross in_proj is applied whole (the q third of the result is unused): every output
// element is its own dot product, so rows [d, 3d) equal linear(mem, w[d:], b[d:]) of torch's _in_projection_packed.
static int memory_kv(vx_engine* e, const std::vector<LayerW>& layers, void* xkv, int S, int d, int H) {
  const int hd = d / H;
  const size_t per_layer = (size_t)2 * d * e->cfg.max_text * e->esz;
  for (size_t li = 0; li < layers.size(); ++li) {
    VXC(gemm_rows(e, e->Hn, layers[li].cin_w, layers[li].cin_b, e->QKV, S, 3 * d, d, GE_BIAS, false, false));
    char* kc = (char*)xkv + li * per_layer;
    char* vc = kc + per_layer / 2;
    if (e->bf16) kv_scatter_kernel<bf16><<<S, 256, 0, e->es>>>((const bf16*)e->QKV, (bf16*)kc, (bf16*)vc, S, d, hd, e->cfg.max_text);
    else kv_scatter_kernel<float><<<S, 256, 0, e->es>>>((const float*)e->QKV, (float*)kc, (float*)vc, S, d, hd, e->cfg.max_text);
  }
  HIPC(hipGetLastError());
  return VX_OK;
}

static int cross_attn_rows(vx_engine* e, const void* q, const void* kc, const void* vc, void* out, int M, int d, int H, int Sk) {
  const int hd = d / H;
  const float scale = 1.0f / sqrtf((float)hd);
  dim3 grid((M + 63) / 64, H);
#define CA(HDV)                                                                                                                  \
  if (hd == HDV) {                                                                                                               \
    if (e->bf16) cross_attn_rows_kernel<bf16, HDV><<<grid, 256, 0, e->es>>>((const bf16*)q, d, (const bf16*)kc, (const bf16*)vc, \
                                                                            e->cfg.max_text, (bf16*)out, d, M, Sk, scale);       \
    else cross_attn_rows_kernel<float, HDV><<<grid, 256, 0, e->es>>>((const float*)q, d, (const float*)kc, (const float*)vc,     \
                                                                     e->cfg.max_text, (float*)out, d, M, Sk, scale);             \
    return VX_OK;                                                                                                                \
  }
  CA(64) CA(32) CA(16) CA(8) CA(4)
#undef CA
  return fail(VX_ERR_UNSUPPORTED, "cross-attention: head_dim %d", hd);
}

// One decoder stack over the M audio rows held in e->X (valle.py:626-632 AR with text_len = 0: causal; valle.py:682-688 NAR
// with text_len < 0: no mask).  Memory K / V of layer li at xkv + li * per_layer, Sk text rows.  Plain epilogue adds (no
// split-K slabs): this variant is built for parity, not tuned.
static int run_stack_f(vx_engine* e, const std::vector<LayerW>& layers, int M, int d, int H, int text_len, int ada_stage,
                       bool fill_cache, const void* xkv, int Sk) {
  const bool post = e->cfg.flags & VX_FLAG_POST_NORM;
  const int hd = d / H;
  const size_t kv_layer = (size_t)2 * H * e->ctx_max * hd * e->esz;
  const size_t per_layer = (size_t)2 * d * e->cfg.max_text * e->esz;
  for (size_t li = 0; li < layers.size(); ++li) {
    const LayerW& l = layers[li];
    const float *aw[3] = {nullptr, nullptr, nullptr}, *ab[3] = {nullptr, nullptr, nullptr};
    if (ada_stage >= 0)
      for (int k = 0; k < 3; ++k) { aw[k] = ada_vec(e, ada_stage, 3 * (int)li + k); ab[k] = aw[k] + d; }
    // self-attention: pre-norm x += sa(norm1(x)) (transformer.py:536-539); post-norm x = norm1(x + sa(x)) (547-550)
    if (!post) VXC(ln_rows(e, e->X, l.n1_g, l.n1_b, aw[0], ab[0], e->Hn, M, d));
    else if (li == 0) VXC(cast_rows(e, e->X, e->Hn, (size_t)M * d));
    VXC(gemm_rows(e, e->Hn, l.in_w, l.in_b, e->QKV, M, 3 * d, d, GE_BIAS, false, use_mfma(e)));
    if (fill_cache) {
      char* kc = (char*)e->kv + li * kv_layer;
      char* vc = kc + kv_layer / 2;
      if (e->bf16) kv_scatter_kernel<bf16><<<M, 256, 0, e->es>>>((const bf16*)e->QKV, (bf16*)kc, (bf16*)vc, M, d, hd, e->ctx_max);
      else kv_scatter_kernel<float><<<M, 256, 0, e->es>>>((const float*)e->QKV, (float*)kc, (float*)vc, M, d, hd, e->ctx_max);
    }
    VXC(attn_rows(e, e->QKV, e->ATT, M, d, H, text_len));
    VXC(gemm_rows(e, e->ATT, l.out_w, l.out_b, e->X, M, d, d, GE_RESID, true));
    // cross-attention over the text: x += mha(norm2(x), memory) (540-545); post-norm x = norm2(x + mha(x, memory)) (551-557)
    if (!post) VXC(ln_rows(e, e->X, l.n2_g, l.n2_b, aw[1], ab[1], e->Hn, M, d));
    else VXC(ln_rows(e, e->X, l.n1_g, l.n1_b, aw[0], ab[0], e->Hn, M, d, e->X));
    VXC(gemm_rows(e, e->Hn, l.cin_w, l.cin_b, e->QKV, M, d, d, GE_BIAS, false, false));  // q = rows [0, d) of the packed in_proj
    const char* xk = (const char*)xkv + li * per_layer;
    VXC(cross_attn_rows(e, e->QKV, xk, xk + per_layer / 2, e->ATT, M, d, H, Sk));
    VXC(gemm_rows(e, e->ATT, l.cout_w, l.cout_b, e->X, M, d, d, GE_RESID, true));
    // feed-forward: x += ff(norm3(x)) (546); post-norm x = norm3(x + ff(x)) (558)
    if (!post) VXC(ln_rows(e, e->X, l.n3_g, l.n3_b, aw[2], ab[2], e->Hn, M, d));
    else VXC(ln_rows(e, e->X, l.n2_g, l.n2_b, aw[1], ab[1], e->Hn, M, d, e->X));
    VXC(gemm_rows(e, e->Hn, l.w1, l.b1, e->FF, M, 4 * d, d, GE_RELU, false));
    VXC(gemm_rows(e, e->FF, l.w2, l.b2, e->X, M, d, 4 * d, GE_RESID, true));
    if (post) VXC(ln_rows(e, e->X, l.n3_g, l.n3_b, aw[2], ab[2], e->Hn, M, d, e->X));
  }
  HIPC(hipGetLastError());
  return VX_OK;
}

static int sync_in(vx_engine* e, void* stream) {
  hipStream_t cs = (hipStream_t)stream;
  if (cs == e->es) return VX_OK;
  HIPC(hipEventRecord(e->ev_in, cs));
  HIPC(hipStreamWaitEvent(e->es, e->ev_in, 0));
  return VX_OK;
}
static int sync_out(vx_engine* e, void* stream) {
  hipStream_t cs = (hipStream_t)stream;
  if (cs == e->es) return VX_OK;
  HIPC(hipEventRecord(e->ev_out, e->es));
  HIPC(hipStreamWaitEvent(cs, e->ev_out, 0));
  return VX_OK;
}

// ------------------------------------------------------------------------------ AR
// prefilled: x is a row of the prefill stack's output.  Pre-norm: the final LayerNorm is fused here (valle.py:1035-1039).
// Post-norm: there is no final norm; a prefill row is already norm2'd, the decode step's x still needs the last
// layer's norm2 (fused here instead).
static int enqueue_head(vx_engine* e, hipStream_t s, const float* x = nullptr, float* logits = nullptr,
                        const ArState* st = nullptr, bool prefilled = false, const void* pfW = nullptr, int pfN = 0, int pfK = 0) {
  const vx_config& c = e->cfg;
  const bool post = c.flags & VX_FLAG_POST_NORM;
  GemvArgs a{};
  a.W = W<void>(e, "ar_predict_layer.weight");
  a.bias = nullptr;
  a.x = x ? x : e->ar_x;
  // post-norm: the last layer's closing norm (norm2 of an encoder layer, norm3 of a VALL-F decoder layer)
  a.gamma = post ? (e->vallf ? e->ar_l.back().n3_g : e->ar_l.back().n2_g) : W<float>(e, "ar_decoder.norm.weight");
  a.beta = post ? (e->vallf ? e->ar_l.back().n3_b : e->ar_l.back().n2_b) : W<float>(e, "ar_decoder.norm.bias");
  a.y = logits ? logits : e->ar_logits;
  a.N = AR_VOCAB; a.K = c.d_model;
  a.pro = (post && prefilled) ? PRO_COPY : PRO_LN; a.epi = EPI_LOGITS;
  a.st = st ? st : e->d_st;
  a.kid = (!prefilled && st == nullptr) ? 61 : -1;  // the decode step's head (probe builds)
  a.nt = (pfW != nullptr && getenv("VX_AR_NT")) ? atoi(getenv("VX_AR_NT")) : 0;
  if (pfW) gemv_prefetch(a, pfW, pfN, pfK, e->bf16, e->num_cu);  // decode step: the next token's first GEMVs
  return launch_gemv(e->bf16, a, e->num_cu, s);
}

// Shared by vx_ar_prefill (slot < 0: the batch-1 buffers) and vx_batch_prefill (slot >= 0).
static int prefill_impl(vx_engine* e, int slot, const int64_t* text, int32_t S, const int64_t* prompt_cb0, int32_t P,
                        void* stream) {
  if (!e || !text || (!prompt_cb0 && P > 0)) return fail(VX_ERR_ARG, "null argument");
  if (!e->finalized) return fail(VX_ERR_STATE, "weights not finalized");
  if (S <= 0 || P < 0) return fail(VX_ERR_ARG, "S must be > 0 (valle.py:991), P >= 0");
  const vx_config& c = e->cfg;
  const bool vf = e->vallf;  // VALL-F: the stack runs over the audio rows only, the text is cross-attention memory (valle.py:598-632)
  const int bos = c.prepend_bos ? 1 : 0, A = bos + P, M = vf ? A : S + A, d = c.d_model;
  if (S > c.max_text || A + 1 > c.max_audio) return fail(VX_ERR_CAPACITY, "S=%d / P=%d exceed capacity", S, P);
  if (A == 0) return fail(VX_ERR_ARG, "empty audio prefix needs prepend_bos");
  if (slot >= e->bmax) return fail(VX_ERR_ARG, "slot %d >= max_batch %d", slot, e->bmax);
  ON_DEVICE(c.device);
  VXC(sync_in(e, stream));
  HIPC(hipEventRecord(e->ev_t[0], e->es));
  HIPC(hipMemcpyAsync(e->ids_text, text, (size_t)S * 8, hipMemcpyDefault, e->es));
  if (bos) {
    static const long long b = NUM_AUDIO_TOKENS + 1;  // valle.py:1006-1007
    HIPC(hipMemcpyAsync(e->ids_audio, &b, 8, hipMemcpyHostToDevice, e->es));
  }
  if (P) HIPC(hipMemcpyAsync(e->ids_audio + bos, prompt_cb0, (size_t)P * 8, hipMemcpyDefault, e->es));
  if (c.flags & VX_FLAG_PRENET) {  // embedding -> prenet -> position (valle.py:995-997, 1013-1015)
    embed_accum_kernel<<<S, 256, 0, e->es>>>(e->ids_text, 1, 0, W<float>(e, "ar_text_embedding.word_embeddings.weight"), 512, d, e->pn_a, S, 1);
    VXC(text_prenet_rows(e, 0, e->pn_a, e->pn_a, S, d));
    add_pos_kernel<<<S, 256, 0, e->es>>>(e->pn_a, d, W<float>(e, "ar_text_position.alpha"), e->pe_ar, 0, e->X, S);
    if (vf) { VXC(cast_rows(e, e->X, e->Hn, (size_t)S * d)); VXC(memory_kv(e, e->ar_l, e->xkv_ar, S, d, c.nhead)); }
    embed_accum_kernel<<<A, 256, 0, e->es>>>(e->ids_audio, 1, 0, W<float>(e, "ar_audio_embedding.word_embeddings.weight"), 1025 + bos, d, e->pn_a, A, 1);
    VXC(audio_prenet_rows(e, 0, e->pn_a, e->pn_b, A, d));
    add_pos_kernel<<<A, 256, 0, e->es>>>(e->pn_b, d, W<float>(e, "ar_audio_position.alpha"), e->pe_ar, 0, e->X + (size_t)(vf ? 0 : S) * d, A);
  } else {
    embed_pos_kernel<<<S, 256, 0, e->es>>>(e->ids_text, 1, 0, W<float>(e, "ar_text_embedding.word_embeddings.weight"), 512, d,
                                           W<float>(e, "ar_text_position.alpha"), e->pe_ar, 0, e->X, S);
    // VALL-F: the text rows become the per-layer memory K / V (projected once, valle.py:598-602), then the audio rows take X
    if (vf) { VXC(cast_rows(e, e->X, e->Hn, (size_t)S * d)); VXC(memory_kv(e, e->ar_l, e->xkv_ar, S, d, c.nhead)); }
    embed_pos_kernel<<<A, 256, 0, e->es>>>(e->ids_audio, 1, 0, W<float>(e, "ar_audio_embedding.word_embeddings.weight"), 1025 + bos, d,
                                           W<float>(e, "ar_audio_position.alpha"), e->pe_ar, 0, e->X + (size_t)(vf ? 0 : S) * d, A);
  }
  char* kv_base = slot < 0 ? (char*)e->kv : (char*)(e->bkv + (size_t)slot * e->bkv_slot);
  float* x_dst = slot < 0 ? e->ar_x : e->bx + (size_t)slot * d;
  float* lg_dst = slot < 0 ? e->ar_logits : e->blogits + (size_t)slot * LOGITS_CUR;
  ArState* st_dst = slot < 0 ? e->d_st : e->bst + slot;
  if (vf) { e->mem_len = S; VXC(run_stack_f(e, e->ar_l, M, d, c.nhead, 0, -1, true, e->xkv_ar, S)); }
  else VXC(run_stack(e, e->ar_l, M, d, c.nhead, S, -1, true, kv_base));
  HIPC(hipMemcpyAsync(x_dst, e->X + (size_t)(M - 1) * d, (size_t)d * 4, hipMemcpyDeviceToDevice, e->es));
  // decode state as of "pass 0 computed"
  ArState& st = slot < 0 ? e->h_st[0] : e->h_bst[slot];
  memset(&st, 0, sizeof st);
  st.S = S; st.bos = bos; st.P = P; st.row = M - 1; st.pass = 0;
  st.kv_text = e->vallf ? 0 : S;
  st.temperature = 1.0f; st.max_new = -1;
  st.trace_logits = (slot < 0 && (c.flags & VX_FLAG_TRACE_LOGITS)) ? 1 : 0;
  HIPC(hipMemcpyAsync(st_dst, &st, sizeof st, hipMemcpyHostToDevice, e->es));
  VXC(enqueue_head(e, e->es, x_dst, lg_dst, st_dst, true));
  HIPC(hipGetLastError());
  HIPC(hipEventRecord(e->ev_t[1], e->es));
  HIPC(hipStreamSynchronize(e->es));  // the staging state is reused by decode
  float ms = 0.f;
  HIPC(hipEventElapsedTime(&ms, e->ev_t[0], e->ev_t[1]));
  e->t_prefill = ms;
  if (slot < 0) {
    e->S = S; e->P = P; e->bos = bos;
    e->prefilled = true; e->decoded = false;
    e->n_gen = 0; e->n_pass = 1; e->stop_reason = 0;
  } else {
    e->bS[slot] = S; e->bP[slot] = P; e->bbos[slot] = bos;
    e->bprefilled[slot] = true; e->bngen[slot] = 0; e->breason[slot] = 0;
  }
  VXC(sync_out(e, stream));
  return VX_OK;
}

extern "C" int vx_ar_prefill(vx_engine* e, const int64_t* text, int32_t S, const int64_t* prompt_cb0, int32_t P,
                             void* stream) {
  return prefill_impl(e, -1, text, S, prompt_cb0, P, stream);
}

extern "C" int vx_batch_prefill(vx_engine* e, int32_t slot, const int64_t* text, int32_t S, const int64_t* prompt_cb0,
                                int32_t P, void* stream) {
  if (!e) return fail(VX_ERR_ARG, "null engine");
  if (slot < 0 || slot >= e->bmax) return fail(VX_ERR_ARG, "slot %d outside [0, max_batch=%d)", slot, e->bmax);
  return prefill_impl(e, slot, text, S, prompt_cb0, P, stream);
}

// All slots' prefills as ONE pass over the concatenated rows (segment z = slot z, starts at a multiple of 64 rows):
// the GEMMs see sum(M_z) rows instead of 32 separate ~270-row problems, attention runs per segment with each
// segment's own prefix mask, K/V go straight to each slot's cache.  Same results as vx_batch_prefill per slot up
// to the GEMM kernel the dispatcher picks for the larger row count.
static int ensure_rows(vx_engine* e, size_t rows, size_t audio_rows, size_t text_rows);
static void launch_ln_batch(float* x, const float* part, int kgroups, const float* pbias, const float* gamma, const float* beta,
                            bf16* h, int B, int d, hipStream_t s);
template <int EPI> static int launch_bgemm(const BgemmArgs& a, hipStream_t s);

extern "C" int vx_batch_prefill_all(vx_engine* e, int32_t n, const int64_t* const* text, const int32_t* S,
                                    const int64_t* const* prompt_cb0, const int32_t* P, void* stream) {
  if (!e || !text || !S || !prompt_cb0 || !P) return fail(VX_ERR_ARG, "null argument");
  if (!e->finalized) return fail(VX_ERR_STATE, "weights not finalized");
  if (n < 1 || n > e->bmax) return fail(VX_ERR_ARG, "n %d outside [1, max_batch=%d]", n, e->bmax);
  if (!e->bf16 || !use_mfma(e)) return fail(VX_ERR_UNSUPPORTED, "vx_batch_prefill_all needs the bf16 MFMA row kernels");
  const vx_config& c = e->cfg;
  const int bos = c.prepend_bos ? 1 : 0, d = c.d_model;
  std::vector<int> start(n), len(n), tlen(n);
  size_t rows = 0;
  int maxlen = 0;
  for (int b = 0; b < n; ++b) {
    if (!text[b] || (!prompt_cb0[b] && P[b] > 0)) return fail(VX_ERR_ARG, "null argument (utterance %d)", b);
    if (S[b] <= 0 || P[b] < 0) return fail(VX_ERR_ARG, "S must be > 0 (valle.py:991), P >= 0");
    const int A = bos + P[b];
    if (S[b] > c.max_text || A + 1 > c.max_audio) return fail(VX_ERR_CAPACITY, "S=%d / P=%d exceed capacity", S[b], P[b]);
    if (A == 0) return fail(VX_ERR_ARG, "empty audio prefix needs prepend_bos");
    start[b] = (int)rows; len[b] = S[b] + A; tlen[b] = S[b];
    rows += (size_t)((len[b] + 63) / 64) * 64;
    if (len[b] > maxlen) maxlen = len[b];
  }
  ON_DEVICE(c.device);
  VXC(ensure_rows(e, rows, e->cap_audio, e->cap_text));
  VXC(sync_in(e, stream));
  HIPC(hipEventRecord(e->ev_t[0], e->es));
  HIPC(hipMemcpyAsync(e->d_seg_start, start.data(), n * sizeof(int), hipMemcpyHostToDevice, e->es));
  HIPC(hipMemcpyAsync(e->d_seg_len, len.data(), n * sizeof(int), hipMemcpyHostToDevice, e->es));
  HIPC(hipMemcpyAsync(e->d_seg_text, tlen.data(), n * sizeof(int), hipMemcpyHostToDevice, e->es));
  HIPC(hipMemsetAsync(e->X, 0, rows * (size_t)d * 4, e->es));  // padding rows must stay finite (they feed V^T columns)
  static const long long bos_id = NUM_AUDIO_TOKENS + 1;  // valle.py:1006-1007
  for (int b = 0; b < n; ++b) {
    const int A = bos + P[b];
    long long* it = e->bp_text + (size_t)b * c.max_text;
    long long* ia = e->bp_audio + (size_t)b * (c.max_audio + 1);
    HIPC(hipMemcpyAsync(it, text[b], (size_t)S[b] * 8, hipMemcpyDefault, e->es));
    if (bos) HIPC(hipMemcpyAsync(ia, &bos_id, 8, hipMemcpyHostToDevice, e->es));
    if (P[b]) HIPC(hipMemcpyAsync(ia + bos, prompt_cb0[b], (size_t)P[b] * 8, hipMemcpyDefault, e->es));
    float* xb = e->X + (size_t)start[b] * d;
    embed_pos_kernel<<<S[b], 256, 0, e->es>>>(it, 1, 0, W<float>(e, "ar_text_embedding.word_embeddings.weight"), 512, d,
                                              W<float>(e, "ar_text_position.alpha"), e->pe_ar, 0, xb, S[b]);
    embed_pos_kernel<<<A, 256, 0, e->es>>>(ia, 1, 0, W<float>(e, "ar_audio_embedding.word_embeddings.weight"), 1025 + bos, d,
                                           W<float>(e, "ar_audio_position.alpha"), e->pe_ar, 0, xb + (size_t)S[b] * d, A);
  }
  e->nseg = n; e->max_seg_len = maxlen; e->seg_text_on = true;
  int rc = run_stack(e, e->ar_l, (int)rows, d, c.nhead, 0, -1, true);
  e->nseg = 0; e->seg_text_on = false;
  VXC(rc);
  // decode state as of "pass 0 computed", last row of every segment = the slot's current activation
  for (int b = 0; b < n; ++b) {
    HIPC(hipMemcpyAsync(e->bx + (size_t)b * d, e->X + (size_t)(start[b] + len[b] - 1) * d, (size_t)d * 4, hipMemcpyDeviceToDevice, e->es));
    ArState& st = e->h_bst[b];
    memset(&st, 0, sizeof st);
    st.S = S[b]; st.bos = bos; st.P = P[b]; st.row = len[b] - 1; st.pass = 0; st.kv_text = S[b];
    st.temperature = 1.0f; st.max_new = -1;
  }
  HIPC(hipMemcpyAsync(e->bst, e->h_bst, (size_t)n * sizeof(ArState), hipMemcpyHostToDevice, e->es));
  // first logits of every slot: final LayerNorm + head, as at the end of a batched step
  launch_ln_batch(e->bx, nullptr, 0, nullptr, W<float>(e, "ar_decoder.norm.weight"), W<float>(e, "ar_decoder.norm.bias"), e->bh, n, d, e->es);
  BgemmArgs hgm{};
  hgm.st = e->bst; hgm.B = n;
  hgm.A = e->bh; hgm.W = W<bf16>(e, "ar_predict_layer.weight"); hgm.N = AR_VOCAB; hgm.K = d; hgm.kgroups = 1;
  hgm.logits = e->blogits; hgm.logits_stride = LOGITS_CUR;
  hgm.trace = e->btrace; hgm.trace_rows = e->btok_stride;
  VXC(launch_bgemm<BE_LOGITS>(hgm, e->es));
  HIPC(hipGetLastError());
  HIPC(hipEventRecord(e->ev_t[1], e->es));
  HIPC(hipStreamSynchronize(e->es));  // the staging state is reused by decode
  float ms = 0.f;
  HIPC(hipEventElapsedTime(&ms, e->ev_t[0], e->ev_t[1]));
  e->t_prefill = ms;
  for (int b = 0; b < n; ++b) {
    e->bS[b] = S[b]; e->bP[b] = P[b]; e->bbos[b] = bos;
    e->bprefilled[b] = true; e->bngen[b] = 0; e->breason[b] = 0;
  }
  VXC(sync_out(e, stream));
  return VX_OK;
}

// One decode step: sample from the newest logits, append, run the 12-layer stack on the new
// token, produce the next logits.  Every kernel reads its position from e->d_st, so the same
// launch sequence (captured once as a hipGraph) serves every pass.
static int enqueue_ar_step_f(vx_engine* e, hipStream_t s);
// The layers and the head of the XCD-sharded step (ar_tp.hpp); the sampling launch in front of them is the caller's (it writes
// the fp32 embedding into ar_x and zeroes accumulator 1).  Launch n = 2 li (attention half), 2 li + 1 (feed-forward half), 2 L
// (head) normalises accumulator n % 3, adds into (n + 1) % 3 and zeroes (n + 2) % 3; layer 0's attention half reads ar_x instead.
static int enqueue_ar_step_tp(vx_engine* e, hipStream_t s) {
  const vx_config& c = e->cfg;
  const int H = c.nhead, hd = c.d_model / H, L = c.num_layers;
  const size_t kv_layer = (size_t)2 * H * e->ctx_max * hd * e->esz;
  const int grid = TP_X * TP_WG;
  auto acc = [&](int n) { return e->tp_xacc + (size_t)(n % 3) * TP_D; };
  for (int li = 0; li < L; ++li) {
    const LayerW& l = e->ar_l[li];
    char* kc = (char*)e->kv + (size_t)li * kv_layer;
    const int n = 2 * li;
    TpAttnArgs a{};
    const float* gbb = e->tp_gbb + (size_t)(2 * li) * 3 * TP_D;
    a.qkv_bias = l.in_b; a.err = e->d_epoch + 1;
    a.gq = e->fq_gq + (size_t)li * H * FQ_QKV; a.gp = e->fq_gp + (size_t)li * H * FQ_G * FQ_PART;
    a.acc.add = acc(n + 1); a.acc.zero = acc(n + 2); a.acc.bias = l.out_b;
    a.kcache = kc; a.vcache = kc + kv_layer / 2; a.ctx_max = e->ctx_max;
    a.scale = 1.0f / sqrtf((float)hd); a.layer = li;
    if (e->bf16) {
      if (li) tp_attn_kernel<bf16, false><<<grid, 256, 0, s>>>(l.in_w, e->ar_x, acc(n), gbb, e->d_st, e->d_epoch, e->tp_wo[li], a);
      else tp_attn_kernel<bf16, true><<<grid, 256, 0, s>>>(l.in_w, e->ar_x, acc(n), gbb, e->d_st, e->d_epoch, e->tp_wo[li], a);
    } else {
      if (li) tp_attn_kernel<float, false><<<grid, 256, 0, s>>>(l.in_w, e->ar_x, acc(n), gbb, e->d_st, e->d_epoch, e->tp_wo[li], a);
      else tp_attn_kernel<float, true><<<grid, 256, 0, s>>>(l.in_w, e->ar_x, acc(n), gbb, e->d_st, e->d_epoch, e->tp_wo[li], a);
    }
    TpFfnArgs f{};
    f.b1 = l.b1; f.err = e->d_epoch + 1; f.gh = e->tp_gh + (size_t)li * TP_X * TP_HID; f.layer = li;
    f.acc.add = acc(n + 2); f.acc.zero = acc(n + 3); f.acc.bias = l.b2;
    if (e->bf16) tp_ffn_kernel<bf16><<<grid, 256, 0, s>>>(l.w1, acc(n + 1), gbb + 3 * TP_D, e->d_epoch, e->tp_w2[li], f);
    else tp_ffn_kernel<float><<<grid, 256, 0, s>>>(l.w1, acc(n + 1), gbb + 3 * TP_D, e->d_epoch, e->tp_w2[li], f);
  }
  TpHeadArgs h{};
  h.logits = e->ar_logits; h.N = AR_VOCAB;
  const int hg = (AR_VOCAB + 15) / 16;
  const float* gbh = e->tp_gbb + (size_t)(2 * L) * 3 * TP_D;
  if (e->bf16) tp_head_kernel<bf16><<<hg, 256, 0, s>>>(W<void>(e, "ar_predict_layer.weight"), acc(2 * L), gbh, e->d_st, h);
  else tp_head_kernel<float><<<hg, 256, 0, s>>>(W<void>(e, "ar_predict_layer.weight"), acc(2 * L), gbh, e->d_st, h);
  return VX_OK;
}

static SampleArgs step_sample_args(vx_engine* e) {
  const vx_config& c = e->cfg;
  SampleArgs sa{};
  sa.logits = e->ar_logits; sa.V = AR_VOCAB; sa.st = e->d_st;
  sa.tokens = e->d_tokens; sa.sampled = e->d_sampled; sa.argmaxes = e->d_argmax;
  sa.emb = W<float>(e, "ar_audio_embedding.word_embeddings.weight");
  sa.alpha = W<float>(e, "ar_audio_position.alpha");
  sa.pe = e->pe_ar; sa.x = e->ar_x; sa.d = c.d_model;
  if (c.flags & VX_FLAG_PRENET) { sa.alpha = e->d_zero; sa.x = e->ar_e; }  // raw embedding; the position is added after the prenet
  sa.kid = 0;  // stamp ids of the step (probe builds): 0 sampling, 1 + 5 l + {0 QKV, 1 attention, 2 out-proj, 3 FFN1, 4 FFN2}, 61 head
  sa.epoch = e->d_epoch;
  sa.zero_acc = e->tp ? e->tp_xacc + TP_D : nullptr;  // accumulator 1: layer 0's attention half adds into it
  return sa;
}
static int enqueue_ar_step(vx_engine* e, hipStream_t s) {
  if (e->vallf) return enqueue_ar_step_f(e, s);
  const vx_config& c = e->cfg;
  const int d = c.d_model, H = c.nhead, hd = d / H;
  const bool post = c.flags & VX_FLAG_POST_NORM;
  const SampleArgs sa = step_sample_args(e);
  const bool prenet = c.flags & VX_FLAG_PRENET;
  sample_embed4_kernel<5, 17><<<1, 256, 0, s>>>(sa);
  if (e->tp) return enqueue_ar_step_tp(e, s);
  if (prenet) {  // y_emb = ar_audio_prenet(E[tok]); x = y_emb + alpha * pe (valle.py:1013-1015): three fp32 GEMVs
    GemvArgs p0{}, p1{}, p2{};
    p0.st = p1.st = p2.st = e->d_st;
    p0.kid = p1.kid = p2.kid = -1;
    p0.W = W<void>(e, "ar_audio_prenet.0.weight"); p0.bias = W<float>(e, "ar_audio_prenet.0.bias");
    p0.x = e->ar_e; p0.y = e->ar_h1; p0.N = PRENET_H; p0.K = d; p0.pro = PRO_COPY; p0.epi = EPI_RELU;
    p1.W = W<void>(e, "ar_audio_prenet.3.weight"); p1.bias = W<float>(e, "ar_audio_prenet.3.bias");
    p1.x = e->ar_h1; p1.y = e->ar_h2; p1.N = PRENET_H; p1.K = PRENET_H; p1.pro = PRO_COPY; p1.epi = EPI_RELU;
    p2.W = W<void>(e, "ar_audio_prenet.6.weight"); p2.bias = W<float>(e, "ar_audio_prenet.6.bias");
    p2.x = e->ar_h2; p2.y = e->ar_x; p2.N = d; p2.K = PRENET_H; p2.pro = PRO_COPY; p2.epi = EPI_POS;
    p2.pe = e->pe_ar; p2.pos_alpha = W<float>(e, "ar_audio_position.alpha");
    VXC(launch_gemv(false, p0, e->num_cu, s));
    VXC(launch_gemv(false, p1, e->num_cu, s));
    VXC(launch_gemv(false, p2, e->num_cu, s));
  }
  const size_t kv_layer = (size_t)2 * H * e->ctx_max * hd * e->esz;
  const float scale = 1.0f / sqrtf((float)hd);
  // L2 / Infinity-Cache warm-up (GemvArgs.pf): GEMV i of the step also requests the weights of GEMV i + dist, in step order
  // [QKV_0, out_0, FFN1_0, FFN2_0, QKV_1, ..., FFN2_{L-1}, head] and wrapping into the next token's step.
  static const int pf_dist = getenv("VX_AR_PREFETCH") ? atoi(getenv("VX_AR_PREFETCH")) : 2;
  static const int ar_nt = getenv("VX_AR_NT") ? atoi(getenv("VX_AR_NT")) : 0;  // A/B switch: non-temporal weight loads
  struct PfW { const void* W; int N, K; };
  std::vector<PfW> seq;
  for (int li = 0; li < c.num_layers; ++li) {
    const LayerW& l = e->ar_l[li];
    seq.push_back({l.in_w, 3 * d, d}); seq.push_back({l.out_w, d, d}); seq.push_back({l.w1, 4 * d, d}); seq.push_back({l.w2, d, 4 * d});
  }
  seq.push_back({W<void>(e, "ar_predict_layer.weight"), AR_VOCAB, d});
  auto warm = [&](GemvArgs& a, int idx) {
    if (pf_dist <= 0) return;
    const PfW& n = seq[(idx + pf_dist) % seq.size()];
    gemv_prefetch(a, n.W, n.N, n.K, e->bf16, e->num_cu);
  };
  for (int li = 0; li < c.num_layers; ++li) {
    const LayerW& l = e->ar_l[li];
    char* kc = (char*)e->kv + (size_t)li * kv_layer;
    char* vc = kc + kv_layer / 2;
    GemvArgs a{};
    a.st = e->d_st; a.d = d; a.hd = hd; a.ctx_max = e->ctx_max; a.nhead = H;
    // qkv = in_proj(LN1(x)); k,v appended to the cache (transformer.py:297-301)
    a.W = l.in_w; a.bias = l.in_b; a.x = e->ar_x; a.gamma = l.n1_g; a.beta = l.n1_b;
    a.N = 3 * d; a.K = d; a.pro = PRO_LN; a.epi = EPI_QKV; a.q = e->ar_q; a.kcache = kc; a.vcache = vc;
    // post-norm (transformer.py:303-308): ar_x holds the raw sum x + block(x) of the previous sub-layer and the norm that
    // follows it is fused into the NEXT kernel's prologue, which also leaves the normalised vector in ar_xn as the
    // base of the next residual add.  Layer 0 reads the fresh embedding as is.
    const float* res = nullptr;  // residual base of this layer's attention block (null: ar_x itself)
    if (post) {
      if (li == 0) { a.pro = PRO_COPY; }
      else { a.gamma = e->ar_l[li - 1].n2_g; a.beta = e->ar_l[li - 1].n2_b; a.xnorm_out = e->ar_xn; res = e->ar_xn; }
    }
    warm(a, 4 * li);
    a.nt = ar_nt; a.kid = 1 + 5 * li;
    VXC(launch_gemv(e->bf16, a, e->num_cu, s));
#define AD(HDV)                                                                                                                                              \
  if (hd == HDV) {                                                                                                                                           \
    if (e->bf16) attn_decode_small_kernel<bf16, HDV><<<H * ATT_NSPLIT + VX_KSTAMP_EXTRA, 256, 0, s>>>(e->ar_q, (const bf16*)kc, (const bf16*)vc, e->ar_part, e->d_st, e->ctx_max, scale, 2 + 5 * li); \
    else attn_decode_small_kernel<float, HDV><<<H * ATT_NSPLIT + VX_KSTAMP_EXTRA, 256, 0, s>>>(e->ar_q, (const float*)kc, (const float*)vc, e->ar_part, e->d_st, e->ctx_max, scale, 2 + 5 * li);      \
  }
    if (hd == 64) {
      if (e->bf16) attn_decode_kernel<bf16, 64><<<H * ATT_NSPLIT + VX_KSTAMP_EXTRA, 256, 0, s>>>(e->ar_q, (const bf16*)kc, (const bf16*)vc, e->ar_part, e->d_st, e->ctx_max, scale, 2 + 5 * li);
      else attn_decode_kernel<float, 64><<<H * ATT_NSPLIT + VX_KSTAMP_EXTRA, 256, 0, s>>>(e->ar_q, (const float*)kc, (const float*)vc, e->ar_part, e->d_st, e->ctx_max, scale, 2 + 5 * li);
    }
    AD(32) AD(16) AD(8) AD(4)
#undef AD
    // x += out_proj(attn)
    GemvArgs o{};
    o.st = e->d_st; o.hd = hd; o.nhead = H;
    o.W = l.out_w; o.bias = l.out_b; o.part = e->ar_part; o.y = e->ar_x; o.N = d; o.K = d; o.pro = PRO_ATTN; o.epi = EPI_RESID;
    o.res = res;
    warm(o, 4 * li + 1);
    o.nt = ar_nt; o.kid = 3 + 5 * li;
    VXC(launch_gemv(e->bf16, o, e->num_cu, s));
    // f = relu(linear1(LN2(x)))
    GemvArgs f{};
    f.st = e->d_st;
    f.W = l.w1; f.bias = l.b1; f.x = e->ar_x; f.gamma = l.n2_g; f.beta = l.n2_b; f.y = e->ar_f;
    f.N = 4 * d; f.K = d; f.pro = PRO_LN; f.epi = EPI_RELU;
    if (post) { f.gamma = l.n1_g; f.beta = l.n1_b; f.xnorm_out = e->ar_xn; }  // x = norm1(x + sa(x)), kept in ar_xn
    warm(f, 4 * li + 2);
    f.nt = ar_nt; f.kid = 4 + 5 * li;
    VXC(launch_gemv(e->bf16, f, e->num_cu, s));
    // x += linear2(f)
    GemvArgs g{};
    g.st = e->d_st;
    g.W = l.w2; g.bias = l.b2; g.x = e->ar_f; g.y = e->ar_x; g.N = d; g.K = 4 * d; g.pro = PRO_COPY; g.epi = EPI_RESID;
    if (post) g.res = e->ar_xn;  // raw sum norm1(..) + ff(..); its norm2 runs in the next layer's (or the head's) prologue
    warm(g, 4 * li + 3);
    g.nt = ar_nt; g.kid = 5 + 5 * li;
    VXC(launch_gemv(e->bf16, g, e->num_cu, s));
  }
  if (pf_dist > 0) {
    const PfW& n = seq[(4 * c.num_layers + pf_dist) % seq.size()];
    VXC(enqueue_head(e, s, nullptr, nullptr, nullptr, false, n.W, n.N, n.K));
  } else {
    VXC(enqueue_head(e, s));
  }
  return VX_OK;
}

// The VALL-F decode step (valle.py:613-647 with a KV cache; TransformerDecoderLayer, modules/transformer.py:536-560): per
// layer QKV GEMV, causal attention over the cached AUDIO rows, out-projection, cross-attention query GEMV (rows [0, d) of the
// packed multihead_attn in_proj), single-query attention over the layer's cached text memory (fixed length), its
// out-projection, FFN1, FFN2: 8 launches per layer.  Post-norm: every norm runs in the prologue of the kernel that consumes it
// and leaves the normalised vector in ar_xn as the next residual base (as in the VALL-E step).
static int enqueue_ar_step_f(vx_engine* e, hipStream_t s) {
  const vx_config& c = e->cfg;
  const int d = c.d_model, H = c.nhead, hd = d / H;
  const bool post = c.flags & VX_FLAG_POST_NORM;
  SampleArgs sa{};
  sa.logits = e->ar_logits; sa.V = AR_VOCAB; sa.st = e->d_st;
  sa.tokens = e->d_tokens; sa.sampled = e->d_sampled; sa.argmaxes = e->d_argmax;
  sa.emb = W<float>(e, "ar_audio_embedding.word_embeddings.weight");
  sa.alpha = W<float>(e, "ar_audio_position.alpha");
  sa.pe = e->pe_ar; sa.x = e->ar_x; sa.d = d; sa.kid = -1;
  const bool prenet = c.flags & VX_FLAG_PRENET;
  if (prenet) { sa.alpha = e->d_zero; sa.x = e->ar_e; }
  sample_embed4_kernel<5, 17><<<1, 256, 0, s>>>(sa);
  if (prenet) {
    GemvArgs p0{}, p1{}, p2{};
    p0.st = p1.st = p2.st = e->d_st;
    p0.kid = p1.kid = p2.kid = -1;
    p0.W = W<void>(e, "ar_audio_prenet.0.weight"); p0.bias = W<float>(e, "ar_audio_prenet.0.bias");
    p0.x = e->ar_e; p0.y = e->ar_h1; p0.N = PRENET_H; p0.K = d; p0.pro = PRO_COPY; p0.epi = EPI_RELU;
    p1.W = W<void>(e, "ar_audio_prenet.3.weight"); p1.bias = W<float>(e, "ar_audio_prenet.3.bias");
    p1.x = e->ar_h1; p1.y = e->ar_h2; p1.N = PRENET_H; p1.K = PRENET_H; p1.pro = PRO_COPY; p1.epi = EPI_RELU;
    p2.W = W<void>(e, "ar_audio_prenet.6.weight"); p2.bias = W<float>(e, "ar_audio_prenet.6.bias");
    p2.x = e->ar_h2; p2.y = e->ar_x; p2.N = d; p2.K = PRENET_H; p2.pro = PRO_COPY; p2.epi = EPI_POS;
    p2.pe = e->pe_ar; p2.pos_alpha = W<float>(e, "ar_audio_position.alpha");
    VXC(launch_gemv(false, p0, e->num_cu, s));
    VXC(launch_gemv(false, p1, e->num_cu, s));
    VXC(launch_gemv(false, p2, e->num_cu, s));
  }
  const size_t kv_layer = (size_t)2 * H * e->ctx_max * hd * e->esz;
  const size_t xkv_layer = (size_t)2 * d * c.max_text * e->esz;
  const float scale = 1.0f / sqrtf((float)hd);
  auto attend = [&](const char* kc, const char* vc, int ctx_max, int fixed_ctx) -> int {
#define ADF(HDV)                                                                                                                       \
  if (hd == HDV) {                                                                                                                     \
    if (e->bf16) attn_decode_small_kernel<bf16, HDV><<<H * ATT_NSPLIT, 256, 0, s>>>(e->ar_q, (const bf16*)kc, (const bf16*)vc, e->ar_part, e->d_st, ctx_max, scale, -1, fixed_ctx); \
    else attn_decode_small_kernel<float, HDV><<<H * ATT_NSPLIT, 256, 0, s>>>(e->ar_q, (const float*)kc, (const float*)vc, e->ar_part, e->d_st, ctx_max, scale, -1, fixed_ctx);      \
    return VX_OK;                                                                                                                      \
  }
    if (hd == 64) {
      if (e->bf16) attn_decode_kernel<bf16, 64><<<H * ATT_NSPLIT, 256, 0, s>>>(e->ar_q, (const bf16*)kc, (const bf16*)vc, e->ar_part, e->d_st, ctx_max, scale, -1, fixed_ctx);
      else attn_decode_kernel<float, 64><<<H * ATT_NSPLIT, 256, 0, s>>>(e->ar_q, (const float*)kc, (const float*)vc, e->ar_part, e->d_st, ctx_max, scale, -1, fixed_ctx);
      return VX_OK;
    }
    ADF(32) ADF(16) ADF(8) ADF(4)
#undef ADF
    return fail(VX_ERR_UNSUPPORTED, "decode attention: head_dim %d", hd);
  };
  for (int li = 0; li < c.num_layers; ++li) {
    const LayerW& l = e->ar_l[li];
    const char* kc = (const char*)e->kv + (size_t)li * kv_layer;
    const char* vc = kc + kv_layer / 2;
    const char* xk = (const char*)e->xkv_ar + (size_t)li * xkv_layer;
    const char* xv = xk + xkv_layer / 2;
    const float* res = nullptr;  // residual base of the sub-block (null: ar_x itself)
    // self-attention
    GemvArgs a{};
    a.st = e->d_st; a.d = d; a.hd = hd; a.ctx_max = e->ctx_max; a.nhead = H; a.kid = -1;
    a.W = l.in_w; a.bias = l.in_b; a.x = e->ar_x; a.gamma = l.n1_g; a.beta = l.n1_b;
    a.N = 3 * d; a.K = d; a.pro = PRO_LN; a.epi = EPI_QKV; a.q = e->ar_q; a.kcache = (void*)kc; a.vcache = (void*)vc;
    if (post) {
      if (li == 0) a.pro = PRO_COPY;  // the fresh embedding, no norm in front of the first block
      else { a.gamma = e->ar_l[li - 1].n3_g; a.beta = e->ar_l[li - 1].n3_b; a.xnorm_out = e->ar_xn; res = e->ar_xn; }
    }
    VXC(launch_gemv(e->bf16, a, e->num_cu, s));
    VXC(attend(kc, vc, e->ctx_max, 0));
    GemvArgs o{};
    o.st = e->d_st; o.hd = hd; o.nhead = H; o.kid = -1;
    o.W = l.out_w; o.bias = l.out_b; o.part = e->ar_part; o.y = e->ar_x; o.N = d; o.K = d; o.pro = PRO_ATTN; o.epi = EPI_RESID; o.res = res;
    VXC(launch_gemv(e->bf16, o, e->num_cu, s));
    // cross-attention: q = in_proj[0:d](norm2(x)) (post-norm: norm1 of the raw sum, which is also the next residual base)
    GemvArgs q{};
    q.st = e->d_st; q.kid = -1;
    q.W = l.cin_w; q.bias = l.cin_b; q.x = e->ar_x; q.y = e->ar_q; q.N = d; q.K = d; q.pro = PRO_LN; q.epi = EPI_BIAS;
    q.gamma = post ? l.n1_g : l.n2_g; q.beta = post ? l.n1_b : l.n2_b;
    if (post) q.xnorm_out = e->ar_xn;
    VXC(launch_gemv(e->bf16, q, e->num_cu, s));
    VXC(attend(xk, xv, c.max_text, 1));  // length = ArState.S (the current utterance's text rows), read on the device
    GemvArgs co{};
    co.st = e->d_st; co.hd = hd; co.nhead = H; co.kid = -1;
    co.W = l.cout_w; co.bias = l.cout_b; co.part = e->ar_part; co.y = e->ar_x; co.N = d; co.K = d; co.pro = PRO_ATTN; co.epi = EPI_RESID;
    co.res = post ? e->ar_xn : nullptr;
    VXC(launch_gemv(e->bf16, co, e->num_cu, s));
    // feed-forward
    GemvArgs f{};
    f.st = e->d_st; f.kid = -1;
    f.W = l.w1; f.bias = l.b1; f.x = e->ar_x; f.y = e->ar_f; f.N = 4 * d; f.K = d; f.pro = PRO_LN; f.epi = EPI_RELU;
    f.gamma = post ? l.n2_g : l.n3_g; f.beta = post ? l.n2_b : l.n3_b;
    if (post) f.xnorm_out = e->ar_xn;
    VXC(launch_gemv(e->bf16, f, e->num_cu, s));
    GemvArgs g{};
    g.st = e->d_st; g.kid = -1;
    g.W = l.w2; g.bias = l.b2; g.x = e->ar_f; g.y = e->ar_x; g.N = d; g.K = 4 * d; g.pro = PRO_COPY; g.epi = EPI_RESID;
    if (post) g.res = e->ar_xn;
    VXC(launch_gemv(e->bf16, g, e->num_cu, s));
  }
  return enqueue_head(e, s);
}

extern "C" int vx_ar_decode(vx_engine* e, const vx_decode_params* p, void* stream) {
  if (!e || !p) return fail(VX_ERR_ARG, "null argument");
  if (p->struct_size != (int32_t)sizeof(vx_decode_params)) return fail(VX_ERR_ARG, "vx_decode_params.struct_size mismatch");
  if (!e->prefilled || e->decoded) return fail(VX_ERR_STATE, "vx_ar_decode needs a fresh vx_ar_prefill");
  if (!(p->temperature > 0.f)) return fail(VX_ERR_ARG, "temperature must be > 0");
  const vx_config& c = e->cfg;
  ON_DEVICE(c.device);
  // upper bound on appended tokens (valle.py:1047: stops once bos + n_gen > 16 S)
  long long max_tok = 16LL * e->S + 1 - e->bos;
  if (p->forced) max_tok = p->n_forced;
  else if (p->max_new_tokens >= 0 && p->max_new_tokens < max_tok) max_tok = p->max_new_tokens;
  // 16 S + 1 is the WORST case (valle.py:1047); a trained model stops at EOS long before it, so a long text must not be
  // refused up front: the launch bound is clamped to the rows the cache has, and only a decode that really fills them while
  // the stop rule has not fired is a capacity error
  const long long room = (long long)c.max_audio - e->bos - e->P;  // >= 1 (checked at prefill)
  bool cap_limited = false;
  if (max_tok > room) {
    if (p->forced) return fail(VX_ERR_CAPACITY, "need %lld audio rows, capacity %d", e->bos + e->P + max_tok, c.max_audio);
    max_tok = room; cap_limited = true;
  }
  VXC(sync_in(e, stream));
  if (p->exp_noise) {
    if (p->noise_rows <= 0) return fail(VX_ERR_ARG, "noise_rows must be > 0");
    const size_t need = (size_t)p->noise_rows * AR_VOCAB;
    if (need > e->noise_cap) {
      if (e->d_noise) HIPC(hipFree(e->d_noise));
      HIPC(hipMalloc((void**)&e->d_noise, need * 4));
      e->noise_cap = need;
    }
    HIPC(hipMemcpyAsync(e->d_noise, p->exp_noise, need * 4, hipMemcpyDefault, e->es));
  }
  if (p->forced && p->n_forced > 0) {
    if ((size_t)p->n_forced > e->forced_cap) {
      if (e->d_forced) HIPC(hipFree(e->d_forced));
      HIPC(hipMalloc((void**)&e->d_forced, (size_t)p->n_forced * 8));
      e->forced_cap = p->n_forced;
    }
    HIPC(hipMemcpyAsync(e->d_forced, p->forced, (size_t)p->n_forced * 8, hipMemcpyDefault, e->es));
  }
  ArState& st = e->h_st[0];
  st.top_k = p->top_k; st.temperature = p->temperature; st.max_new = cap_limited ? (int)room : p->max_new_tokens;
  st.exp_noise = p->exp_noise ? e->d_noise : nullptr;
  st.noise_rows = p->noise_rows; st.seed = p->seed;
  st.forced = p->forced ? (p->n_forced > 0 ? e->d_forced : (const long long*)e->d_tokens) : nullptr;
  st.n_forced = p->forced ? p->n_forced : 0;
  HIPC(hipMemcpyAsync(e->d_st, &st, sizeof st, hipMemcpyHostToDevice, e->es));

  const bool graph = !(c.flags & VX_FLAG_NO_GRAPH);
  if (graph && !e->gexec) {
    HIPC(hipStreamBeginCapture(e->es, hipStreamCaptureModeThreadLocal));
    int r = enqueue_ar_step(e, e->es);
    hipError_t ce = hipStreamEndCapture(e->es, &e->graph);
    if (r != VX_OK) return r;
    HIPC(ce);
    HIPC(hipGraphInstantiate(&e->gexec, e->graph, nullptr, nullptr, 0));
  }
  HIPC(hipEventRecord(e->ev_t[2], e->es));
  // step j samples from logits j and appends token j+1; the step that appends the last
  // admissible token also raises the stop flag, so max_tok launches suffice (teacher forcing
  // needs one more to close the sequence); EOS can only end it earlier.
  long long bound = max_tok + (p->forced ? 1 : 0);
  if (bound < 1) bound = 1;
  long long launched = 0;
  int slot = 0;
  bool done = false;
  bool pending[2] = {false, false};
  while (!done) {
    const long long n = (bound - launched) < POLL_CHUNK ? (bound - launched) : POLL_CHUNK;
    for (long long i = 0; i < n; ++i) {
      if (graph) HIPC(hipGraphLaunch(e->gexec, e->es));
      else VXC(enqueue_ar_step(e, e->es));
    }
    launched += n;
    HIPC(hipMemcpyAsync(&e->h_st[1 + slot], e->d_st, sizeof(ArState), hipMemcpyDeviceToHost, e->es));
    HIPC(hipEventRecord(e->ev_poll[slot], e->es));
    pending[slot] = true;
    const int other = slot ^ 1;
    // keep one chunk in flight while the previous one is inspected
    if (pending[other]) {
      HIPC(hipEventSynchronize(e->ev_poll[other]));
      pending[other] = false;
      if (e->h_st[1 + other].done) done = true;
    }
    if (!done && launched >= bound) {
      HIPC(hipEventSynchronize(e->ev_poll[slot]));
      pending[slot] = false;
      if (!e->h_st[1 + slot].done) return fail(VX_ERR_STATE, "decode did not terminate within %lld steps", bound);
      done = true;
    }
    slot = other;
  }
  HIPC(hipEventRecord(e->ev_t[3], e->es));
  HIPC(hipMemcpyAsync(&e->h_st[1], e->d_st, sizeof(ArState), hipMemcpyDeviceToHost, e->es));
  unsigned* h_ep = reinterpret_cast<unsigned*>(e->h_st + 3);
  HIPC(hipMemcpyAsync(h_ep, e->d_epoch, 2 * sizeof(unsigned), hipMemcpyDeviceToHost, e->es));
  HIPC(hipStreamSynchronize(e->es));
  HIPC(hipGetLastError());
  if (h_ep[1] != 0) {  // a bounded spin of the sharded step ran out: its workgroups were not all resident (another process on the GPU?)
    const unsigned code = h_ep[1];
    HIPC(hipMemsetAsync(e->d_epoch + 1, 0, sizeof(unsigned), e->es));
    HIPC(hipStreamSynchronize(e->es));
    return fail(VX_ERR_STATE, "decode step: hand-over %u of the sharded decode step timed out (set VX_AR_TP=0)", code);
  }
  float ms = 0.f;
  HIPC(hipEventElapsedTime(&ms, e->ev_t[2], e->ev_t[3]));
  e->t_decode = ms;
  e->n_launch = (double)launched;
  e->n_gen = e->h_st[1].n_gen;
  e->stop_reason = e->h_st[1].stop_reason;
  e->n_pass = e->h_st[1].pass + 1;
  e->decoded = true;
  VXC(sync_out(e, stream));
  if (cap_limited && e->stop_reason == VX_STOP_MAX_NEW)
    return fail(VX_ERR_CAPACITY, "capacity exceeded: the KV cache filled (max_audio = %d rows: %d prompt + %d generated) before the stop rule fired; "
                "raise max_audio", c.max_audio, e->bos + e->P, e->n_gen);
  return VX_OK;
}

extern "C" int vx_ar_result(vx_engine* e, int64_t* tokens, int32_t capacity, int32_t* n_tokens, int32_t* stop_reason,
                            int32_t* n_pass) {
  if (!e) return fail(VX_ERR_ARG, "null engine");
  if (!e->decoded) return fail(VX_ERR_STATE, "no finished decode");
  ON_DEVICE(e->cfg.device);
  if (n_tokens) *n_tokens = e->n_gen;
  if (stop_reason) *stop_reason = e->stop_reason;
  if (n_pass) *n_pass = e->n_pass;
  if (tokens) {
    if (capacity < e->n_gen) return fail(VX_ERR_CAPACITY, "token buffer too small (%d < %d)", capacity, e->n_gen);
    std::vector<int> tmp(e->n_gen);
    if (e->n_gen) HIPC(hipMemcpy(tmp.data(), e->d_tokens, (size_t)e->n_gen * 4, hipMemcpyDeviceToHost));
    for (int i = 0; i < e->n_gen; ++i) tokens[i] = tmp[i];
  }
  return VX_OK;
}

// ------------------------------------------------------------------------------ batched AR decode
template <int EPI, int NH> static int launch_bgemm_h(const BgemmArgs& a, int ns, int grid, hipStream_t s) {
  if (a.N > 65535 || a.K > 65535) return fail(VX_ERR_UNSUPPORTED, "bgemm: N=%d K=%d", a.N, a.K);  // (N << 16) | K travels as one argument
  const unsigned nk = ((unsigned)a.N << 16) | (unsigned)a.K;
#define BG(NSV)                                                                                              \
  if (ns == NSV) {                                                                                           \
    if (a.pf != nullptr) bgemm_kernel<EPI, NSV, NH, true><<<grid, 256, 0, s>>>(a.A, a.W, nk, a.kgroups, a);   \
    else bgemm_kernel<EPI, NSV, NH><<<grid, 256, 0, s>>>(a.A, a.W, nk, a.kgroups, a);                         \
    return VX_OK;                                                                                            \
  }
  BG(1) BG(2) BG(4) BG(8)
#undef BG
  return fail(VX_ERR_UNSUPPORTED, "bgemm: %d steps per wave", ns);
}
// Points `a` at the weights of a later GEMM of the batched step (BgemmArgs.pf): an even share per workgroup, 32 KB at most.
static void bgemm_prefetch(BgemmArgs& a, const void* Wn, int Nn, int Kn) {
  const size_t total = (size_t)Nn * Kn * 2;
  const int grid = ((a.N + 15) / 16) * a.kgroups;
  if (Wn == nullptr || total >= (1ull << 32) || total < 16 || grid <= 0) return;
  size_t slice = ((total + grid - 1) / grid + 15) & ~(size_t)15;
  if (slice > 32768) slice = 32768;
  a.pf = Wn; a.pf_slice = (unsigned)slice; a.pf_total = (unsigned)total;
}
template <int EPI> static int launch_bgemm(const BgemmArgs& a, hipStream_t s) {
  const int ns = a.K / (a.kgroups * 128);
  const int grid = ((a.N + 15) / 16) * a.kgroups;
  if (ns * a.kgroups * 128 != a.K) return fail(VX_ERR_UNSUPPORTED, "bgemm: K=%d kgroups=%d", a.K, a.kgroups);
  // two 16-slot MFMA halves up to 32 slots, four up to 64
  return a.B <= 32 ? launch_bgemm_h<EPI, 2>(a, ns, grid, s) : launch_bgemm_h<EPI, 4>(a, ns, grid, s);
}
static int kgroups_for(int K) { return (K / 128) >= 4 ? 4 : 1; }
static void launch_ln_batch(float* x, const float* part, int kgroups, const float* pbias, const float* gamma, const float* beta,
                            bf16* h, int B, int d, hipStream_t s) {
  if (part == nullptr) ln_batch_kernel<0><<<B, 256, 0, s>>>(x, nullptr, nullptr, gamma, beta, h, d);
  else if (kgroups == 4) ln_batch_kernel<4><<<B, 256, 0, s>>>(x, part, pbias, gamma, beta, h, d);
  else ln_batch_kernel<1><<<B, 256, 0, s>>>(x, part, pbias, gamma, beta, h, d);
}

// One batched step: every slot samples its next token, then the L layers run once over all B slots.
static int enqueue_batch_step(vx_engine* e, int B, hipStream_t s) {
  const vx_config& c = e->cfg;
  const int d = c.d_model, H = c.nhead, hd = 64, L = c.num_layers;
  SampleArgs sa{};
  sa.logits = e->blogits; sa.V = AR_VOCAB; sa.st = e->bst;
  sa.tokens = e->btok; sa.sampled = e->bsamp; sa.argmaxes = e->bargm;
  sa.emb = W<float>(e, "ar_audio_embedding.word_embeddings.weight");
  sa.alpha = W<float>(e, "ar_audio_position.alpha");
  sa.pe = e->pe_ar; sa.x = e->bx; sa.d = d;
  sa.logits_stride = LOGITS_CUR; sa.tok_stride = e->btok_stride;
  sample_embed4_kernel<5, 17><<<B, 256, 0, s>>>(sa);
  const size_t kv_layer = (size_t)2 * H * e->ctx_max * hd;  // elements
  const float scale = 1.0f / sqrtf((float)hd);
  const int kg_d = kgroups_for(d), kg_ff = kgroups_for(4 * d);
  // cache warm-up as in the batch-1 step: GEMM i of the step also requests the weights of GEMM i + dist, in step order
  // [QKV_0, out_0, FFN1_0, FFN2_0, QKV_1, ..., FFN2_{L-1}, head], wrapping into the next step.  OFF by default
  // (VX_BATCH_PREFETCH=<dist> turns it on): at 32 slots it measured 528-531 us per step against 524-526 without
  // (profiles/r02_ab_batch_prefetch.log) - between two GEMMs of the batched step sit an attention kernel that streams 87 MB of
  // K / V and the LayerNorm, and the memory system is not idle as in the batch-1 step.
  static const int pf_dist = getenv("VX_BATCH_PREFETCH") ? atoi(getenv("VX_BATCH_PREFETCH")) : 0;
  struct PfW { const void* W; int N, K; };
  std::vector<PfW> seq;
  for (int li = 0; li < L; ++li) {
    const LayerW& l = e->ar_l[li];
    seq.push_back({l.in_w, 3 * d, d}); seq.push_back({l.out_w, d, d}); seq.push_back({l.w1, 4 * d, d}); seq.push_back({l.w2, d, 4 * d});
  }
  seq.push_back({W<void>(e, "ar_predict_layer.weight"), AR_VOCAB, d});
  auto warm = [&](BgemmArgs& g, int idx) {
    if (pf_dist <= 0) return;
    const PfW& n = seq[(idx + pf_dist) % seq.size()];
    bgemm_prefetch(g, n.W, n.N, n.K);
  };
  for (int li = 0; li < L; ++li) {
    const LayerW& l = e->ar_l[li];
    // LN1 (+ the FFN2 partial sums of the previous layer)
    const bool prev = li > 0;
    launch_ln_batch(e->bx, prev ? e->bpart : nullptr, kg_ff, prev ? e->ar_l[li - 1].b2 : nullptr, l.n1_g, l.n1_b, e->bh, B, d, s);
    BgemmArgs a{};
    a.st = e->bst; a.B = B; a.d = d; a.hd = hd; a.ctx_max = e->ctx_max;
    a.A = e->bh; a.W = (const bf16*)l.in_w; a.bias = l.in_b; a.N = 3 * d; a.K = d; a.kgroups = 1;
    a.q = e->bq; a.kv = e->bkv + (size_t)li * kv_layer; a.kv_slot_stride = e->bkv_slot; a.kv_v_offset = kv_layer / 2;
    warm(a, 4 * li);
    VXC(launch_bgemm<BE_QKV>(a, s));
    attn_batch_kernel<64><<<dim3(H, B), 256, 0, s>>>(e->bq, e->bkv + (size_t)li * kv_layer, e->bkv_slot, kv_layer / 2, e->bst,
                                                     e->ctx_max, d, scale, e->batt);
    BgemmArgs o{};
    o.st = e->bst; o.B = B;
    o.A = e->batt; o.W = (const bf16*)l.out_w; o.N = d; o.K = d; o.kgroups = kg_d; o.part = e->bpart;
    warm(o, 4 * li + 1);
    VXC(launch_bgemm<BE_PARTIAL>(o, s));
    launch_ln_batch(e->bx, e->bpart, kg_d, l.out_b, l.n2_g, l.n2_b, e->bh, B, d, s);
    BgemmArgs f{};
    f.st = e->bst; f.B = B;
    f.A = e->bh; f.W = (const bf16*)l.w1; f.bias = l.b1; f.N = 4 * d; f.K = d; f.kgroups = 1; f.f = e->bff;
    warm(f, 4 * li + 2);
    VXC(launch_bgemm<BE_RELU>(f, s));
    BgemmArgs g{};
    g.st = e->bst; g.B = B;
    g.A = e->bff; g.W = (const bf16*)l.w2; g.N = d; g.K = 4 * d; g.kgroups = kg_ff; g.part = e->bpart;
    warm(g, 4 * li + 3);
    VXC(launch_bgemm<BE_PARTIAL>(g, s));
  }
  launch_ln_batch(e->bx, e->bpart, kg_ff, e->ar_l[L - 1].b2, W<float>(e, "ar_decoder.norm.weight"),
                  W<float>(e, "ar_decoder.norm.bias"), e->bh, B, d, s);
  BgemmArgs hgm{};
  hgm.st = e->bst; hgm.B = B;
  hgm.A = e->bh; hgm.W = W<bf16>(e, "ar_predict_layer.weight"); hgm.N = AR_VOCAB; hgm.K = d; hgm.kgroups = 1;
  hgm.logits = e->blogits; hgm.logits_stride = LOGITS_CUR;
  hgm.trace = e->btrace; hgm.trace_rows = e->btok_stride;
  warm(hgm, 4 * L);
  VXC(launch_bgemm<BE_LOGITS>(hgm, s));
  return VX_OK;
}

extern "C" int vx_batch_decode(vx_engine* e, int32_t B, const vx_decode_params* params, void* stream) {
  if (!e || !params) return fail(VX_ERR_ARG, "null argument");
  if (B < 1 || B > e->bmax) return fail(VX_ERR_ARG, "n_slots %d outside [1, max_batch=%d]", B, e->bmax);
  const vx_config& c = e->cfg;
  ON_DEVICE(c.device);
  long long bound = 1;
  bool cap_limited[BMAX] = {};
  for (int b = 0; b < B; ++b) {
    const vx_decode_params& p = params[b];
    if (p.struct_size != (int32_t)sizeof(vx_decode_params)) return fail(VX_ERR_ARG, "vx_decode_params.struct_size mismatch");
    if (!e->bprefilled[b]) return fail(VX_ERR_STATE, "slot %d needs a fresh vx_batch_prefill", b);
    if (!(p.temperature > 0.f)) return fail(VX_ERR_ARG, "temperature must be > 0");
    long long max_tok = 16LL * e->bS[b] + 1 - e->bbos[b];
    if (p.forced) max_tok = p.n_forced;
    else if (p.max_new_tokens >= 0 && p.max_new_tokens < max_tok) max_tok = p.max_new_tokens;
    const long long room = (long long)c.max_audio - e->bbos[b] - e->bP[b];  // as in vx_ar_decode: clamp, fail only if it fills
    cap_limited[b] = false;
    if (max_tok > room) {
      if (p.forced) return fail(VX_ERR_CAPACITY, "slot %d needs %lld audio rows, capacity %d", b, e->bbos[b] + e->bP[b] + max_tok, c.max_audio);
      max_tok = room; cap_limited[b] = true;
    }
    const long long steps = max_tok + (p.forced ? 1 : 0);
    if (steps > bound) bound = steps;
    ArState& st = e->h_bst[b];
    st.top_k = p.top_k; st.temperature = p.temperature; st.max_new = cap_limited[b] ? (int)room : p.max_new_tokens;
    st.exp_noise = p.exp_noise; st.noise_rows = p.noise_rows; st.seed = p.seed;
    st.forced = p.forced ? (p.n_forced > 0 ? (const long long*)p.forced : (const long long*)e->btok) : nullptr;
    st.n_forced = p.forced ? p.n_forced : 0;
    if (p.exp_noise && p.noise_rows <= 0) return fail(VX_ERR_ARG, "noise_rows must be > 0");
  }
  VXC(sync_in(e, stream));
  HIPC(hipMemcpyAsync(e->bst, e->h_bst, (size_t)B * sizeof(ArState), hipMemcpyHostToDevice, e->es));
  const bool graph = !(c.flags & VX_FLAG_NO_GRAPH);
  hipGraphExec_t gx = nullptr;
  if (graph) {
    auto it = e->bgraphs.find(B);
    if (it == e->bgraphs.end()) {
      hipGraph_t gr = nullptr;
      HIPC(hipStreamBeginCapture(e->es, hipStreamCaptureModeThreadLocal));
      int r = enqueue_batch_step(e, B, e->es);
      hipError_t ce = hipStreamEndCapture(e->es, &gr);
      if (r != VX_OK) return r;
      HIPC(ce);
      HIPC(hipGraphInstantiate(&gx, gr, nullptr, nullptr, 0));
      (void)hipGraphDestroy(gr);
      e->bgraphs[B] = gx;
    } else {
      gx = it->second;
    }
  }
  HIPC(hipEventRecord(e->ev_t[2], e->es));
  long long launched = 0;
  int slot = 0;
  bool done = false, pending[2] = {false, false};
  auto all_done = [&](int sl) {
    const ArState* hs = e->h_bst + (size_t)(1 + sl) * BMAX;
    for (int b = 0; b < B; ++b) if (!hs[b].done) return false;
    return true;
  };
  while (!done) {
    const long long n = (bound - launched) < POLL_CHUNK ? (bound - launched) : POLL_CHUNK;
    for (long long i = 0; i < n; ++i) {
      if (graph) HIPC(hipGraphLaunch(gx, e->es));
      else VXC(enqueue_batch_step(e, B, e->es));
    }
    launched += n;
    HIPC(hipMemcpyAsync(e->h_bst + (size_t)(1 + slot) * BMAX, e->bst, (size_t)B * sizeof(ArState), hipMemcpyDeviceToHost, e->es));
    HIPC(hipEventRecord(e->ev_poll[slot], e->es));
    pending[slot] = true;
    const int other = slot ^ 1;
    if (pending[other]) {
      HIPC(hipEventSynchronize(e->ev_poll[other]));
      pending[other] = false;
      if (all_done(other)) done = true;
    }
    if (!done && launched >= bound) {
      HIPC(hipEventSynchronize(e->ev_poll[slot]));
      pending[slot] = false;
      if (!all_done(slot)) return fail(VX_ERR_STATE, "batched decode did not terminate within %lld steps", bound);
      done = true;
    }
    slot = other;
  }
  HIPC(hipEventRecord(e->ev_t[3], e->es));
  HIPC(hipMemcpyAsync(e->h_bst + BMAX, e->bst, (size_t)B * sizeof(ArState), hipMemcpyDeviceToHost, e->es));
  HIPC(hipStreamSynchronize(e->es));
  HIPC(hipGetLastError());
  float ms = 0.f;
  HIPC(hipEventElapsedTime(&ms, e->ev_t[2], e->ev_t[3]));
  e->t_bdecode = ms;
  e->n_blaunch = (double)launched;
  for (int b = 0; b < B; ++b) {
    e->bngen[b] = e->h_bst[BMAX + b].n_gen;
    e->breason[b] = e->h_bst[BMAX + b].stop_reason;
    e->bprefilled[b] = false;
  }
  VXC(sync_out(e, stream));
  for (int b = 0; b < B; ++b)
    if (cap_limited[b] && e->breason[b] == VX_STOP_MAX_NEW)
      return fail(VX_ERR_CAPACITY, "capacity exceeded in slot %d: the KV cache filled (max_audio = %d rows) before the stop rule fired; raise max_audio", b, c.max_audio);
  return VX_OK;
}

extern "C" int vx_batch_result(vx_engine* e, int32_t slot, int64_t* tokens, int32_t capacity, int32_t* n_tokens,
                               int32_t* stop_reason) {
  if (!e) return fail(VX_ERR_ARG, "null engine");
  if (slot < 0 || slot >= e->bmax) return fail(VX_ERR_ARG, "bad slot");
  ON_DEVICE(e->cfg.device);
  const int n = e->bngen[slot];
  if (n_tokens) *n_tokens = n;
  if (stop_reason) *stop_reason = e->breason[slot];
  if (tokens) {
    if (capacity < n) return fail(VX_ERR_CAPACITY, "token buffer too small (%d < %d)", capacity, n);
    std::vector<int> tmp(n);
    if (n) HIPC(hipMemcpy(tmp.data(), e->btok + (size_t)slot * e->btok_stride, (size_t)n * 4, hipMemcpyDeviceToHost));
    for (int i = 0; i < n; ++i) tokens[i] = tmp[i];
  }
  return VX_OK;
}

// ------------------------------------------------------------------------------ NAR
// forced (optional, (T, Q)): stage i's argmax is still what codes_out reports, but the embedding that feeds stage i+1 is
// taken from forced[:, i+1] - the input the reference itself gave that stage when `forced` are its codes (valle.py:1133-1134).
// stage_logits (optional, (Q-1, T, 1024) fp32, host or device): every stage's logits rows (valle.py:1128).
static int nar_impl(vx_engine* e, const int64_t* text_nar, int32_t S2, const int64_t* prompts, int32_t P,
                    const int64_t* ar_tokens, int32_t T, int64_t* codes_out, void* stream, bool pos_before_prenet,
                    const int64_t* forced = nullptr, float* stage_logits = nullptr) {
  if (!e || !text_nar || !ar_tokens || !codes_out || (P > 0 && !prompts)) return fail(VX_ERR_ARG, "null argument");
  if (!e->finalized) return fail(VX_ERR_STATE, "weights not finalized");
  const vx_config& c = e->cfg;
  const int Q = c.num_quantizers;
  if (S2 <= 0 || T <= 0 || P < 0) return fail(VX_ERR_ARG, "bad S2/P/T");
  if (S2 > c.max_text || P + T > c.max_audio) return fail(VX_ERR_CAPACITY, "S2=%d P+T=%d exceed capacity", S2, P + T);
  ON_DEVICE(c.device);
  VXC(sync_in(e, stream));
  HIPC(hipEventRecord(e->ev_t[4], e->es));
  const bool vf = e->vallf;  // VALL-F (valle.py:650-708): the stack runs over the audio rows, the NAR text is cross-attention memory
  const int dn = c.nar_d_model, A = P + T, N = vf ? A : S2 + A, tx = vf ? 0 : S2;  // tx = text rows in front of the audio rows in X
  // y = [prompt codebook 0 | AR tokens] (valle.py:1064-1066)
  if (P) HIPC(hipMemcpyAsync(e->ids_prompts, prompts, (size_t)P * Q * 8, hipMemcpyDefault, e->es));
  HIPC(hipMemcpyAsync(e->ids_samples, ar_tokens, (size_t)T * 8, hipMemcpyDefault, e->es));
  copy_col_kernel<<<(T + 255) / 256, 256, 0, e->es>>>(e->ids_samples, e->d_codes, T, Q, 0);
  if (forced) HIPC(hipMemcpyAsync(e->d_fcodes, forced, (size_t)T * Q * 8, hipMemcpyDefault, e->es));
  if (Q > 1) {
    HIPC(hipMemcpyAsync(e->ids_text, text_nar, (size_t)S2 * 8, hipMemcpyDefault, e->es));
    auto emb = [&](int j) { return W<float>(e, "nar_audio_embeddings." + std::to_string(j) + ".word_embeddings.weight"); };
    if (P) embed_accum_kernel<<<P, 256, 0, e->es>>>(e->ids_prompts, Q, 0, emb(0), 1025, dn, e->yemb, P, 1);
    embed_accum_kernel<<<T, 256, 0, e->es>>>(e->ids_samples, 1, 0, emb(0), 1025, dn, e->yemb + (size_t)P * dn, T, 1);
    if (c.prefix_mode != 0 && P)  // valle.py:1110-1113
      for (int j = 1; j < Q; ++j) embed_accum_kernel<<<P, 256, 0, e->es>>>(e->ids_prompts, Q, j, emb(j), 1024, dn, e->yemb, P, 0);
    const float* a_txt = W<float>(e, "nar_text_position.alpha");
    const float* a_aud = W<float>(e, "nar_audio_position.alpha");
    const bool prenet = c.flags & VX_FLAG_PRENET;
    if (prenet) {  // x = position(nar_text_prenet(embedding)) once (valle.py:1081-1083); kept in pn_text for every stage
      embed_accum_kernel<<<S2, 256, 0, e->es>>>(e->ids_text, 1, 0, W<float>(e, "nar_text_embedding.word_embeddings.weight"), 512, dn, e->pn_a, S2, 1);
      VXC(text_prenet_rows(e, 1, e->pn_a, e->pn_a, S2, dn));
      add_pos_kernel<<<S2, 256, 0, e->es>>>(e->pn_a, dn, a_txt, e->pe_nar, 0, e->pn_text, S2);
    }
    if (vf) {  // the text memory's K / V per layer, once for all stages (same memory and weights in every stage, valle.py:664-688)
      if (prenet) HIPC(hipMemcpyAsync(e->X, e->pn_text, (size_t)S2 * dn * 4, hipMemcpyDeviceToDevice, e->es));
      else embed_pos_kernel<<<S2, 256, 0, e->es>>>(e->ids_text, 1, 0, W<float>(e, "nar_text_embedding.word_embeddings.weight"), 512, dn,
                                                   a_txt, e->pe_nar, 0, e->X, S2);
      VXC(cast_rows(e, e->X, e->Hn, (size_t)S2 * dn));
      VXC(memory_kv(e, e->nar_l, e->xkv_nar, S2, dn, c.nar_nhead));
    }
    for (int i = 0; i < Q - 1; ++i) {
      if (prenet) {
        if (!vf) HIPC(hipMemcpyAsync(e->X, e->pn_text, (size_t)S2 * dn * 4, hipMemcpyDeviceToDevice, e->es));
        if (pos_before_prenet) {  // VALLE.continual, prefix mode 0 (valle.py:1193-1194)
          add_pos_kernel<<<A, 256, 0, e->es>>>(e->yemb, dn, a_aud, e->pe_nar, 0, e->pn_a, A);
          VXC(audio_prenet_rows(e, 1, e->pn_a, e->X + (size_t)tx * dn, A, dn));
        } else {  // valle.py:1092-1093, 1121-1122
          VXC(audio_prenet_rows(e, 1, e->yemb, e->pn_b, A, dn));
          add_pos_kernel<<<A, 256, 0, e->es>>>(e->pn_b, dn, a_aud, e->pe_nar, 0, e->X + (size_t)tx * dn, A);
        }
      } else {
        if (!vf) embed_pos_kernel<<<S2, 256, 0, e->es>>>(e->ids_text, 1, 0, W<float>(e, "nar_text_embedding.word_embeddings.weight"), 512, dn,
                                                         a_txt, e->pe_nar, 0, e->X, S2);
        add_pos_kernel<<<A, 256, 0, e->es>>>(e->yemb, dn, a_aud, e->pe_nar, 0, e->X + (size_t)tx * dn, A);
      }
      if (vf) VXC(run_stack_f(e, e->nar_l, N, dn, c.nar_nhead, -1, i, false, e->xkv_nar, S2));
      else VXC(run_stack(e, e->nar_l, N, dn, c.nar_nhead, -1, i, false));
      // final AdaLN + predict layer on the T generated rows only (valle.py:1128)
      if (c.flags & VX_FLAG_POST_NORM) {  // no final norm (valle.py:242-246): the rows are already norm2'd
        VXC(cast_rows(e, e->X + (size_t)(tx + P) * dn, e->Hn, (size_t)T * dn));
      } else {
        const float* fw = ada_vec(e, i, e->npl * c.nar_num_layers);
        VXC(ln_rows(e, e->X + (size_t)(tx + P) * dn, W<float>(e, "nar_decoder.norm.norm.weight"),
                    W<float>(e, "nar_decoder.norm.norm.bias"), fw, fw + dn, e->Hn, T, dn));
      }
      VXC(gemm_rows(e, e->Hn, W<void>(e, "nar_predict_layers." + std::to_string(i) + ".weight"), nullptr, e->nar_logits,
                    T, 1024, dn, GE_PLAIN, true));
      argmax_rows_kernel<<<(T + 3) / 4, 256, 0, e->es>>>(e->nar_logits, 1024, T, e->ids_samples, e->d_codes, Q, i + 1);
      if (stage_logits) HIPC(hipMemcpyAsync(stage_logits + (size_t)i * T * 1024, e->nar_logits, (size_t)T * 1024 * 4, hipMemcpyDefault, e->es));
      if (forced && i < Q - 2) pick_col_kernel<<<(T + 255) / 256, 256, 0, e->es>>>(e->d_fcodes, Q, i + 1, e->ids_samples, T);
      if (i < Q - 2) {  // valle.py:1104-1108 / 1133-1134
        if (c.prefix_mode == 0 && P)
          embed_accum_kernel<<<P, 256, 0, e->es>>>(e->ids_prompts, Q, i + 1, emb(i + 1), 1024, dn, e->yemb, P, 0);
        embed_accum_kernel<<<T, 256, 0, e->es>>>(e->ids_samples, 1, 0, emb(i + 1), 1024, dn, e->yemb + (size_t)P * dn, T, 0);
      }
    }
  }
  HIPC(hipGetLastError());
  HIPC(hipEventRecord(e->ev_t[5], e->es));
  HIPC(hipMemcpyAsync(codes_out, e->d_codes, (size_t)T * Q * 8, hipMemcpyDefault, e->es));
  HIPC(hipStreamSynchronize(e->es));
  float ms = 0.f;
  HIPC(hipEventElapsedTime(&ms, e->ev_t[4], e->ev_t[5]));
  e->t_nar = ms;
  gemm_collect(e);
  e->last_T = T; e->last_N = N;
  VXC(sync_out(e, stream));
  return VX_OK;
}

extern "C" int vx_nar(vx_engine* e, const int64_t* text_nar, int32_t S2, const int64_t* prompts, int32_t P,
                      const int64_t* ar_tokens, int32_t T, int64_t* codes_out, void* stream) {
  return nar_impl(e, text_nar, S2, prompts, P, ar_tokens, T, codes_out, stream, false);
}

extern "C" int vx_nar_ex(vx_engine* e, const int64_t* text_nar, int32_t S2, const int64_t* prompts, int32_t P,
                         const int64_t* ar_tokens, int32_t T, int64_t* codes_out, const int64_t* forced_codes,
                         float* stage_logits, int32_t continual, void* stream) {
  return nar_impl(e, text_nar, S2, prompts, P, ar_tokens, T, codes_out, stream, continual && e && e->cfg.prefix_mode == 0,
                  forced_codes, stage_logits);
}

extern "C" int vx_nar_continual(vx_engine* e, const int64_t* text_nar, int32_t S2, const int64_t* prompts, int32_t P,
                                const int64_t* ar_tokens, int32_t T, int64_t* codes_out, void* stream) {
  return nar_impl(e, text_nar, S2, prompts, P, ar_tokens, T, codes_out, stream, e && e->cfg.prefix_mode == 0);
}

// Row buffers are sized for one utterance at vx_create; the batched NAR concatenates up to max_batch of them.
static int ensure_rows(vx_engine* e, size_t rows, size_t audio_rows, size_t text_rows) {
  const vx_config& c = e->cfg;
  const size_t dmax = c.d_model > c.nar_d_model ? c.d_model : c.nar_d_model;
  if (rows <= (size_t)e->n_max && audio_rows <= e->cap_audio && text_rows <= e->cap_text) return VX_OK;
  if (rows < (size_t)e->n_max) rows = e->n_max;
  if (audio_rows < e->cap_audio) audio_rows = e->cap_audio;
  if (text_rows < e->cap_text) text_rows = e->cap_text;
  e->cap_audio = audio_rows; e->cap_text = text_rows;
  HIPC(hipStreamSynchronize(e->es));
  auto regrow = [&](void** p, size_t bytes) -> int {
    for (auto& q : e->allocs) if (q == *p) { (void)hipFree(q); q = nullptr; }
    HIPC(hipMalloc(p, bytes));
    if (poison_on()) { HIPC(hipMemsetAsync(*p, 0xFF, bytes, e->es)); HIPC(hipStreamSynchronize(e->es)); }
    e->allocs.push_back(*p);
    return VX_OK;
  };
  e->n_max = (int)rows;
  e->vt_ld = (int)(((rows + 63) / 64) * 64 + 64);
  VXC(regrow((void**)&e->X, rows * dmax * 4));
  VXC(regrow(&e->Hn, rows * dmax * e->esz));
  VXC(regrow(&e->QKV, rows * 3 * dmax * e->esz));
  VXC(regrow(&e->ATT, rows * dmax * e->esz));
  HIPC(hipMemsetAsync(e->ATT, 0, rows * dmax * e->esz, e->es));  // padding rows between segments are never written: keep them finite
  VXC(regrow(&e->FF, rows * 4 * dmax * e->esz));
  VXC(regrow(&e->VT, dmax * (size_t)e->vt_ld * 2));
  HIPC(hipMemsetAsync(e->VT, 0, dmax * (size_t)e->vt_ld * 2, e->es));
  HIPC(hipMemsetAsync(e->X, 0, rows * dmax * 4, e->es));
  if (e->fp8nar) {
    e->mx_ld = (int)((rows + 255) / 256 * 256);
    VXC(regrow((void**)&e->Hn8, rows * dmax));
    VXC(regrow((void**)&e->FF8, rows * 4 * dmax));
    VXC(regrow((void**)&e->SHn, (dmax / 32) * (size_t)e->mx_ld));
    VXC(regrow((void**)&e->SFF, (4 * dmax / 32) * (size_t)e->mx_ld));
    HIPC(hipMemsetAsync(e->SHn, 0, (dmax / 32) * (size_t)e->mx_ld, e->es));
    HIPC(hipMemsetAsync(e->SFF, 0, (4 * dmax / 32) * (size_t)e->mx_ld, e->es));
  }
  if (e->slab != nullptr) {  // keep the split-K path available for concatenated rows below the 256^2 threshold
    e->slab_rows = rows < 4095 ? (int)rows : 4095;
    VXC(regrow((void**)&e->slab, (size_t)4 * e->slab_rows * dmax * 4));
  }
  VXC(regrow((void**)&e->yemb, audio_rows * dmax * 4));
  VXC(regrow((void**)&e->nar_logits, audio_rows * 1024 * 4));
  VXC(regrow((void**)&e->ids_text, text_rows * 8));
  VXC(regrow((void**)&e->ids_prompts, audio_rows * 8 * 8));
  VXC(regrow((void**)&e->ids_samples, audio_rows * 8));
  VXC(regrow((void**)&e->d_codes, audio_rows * 8 * 8));
  VXC(regrow((void**)&e->d_fcodes, audio_rows * 8 * 8));
  return VX_OK;
}

// The NAR stages of n utterances at once (valle.py:1063-1134 per utterance): rows of all utterances are
// concatenated (each segment starts at a multiple of 64 rows), so the GEMMs run at M ~ n x 1k rows where the
// MFMA kernels are efficient, and attention runs per segment in one launch.
static int nar_batch_impl(vx_engine* e, int32_t n, const int64_t* const* text_nar, const int32_t* S2,
                          const int64_t* const* prompts, const int32_t* P, const int64_t* const* ar_tokens,
                          const int32_t* T, int64_t* const* codes_out, const int64_t* const* forced, void* stream) {
  if (e && e->vallf) return fail(VX_ERR_UNSUPPORTED, "vx_nar_batch: VALL-F runs on the batch-1 path only");
  if (!e || !text_nar || !S2 || !prompts || !P || !ar_tokens || !T || !codes_out) return fail(VX_ERR_ARG, "null argument");
  if (!e->finalized) return fail(VX_ERR_STATE, "weights not finalized");
  const vx_config& c = e->cfg;
  const int Q = c.num_quantizers, dn = c.nar_d_model;
  if (n < 1 || n > BMAX) return fail(VX_ERR_ARG, "n must be 1..%d", BMAX);
  if (!e->bf16 || !use_mfma(e) || Q < 2) return fail(VX_ERR_UNSUPPORTED, "vx_nar_batch needs bf16 MFMA rows and num_quantizers > 1");
  if (c.flags & VX_FLAG_PRENET) return fail(VX_ERR_UNSUPPORTED, "vx_nar_batch: prenet models run on the batch-1 path only");
  ON_DEVICE(c.device);
  std::vector<int> start(n), len(n), aoff(n), toff(n), soff(n);
  size_t rows = 0, arows = 0, trows = 0, srows = 0;
  int maxlen = 0;
  for (int b = 0; b < n; ++b) {
    if (S2[b] <= 0 || T[b] <= 0 || P[b] < 0 || !text_nar[b] || !ar_tokens[b] || !codes_out[b] || (P[b] > 0 && !prompts[b]) ||
        (forced && !forced[b]))
      return fail(VX_ERR_ARG, "bad utterance %d", b);
    // per-utterance limits: positions index the sine table (pe_rows rows) separately for text and audio
    if (S2[b] > e->pe_rows || P[b] + T[b] > e->pe_rows)
      return fail(VX_ERR_CAPACITY, "utterance %d: S2=%d / P+T=%d exceed the %d positions of the sine table", b, S2[b], P[b] + T[b], e->pe_rows);
    start[b] = (int)rows; len[b] = S2[b] + P[b] + T[b];
    aoff[b] = (int)arows; toff[b] = (int)trows; soff[b] = (int)srows;
    rows += (size_t)((len[b] + 63) / 64) * 64;
    arows += P[b] + T[b]; trows += T[b]; srows += S2[b];
    if (len[b] > maxlen) maxlen = len[b];
  }
  VXC(ensure_rows(e, rows, arows, srows));
  VXC(sync_in(e, stream));
  HIPC(hipEventRecord(e->ev_t[4], e->es));
  if (!e->d_seg_start) {  // engines created with max_batch == 1 may still batch their NAR stages
    HIPC(hipMalloc((void**)&e->d_seg_start, BMAX * sizeof(int)));
    HIPC(hipMalloc((void**)&e->d_seg_len, BMAX * sizeof(int)));
    e->allocs.push_back(e->d_seg_start); e->allocs.push_back(e->d_seg_len);
  }
  HIPC(hipMemcpyAsync(e->d_seg_start, start.data(), n * sizeof(int), hipMemcpyHostToDevice, e->es));
  HIPC(hipMemcpyAsync(e->d_seg_len, len.data(), n * sizeof(int), hipMemcpyHostToDevice, e->es));
  HIPC(hipMemsetAsync(e->X, 0, rows * (size_t)dn * 4, e->es));  // padding rows must stay finite (they feed V^T columns)
  auto emb = [&](int j) { return W<float>(e, "nar_audio_embeddings." + std::to_string(j) + ".word_embeddings.weight"); };
  for (int b = 0; b < n; ++b) {
    long long* idp = e->ids_prompts + (size_t)aoff[b] * Q;   // (P_b, Q) rows; region sized for P+T rows per utterance
    long long* ids = e->ids_samples + toff[b];
    if (P[b]) HIPC(hipMemcpyAsync(idp, prompts[b], (size_t)P[b] * Q * 8, hipMemcpyDefault, e->es));
    HIPC(hipMemcpyAsync(ids, ar_tokens[b], (size_t)T[b] * 8, hipMemcpyDefault, e->es));
    if (forced) HIPC(hipMemcpyAsync(e->d_fcodes + (size_t)toff[b] * Q, forced[b], (size_t)T[b] * Q * 8, hipMemcpyDefault, e->es));
    HIPC(hipMemcpyAsync(e->ids_text + soff[b], text_nar[b], (size_t)S2[b] * 8, hipMemcpyDefault, e->es));
    copy_col_kernel<<<(T[b] + 255) / 256, 256, 0, e->es>>>(ids, e->d_codes + (size_t)toff[b] * Q, T[b], Q, 0);
    float* ye = e->yemb + (size_t)aoff[b] * dn;
    if (P[b]) embed_accum_kernel<<<P[b], 256, 0, e->es>>>(idp, Q, 0, emb(0), 1025, dn, ye, P[b], 1);
    embed_accum_kernel<<<T[b], 256, 0, e->es>>>(ids, 1, 0, emb(0), 1025, dn, ye + (size_t)P[b] * dn, T[b], 1);
    if (c.prefix_mode != 0 && P[b])
      for (int j = 1; j < Q; ++j) embed_accum_kernel<<<P[b], 256, 0, e->es>>>(idp, Q, j, emb(j), 1024, dn, ye, P[b], 0);
  }
  const float* a_txt = W<float>(e, "nar_text_position.alpha");
  const float* a_aud = W<float>(e, "nar_audio_position.alpha");
  e->nseg = n; e->max_seg_len = maxlen;
  int rc = VX_OK;
  for (int i = 0; i < Q - 1 && rc == VX_OK; ++i) {
    for (int b = 0; b < n; ++b) {
      float* xb = e->X + (size_t)start[b] * dn;
      embed_pos_kernel<<<S2[b], 256, 0, e->es>>>(e->ids_text + soff[b], 1, 0, W<float>(e, "nar_text_embedding.word_embeddings.weight"),
                                                 512, dn, a_txt, e->pe_nar, 0, xb, S2[b]);
      add_pos_kernel<<<P[b] + T[b], 256, 0, e->es>>>(e->yemb + (size_t)aoff[b] * dn, dn, a_aud, e->pe_nar, 0,
                                                     xb + (size_t)S2[b] * dn, P[b] + T[b]);
    }
    rc = run_stack(e, e->nar_l, (int)rows, dn, c.nar_nhead, -1, i, false);
    if (rc != VX_OK) break;
    const bool post = c.flags & VX_FLAG_POST_NORM;
    const float* fw = post ? nullptr : ada_vec(e, i, e->npl * c.nar_num_layers);
    for (int b = 0; b < n && rc == VX_OK; ++b) {  // final AdaLN (pre-norm only) on the generated rows, compacted to [sum T][dn]
      const float* xr = e->X + (size_t)(start[b] + S2[b] + P[b]) * dn;
      bf16* hr = (bf16*)e->Hn + (size_t)toff[b] * dn;
      rc = post ? cast_rows(e, xr, hr, (size_t)T[b] * dn)
                : ln_rows(e, xr, W<float>(e, "nar_decoder.norm.norm.weight"), W<float>(e, "nar_decoder.norm.norm.bias"), fw,
                          fw + dn, hr, T[b], dn);
    }
    if (rc != VX_OK) break;
    const int nseg_keep = e->nseg;
    e->nseg = 0;  // the predict GEMM below is a plain GEMM
    rc = gemm_rows(e, e->Hn, W<void>(e, "nar_predict_layers." + std::to_string(i) + ".weight"), nullptr, e->nar_logits,
                   (int)trows, 1024, dn, GE_PLAIN, true);
    e->nseg = nseg_keep;
    if (rc != VX_OK) break;
    argmax_rows_kernel<<<((int)trows + 3) / 4, 256, 0, e->es>>>(e->nar_logits, 1024, (int)trows, e->ids_samples, e->d_codes, Q, i + 1);
    if (forced && i < Q - 2)  // teacher forcing: the next stage sees the caller's codes of this stage (all segments at once)
      pick_col_kernel<<<((int)trows + 255) / 256, 256, 0, e->es>>>(e->d_fcodes, Q, i + 1, e->ids_samples, (int)trows);
    if (i < Q - 2)
      for (int b = 0; b < n; ++b) {
        float* ye = e->yemb + (size_t)aoff[b] * dn;
        if (c.prefix_mode == 0 && P[b])
          embed_accum_kernel<<<P[b], 256, 0, e->es>>>(e->ids_prompts + (size_t)aoff[b] * Q, Q, i + 1, emb(i + 1), 1024, dn, ye, P[b], 0);
        embed_accum_kernel<<<T[b], 256, 0, e->es>>>(e->ids_samples + toff[b], 1, 0, emb(i + 1), 1024, dn, ye + (size_t)P[b] * dn, T[b], 0);
      }
  }
  e->nseg = 0;
  VXC(rc);
  HIPC(hipGetLastError());
  HIPC(hipEventRecord(e->ev_t[5], e->es));
  for (int b = 0; b < n; ++b)
    HIPC(hipMemcpyAsync(codes_out[b], e->d_codes + (size_t)toff[b] * Q, (size_t)T[b] * Q * 8, hipMemcpyDefault, e->es));
  HIPC(hipStreamSynchronize(e->es));
  float ms = 0.f;
  HIPC(hipEventElapsedTime(&ms, e->ev_t[4], e->ev_t[5]));
  e->t_nar = ms;
  gemm_collect(e);
  e->last_T = (int)trows; e->last_N = (int)rows;
  VXC(sync_out(e, stream));
  return VX_OK;
}

extern "C" int vx_nar_batch(vx_engine* e, int32_t n, const int64_t* const* text_nar, const int32_t* S2,
                            const int64_t* const* prompts, const int32_t* P, const int64_t* const* ar_tokens,
                            const int32_t* T, int64_t* const* codes_out, void* stream) {
  return nar_batch_impl(e, n, text_nar, S2, prompts, P, ar_tokens, T, codes_out, nullptr, stream);
}
extern "C" int vx_nar_batch_ex(vx_engine* e, int32_t n, const int64_t* const* text_nar, const int32_t* S2,
                               const int64_t* const* prompts, const int32_t* P, const int64_t* const* ar_tokens,
                               const int32_t* T, int64_t* const* codes_out, const int64_t* const* forced_codes, void* stream) {
  return nar_batch_impl(e, n, text_nar, S2, prompts, P, ar_tokens, T, codes_out, forced_codes, stream);
}

extern "C" int vx_get_timings(vx_engine* e, double* out, int32_t n) {
  if (!e || !out) return fail(VX_ERR_ARG, "null argument");
  const double v[9] = {e->t_prefill, e->t_decode, e->t_nar, (double)e->n_pass, e->n_launch, e->t_bdecode, e->n_blaunch,
                       e->t_gemm, e->gemm_flops_done};
  for (int i = 0; i < n && i < 9; ++i) out[i] = v[i];
  return VX_OK;
}

extern "C" int vx_read_buffer(vx_engine* e, const char* name, void* dst, int64_t off, int64_t nbytes) {
  if (!e || !name || !dst) return fail(VX_ERR_ARG, "null argument");
  ON_DEVICE(e->cfg.device);
  HIPC(hipStreamSynchronize(e->es));
  const std::string n = name;
  const char* src = nullptr;
  int64_t size = 0;
  const bool trace = e->cfg.flags & VX_FLAG_TRACE_LOGITS;
  if (n == "ar_logits") {
    src = (const char*)(trace ? e->ar_logits + LOGITS_CUR : e->ar_logits);
    size = (int64_t)(trace ? e->n_pass : 1) * AR_VOCAB * 4;
  }
  else if (n == "ar_sampled") { src = (const char*)e->d_sampled; size = (int64_t)e->n_pass * 4; }
  else if (n == "ar_argmax") { src = (const char*)e->d_argmax; size = (int64_t)e->n_pass * 4; }
  else if (n == "nar_logits") { src = (const char*)e->nar_logits; size = (int64_t)e->last_T * 1024 * 4; }
  else if (n == "ar_x") { src = (const char*)e->ar_x; size = (int64_t)e->cfg.d_model * 4; }
  else if (n == "nar_x") { src = (const char*)e->X; size = (int64_t)e->last_N * e->cfg.nar_d_model * 4; }
  else if (n == "batch_trace" && e->btrace) { src = (const char*)e->btrace; size = (int64_t)e->bmax * e->btok_stride * AR_VOCAB * 4; }
  else if (n == "batch_logits" && e->bmax > 1) { src = (const char*)e->blogits; size = (int64_t)BMAX * LOGITS_CUR * 4; }
  else if (n == "batch_argmax" && e->bmax > 1) { src = (const char*)e->bargm; size = (int64_t)BMAX * e->btok_stride * 4; }
  else if (n == "batch_sampled" && e->bmax > 1) { src = (const char*)e->bsamp; size = (int64_t)BMAX * e->btok_stride * 4; }
  else return fail(VX_ERR_ARG, "unknown buffer '%s'", name);
  if (off < 0 || nbytes < 0 || off + nbytes > size) return fail(VX_ERR_ARG, "read of '%s' out of range (%lld+%lld > %lld)", name, (long long)off, (long long)nbytes, (long long)size);
  HIPC(hipStreamSynchronize(e->es));  // the copy below runs on the null stream, which the engine's non-blocking stream does not order with
  HIPC(hipMemcpy(dst, src + off, (size_t)nbytes, hipMemcpyDeviceToHost));
  return VX_OK;
}

// ------------------------------------------------------------------------------ kernel-level ops
extern "C" int vx_op_convert_bf16(const float* src, void* dst, int64_t n, void* stream) {
  convert_kernel<bf16><<<1024, 256, 0, (hipStream_t)stream>>>(src, (bf16*)dst, (size_t)n);
  HIPC(hipGetLastError());
  return VX_OK;
}

extern "C" int vx_op_layernorm(int32_t prec, const float* x, const float* gamma, const float* beta, const float* ada_w,
                               const float* ada_b, void* out, int32_t rows, int32_t d, void* stream) {
  if (d % 4 || d > 2048) return fail(VX_ERR_UNSUPPORTED, "layernorm: d=%d", d);
  hipStream_t s = (hipStream_t)stream;
  if (d <= 1024) {
    if (prec == VX_PREC_BF16) layernorm_rows_kernel<bf16, 4><<<(rows + 3) / 4, 256, 0, s>>>(x, gamma, beta, ada_w, ada_b, (bf16*)out, rows, d);
    else layernorm_rows_kernel<float, 4><<<(rows + 3) / 4, 256, 0, s>>>(x, gamma, beta, ada_w, ada_b, (float*)out, rows, d);
  } else {
    if (prec == VX_PREC_BF16) layernorm_rows_kernel<bf16, 8><<<(rows + 3) / 4, 256, 0, s>>>(x, gamma, beta, ada_w, ada_b, (bf16*)out, rows, d);
    else layernorm_rows_kernel<float, 8><<<(rows + 3) / 4, 256, 0, s>>>(x, gamma, beta, ada_w, ada_b, (float*)out, rows, d);
  }
  HIPC(hipGetLastError());
  return VX_OK;
}

extern "C" int vx_op_gemv(int32_t prec, const void* Wp, const float* bias, const float* x, float* y, int32_t N, int32_t K,
                          int32_t relu, void* stream) {
  GemvArgs a{};
  a.kid = -1;
  a.W = Wp; a.bias = bias; a.x = x; a.y = y; a.N = N; a.K = K;
  a.pro = PRO_COPY; a.epi = relu ? EPI_RELU : (bias ? EPI_BIAS : EPI_PLAIN);
  int dev = 0, cu = 256;
  (void)hipGetDevice(&dev);
  (void)hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, dev);
  VXC(launch_gemv(prec == VX_PREC_BF16, a, cu, (hipStream_t)stream));
  HIPC(hipGetLastError());
  return VX_OK;
}

extern "C" int vx_op_gemm(int32_t prec, int32_t mfma, const void* A, const void* Wp, const float* bias, float* C, int32_t M,
                          int32_t N, int32_t K, int32_t relu, void* stream) {
  const int epi = relu ? GE_RELU : (bias ? GE_BIAS : GE_PLAIN);
  hipStream_t s = (hipStream_t)stream;
  if (prec == VX_PREC_BF16) VXC(gemm_rows_t<bf16>(mfma != 0, (const bf16*)A, (const bf16*)Wp, bias, C, M, N, K, epi, true, s));
  else {
    if (mfma) return fail(VX_ERR_UNSUPPORTED, "MFMA GEMM is bf16 only");
    VXC(gemm_rows_t<float>(false, (const float*)A, (const float*)Wp, bias, C, M, N, K, epi, true, s));
  }
  HIPC(hipGetLastError());
  return VX_OK;
}

// The row path's own GEMM forms on caller data (bf16 output + V^T copy, fp32 residual update): what run_stack launches for QKV /
// FFN1 and for the out-projection / FFN2 at M >= 4096.
extern "C" int vx_op_gemm_rows(int32_t form, const void* A, const void* Wp, const float* bias, void* C, int32_t M, int32_t N,
                               int32_t K, int32_t relu, void* vt, int32_t vt_n0, int32_t vt_ld, void* stream) {
  if (!A || !Wp || !C || !bias || M < 1 || N < 1 || K < 1) return fail(VX_ERR_ARG, "gemm_rows: null operand or empty shape");
  if (form != 0 && form != 1) return fail(VX_ERR_ARG, "gemm_rows: form %d", form);
  if (vt && (form != 0 || vt_n0 < 0 || vt_n0 >= N || vt_n0 % 64 != 0 || vt_ld < M))
    return fail(VX_ERR_ARG, "gemm_rows: V^T copy needs form 0, 0 <= vt_n0 < N, vt_n0 %% 64 == 0 (the kernels test whole column groups), vt_ld >= M");
  hipStream_t s = (hipStream_t)stream;
  if (mfma_gemm_dispatch((const bf16*)A, (const bf16*)Wp, bias, C, M, N, K, form == 1 ? GE_RESID : (relu ? GE_RELU : GE_BIAS), form == 1, s,
                         (bf16*)vt, vt_n0, vt_ld))
    return fail(VX_ERR_UNSUPPORTED, "gemm_rows: no kernel instance");
  HIPC(hipGetLastError());
  return VX_OK;
}

// MXFP8 GEMM of the NAR stages (mx_kernels.hpp) on caller-supplied fp32 operands: A (M, K) and W (N, K) are quantised on the
// device exactly as the engine quantises activations / weights, then multiplied by mx256_kernel.  out_mode 0: C (M, N) fp32
// [bias / ReLU]; 2: the FFN1 form - C as e4m3 bytes (M, N) in c_out and its E8M0 block scales (N/32, ld) in sc_out, ld = M
// rounded up to 256.  qa_out / sa_out (optional): the quantised A bytes (M, K) and scales (K/32, ld), so that the quantiser
// itself can be compared bit for bit with the host emulation (tests/mx_ref.py).
extern "C" int vx_op_gemm_mx(const float* A, const float* Wt, const float* bias, void* c_out, void* sc_out, int32_t M, int32_t N,
                             int32_t K, int32_t relu, int32_t out_mode, void* qa_out, void* sa_out, void* stream) {
  if (!A || !Wt || !c_out || M < 1 || N % 256 || K % 128) return fail(VX_ERR_UNSUPPORTED, "mx gemm: M=%d N=%d K=%d (N %% 256, K %% 128)", M, N, K);
  if (out_mode == 2 && !sc_out) return fail(VX_ERR_ARG, "mx gemm: out_mode 2 needs sc_out");  // arguments first, allocations after
  hipStream_t s = (hipStream_t)stream;
  const int ld = (M + 255) / 256 * 256;
  struct Scratch {  // freed on every exit path
    uint8_t *qa = nullptr, *sa = nullptr, *qw = nullptr, *sw = nullptr;
    ~Scratch() { (void)hipFree(qa); (void)hipFree(sa); (void)hipFree(qw); (void)hipFree(sw); }
  } t;
  HIPC(hipMalloc((void**)&t.qa, (size_t)M * K)); HIPC(hipMalloc((void**)&t.sa, (size_t)(K / 32) * ld));
  HIPC(hipMalloc((void**)&t.qw, (size_t)N * K)); HIPC(hipMalloc((void**)&t.sw, (size_t)(K / 32) * N));
  HIPC(hipMemsetAsync(t.sa, 0, (size_t)(K / 32) * ld, s));
  mx_quant_rows_kernel<<<(M + 3) / 4, 256, 0, s>>>(A, t.qa, t.sa, M, K, ld);
  mx_quant_rows_kernel<<<(N + 3) / 4, 256, 0, s>>>(Wt, t.qw, t.sw, N, K, N);
  HIPC(hipGetLastError());  // a quantiser launch failure is reported as such, not as the GEMM's
  int rc;
  if (out_mode == 2) {
    HIPC(hipMemsetAsync(sc_out, 0, (size_t)(N / 32) * ld, s));
    rc = mx_gemm_dispatch(t.qa, t.sa, ld, t.qw, t.sw, N, bias, c_out, (uint8_t*)sc_out, ld, M, N, K, GE_RELU, MX_OUT_MX, s);
  } else {
    rc = mx_gemm_dispatch(t.qa, t.sa, ld, t.qw, t.sw, N, bias, c_out, nullptr, 0, M, N, K, relu ? GE_RELU : (bias ? GE_BIAS : GE_PLAIN), MX_OUT_F32, s);
  }
  hipError_t le = hipGetLastError();
  if (rc == 0 && le == hipSuccess) {
    if (qa_out) le = hipMemcpyAsync(qa_out, t.qa, (size_t)M * K, hipMemcpyDefault, s);
    if (le == hipSuccess && sa_out) le = hipMemcpyAsync(sa_out, t.sa, (size_t)(K / 32) * ld, hipMemcpyDefault, s);
  }
  const hipError_t se = hipStreamSynchronize(s);  // the scratch is in use until the stream has drained, whatever happened
  if (rc) return fail(VX_ERR_UNSUPPORTED, "mx gemm: no kernel instance (rc %d)", rc);
  HIPC(le);
  HIPC(se);
  return VX_OK;
}

// (Adaptive)LayerNorm with an MXFP8 result (layernorm_rows_mx_kernel): q_out (rows, d) e4m3 bytes, s_out (d/32, ld) E8M0 bytes,
// ld = rows rounded up to 256 (pad entries zeroed).
extern "C" int vx_op_layernorm_mx(const float* x, const float* gamma, const float* beta, const float* ada_w, const float* ada_b,
                                  void* q_out, void* s_out, int32_t rows, int32_t d, void* stream) {
  if (d % 32 || d > 1024 || rows < 1) return fail(VX_ERR_UNSUPPORTED, "layernorm_mx: d=%d", d);
  hipStream_t s = (hipStream_t)stream;
  const int ld = (rows + 255) / 256 * 256;
  HIPC(hipMemsetAsync(s_out, 0, (size_t)(d / 32) * ld, s));
  layernorm_rows_mx_kernel<4><<<(rows + 3) / 4, 256, 0, s>>>(x, gamma, beta, ada_w, ada_b, (uint8_t*)q_out, (uint8_t*)s_out, rows, d, ld);
  HIPC(hipGetLastError());
  return VX_OK;
}

extern "C" int vx_op_attention(int32_t prec, int32_t mfma, const void* qkv, void* out, int32_t rows, int32_t nhead, int32_t hd,
                               int32_t text_len, void* stream) {
  if (hd != 64) return fail(VX_ERR_UNSUPPORTED, "attention: head_dim %d", hd);
  hipStream_t s = (hipStream_t)stream;
  const int d = nhead * hd;
  const float scale = 1.0f / sqrtf((float)hd);
  dim3 grid((rows + 63) / 64, nhead);
  if (prec == VX_PREC_BF16) {
    if (mfma) {
      const int vt_ld = ((rows + 63) / 64) * 64 + 64;
      bf16* vt = nullptr;
      HIPC(hipMalloc((void**)&vt, (size_t)d * vt_ld * 2));
      vt_from_qkv_kernel<<<dim3((vt_ld + 255) / 256, d), 256, 0, s>>>((const bf16*)qkv, vt, rows, d, vt_ld);
      if (mfma_attn_dispatch((const bf16*)qkv, vt, vt_ld, (bf16*)out, rows, d, nhead, text_len, s) != 0) {
        (void)hipFree(vt);
        return fail(VX_ERR_UNSUPPORTED, "attention: rows x 3 d or d x vt_ld exceed 4 GB");
      }
      HIPC(hipStreamSynchronize(s));
      HIPC(hipFree(vt));
    }
    else attn_rows_simple_kernel<bf16, 64><<<grid, 256, 0, s>>>((const bf16*)qkv, (bf16*)out, rows, d, text_len, scale);
  } else {
    if (mfma) return fail(VX_ERR_UNSUPPORTED, "MFMA attention is bf16 only");
    attn_rows_simple_kernel<float, 64><<<grid, 256, 0, s>>>((const float*)qkv, (float*)out, rows, d, text_len, scale);
  }
  HIPC(hipGetLastError());
  return VX_OK;
}

// Stand-alone sampling check: runs the step's sampling kernel on caller logits with a scratch
// state (no stop-rule side effects are reported; out[0] = sampled index, out[1] = argmax).
extern "C" int vx_op_sample(const float* logits, int32_t V, int32_t top_k, float temperature, const float* exp_noise,
                            int32_t* out, void* stream) {
  if (V < 2 || V > 2048) return fail(VX_ERR_UNSUPPORTED, "sample: V=%d", V);
  hipStream_t s = (hipStream_t)stream;
  ArState h{};
  h.S = 1 << 20; h.kv_text = h.S; h.top_k = top_k; h.temperature = temperature; h.max_new = -1;
  h.exp_noise = exp_noise; h.noise_rows = 1; h.seed = 1;
  ArState* dst = nullptr;
  int* scratch = nullptr;
  float* fz = nullptr;
  HIPC(hipMalloc((void**)&dst, sizeof h));
  HIPC(hipMalloc((void**)&scratch, 16 * sizeof(int)));
  HIPC(hipMalloc((void**)&fz, 4096 * sizeof(float)));
  HIPC(hipMemsetAsync(fz, 0, 4096 * sizeof(float), s));
  HIPC(hipMemcpyAsync(dst, &h, sizeof h, hipMemcpyHostToDevice, s));
  SampleArgs sa{};
  sa.logits = logits; sa.V = V; sa.st = dst;
  sa.tokens = scratch; sa.sampled = scratch + 4; sa.argmaxes = scratch + 8;
  sa.emb = fz; sa.alpha = fz; sa.pe = fz; sa.x = fz + 2048; sa.d = 0;
  // both variants are exercised by the parity test: the 4-wave kernel of the decode step for the model's vocabulary,
  // the single-wave one for larger test vocabularies
  if (V <= 17 * 64) sample_embed4_kernel<5, 17><<<1, 256, 0, s>>>(sa);
  else sample_embed_kernel<32><<<1, 64, 0, s>>>(sa);
  HIPC(hipGetLastError());
  int host[16];
  HIPC(hipMemcpyAsync(host, scratch, sizeof host, hipMemcpyDeviceToHost, s));
  HIPC(hipStreamSynchronize(s));
  out[0] = host[4];
  out[1] = host[8];
  (void)hipFree(dst); (void)hipFree(scratch); (void)hipFree(fz);
  return VX_OK;
}

#ifdef VX_PROBES
#include "probe_entry.hpp"
#endif
