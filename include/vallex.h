/*
 * vallex.h — C ABI of the MI355X-native VALL-E inference engine (libvallex.so).
 *
 * The reference (RuntimeRacer/vall-e) is pure Python and has no FFI: its "operator interface"
 * for this path is the method valle.models.VALLE.inference (valle/models/valle.py:961-1137)
 * called from valle/bin/infer.py:197-204 and :241-248.  This header is the boundary a
 * maintainer binds instead (ctypes stub in INTEGRATION.md); every entry point cites the piece
 * of the reference it replaces.  Plain C types only: no torch / HIP types in any signature
 * (a stream is passed as void* = hipStream_t, NULL = the default stream).
 *
 * Conventions
 *   - every function returns VX_OK (0) or an error code; vx_last_error() gives the message of
 *     the last failure on the calling thread.
 *   - id / code arrays are int64 (torch.LongTensor layout, as the reference passes them) and
 *     may live in device OR host memory — the engine detects which.  The caller owns them.
 *   - one engine per device per thread; calls on one engine must not overlap.
 *   - work is ordered after everything already enqueued on `stream`, and `stream` is made to
 *     wait for the engine's work before the call returns (the engine runs on its own stream
 *     so that the AR step can be replayed as a hipGraph).
 */
#ifndef VALLEX_H
#define VALLEX_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct vx_engine vx_engine;

enum vx_status {
  VX_OK = 0,
  VX_ERR_ARG = 1,         /* bad argument (the reference would trip an assert, valle.py:986-991) */
  VX_ERR_HIP = 2,         /* a HIP runtime call failed */
  VX_ERR_STATE = 3,       /* call order violated (e.g. decode before prefill) */
  VX_ERR_CAPACITY = 4,    /* S / P / T exceed the capacities given at vx_create, or a decode filled the KV cache (max_audio
                             rows) before the reference's stop rule fired; the worst case 16 S + 1 alone is NOT refused */
  VX_ERR_UNSUPPORTED = 5, /* configuration outside the built scope */
  VX_ERR_WEIGHTS = 6      /* unknown key, wrong shape or missing tensor */
};

enum vx_precision {
  VX_PREC_F32 = 0, /* fp32 weights, KV cache and activations: the token-exact parity mode */
  VX_PREC_BF16 = 1, /* bf16 matrices / KV cache / GEMM operands, fp32 residual stream + accumulate */
  VX_PREC_FP8_NAR = 2 /* VX_PREC_BF16, and the NAR stages' QKV / FFN1 / FFN2 GEMMs (valle.py:1115-1134 through
                         transformer.py:296-334) on OCP e4m3 operands with E8M0 scales per 32 k (MXFP8, the matrix cores'
                         block-scaled fp8 form) wherever the stage runs at >= 4096 concatenated rows (vx_nar_batch);
                         attention, out-projection, predict layers and the whole AR path stay bf16 (BASELINE configs[4]) */
};

enum vx_stop_reason {
  VX_STOP_NONE = 0,
  VX_STOP_EOS_ARGMAX = 1, /* argmax(logits) == 1024      (valle.py:1045) */
  VX_STOP_EOS_SAMPLE = 2, /* sampled token == 1024       (valle.py:1046) */
  VX_STOP_LENGTH = 3,     /* generated > 16 * text_len   (valle.py:1047) */
  VX_STOP_MAX_NEW = 4     /* engine-level max_new_tokens (not in the reference) */
};

enum vx_flags {
  VX_FLAG_TRACE_LOGITS = 1, /* keep the (1025,) AR logits of every pass (parity tests) */
  VX_FLAG_NO_GRAPH = 2,     /* launch the AR step kernel by kernel instead of as a hipGraph */
  VX_FLAG_SIMPLE_ROWS = 4,  /* bf16 mode: use the scalar-FMA row kernels instead of MFMA (A/B checks) */
  VX_FLAG_PRENET = 16,      /* add_prenet=True (valle.py:96-123, 181-213): conv/BatchNorm text prenets and MLP audio
                               prenets in front of the position embeddings, fp32; batch-1 path only */
  VX_FLAG_POST_NORM = 8,    /* norm_first=False (valle.py:60, transformer.py:303-308): x = norm(x + block(x)), no final
                               encoder norms; batch-1 path only (max_batch must be <= 1) */
  VX_FLAG_VALLF = 32        /* the cross-attention variant VALLF (valle.py:49-719, --model-name VALL-F): both stacks are
                               TransformerDecoderLayers (modules/transformer.py:409-601) - causal / unmasked self-attention over
                               the AUDIO rows only, cross-attention over the embedded text, three norms per layer.  Same entry
                               points (VALLF.inference, valle.py:566-710, has VALLE.inference's signature); the state_dict gains
                               layers.N.multihead_attn.* and layers.N.norm3.*.  Batch-1 path only */
};

/* Mirrors VALLE.__init__ (valle.py:727-760) / get_model (models/__init__.py:112-124). */
typedef struct vx_config {
  int32_t struct_size;     /* = sizeof(vx_config) */
  int32_t d_model;         /* --decoder-dim: multiple of 8, <= 1024 */
  int32_t nhead;           /* --nhead: d_model / nhead in {4, 8, 16, 32, 64}; 64 is the tuned geometry, the others run on plain kernels (batch-1) */
  int32_t num_layers;      /* --num-decoder-layers */
  int32_t nar_d_model;     /* int(d_model * scale_factor), valle.py:83 */
  int32_t nar_nhead;       /* int(nhead * scale_factor),   valle.py:234 */
  int32_t nar_num_layers;  /* int(L * scale_factor),       valle.py:241 */
  int32_t num_quantizers;  /* --num-quantizers (1..8) */
  int32_t prefix_mode;     /* --prefix-mode 0/1/2/4 */
  int32_t prepend_bos;     /* --prepend-bos */
  int32_t precision;       /* enum vx_precision */
  int32_t max_text;        /* capacity: phoneme ids per utterance */
  int32_t max_audio;       /* capacity: audio rows = [BOS] + prompt frames + generated frames */
  int32_t device;          /* HIP device ordinal */
  int32_t flags;           /* enum vx_flags */
  int32_t max_batch;       /* slots for batched AR decode (vx_batch_*): 0/1 = batch-1 only, <= 64; bf16 only */
} vx_config;

/* Sampling / stop-rule parameters of one AR decode (VALLE.inference args top_k, temperature,
 * valle.py:967-968; topk_sampling valle.py:1287-1302). */
typedef struct vx_decode_params {
  int32_t struct_size;
  int32_t top_k;            /* <= 0: no filtering (the reference default -100) */
  float temperature;        /* logits / temperature when != 1.0 */
  int32_t max_new_tokens;   /* < 0: reference stop rule only */
  const float* exp_noise;   /* optional (noise_rows, 1025) Exp(1) draws, row i feeds pass i:      */
  int64_t noise_rows;       /*   sample = argmax(p / q) == torch.multinomial(p, 1) on that stream  */
  uint64_t seed;            /* device counter RNG seed, used when exp_noise == NULL */
  const int64_t* forced;    /* optional teacher forcing: token appended at pass i = forced[i]     */
  int32_t n_forced;         /*   (the sample is still drawn and recorded); decode ends after them */
} vx_decode_params;

const char* vx_last_error(void);

/* VALLE(...) constructor + .to(device) (valle.py:727-760; bin/infer.py:137,147). */
int vx_create(const vx_config* cfg, vx_engine** out);
void vx_destroy(vx_engine* e);

/* model.load_state_dict(checkpoint["model"], strict=True) (bin/infer.py:139-143), one tensor
 * at a time.  `key` is the reference state_dict key, `data` fp32 (host or device), row-major. */
int vx_set_weight(vx_engine* e, const char* key, const float* data, const int64_t* shape, int32_t ndim);
/* Optional: the fp32 sine table of SinePositionalEmbedding (modules/embedding.py:75-88),
 * (rows, dim) with rows >= max(4000, max_text + max_audio).  which: 0 = AR width, 1 = NAR width.  If never set, the engine fills it with
 * host sinf/cosf (same formula; may differ from torch's table in the last ulp). */
int vx_set_sine_table(vx_engine* e, int32_t which, const float* data, int64_t rows, int64_t dim);
/* strict=True check (every key present) + model.eval(); precomputes the per-stage AdaptiveLayerNorm
 * scale/shift vectors (modules/transformer.py:93-108, constant per stage at inference). */
int vx_finalize_weights(vx_engine* e);

/* valle.py:994-1010 + the first pass of the AR loop (valle.py:1012-1039): embeds text and
 * codebook-0 prompt (+BOS), runs the AR stack under the reference mask, fills the KV cache and
 * leaves the logits of pass 0.  text: (S,), prompt_cb0: (P,). */
int vx_ar_prefill(vx_engine* e, const int64_t* text, int32_t S, const int64_t* prompt_cb0, int32_t P, void* stream);

/* The AR while-loop (valle.py:1012-1057): sample, stop test, append, next pass.  Asynchronous
 * w.r.t. the host except for periodic polls of the stop flag. */
int vx_ar_decode(vx_engine* e, const vx_decode_params* p, void* stream);

/* Blocks until the decode has finished.  tokens: host buffer of `capacity` int64 (generated
 * codebook-0 tokens, valle.py:1059).  n_pass = number of logits rows produced. */
int vx_ar_result(vx_engine* e, int64_t* tokens, int32_t capacity, int32_t* n_tokens, int32_t* stop_reason,
                 int32_t* n_pass);

/* The NAR stages (valle.py:1063-1134).  text_nar: (S2,) phoneme ids after the prefix_mode 2/4
 * trim (done by the host shim, valle.py:1068-1079); prompts: (P, Q) row-major; ar_tokens: (T,);
 * codes_out: (T, Q) int64 row-major, column 0 = ar_tokens (valle.py:1136-1137). */
int vx_nar(vx_engine* e, const int64_t* text_nar, int32_t S2, const int64_t* prompts, int32_t P,
           const int64_t* ar_tokens, int32_t T, int64_t* codes_out, void* stream);
/* vx_nar / vx_nar_continual (continual != 0) with two parity-test options.  forced_codes (optional, (T, Q) int64): stage i
 * still reports its own argmax in codes_out[:, i+1], but the embedding fed to the later stages (valle.py:1133-1134) is taken from
 * forced_codes[:, i+1] - with the reference's codes this gives every stage exactly the input the reference gave it, so bf16 / fp8
 * engines can be compared stage by stage.  stage_logits (optional, (Q-1, T, 1024) fp32, host or device): the logits rows of every
 * stage (valle.py:1128). */
int vx_nar_ex(vx_engine* e, const int64_t* text_nar, int32_t S2, const int64_t* prompts, int32_t P,
              const int64_t* ar_tokens, int32_t T, int64_t* codes_out, const int64_t* forced_codes, float* stage_logits,
              int32_t continual, void* stream);
/* VALLE.continual's NAR body (valle.py:1185-1236): same arguments as vx_nar.  Differs from vx_nar only for models with
 * prenets in prefix mode 0, where continual() applies the audio position BEFORE the audio prenet (valle.py:1193-1194). */
int vx_nar_continual(vx_engine* e, const int64_t* text_nar, int32_t S2, const int64_t* prompts, int32_t P,
           const int64_t* ar_tokens, int32_t T, int64_t* codes_out, void* stream);

/* ---- batched AR decode (BASELINE configs[2]): up to max_batch utterances ("slots") advance one token per
 * step and share one stream of the weights; every slot has its own padded KV cache, stop rule and sampler
 * (the reference is batch-1, valle.py:989: each slot reproduces one independent inference() call).
 * vx_batch_prefill = vx_ar_prefill into a slot; vx_batch_decode runs the shared step until every one of
 * slots [0, n_slots) has stopped (params[i] for slot i; exp_noise / forced must be DEVICE pointers that stay
 * valid during the call); vx_batch_result = vx_ar_result of a slot.  The NAR stages then run per utterance
 * with vx_nar. */
int vx_batch_prefill(vx_engine* e, int32_t slot, const int64_t* text, int32_t S, const int64_t* prompt_cb0, int32_t P,
                     void* stream);
/* All n slots' prefills in one pass over the concatenated rows (slot z = utterance z): same contract as n calls of
 * vx_batch_prefill(z, text[z], S[z], prompt_cb0[z], P[z]); bf16 engines only.  The pointer arrays and S / P live on
 * the host, text[z] / prompt_cb0[z] may be host or device pointers. */
int vx_batch_prefill_all(vx_engine* e, int32_t n, const int64_t* const* text, const int32_t* S,
                         const int64_t* const* prompt_cb0, const int32_t* P, void* stream);
int vx_batch_decode(vx_engine* e, int32_t n_slots, const vx_decode_params* params, void* stream);
int vx_batch_result(vx_engine* e, int32_t slot, int64_t* tokens, int32_t capacity, int32_t* n_tokens,
                    int32_t* stop_reason);

/* The NAR stages of n (<= 32) utterances in one pass: rows are concatenated so the GEMMs run at M ~ n x 1k.
 * Arguments are arrays of n pointers / sizes with the meaning of vx_nar's. */
int vx_nar_batch(vx_engine* e, int32_t n, const int64_t* const* text_nar, const int32_t* S2,
                 const int64_t* const* prompts, const int32_t* P, const int64_t* const* ar_tokens, const int32_t* T,
                 int64_t* const* codes_out, void* stream);

/* vx_nar_batch with per-stage teacher forcing: forced_codes[i] = (T_i, Q) int64 as in vx_nar_ex, or forced_codes == NULL. */
int vx_nar_batch_ex(vx_engine* e, int32_t n, const int64_t* const* text_nar, const int32_t* S2,
                    const int64_t* const* prompts, const int32_t* P, const int64_t* const* ar_tokens, const int32_t* T,
                    int64_t* const* codes_out, const int64_t* const* forced_codes, void* stream);

/* Device-time of the last calls, measured with HIP events on the engine's stream:
 * out[0] prefill ms, out[1] AR decode ms, out[2] NAR ms, out[3] AR passes, out[4] graph launches, out[5] batched decode ms,
 * out[6] batched graph launches; with VX_TIME_GEMMS=1 in the environment also out[7] = ms spent in the QKV / out-projection / FFN
 * GEMM launches of the last NAR call and out[8] = their FLOPs (2 M N K each). */
int vx_get_timings(vx_engine* e, double* out, int32_t n);

/* Parity-test taps: copies an internal buffer to host memory (synchronises the engine stream).
 * names: "ar_logits" (n_pass x 1025 fp32 with VX_FLAG_TRACE_LOGITS, else the last row),
 * "ar_sampled" / "ar_argmax" (int32 per pass), "nar_logits" (T x 1024 fp32 of the last stage),
 * "ar_x" (d fp32 residual stream of the last AR row), "nar_x" (N x d fp32 after the last stage),
 * "batch_logits" (64 x 1088 fp32: newest logits row of every slot), "batch_argmax" / "batch_sampled"
 * (64 x (max_audio+2) int32 per pass), "batch_trace" (max_batch x (max_audio+2) x 1025 fp32: every pass's logits row of every
 * slot, engines created with VX_FLAG_TRACE_LOGITS and max_batch > 1). */
int vx_read_buffer(vx_engine* e, const char* name, void* dst, int64_t offset_bytes, int64_t nbytes);

/* Kernel-level entry points (device pointers, fp32 unless noted) used by tests/ to check each
 * HIP kernel against the oracle's corresponding torch op.  prec selects the storage type the
 * kernel is instantiated for (weights / KV: fp32 or bf16).  All run on `stream` and return
 * after enqueueing. */
int vx_op_layernorm(int32_t prec, const float* x, const float* gamma, const float* beta, const float* ada_w,
                    const float* ada_b, void* out /* fp32 or bf16 per prec */, int32_t rows, int32_t d, void* stream);
int vx_op_gemv(int32_t prec, const void* W, const float* bias, const float* x, float* y, int32_t N, int32_t K,
               int32_t relu, void* stream);
int vx_op_gemm(int32_t prec, int32_t use_mfma, const void* A, const void* W, const float* bias, float* C_f32,
               int32_t M, int32_t N, int32_t K, int32_t relu, void* stream);
/* The bf16 MFMA GEMM in the forms the engine's row path launches (mfma_gemm_dispatch; reference op: F.linear inside
 * valle/modules/transformer.py:296-334): form 0: C (M, N) bf16 = A.W^T + bias [ReLU] (QKV / FFN1), optionally with the last
 * N - vt_n0 columns (vt_n0 a multiple of 64) also written transposed to vt (N - vt_n0, vt_ld) bf16 (the V^T copy the attention
 * kernel reads; vt may be NULL); form 1: C (M, N) fp32 += A.W^T + bias (out-projection / FFN2 onto the residual stream). */
int vx_op_gemm_rows(int32_t form, const void* A_bf16, const void* W_bf16, const float* bias, void* C, int32_t M, int32_t N,
                    int32_t K, int32_t relu, void* vt_bf16, int32_t vt_n0, int32_t vt_ld, void* stream);
/* The MXFP8 (VX_PREC_FP8_NAR) kernels on caller data.  vx_op_gemm_mx: A (M, K) / W (N, K) fp32 device pointers are quantised
 * on the device as the engine does (e4m3 + one E8M0 scale per 32 k) and multiplied on the block-scaled matrix cores; out_mode 0:
 * c_out = C (M, N) fp32 (+bias, ReLU); out_mode 2 (FFN1's form): c_out = ReLU(C + bias) as e4m3 bytes (M, N), sc_out = its block
 * scales (N/32, ld), ld = M rounded up to 256.  qa_out / sa_out (optional): the quantised A and its scales (K/32, ld).
 * vx_op_layernorm_mx: LayerNorm / AdaLN with the quantised result, q_out (rows, d) bytes + s_out (d/32, ld) scales. */
int vx_op_gemm_mx(const float* A, const float* W, const float* bias, void* c_out, void* sc_out, int32_t M, int32_t N, int32_t K,
                  int32_t relu, int32_t out_mode, void* qa_out, void* sa_out, void* stream);
int vx_op_layernorm_mx(const float* x, const float* gamma, const float* beta, const float* ada_w, const float* ada_b, void* q_out,
                       void* s_out, int32_t rows, int32_t d, void* stream);
int vx_op_attention(int32_t prec, int32_t use_mfma, const void* qkv, void* out, int32_t rows, int32_t nhead,
                    int32_t hd, int32_t text_len_for_ar_mask /* <0: no mask */, void* stream);
int vx_op_sample(const float* logits, int32_t V, int32_t top_k, float temperature, const float* exp_noise,
                 int32_t* out_token_argmax /* [2]: sampled, argmax */, void* stream);
int vx_op_convert_bf16(const float* src, void* dst_bf16, int64_t n, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* VALLEX_H */
