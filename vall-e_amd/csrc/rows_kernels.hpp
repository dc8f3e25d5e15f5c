// Multi-row kernels shared by the AR prefill (valle.py:995-1039, first pass) and the seven NAR
// stages (valle.py:1063-1134): embedding + sine position, (Adaptive)LayerNorm, a scalar-FMA
// tiled GEMM and a tiled attention.  These are the straightforward fp32-accumulate versions:
// they ARE the product path in fp32 precision (token-exact parity mode) and the A/B reference
// for the MFMA kernels (mfma_kernels.hpp) that replace them in bf16 precision.
#pragma once
#include "common.hpp"

namespace vx {

// out[r] = table[ids[r*stride+off]] * 1.0 + alpha * pe[pos0 + r]   (embedding.py:93-97)
__global__ __launch_bounds__(256) void embed_pos_kernel(const long long* __restrict__ ids, int id_stride, int id_off,
                                                        const float* __restrict__ table, int table_rows, int d,
                                                        const float* __restrict__ alpha, const float* __restrict__ pe,
                                                        int pos0, float* __restrict__ out, int rows) {
  const int r = blockIdx.x;
  if (r >= rows) return;
  // ids are validated by the host shim (IndexError like nn.Embedding); the clamp only keeps a
  // bad id from a raw C-ABI caller inside the table
  const long long id = min(max(ids[(size_t)r * id_stride + id_off], 0ll), (long long)table_rows - 1);
  const float al = alpha[0];
  for (int c = threadIdx.x; c < d; c += 256)
    out[(size_t)r * d + c] = __fadd_rn(table[(size_t)id * d + c], __fmul_rn(al, pe[(size_t)(pos0 + r) * d + c]));
}

// out[r] = src[r] * 1.0 + alpha * pe[pos0 + r]
__global__ __launch_bounds__(256) void add_pos_kernel(const float* __restrict__ src, int d,
                                                      const float* __restrict__ alpha, const float* __restrict__ pe,
                                                      int pos0, float* __restrict__ out, int rows) {
  const int r = blockIdx.x;
  if (r >= rows) return;
  const float al = alpha[0];
  for (int c = threadIdx.x; c < d; c += 256)
    out[(size_t)r * d + c] = __fadd_rn(src[(size_t)r * d + c], __fmul_rn(al, pe[(size_t)(pos0 + r) * d + c]));
}

// acc[r] (=|+=) table[ids[r*stride+off]]   (valle.py:1064-1066, 1105-1113, 1134)
__global__ __launch_bounds__(256) void embed_accum_kernel(const long long* __restrict__ ids, int id_stride, int id_off,
                                                          const float* __restrict__ table, int table_rows, int d,
                                                          float* __restrict__ acc, int rows, int init) {
  const int r = blockIdx.x;
  if (r >= rows) return;
  const long long id = min(max(ids[(size_t)r * id_stride + id_off], 0ll), (long long)table_rows - 1);
  for (int c = threadIdx.x; c < d; c += 256) {
    const float e = table[(size_t)id * d + c];
    acc[(size_t)r * d + c] = init ? e : __fadd_rn(acc[(size_t)r * d + c], e);
  }
}

// ---- text prenet (valle.py:97-113): Conv1d(d, d, kernel 5, padding same) + BatchNorm1d (running statistics) + ReLU ----
// x, y: (S, d) row-major (the reference transposes to channels-first around the convolutions; same numbers).
// wt: the Conv1d weight re-laid out as [k][ci][co] (vx_finalize_weights), so a wave reads consecutive co.
// One workgroup = CONV_TT consecutive rows t, all co (thread = co, strided); the CONV_TT + 4 input rows it needs sit
// in LDS and every weight is used CONV_TT times.
constexpr int CONV_TT = 4;
__global__ __launch_bounds__(256) void conv5_bn_relu_kernel(const float* __restrict__ x, const float* __restrict__ wt,
                                                            const float* __restrict__ bias, const float* __restrict__ bn_g,
                                                            const float* __restrict__ bn_b, const float* __restrict__ bn_mean,
                                                            const float* __restrict__ bn_var, float* __restrict__ y, int S,
                                                            int d) {
  extern __shared__ float xs[];  // [(CONV_TT + 4)][d]
  const int t0 = blockIdx.x * CONV_TT;
  for (int i = threadIdx.x; i < (CONV_TT + 4) * d; i += 256) {
    const int rr = i / d, c = i - rr * d, t = t0 + rr - 2;
    xs[i] = (t >= 0 && t < S) ? x[(size_t)t * d + c] : 0.f;  // zero padding ("same")
  }
  __syncthreads();
  for (int co = threadIdx.x; co < d; co += 256) {
    float acc[CONV_TT];
#pragma unroll
    for (int j = 0; j < CONV_TT; ++j) acc[j] = 0.f;
    for (int ci = 0; ci < d; ++ci) {
      float w[5];
#pragma unroll
      for (int k = 0; k < 5; ++k) w[k] = wt[((size_t)k * d + ci) * d + co];
#pragma unroll
      for (int j = 0; j < CONV_TT; ++j)
#pragma unroll
        for (int k = 0; k < 5; ++k) acc[j] = fmaf(w[k], xs[(j + k) * d + ci], acc[j]);
    }
    const float inv = 1.0f / sqrtf(bn_var[co] + 1e-5f);  // BatchNorm1d eps (torch default)
#pragma unroll
    for (int j = 0; j < CONV_TT; ++j) {
      if (t0 + j >= S) break;
      const float v = (acc[j] + bias[co] - bn_mean[co]) * inv * bn_g[co] + bn_b[co];
      y[(size_t)(t0 + j) * d + co] = fmaxf(v, 0.f);
    }
  }
}

// [co][ci][k] (nn.Conv1d) -> [k][ci][co]
__global__ __launch_bounds__(256) void conv_weight_relayout_kernel(const float* __restrict__ w, float* __restrict__ wt, int d) {
  const size_t n = (size_t)d * d * 5, i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int k = (int)(i % 5), ci = (int)((i / 5) % d), co = (int)(i / (5 * (size_t)d));
  wt[((size_t)k * d + ci) * d + co] = w[i];
}

// (Adaptive)LayerNorm, one wave per row: out = [w *] (LN(x) * gamma + beta) [+ b]
// (modules/transformer.py:57-74, 93-108).  OT = float or bf16 (the GEMM A-operand type).
template <typename OT, int MAXV>  // MAXV float4 per lane: d <= 256 * MAXV
__global__ __launch_bounds__(256) void layernorm_rows_kernel(const float* x, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta,
                                                             const float* __restrict__ ada_w,
                                                             const float* __restrict__ ada_b, OT* __restrict__ out,
                                                             int rows, int d, float* xout = nullptr,
                                                             const float* __restrict__ part = nullptr, int nsplit = 0,
                                                             size_t part_stride = 0, const float* __restrict__ pbias = nullptr) {
  // part != nullptr: the row first receives the preceding split-K GEMM, x += pbias + sum_z part[z] (fixed order); the
  // sum is written back to x unless xout redirects the normalised row there (post-norm).  out == nullptr: fold only.
  // xout (post-norm layers, transformer.py:303-308): the normalised row also replaces the residual stream; it may
  // alias x (every element is read into registers by its own lane before anything is stored).
  // Every load of the kernel (row, slabs, affine and AdaLN vectors) is issued before the first reduction: the kernel
  // is one memory round trip, not one per phase.
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  const float* xr = x + (size_t)r * d;
  const bool fold = part != nullptr, norm = out != nullptr, ada = ada_w != nullptr;  // wave-uniform
  float4 v[MAXV], g[MAXV], b[MAXV], w[MAXV], c[MAXV];
  int kk[MAXV];
  bool ok[MAXV];
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int k = (i * 64 + lane) * 4;
    ok[i] = k < d;
    kk[i] = ok[i] ? k : 0;  // clamped: the loads below are unconditional
    v[i] = *reinterpret_cast<const float4*>(xr + kk[i]);
    if (norm) {
      g[i] = *reinterpret_cast<const float4*>(gamma + kk[i]);
      b[i] = *reinterpret_cast<const float4*>(beta + kk[i]);
      if (ada) {
        w[i] = *reinterpret_cast<const float4*>(ada_w + kk[i]);
        c[i] = *reinterpret_cast<const float4*>(ada_b + kk[i]);
      }
    }
  }
  if (fold) {
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      float4 t = *reinterpret_cast<const float4*>(pbias + kk[i]);
      float4 p[4];  // nsplit <= 4; all loads go out together (clamped slab index, masked sum)
#pragma unroll
      for (int z = 0; z < 4; ++z)
        p[z] = *reinterpret_cast<const float4*>(part + (size_t)min(z, nsplit - 1) * part_stride + (size_t)r * d + kk[i]);
#pragma unroll
      for (int z = 0; z < 4; ++z)
        if (z < nsplit) { t.x += p[z].x; t.y += p[z].y; t.z += p[z].z; t.w += p[z].w; }
      v[i].x += t.x; v[i].y += t.y; v[i].z += t.z; v[i].w += t.w;
      if (ok[i] && (xout == nullptr || !norm)) *reinterpret_cast<float4*>(const_cast<float*>(xr) + kk[i]) = v[i];
    }
  }
  if (!norm) return;  // fold-only pass
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) s += ok[i] ? (v[i].x + v[i].y) + (v[i].z + v[i].w) : 0.f;
  const float mean = wave_sum(s) / (float)d;
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const float d0 = v[i].x - mean, d1 = v[i].y - mean, d2 = v[i].z - mean, d3 = v[i].w - mean;
    ss += ok[i] ? (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3) : 0.f;
  }
  const float rstd = 1.0f / sqrtf(wave_sum(ss) / (float)d + LN_EPS);
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    if (!ok[i]) continue;
    float o[4] = {(v[i].x - mean) * rstd * g[i].x + b[i].x, (v[i].y - mean) * rstd * g[i].y + b[i].y,
                  (v[i].z - mean) * rstd * g[i].z + b[i].z, (v[i].w - mean) * rstd * g[i].w + b[i].w};
    if (ada) {
      o[0] = __fadd_rn(__fmul_rn(w[i].x, o[0]), c[i].x); o[1] = __fadd_rn(__fmul_rn(w[i].y, o[1]), c[i].y);
      o[2] = __fadd_rn(__fmul_rn(w[i].z, o[2]), c[i].z); o[3] = __fadd_rn(__fmul_rn(w[i].w, o[3]), c[i].w);
    }
    union { OT e[4]; uint2 u2; float4 f4; } pk;
#pragma unroll
    for (int j = 0; j < 4; ++j) pk.e[j] = from_f32<OT>(o[j]);
    OT* op = out + (size_t)r * d + kk[i];
    if (sizeof(OT) == 2) *reinterpret_cast<uint2*>(op) = pk.u2;
    else *reinterpret_cast<float4*>(op) = pk.f4;
    if (xout != nullptr) *reinterpret_cast<float4*>(xout + (size_t)r * d + kk[i]) = make_float4(o[0], o[1], o[2], o[3]);
  }
}

// out[row] = W[row,:] . e + b[row]   (AdaptiveLayerNorm.project_layer on the 1 x d stage
// embedding, modules/transformer.py:96-100; run once per stage/site at finalize time)
__global__ __launch_bounds__(256) void project_vec_kernel(const float* __restrict__ W, const float* __restrict__ b,
                                                          const float* __restrict__ e, float* __restrict__ out,
                                                          int N, int K) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= N) return;
  float s = 0.f;
  for (int k = lane; k < K; k += 64) s = fmaf(W[(size_t)row * K + k], e[k], s);
  s = wave_sum(s);
  if (lane == 0) out[row] = s + b[row];
}

// ---- scalar-FMA tiled GEMM: C[M,N] = A[M,K] . W[N,K]^T (+bias)(relu)(+residual) ------------
enum GemmEpi { GE_PLAIN = 0, GE_BIAS = 1, GE_RELU = 2, GE_RESID = 3 };

template <typename T, typename OT, int EPI>
__global__ __launch_bounds__(256) void gemm_simple_kernel(const T* __restrict__ A, const T* __restrict__ W,
                                                          const float* __restrict__ bias, OT* __restrict__ C,
                                                          int M, int N, int K) {
  constexpr int BM = 64, BN = 64, BK = 32, LD = BM + 4;
  __shared__ __attribute__((aligned(16))) float As[BK][LD];
  __shared__ __attribute__((aligned(16))) float Ws[BK][LD];
  const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  const int lr = tid >> 2, lk = (tid & 3) * 8;  // loader: row lr, 8 consecutive k from lk
  float acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;

  for (int k0 = 0; k0 < K; k0 += BK) {
    float av[8], wv[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int k = k0 + lk + j;
      av[j] = (m0 + lr < M && k < K) ? to_f32(A[(size_t)(m0 + lr) * K + k]) : 0.f;
      wv[j] = (n0 + lr < N && k < K) ? to_f32(W[(size_t)(n0 + lr) * K + k]) : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 8; ++j) { As[lk + j][lr] = av[j]; Ws[lk + j][lr] = wv[j]; }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < BK; ++k) {
      const float4 a4 = *reinterpret_cast<const float4*>(&As[k][ty * 4]);
      const float4 b4 = *reinterpret_cast<const float4*>(&Ws[k][tx * 4]);
      const float a[4] = {a4.x, a4.y, a4.z, a4.w}, b[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(a[i], b[j], acc[i][j]);
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + ty * 4 + i;
    if (m >= M) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + tx * 4 + j;
      if (n >= N) continue;
      float v = acc[i][j];
      if (EPI != GE_PLAIN) v += bias[n];
      if (EPI == GE_RELU) v = fmaxf(v, 0.f);
      OT* c = C + (size_t)m * N + n;
      if (EPI == GE_RESID) *c = from_f32<OT>(to_f32(*c) + v);
      else *c = from_f32<OT>(v);
    }
  }
}

// ---- tiled attention over rows (prefill mask or none) --------------------------------------
// qkv: (M, 3d) rows [q | k | v], head h = channels [h*HD, (h+1)*HD) (functional.py:5785-5830).
// mask: text_len < 0 -> none (NAR, valle.py:1125-1127); else rows < text_len see keys
// [0, text_len), rows >= text_len see keys [0, row] (valle.py:1019-1033).
// Workgroup = 64 queries of one head; 4 threads per query, each owning every 4th key of a
// 64-key tile with its own online-softmax state, merged by shuffles at the end.
template <typename T, int HD>
__global__ __launch_bounds__(256) void attn_rows_simple_kernel(const T* __restrict__ qkv, T* __restrict__ out, int M,
                                                               int d, int text_len, float scale) {
  constexpr int LD = HD + 4;
  __shared__ __attribute__((aligned(16))) float Ks[64][LD];
  __shared__ __attribute__((aligned(16))) float Vs[64][LD];
  const int tid = threadIdx.x, qi = tid >> 2, s = tid & 3;
  const int h = blockIdx.y, q0 = blockIdx.x * 64;
  const int row = q0 + qi;
  const bool qvalid = row < M;
  const int ld3 = 3 * d;
  float q[HD], o[HD];
#pragma unroll
  for (int c = 0; c < HD; ++c) {
    q[c] = qvalid ? to_f32(qkv[(size_t)row * ld3 + h * HD + c]) : 0.f;
    o[c] = 0.f;
  }
  const int limit = !qvalid ? 0 : (text_len < 0 ? M : (row < text_len ? text_len : row + 1));  // keys [0, limit)
  // highest key any query of this block may see
  int blk_limit = M;
  if (text_len >= 0) {
    const int last = min(M, q0 + 64) - 1;
    blk_limit = last < text_len ? text_len : last + 1;
  }
  float m = -INFINITY, l = 0.f;
  constexpr int CPT = HD / 4;  // channels each of a row's 4 loader threads copies
  const int lr = tid >> 2, lc = (tid & 3) * CPT;
  for (int kt = 0; kt < blk_limit; kt += 64) {
    __syncthreads();
    {
      const int kr = kt + lr;
#pragma unroll
      for (int j = 0; j < CPT; ++j) {
        Ks[lr][lc + j] = (kr < M) ? to_f32(qkv[(size_t)kr * ld3 + d + h * HD + lc + j]) : 0.f;
        Vs[lr][lc + j] = (kr < M) ? to_f32(qkv[(size_t)kr * ld3 + 2 * d + h * HD + lc + j]) : 0.f;
      }
    }
    __syncthreads();
#pragma unroll 4
    for (int kk = 0; kk < 16; ++kk) {
      const int kl = kk * 4 + s, kg = kt + kl;
      float dot = 0.f;
#pragma unroll
      for (int c = 0; c < HD; c += 4) {
        const float4 kv = *reinterpret_cast<const float4*>(&Ks[kl][c]);
        dot = fmaf(q[c], kv.x, dot); dot = fmaf(q[c + 1], kv.y, dot);
        dot = fmaf(q[c + 2], kv.z, dot); dot = fmaf(q[c + 3], kv.w, dot);
      }
      if (kg < limit) {
        const float sc = dot * scale;
        const float mn = fmaxf(m, sc);
        const float corr = expf(m - mn), p = expf(sc - mn);
        l = l * corr + p;
#pragma unroll
        for (int c = 0; c < HD; c += 4) {
          const float4 vv = *reinterpret_cast<const float4*>(&Vs[kl][c]);
          o[c] = o[c] * corr + p * vv.x; o[c + 1] = o[c + 1] * corr + p * vv.y;
          o[c + 2] = o[c + 2] * corr + p * vv.z; o[c + 3] = o[c + 3] * corr + p * vv.w;
        }
        m = mn;
      }
    }
  }
  // merge the 4 threads of a query (adjacent lanes)
#pragma unroll
  for (int off = 1; off < 4; off <<= 1) {
    const float m2 = __shfl_xor(m, off, WAVE), l2 = __shfl_xor(l, off, WAVE);
    const float mn = fmaxf(m, m2);
    const float c1 = (m == -INFINITY) ? 0.f : expf(m - mn), c2 = (m2 == -INFINITY) ? 0.f : expf(m2 - mn);
    l = l * c1 + l2 * c2;
#pragma unroll
    for (int c = 0; c < HD; ++c) o[c] = o[c] * c1 + __shfl_xor(o[c], off, WAVE) * c2;
    m = mn;
  }
  if (qvalid) {
    const float inv = 1.0f / l;
    // thread s writes channels [s*HD/4, (s+1)*HD/4)
#pragma unroll
    for (int c = 0; c < HD; ++c)
      if ((c / (HD / 4)) == s) out[(size_t)row * d + h * HD + c] = from_f32<T>(o[c] * inv);
  }
}

// Cross-attention rows (VALL-F: TransformerDecoderLayer._mha_block, modules/transformer.py:583-596): queries are rows of a
// (M, ldq) buffer (the projected norm2(x)), keys / values are the text memory of this layer in the decode-cache layout
// (nhead, ctx_max, HD) - projected once per utterance - and every query sees all Sk keys (no mask: valle.py:631, 687 pass only
// an all-false key padding mask).  Same walk as attn_rows_simple_kernel: 4 threads per query, 64-key LDS tiles, online softmax.
template <typename T, int HD>
__global__ __launch_bounds__(256) void cross_attn_rows_kernel(const T* __restrict__ q_, int ldq, const T* __restrict__ kc,
                                                              const T* __restrict__ vc, int ctx_max, T* __restrict__ out,
                                                              int ldo, int M, int Sk, float scale) {
  constexpr int LD = HD + 4;
  __shared__ __attribute__((aligned(16))) float Ks[64][LD];
  __shared__ __attribute__((aligned(16))) float Vs[64][LD];
  const int tid = threadIdx.x, qi = tid >> 2, s = tid & 3;
  const int h = blockIdx.y, q0 = blockIdx.x * 64;
  const int row = q0 + qi;
  const bool qvalid = row < M;
  float q[HD], o[HD];
#pragma unroll
  for (int c = 0; c < HD; ++c) {
    q[c] = qvalid ? to_f32(q_[(size_t)row * ldq + h * HD + c]) : 0.f;
    o[c] = 0.f;
  }
  float m = -INFINITY, l = 0.f;
  constexpr int CPT = HD / 4;
  const int lr = tid >> 2, lc = (tid & 3) * CPT;
  const T* kh = kc + (size_t)h * ctx_max * HD;
  const T* vh = vc + (size_t)h * ctx_max * HD;
  for (int kt = 0; kt < Sk; kt += 64) {
    __syncthreads();
    {
      const int kr = kt + lr;
#pragma unroll
      for (int j = 0; j < CPT; ++j) {
        Ks[lr][lc + j] = (kr < Sk) ? to_f32(kh[(size_t)kr * HD + lc + j]) : 0.f;
        Vs[lr][lc + j] = (kr < Sk) ? to_f32(vh[(size_t)kr * HD + lc + j]) : 0.f;
      }
    }
    __syncthreads();
#pragma unroll 4
    for (int kk = 0; kk < 16; ++kk) {
      const int kl = kk * 4 + s, kg = kt + kl;
      float dot = 0.f;
#pragma unroll
      for (int c = 0; c < HD; c += 4) {
        const float4 kv = *reinterpret_cast<const float4*>(&Ks[kl][c]);
        dot = fmaf(q[c], kv.x, dot); dot = fmaf(q[c + 1], kv.y, dot);
        dot = fmaf(q[c + 2], kv.z, dot); dot = fmaf(q[c + 3], kv.w, dot);
      }
      if (qvalid && kg < Sk) {
        const float sc = dot * scale;
        const float mn = fmaxf(m, sc);
        const float corr = expf(m - mn), p = expf(sc - mn);
        l = l * corr + p;
#pragma unroll
        for (int c = 0; c < HD; c += 4) {
          const float4 vv = *reinterpret_cast<const float4*>(&Vs[kl][c]);
          o[c] = o[c] * corr + p * vv.x; o[c + 1] = o[c + 1] * corr + p * vv.y;
          o[c + 2] = o[c + 2] * corr + p * vv.z; o[c + 3] = o[c + 3] * corr + p * vv.w;
        }
        m = mn;
      }
    }
  }
#pragma unroll
  for (int off = 1; off < 4; off <<= 1) {
    const float m2 = __shfl_xor(m, off, WAVE), l2 = __shfl_xor(l, off, WAVE);
    const float mn = fmaxf(m, m2);
    const float c1 = (m == -INFINITY) ? 0.f : expf(m - mn), c2 = (m2 == -INFINITY) ? 0.f : expf(m2 - mn);
    l = l * c1 + l2 * c2;
#pragma unroll
    for (int c = 0; c < HD; ++c) o[c] = o[c] * c1 + __shfl_xor(o[c], off, WAVE) * c2;
    m = mn;
  }
  if (qvalid) {
    const float inv = 1.0f / l;
#pragma unroll
    for (int c = 0; c < HD; ++c)
      if ((c / CPT) == s) out[(size_t)row * ldo + h * HD + c] = from_f32<T>(o[c] * inv);
  }
}

// K/V rows of the prefill -> per-head cache layout (nhead, ctx_max, HD)
template <typename T>
__global__ __launch_bounds__(256) void kv_scatter_kernel(const T* __restrict__ qkv, T* __restrict__ kc,
                                                         T* __restrict__ vc, int rows, int d, int hd, int ctx_max) {
  const int r = blockIdx.x;
  if (r >= rows) return;
  for (int i = threadIdx.x; i < d; i += 256) {
    const int h = i / hd, c = i - h * hd;
    const size_t dst = ((size_t)h * ctx_max + r) * hd + c;
    kc[dst] = qkv[(size_t)r * 3 * d + d + i];
    vc[dst] = qkv[(size_t)r * 3 * d + 2 * d + i];
  }
}

// Batched prefill: row j of segment z (rows seg_start[z] + j of the concatenated buffer) -> position j of slot z's
// cache.  grid = (max segment length, segments).
template <typename T>
__global__ __launch_bounds__(256) void kv_scatter_seg_kernel(const T* __restrict__ qkv, T* __restrict__ kv_layer,
                                                             size_t slot_stride, size_t v_offset,
                                                             const int* __restrict__ seg_start,
                                                             const int* __restrict__ seg_len, int d, int hd, int ctx_max) {
  const int j = blockIdx.x, z = blockIdx.y;
  if (j >= seg_len[z]) return;
  const size_t r = (size_t)seg_start[z] + j;
  T* kc = kv_layer + (size_t)z * slot_stride;
  T* vc = kc + v_offset;
  for (int i = threadIdx.x; i < d; i += 256) {
    const int h = i / hd, c = i - h * hd;
    const size_t dst = ((size_t)h * ctx_max + j) * hd + c;
    kc[dst] = qkv[r * 3 * d + d + i];
    vc[dst] = qkv[r * 3 * d + 2 * d + i];
  }
}

// samples[t] = argmax(logits[t]) (first max wins, torch.argmax valle.py:1130); also written into
// column `col` of the (T, Q) int64 code matrix (valle.py:1136-1137).
__global__ __launch_bounds__(256) void argmax_rows_kernel(const float* __restrict__ logits, int N, int rows,
                                                          long long* __restrict__ samples,
                                                          long long* __restrict__ codes, int Q, int col) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  ValIdx best{-INFINITY, 0x7fffffff};
  for (int i = lane; i < N; i += 64) {
    ValIdx c{logits[(size_t)r * N + i], i};
    best = better(best, c);
  }
  best = wave_argmax(best);
  if (lane == 0) {
    samples[r] = best.i;
    codes[(size_t)r * Q + col] = best.i;
  }
}

__global__ void copy_col_kernel(const long long* __restrict__ src, long long* __restrict__ codes, int rows, int Q, int col) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r < rows) codes[(size_t)r * Q + col] = src[r];
}

// dst[r] = codes[r][col] of a (rows, Q) code matrix (teacher-forced NAR stages)
__global__ void pick_col_kernel(const long long* __restrict__ codes, int Q, int col, long long* __restrict__ dst, int rows) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r < rows) dst[r] = min(max(codes[(size_t)r * Q + col], 0ll), 1023ll);
}

template <typename OT>
__global__ void convert_kernel(const float* __restrict__ src, OT* __restrict__ dst, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) dst[i] = from_f32<OT>(src[i]);
}

}  // namespace vx
