// Kernels of the batch-1 AR decode step (valle/models/valle.py:1012-1057 with a KV cache).
//
// The step is HBM-bound: 12 * d^2 weights per layer are streamed once per token.  Each GEMV
// workgroup issues its 16-byte weight loads FIRST and only then computes the activation-side
// prologue (LayerNorm / attention-split combine), so the HBM latency of the weight stream is
// overlapped with the small dependent work (cdna_hip_programming.md §5 "GEMV / M<=16" row:
// weights straight to VGPRs, deep unroll, late wait).
#pragma once
#include "common.hpp"

namespace vx {

enum GemvPro { PRO_COPY = 0, PRO_LN = 1, PRO_ATTN = 2 };
enum GemvEpi { EPI_PLAIN = 0, EPI_BIAS = 1, EPI_RELU = 2, EPI_RESID = 3, EPI_QKV = 4, EPI_LOGITS = 5 };

struct GemvArgs {
  const void* W;       // (N, K) row-major, WT
  const float* bias;   // (N,) or null
  const float* x;      // (K,) input vector [PRO_COPY, PRO_LN]
  const float* gamma;  // PRO_LN
  const float* beta;
  const float* part;   // PRO_ATTN: (nhead, nsplit, 4 + hd) split-KV partials {m, l, -, -, o[hd]}
  float* y;            // output vector / residual stream / logits base
  int N, K;
  int pro, epi;
  // EPI_QKV
  float* q;            // (d,)
  void* kcache;        // this layer's K: (nhead, ctx_max, hd) WT
  void* vcache;
  int d, hd, ctx_max, nhead, nsplit;
  const ArState* st;
};

// y = W x (+epilogue).  One wave owns RPW rows at a time; a row is KCH 16-byte loads per lane.
template <typename WT, int KCH, int RPW>
__global__ __launch_bounds__(256) void gemv_kernel(const GemvArgs a) {
  constexpr int VEC = Vec16<WT>::N;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int K = a.K, N = a.N;
  const int Kpad = (K + 3) & ~3;
  float* xs = smem;           // K floats
  float* red = smem + Kpad;   // 8 floats scratch
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nwaves = gridDim.x * 4;
  const WT* __restrict__ W = reinterpret_cast<const WT*>(a.W);

  // ---- prologue part A: issue the (tiny, L2-resident) activation loads first: vmcnt retires
  // in order, so they must be older than the weight loads to be waited on separately.
  float4 xv[4];
  const int n4 = (K + 1023) >> 10;  // float4 per thread (K <= 4096)
  if (a.pro != PRO_ATTN) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int k = (i * 256 + tid) * 4;
      xv[i] = (i < n4 && k < K) ? *reinterpret_cast<const float4*>(a.x + k) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  int st_row = 0, st_pass = 0, st_trace = 0, st_done = 0;
  if (a.st) { st_row = a.st->row; st_pass = a.st->pass; st_trace = a.st->trace_logits; st_done = a.st->done; }

  // ---- weight loads of this wave's first row group
  uint4 w[RPW][KCH];
  int g = blockIdx.x * 4 + wave;
  auto issue = [&](int grp) {
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
      const int row = grp * RPW + r;
#pragma unroll
      for (int c = 0; c < KCH; ++c) {
        const int k = (c * 64 + lane) * VEC;
        w[r][c] = (row < N && k < K) ? ld16(W + (size_t)row * K + k) : make_uint4(0u, 0u, 0u, 0u);
      }
    }
  };
  issue(g);

  // ---- prologue part B: build the input vector in LDS
  if (a.pro == PRO_COPY) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int k = (i * 256 + tid) * 4;
      if (i < n4 && k < K) *reinterpret_cast<float4*>(xs + k) = xv[i];
    }
  } else if (a.pro == PRO_LN) {
    // F.layer_norm over K channels (modules/transformer.py:57-74), two-pass in registers
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) s += (xv[i].x + xv[i].y) + (xv[i].z + xv[i].w);
    const float mean = block_sum<4>(s, red) / (float)K;
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int k = (i * 256 + tid) * 4;
      if (i < n4 && k < K) {
        const float d0 = xv[i].x - mean, d1 = xv[i].y - mean, d2 = xv[i].z - mean, d3 = xv[i].w - mean;
        ss += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
      }
    }
    const float var = block_sum<4>(ss, red) / (float)K;
    const float rstd = 1.0f / sqrtf(var + LN_EPS);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int k = (i * 256 + tid) * 4;
      if (i < n4 && k < K) {
        const float4 gm = *reinterpret_cast<const float4*>(a.gamma + k);
        const float4 bt = *reinterpret_cast<const float4*>(a.beta + k);
        float4 o;
        o.x = (xv[i].x - mean) * rstd * gm.x + bt.x;
        o.y = (xv[i].y - mean) * rstd * gm.y + bt.y;
        o.z = (xv[i].z - mean) * rstd * gm.z + bt.z;
        o.w = (xv[i].w - mean) * rstd * gm.w + bt.w;
        *reinterpret_cast<float4*>(xs + k) = o;
      }
    }
  } else {  // PRO_ATTN: merge the nsplit partial softmaxes of every head (flash-decoding combine)
    const int hd = a.hd, ns = a.nsplit, stride = 4 + hd;  // 16-byte aligned o[]
    for (int k = tid * 4; k < K; k += 1024) {
      const int h = k / hd, c = k - h * hd;
      const float* p = a.part + (size_t)h * ns * stride;
      float M = -INFINITY;
      for (int s = 0; s < ns; ++s) M = fmaxf(M, p[s * stride]);
      float L = 0.f;
      float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int s = 0; s < ns; ++s) {
        const float ms = p[s * stride];
        const float f = (ms == -INFINITY) ? 0.f : expf(ms - M);
        L += p[s * stride + 1] * f;
        const float4 ov = *reinterpret_cast<const float4*>(p + s * stride + 4 + c);
        o.x += ov.x * f; o.y += ov.y * f; o.z += ov.z * f; o.w += ov.w * f;
      }
      const float inv = 1.0f / L;
      *reinterpret_cast<float4*>(xs + k) = make_float4(o.x * inv, o.y * inv, o.z * inv, o.w * inv);
    }
  }
  __syncthreads();

  // ---- this lane's slice of x, kept in registers across row groups
  float xr[KCH][VEC];
#pragma unroll
  for (int c = 0; c < KCH; ++c) {
    const int k = (c * 64 + lane) * VEC;
#pragma unroll
    for (int j = 0; j < VEC; j += 4) {
      const float4 t = (k < K) ? *reinterpret_cast<const float4*>(xs + k + j) : make_float4(0.f, 0.f, 0.f, 0.f);
      xr[c][j] = t.x; xr[c][j + 1] = t.y; xr[c][j + 2] = t.z; xr[c][j + 3] = t.w;
    }
  }

  for (;;) {
    float acc[RPW];
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
      float s = 0.f;
#pragma unroll
      for (int c = 0; c < KCH; ++c) {
        float wf[VEC];
        unpack<WT>(w[r][c], wf);
#pragma unroll
        for (int j = 0; j < VEC; ++j) s = fmaf(wf[j], xr[c][j], s);
      }
      acc[r] = wave_sum(s);
    }
    // epilogue: lane r finishes row r
    float mine = 0.f;
#pragma unroll
    for (int r = 0; r < RPW; ++r) mine = (lane == r) ? acc[r] : mine;
    const int row = g * RPW + lane;
    if (lane < RPW && row < N) {
      float v = mine + ((a.bias != nullptr) ? a.bias[row] : 0.f);
      switch (a.epi) {
        case EPI_RELU: a.y[row] = fmaxf(v, 0.f); break;
        case EPI_RESID: a.y[row] = a.y[row] + v; break;
        case EPI_LOGITS:  // a finished decode keeps replaying the step: leave its last logits row intact
          if (!st_done) a.y[(size_t)(st_trace ? st_pass : 0) * N + row] = v;
          break;
        case EPI_QKV: {
          const int sec = row / a.d, i = row - sec * a.d;
          if (sec == 0) {
            a.q[i] = v;
          } else if (!st_done) {
            const int h = i / a.hd, c = i - h * a.hd;
            WT* cache = reinterpret_cast<WT*>(sec == 1 ? a.kcache : a.vcache);
            cache[((size_t)h * a.ctx_max + st_row) * a.hd + c] = from_f32<WT>(v);
          }
        } break;
        default: a.y[row] = v; break;
      }
    }
    g += nwaves;
    if (g * RPW >= N) break;
    issue(g);
  }
}

// ---- single-query attention over the KV cache, split over keys (flash-decoding) -----------
// grid = nhead * nsplit.  Cache layout (nhead, ctx_max, HD): one head's keys are contiguous, so a
// wave-load covers 64/LPK whole keys with 16-byte lanes.  Each group of LPK lanes keeps an
// online-softmax state; groups are merged with wave shuffles, waves through LDS.
template <typename T, int HD>
__global__ __launch_bounds__(256) void attn_decode_kernel(const float* __restrict__ q, const T* __restrict__ kc,
                                                          const T* __restrict__ vc, float* __restrict__ part,
                                                          const ArState* __restrict__ st, int ctx_max, int nsplit,
                                                          float scale) {
  constexpr int VEC = Vec16<T>::N;
  constexpr int LPK = HD / VEC;   // lanes per key
  constexpr int KPW = 64 / LPK;   // keys per wave-iteration
  constexpr int UNR = 4;
  __shared__ float sm_m[4], sm_l[4], sm_o[4][HD];
  const int h = blockIdx.x / nsplit, s = blockIdx.x - h * nsplit;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int sub = lane % LPK, grp = lane / LPK;
  const int ctx = st->row + 1;
  const int chunk = (ctx + nsplit - 1) / nsplit;
  const int j0 = s * chunk, j1 = min(ctx, j0 + chunk);

  float qv[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) qv[i] = q[h * HD + sub * VEC + i];

  const T* kb = kc + (size_t)h * ctx_max * HD + sub * VEC;
  const T* vb = vc + (size_t)h * ctx_max * HD + sub * VEC;
  float m = -INFINITY, l = 0.f, acc[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) acc[i] = 0.f;

  for (int jb = j0 + wave * KPW + grp; jb < j1; jb += 4 * KPW * UNR) {
    uint4 kr[UNR], vr[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int j = jb + u * 4 * KPW;
      if (j < j1) { kr[u] = ld16(kb + (size_t)j * HD); vr[u] = ld16(vb + (size_t)j * HD); }
      else { kr[u] = make_uint4(0u, 0u, 0u, 0u); vr[u] = kr[u]; }
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int j = jb + u * 4 * KPW;
      float kf[VEC], vf[VEC];
      unpack<T>(kr[u], kf);
      unpack<T>(vr[u], vf);
      float dot = 0.f;
#pragma unroll
      for (int i = 0; i < VEC; ++i) dot = fmaf(kf[i], qv[i], dot);
#pragma unroll
      for (int o = 1; o < LPK; o <<= 1) dot += __shfl_xor(dot, o, WAVE);
      if (j < j1) {  // uniform within the LPK-lane group
        const float sc = dot * scale;
        const float mn = fmaxf(m, sc);
        const float corr = expf(m - mn);  // m = -inf -> 0
        const float p = expf(sc - mn);
        l = l * corr + p;
#pragma unroll
        for (int i = 0; i < VEC; ++i) acc[i] = acc[i] * corr + p * vf[i];
        m = mn;
      }
    }
  }
  // merge the KPW groups of this wave
#pragma unroll
  for (int o = LPK; o < 64; o <<= 1) {
    const float m2 = __shfl_xor(m, o, WAVE), l2 = __shfl_xor(l, o, WAVE);
    const float mn = fmaxf(m, m2);
    const float c1 = (m == -INFINITY) ? 0.f : expf(m - mn);
    const float c2 = (m2 == -INFINITY) ? 0.f : expf(m2 - mn);
    l = l * c1 + l2 * c2;
#pragma unroll
    for (int i = 0; i < VEC; ++i) acc[i] = acc[i] * c1 + __shfl_xor(acc[i], o, WAVE) * c2;
    m = mn;
  }
  if (grp == 0) {
    if (sub == 0) { sm_m[wave] = m; sm_l[wave] = l; }
#pragma unroll
    for (int i = 0; i < VEC; ++i) sm_o[wave][sub * VEC + i] = acc[i];
  }
  __syncthreads();
  if (tid < HD) {
    float M = fmaxf(fmaxf(sm_m[0], sm_m[1]), fmaxf(sm_m[2], sm_m[3]));
    float L = 0.f, o = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const float f = (sm_m[w] == -INFINITY) ? 0.f : expf(sm_m[w] - M);
      L += sm_l[w] * f;
      o += sm_o[w][tid] * f;
    }
    float* p = part + (size_t)blockIdx.x * (4 + HD);
    if (tid == 0) { p[0] = M; p[1] = L; }
    p[4 + tid] = o;
  }
}

// ---- sampling + stop rule + next-token embedding (valle.py:1040-1057, 1287-1302) -----------
struct SampleArgs {
  const float* logits;   // base of the logits rows (V per pass when tracing, else one row)
  int V;                 // 1025
  ArState* st;
  int* tokens;           // appended tokens (n_gen)
  int* sampled;          // per pass: what the multinomial drew
  int* argmaxes;         // per pass: argmax of the raw logits
  const float* emb;      // ar_audio_embedding (rows, d) fp32
  const float* alpha;    // ar_audio_position.alpha (1,)
  const float* pe;       // sine table (pe_rows, d) fp32
  float* x;              // (d,) residual stream input of the next pass
  int d;
};

__device__ __forceinline__ uint32_t order_key(float v) {
  const uint32_t u = __float_as_uint(v);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__device__ __forceinline__ float device_exp1(unsigned long long seed, int pass, int i) {
  unsigned long long z = seed + 0x9E3779B97F4A7C15ull * ((unsigned long long)pass * 2048ull + (unsigned long long)i + 1ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z ^= z >> 31;
  const float u = ((float)(z >> 40) + 0.5f) * (1.0f / 16777216.0f);  // (0,1)
  return -logf(u);
}

// One workgroup of 1024 threads; thread t owns logits t and t+1024.
__global__ __launch_bounds__(1024) void sample_embed_kernel(const SampleArgs a) {
  __shared__ float redv[16];
  __shared__ int redi[16];
  __shared__ int cnt[2][16];
  __shared__ int s_tok, s_go;
  ArState* st = a.st;
  if (st->done) return;  // uniform
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int V = a.V;
  const int pass = st->pass;
  const float* lg = a.logits + (st->trace_logits ? (size_t)pass * V : 0);
  const bool has0 = t < V, has1 = (t + 1024) < V;
  float v0 = has0 ? lg[t] : -INFINITY;
  float v1 = has1 ? lg[t + 1024] : -INFINITY;

  // argmax of the raw logits (valle.py:1045); first index on ties
  ValIdx c0{v0, t}, c1{v1, t + 1024};
  const ValIdx am = block_argmax<16>(better(c0, c1), redv, redi);

  const float temp = st->temperature;
  if (temp != 1.0f) { v0 = v0 / temp; v1 = v1 / temp; }  // valle.py:1296-1297

  // top-k: keep v >= (k-th largest value), ties kept (valle.py:1254-1260).  The threshold is
  // found by a 32-step bitwise select on order-preserving integer keys: exact, k-independent.
  bool keep0 = has0, keep1 = has1;
  int k = st->top_k;
  if (k > 0 && k < V) {
    const uint32_t key0 = has0 ? order_key(v0) : 0u, key1 = has1 ? order_key(v1) : 0u;
    uint32_t T = 0u;
    for (int b = 31; b >= 0; --b) {
      const uint32_t cand = T | (1u << b);
      const int c = __popcll(__ballot(key0 >= cand)) + __popcll(__ballot(key1 >= cand));
      if (lane == 0) cnt[b & 1][wave] = c;
      __syncthreads();
      int tot = 0;
#pragma unroll
      for (int i = 0; i < 16; ++i) tot += cnt[b & 1][i];
      if (tot >= k) T = cand;
    }
    keep0 = has0 && key0 >= T;
    keep1 = has1 && key1 >= T;
  }

  // softmax over the kept entries (F.softmax, valle.py:1301)
  const float mx = block_max<16>(fmaxf(keep0 ? v0 : -INFINITY, keep1 ? v1 : -INFINITY), redv);
  const float e0 = keep0 ? expf(v0 - mx) : 0.f, e1 = keep1 ? expf(v1 - mx) : 0.f;
  const float Z = block_sum<16>(e0 + e1, redv);
  const float p0 = e0 / Z, p1 = e1 / Z;

  // multinomial(p, 1) == argmax(p / q), q ~ Exp(1)
  float q0, q1;
  if (a.st->exp_noise != nullptr) {
    const float* nz = a.st->exp_noise + (size_t)min((long long)pass, a.st->noise_rows - 1) * V;
    q0 = has0 ? nz[t] : 1.f;
    q1 = has1 ? nz[t + 1024] : 1.f;
  } else {
    q0 = device_exp1(st->seed, pass, t);
    q1 = device_exp1(st->seed, pass, t + 1024);
  }
  ValIdx r0{has0 ? p0 / q0 : -1.f, t}, r1{has1 ? p1 / q1 : -1.f, t + 1024};
  const ValIdx smp = block_argmax<16>(better(r0, r1), redv, redi);

  // stop rule + append (valle.py:1044-1057); thread 0 owns the state
  if (t == 0) {
    a.sampled[pass] = smp.i;
    a.argmaxes[pass] = am.i;
    int tok = smp.i, go = 0, reason = 0;
    const int n_gen = st->n_gen;
    const bool forcing = st->forced != nullptr;
    if (forcing) {
      if (pass >= st->n_forced) reason = 4; else tok = (int)st->forced[pass];
    } else if (am.i == NUM_AUDIO_TOKENS) reason = 1;
    else if (smp.i == NUM_AUDIO_TOKENS) reason = 2;
    else if (st->bos + n_gen > 16 * st->S) reason = 3;
    else if (st->max_new >= 0 && n_gen >= st->max_new) reason = 4;
    if (reason == 0) {
      a.tokens[n_gen] = tok;
      st->n_gen = n_gen + 1;
      // after this append the next pass can only stop (valle.py:1047): skip computing it
      if (!forcing && st->bos + n_gen + 1 > 16 * st->S) reason = 3;
      else if (!forcing && st->max_new >= 0 && n_gen + 1 >= st->max_new) reason = 4;
      if (reason == 0) go = 1;
    }
    if (reason != 0) { st->done = 1; st->stop_reason = reason; }
    s_tok = tok;
    s_go = go;
  }
  __syncthreads();
  if (!s_go) return;
  // x = E[tok] * 1.0 + alpha * pe[audio position] (valle.py:1013-1015; embedding.py:93-97)
  const int tok = s_tok;
  const int row = st->row + 1;                 // KV row of the new token
  const int apos = row - st->S;                // position inside the audio sub-sequence
  const float alpha = a.alpha[0];
  for (int c = t; c < a.d; c += 1024)
    a.x[c] = __fadd_rn(a.emb[(size_t)tok * a.d + c], __fmul_rn(alpha, a.pe[(size_t)apos * a.d + c]));
  __syncthreads();
  if (t == 0) { st->row = row; st->pass = pass + 1; }
}

}  // namespace vx
