// Kernels of the batch-1 AR decode step (valle/models/valle.py:1012-1057 with a KV cache).
//
// One token = 12 layers x {QKV GEMV, split-KV attention, out-proj GEMV, FFN1 GEMV, FFN2 GEMV}
// + head GEMV + sampling = 62 dependent launches over 304 MB of bf16 weights, so the step is
// bounded by HBM streaming plus one launch boundary per all-to-all dependency.  Each kernel is
// built to have ONE memory round trip on its critical path:
//   * every small activation-side load (x, gamma, beta, bias, residual, state) is issued first,
//     the 16-byte weight loads right after; vmcnt retires in order, so the LayerNorm runs on the
//     early loads while the weight stream is still in flight;
//   * LayerNorm is recomputed per wave in registers (x is 4 KB, L2-resident) with DPP
//     reductions: no LDS, no workgroup barrier in the GEMV (the attention-combine prologue of
//     the out-projection is the one exception, one barrier);
//   * weights go straight to VGPRs (GEMV / M<=16 row of cdna_hip_programming.md §5).
#pragma once
#include "common.hpp"

namespace vx {

enum GemvPro { PRO_COPY = 0, PRO_LN = 1, PRO_ATTN = 2 };
enum GemvEpi { EPI_PLAIN = 0, EPI_BIAS = 1, EPI_RELU = 2, EPI_RESID = 3, EPI_QKV = 4, EPI_LOGITS = 5, EPI_POS = 6 };  // POS: + bias + alpha * pe[audio position] (last prenet layer)

constexpr int LOGITS_CUR = 1088;     // floats reserved for the newest logits row at the buffer head; trace rows follow
#ifndef VX_ATT_NSPLIT
#define VX_ATT_NSPLIT 8  // A/B builds: hipcc -DVX_ATT_NSPLIT=4 (profiles/r02_notes.md)
#endif
constexpr int ATT_NSPLIT = VX_ATT_NSPLIT;  // key splits per head in the decode attention
constexpr int ATT_PSTRIDE = 4 + 64;  // floats per partial: {m, l, -, -, o[64]}

struct GemvArgs {
  const void* W;       // (N, K) row-major, WT
  const float* bias;   // (N,) or null
  const float* x;      // (K,) input vector [PRO_COPY, PRO_LN]
  const float* gamma;  // PRO_LN
  const float* beta;
  const float* part;   // PRO_ATTN: (nhead, ATT_NSPLIT, ATT_PSTRIDE) split-KV partials
  float* y;            // output vector / residual stream / logits base
  const float* res;    // EPI_RESID: residual source (null: y itself).  Post-norm layers add to the NORMALISED stream
  const float* pe;     // EPI_POS: sine table (rows, N) and its alpha (valle.py:1014-1015 with add_prenet)
  const float* pos_alpha;
  float* xnorm_out;    // PRO_LN: if set, workgroup 0 / wave 0 also stores LN(x) here (post-norm: the next residual base)
  int N, K;
  int pro, epi;
  // EPI_QKV
  float* q;            // (d,)
  void* kcache;        // this layer's K: (nhead, ctx_max, hd) WT
  void* vcache;
  int d, hd, ctx_max, nhead;
  const ArState* st;
  // NPF > 0: cache warm-up for a LATER GEMV of the decode step (engine.hip picks which: two places ahead).  Workgroup b
  // touches bytes [b pf_slice, (b+1) pf_slice) of `pf` - the slice workgroup b of that GEMV streams.  The per-XCD L2s drop the
  // lines at the kernel boundary; what the consumer gains is a hit in the memory-side Infinity Cache.  Speed only: the values
  // are never used.
  const void* pf;
  unsigned pf_slice, pf_total;
  int kid;  // position in the decode step (probe builds: VX_KSTAMP; -1 = not stamped)
  int nt;   // non-temporal weight loads (decode step, VX_AR_NT)
};

// y = W x (+epilogue).  One wave owns RPW rows at a time; a row is KCH 16-byte loads per lane;
// lane l holds x[(c*64 + l)*VEC .. +VEC) for chunk c.
// The operands of the kernel's FIRST loads (activation vector, LayerNorm affine, weights, sizes) are explicit leading
// arguments: with the build's `-amdgpu-kernarg-preload-count` the dispatcher places them in SGPRs at wave launch (14 are
// available), so those loads do not wait for a kernarg fetch - one memory round trip per wave otherwise.  hipcc does not
// preload by-value structs, so GemvArgs alone would not qualify; everything else is read from it after the weight loads
// are out.   xin = a.part for PRO_ATTN, a.x otherwise;  nk = (N << 16) | K.
// NT: the weight stream uses non-temporal loads (global_load_dwordx4 ... nt): each weight byte is read once per token by one CU.
template <typename WT, int KCH, int RPW, int PRO, int NPF = 0, bool NT = false>
__global__ __launch_bounds__(256) void gemv_kernel(const void* __restrict__ W_, const float* __restrict__ xin,
                                                   const float* __restrict__ gamma_, const float* __restrict__ beta_,
                                                   unsigned nk, const GemvArgs a) {
  VX_KSTAMP_WG(a.kid, a.st);  // probe builds: the extra last workgroup of a stamped launch only records the time
  constexpr int VEC = Vec16<WT>::N;
  constexpr int V4 = VEC / 4;
  __shared__ __attribute__((aligned(16))) float xs[PRO == PRO_ATTN ? 1024 : 4];
  const int K = (int)(nk & 0xffffu), N = (int)(nk >> 16);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nwaves = (gridDim.x - (a.kid >= 0 ? VX_KSTAMP_EXTRA : 0)) * 4;
  const WT* __restrict__ W = reinterpret_cast<const WT*>(W_);
  int g = blockIdx.x * 4 + wave;

  // ---- (A) small loads, issued first.  Every load is unconditional on a clamped (in-range)
  // address and masked afterwards: a predicated load would put an exec-masked branch and a
  // vmcnt(0) in front of the weight stream.
  float4 x4[KCH][V4], g4[KCH][V4], b4[KCH][V4];
  bool kok[KCH];
#pragma unroll
  for (int c = 0; c < KCH; ++c) kok[c] = (c * 64 + lane) * VEC < K;
  if (PRO != PRO_ATTN) {
#pragma unroll
    for (int c = 0; c < KCH; ++c) {
      const int k = min((c * 64 + lane) * VEC, K - VEC);
#pragma unroll
      for (int j = 0; j < V4; ++j) {
        x4[c][j] = *reinterpret_cast<const float4*>(xin + k + 4 * j);
        if (PRO == PRO_LN) {
          g4[c][j] = *reinterpret_cast<const float4*>(gamma_ + k + 4 * j);
          b4[c][j] = *reinterpret_cast<const float4*>(beta_ + k + 4 * j);
        }
      }
    }
  }
  // attention partials: thread t combines channels [4t, 4t+4) (K = d <= 1024)
  float pm[ATT_NSPLIT], pl[ATT_NSPLIT];
  float4 po[ATT_NSPLIT];
  const int ka = min(tid * 4, K - 4);
  if (PRO == PRO_ATTN) {
    const int h = ka / a.hd, c = ka - h * a.hd;
    const float* p = xin + (size_t)h * ATT_NSPLIT * ATT_PSTRIDE;
#pragma unroll
    for (int s = 0; s < ATT_NSPLIT; ++s) {
      const float2 ml = *reinterpret_cast<const float2*>(p + s * ATT_PSTRIDE);
      pm[s] = ml.x; pl[s] = ml.y;
      po[s] = *reinterpret_cast<const float4*>(p + s * ATT_PSTRIDE + 4 + c);
    }
  }

  // ---- (B) weight stream of the first row group (rows/k clamped; x is zero where k >= K) ----
  uint4 w[RPW][KCH];
  auto issue = [&](int grp) {
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
      const int row = min(grp * RPW + r, N - 1);
#pragma unroll
      for (int c = 0; c < KCH; ++c) {
        const int k = min((c * 64 + lane) * VEC, K - VEC);
        w[r][c] = NT ? ld16nt(W + (size_t)row * K + k) : ld16(W + (size_t)row * K + k);
      }
    }
  };
  issue(g);
  __builtin_amdgcn_sched_barrier(0);
  // epilogue operands of this wave's first row group: lane r owns row r (clamped, masked at the store).  They come from
  // the by-value struct, i.e. behind a kernarg fetch: placed after the weight loads, in front of the warm-up loads (the
  // epilogue's wait must not cover those)
  const float* __restrict__ bias_ = a.bias;
  const float* rsrc = a.epi == EPI_RESID ? (a.res ? a.res : a.y) : nullptr;
  const ArState* __restrict__ st_ = a.st;
  const bool has_bias = bias_ != nullptr, has_res = rsrc != nullptr;  // wave-uniform
  float e_bias = 0.f, e_res = 0.f;
  {
    const int rc = min(g * RPW + min(lane, RPW - 1), N - 1);
    if (has_bias) e_bias = bias_[rc];
    if (has_res) e_res = rsrc[rc];
  }
  // next kernel's weights, behind this kernel's own stream (vmcnt retires in order: the waits below do not cover them)
  uint4 pfv[NPF > 0 ? NPF : 1];
  if (NPF > 0) {
    const unsigned lim = min(a.pf_slice, a.pf_total - min(a.pf_total, blockIdx.x * a.pf_slice));  // bytes of this slice inside pf
    const char* pb = reinterpret_cast<const char*>(a.pf) + min((size_t)blockIdx.x * a.pf_slice, (size_t)a.pf_total - 16);
#pragma unroll
    for (int i = 0; i < NPF; ++i)
      pfv[i] = *reinterpret_cast<const uint4*>(pb + min((unsigned)(i * 4096 + tid * 16), max(lim, 16u) - 16u));
  }
  // pin the weight loads HERE: without this the scheduler sinks them below the prologue, next to
  // their first use, and the HBM round trip is serialised behind the LayerNorm
  __builtin_amdgcn_sched_barrier(0);
  // decode state (scalar loads, used by the epilogues only): after the warm-up loads, so that the wait for the warm-up's
  // own arguments does not also wait for these
  int st_row = 0, st_pass = 0, st_trace = 0, st_done = 0, st_S = 0;
  if (st_) { st_row = st_->row; st_pass = st_->pass; st_trace = st_->trace_logits; st_done = st_->done; st_S = st_->kv_text; }

  // ---- (C) activation prologue in registers ---------------------------------------------------
  float xr[KCH][VEC];
  if (PRO == PRO_ATTN) {
    {  // flash-decoding combine of the ATT_NSPLIT partial softmaxes
      float M = pm[0];
#pragma unroll
      for (int s = 1; s < ATT_NSPLIT; ++s) M = fmaxf(M, pm[s]);
      float L = 0.f;
      float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int s = 0; s < ATT_NSPLIT; ++s) {
        const float f = (pm[s] == -INFINITY) ? 0.f : expf(pm[s] - M);
        L += pl[s] * f;
        o.x += po[s].x * f; o.y += po[s].y * f; o.z += po[s].z * f; o.w += po[s].w * f;
      }
      const float inv = 1.0f / L;
      // threads past K recompute the last quad and store the same values (benign)
      *reinterpret_cast<float4*>(xs + ka) = make_float4(o.x * inv, o.y * inv, o.z * inv, o.w * inv);
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < KCH; ++c) {
      const int k2 = min((c * 64 + lane) * VEC, K - VEC);
#pragma unroll
      for (int j = 0; j < V4; ++j) {
        const float4 t = *reinterpret_cast<const float4*>(xs + k2 + 4 * j);
        xr[c][4 * j] = kok[c] ? t.x : 0.f; xr[c][4 * j + 1] = kok[c] ? t.y : 0.f;
        xr[c][4 * j + 2] = kok[c] ? t.z : 0.f; xr[c][4 * j + 3] = kok[c] ? t.w : 0.f;
      }
    }
  } else {
#pragma unroll
    for (int c = 0; c < KCH; ++c)
#pragma unroll
      for (int j = 0; j < V4; ++j) {
        xr[c][4 * j] = kok[c] ? x4[c][j].x : 0.f; xr[c][4 * j + 1] = kok[c] ? x4[c][j].y : 0.f;
        xr[c][4 * j + 2] = kok[c] ? x4[c][j].z : 0.f; xr[c][4 * j + 3] = kok[c] ? x4[c][j].w : 0.f;
      }
    if (PRO == PRO_LN) {  // F.layer_norm (modules/transformer.py:57-74), two-pass, whole row in this wave
      float s = 0.f;
#pragma unroll
      for (int c = 0; c < KCH; ++c)
#pragma unroll
        for (int j = 0; j < VEC; ++j) s += xr[c][j];  // out-of-range lanes hold zeros
      const float mean = wave_sum_dpp(s) / (float)K;
      float ss = 0.f;
#pragma unroll
      for (int c = 0; c < KCH; ++c) {
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
          const float dv = xr[c][j] - mean;
          ss += kok[c] ? dv * dv : 0.f;
        }
      }
      const float rstd = 1.0f / sqrtf(wave_sum_dpp(ss) / (float)K + LN_EPS);
#pragma unroll
      for (int c = 0; c < KCH; ++c)
#pragma unroll
        for (int j = 0; j < V4; ++j) {
          const float o0 = (xr[c][4 * j] - mean) * rstd * g4[c][j].x + b4[c][j].x;
          const float o1 = (xr[c][4 * j + 1] - mean) * rstd * g4[c][j].y + b4[c][j].y;
          const float o2 = (xr[c][4 * j + 2] - mean) * rstd * g4[c][j].z + b4[c][j].z;
          const float o3 = (xr[c][4 * j + 3] - mean) * rstd * g4[c][j].w + b4[c][j].w;
          xr[c][4 * j] = kok[c] ? o0 : 0.f; xr[c][4 * j + 1] = kok[c] ? o1 : 0.f;
          xr[c][4 * j + 2] = kok[c] ? o2 : 0.f; xr[c][4 * j + 3] = kok[c] ? o3 : 0.f;
        }
      if (a.xnorm_out != nullptr && blockIdx.x == 0 && wave == 0) {  // wave-uniform; stores only, after every load was issued
#pragma unroll
        for (int c = 0; c < KCH; ++c)
          if (kok[c]) {
#pragma unroll
            for (int j = 0; j < V4; ++j)
              *reinterpret_cast<float4*>(a.xnorm_out + (c * 64 + lane) * VEC + 4 * j) =
                  make_float4(xr[c][4 * j], xr[c][4 * j + 1], xr[c][4 * j + 2], xr[c][4 * j + 3]);
          }
      }
    }
  }

  // ---- (D) dot products, wave reduction, epilogue -------------------------------------------
  // The first row group runs outside the loop: merged into it, the wait in front of the dot products is the loop's
  // (`vmcnt(7..0)`, sized for a later pass) and covers the warm-up loads as well.
  auto rows = [&]() {
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
      acc[r] = s;
    }
#pragma unroll
    for (int r = 0; r < RPW; ++r) acc[r] = wave_sum_dpp(acc[r]);
    float mine = 0.f;
#pragma unroll
    for (int r = 0; r < RPW; ++r) mine = (lane == r) ? acc[r] : mine;
    const int row = g * RPW + lane;
    if (lane < RPW && row < N) {
      const float v = mine + e_bias;
      switch (a.epi) {
        case EPI_RELU: a.y[row] = fmaxf(v, 0.f); break;
        case EPI_RESID: a.y[row] = e_res + v; break;
        case EPI_POS: a.y[row] = __fadd_rn(v, __fmul_rn(a.pos_alpha[0], a.pe[(size_t)(st_row - st_S) * N + row])); break;
        case EPI_LOGITS:  // a finished decode keeps replaying the step: leave its last logits row intact
          if (!st_done) {
            a.y[row] = v;  // fixed address: the sampling kernel's loads do not wait for the step counter
            if (st_trace) a.y[LOGITS_CUR + (size_t)st_pass * N + row] = v;
          }
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
  };
  rows();
  for (;;) {
    g += nwaves;
    if (g * RPW >= N) break;
    {
      const int rc = min(g * RPW + min(lane, RPW - 1), N - 1);
      if (has_bias) e_bias = bias_[rc];
      if (has_res) e_res = rsrc[rc];
    }
    issue(g);
    rows();
  }
  if (NPF > 0) {  // the warm-up loads stay live (and unwaited) until here
#pragma unroll
    for (int i = 0; i < NPF; ++i) asm volatile("" ::"v"(pfv[i].x), "v"(pfv[i].y), "v"(pfv[i].z), "v"(pfv[i].w));
  }
}

// ---- single-query attention over the KV cache, split over keys (flash-decoding) -----------
// grid = nhead * ATT_NSPLIT workgroups of 256.  Cache layout (nhead, ctx_max, 64): one head's keys
// are contiguous, a wave-load covers 64/LPK whole keys with 16-byte lanes.  Two passes over
// registers: scores of all of this workgroup's keys (K and V loads issued together up front),
// one workgroup max, then p = exp(s - max) and P.V with plain sums — no per-key rescale.
template <typename T, int HD>
__global__ __launch_bounds__(256) void attn_decode_kernel(const float* __restrict__ q, const T* __restrict__ kc,
                                                          const T* __restrict__ vc, float* __restrict__ part,
                                                          const ArState* __restrict__ st, int ctx_max, float scale,
                                                          int kid, int fixed_ctx = 0) {
  VX_KSTAMP_WG(kid, st);
  constexpr int VEC = Vec16<T>::N;
  constexpr int LPK = HD / VEC;        // lanes per key: 8 (bf16) / 16 (fp32)
  constexpr int KPW = 64 / LPK;        // keys per wave-load
  constexpr int KPB = 4 * KPW;         // keys per workgroup round
  constexpr int UNR = 6;               // rounds held in registers: UNR * KPB keys per outer pass
  __shared__ float sm_red[4];
  __shared__ __attribute__((aligned(16))) float sm_o[4 * KPW][HD + 1];
  __shared__ float sm_l[4 * KPW];
  const int h = blockIdx.x / ATT_NSPLIT, s = blockIdx.x - h * ATT_NSPLIT;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int sub = lane % LPK, grp = lane / LPK;
  // fixed_ctx != 0: cross-attention over the cached text memory (VALL-F).  Its length is read from the decode state (uploaded by
  // every prefill), never taken by value: the step is captured ONCE per engine and replayed for every later utterance
  const int ctx = fixed_ctx != 0 ? st->S : st->row + 1;
  const int chunk = (ctx + ATT_NSPLIT - 1) / ATT_NSPLIT;
  const int j0 = s * chunk, j1 = min(ctx, j0 + chunk);
  float qv[VEC];
#pragma unroll
  for (int i = 0; i < VEC; i += 4) {
    const float4 t = *reinterpret_cast<const float4*>(q + h * HD + sub * VEC + i);
    qv[i] = t.x; qv[i + 1] = t.y; qv[i + 2] = t.z; qv[i + 3] = t.w;
  }
  const T* kb = kc + (size_t)h * ctx_max * HD + sub * VEC;
  const T* vb = vc + (size_t)h * ctx_max * HD + sub * VEC;

  float M = -INFINITY, L = 0.f, acc[VEC];  // carried across outer passes (one pass while ctx <= NSPLIT*UNR*KPB)
#pragma unroll
  for (int i = 0; i < VEC; ++i) acc[i] = 0.f;

  for (int base = j0; base < j1; base += UNR * KPB) {
    uint4 kr[UNR], vr[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int j = base + u * KPB + wave * KPW + grp;
      if (j < j1) { kr[u] = ld16(kb + (size_t)j * HD); vr[u] = ld16(vb + (size_t)j * HD); }
      else { kr[u] = make_uint4(0u, 0u, 0u, 0u); vr[u] = kr[u]; }
    }
    float sc[UNR], mloc = -INFINITY;
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int j = base + u * KPB + wave * KPW + grp;
      float kf[VEC];
      unpack<T>(kr[u], kf);
      float dot = 0.f;
#pragma unroll
      for (int i = 0; i < VEC; ++i) dot = fmaf(kf[i], qv[i], dot);
      dot = (LPK == 8) ? group8_sum_dpp(dot) : group16_sum_dpp(dot);
      sc[u] = (j < j1) ? dot * scale : -INFINITY;
      mloc = fmaxf(mloc, sc[u]);
    }
    // workgroup max of this pass (uniform trip count: base/j1 are workgroup-uniform)
    mloc = wave_max_dpp(mloc);
    __syncthreads();
    if (lane == 0) sm_red[wave] = mloc;
    __syncthreads();
    const float mb = fmaxf(fmaxf(sm_red[0], sm_red[1]), fmaxf(sm_red[2], sm_red[3]));
    const float Mn = fmaxf(M, mb);
    const float corr = (M == -INFINITY) ? 0.f : expf(M - Mn);  // 0 on the first pass
    L *= corr;
#pragma unroll
    for (int i = 0; i < VEC; ++i) acc[i] *= corr;
    M = Mn;
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      float vf[VEC];
      unpack<T>(vr[u], vf);
      const float p = (sc[u] == -INFINITY) ? 0.f : expf(sc[u] - M);
      L += p;
#pragma unroll
      for (int i = 0; i < VEC; ++i) acc[i] = fmaf(p, vf[i], acc[i]);
    }
  }
  // sum the 4*KPW key groups: LDS transpose, thread c sums channel c
  const int gi = wave * KPW + grp;
#pragma unroll
  for (int i = 0; i < VEC; ++i) sm_o[gi][sub * VEC + i] = acc[i];
  if (sub == 0) sm_l[gi] = L;
  __syncthreads();
  if (tid < HD) {
    float o = 0.f, l = 0.f;
#pragma unroll
    for (int gidx = 0; gidx < 4 * KPW; ++gidx) { o += sm_o[gidx][tid]; l += sm_l[gidx]; }
    float* p = part + (size_t)blockIdx.x * ATT_PSTRIDE;
    if (tid == 0) { p[0] = M; p[1] = l; }
    p[4 + tid] = o;
  }
}

// ---- the same split-KV partials for head sizes other than 64 (4 .. 32: the reference's own tests run head_dim 4,
// valle_test.py:93-95).  Plain per-thread online softmax over every 256th key of the split, merged through LDS;
// nothing here is tuned - these geometries are test-sized.
template <typename T, int HD>
__global__ __launch_bounds__(256) void attn_decode_small_kernel(const float* __restrict__ q, const T* __restrict__ kc,
                                                                const T* __restrict__ vc, float* __restrict__ part,
                                                                const ArState* __restrict__ st, int ctx_max, float scale,
                                                                int kid, int fixed_ctx = 0) {
  __shared__ float sm_m[256], sm_l[256];
  __shared__ float sm_o[256][HD + 1];
  const int h = blockIdx.x / ATT_NSPLIT, s = blockIdx.x - h * ATT_NSPLIT;
  const int tid = threadIdx.x;
  const int ctx = fixed_ctx != 0 ? st->S : st->row + 1;  // see attn_decode_kernel
  const int chunk = (ctx + ATT_NSPLIT - 1) / ATT_NSPLIT;
  const int j0 = s * chunk, j1 = min(ctx, j0 + chunk);
  float qv[HD], o[HD];
#pragma unroll
  for (int c = 0; c < HD; ++c) { qv[c] = q[h * HD + c]; o[c] = 0.f; }
  float m = -INFINITY, l = 0.f;
  for (int j = j0 + tid; j < j1; j += 256) {
    const T* kr = kc + ((size_t)h * ctx_max + j) * HD;
    const T* vr = vc + ((size_t)h * ctx_max + j) * HD;
    float dot = 0.f;
#pragma unroll
    for (int c = 0; c < HD; ++c) dot = fmaf(to_f32(kr[c]), qv[c], dot);
    const float sc = dot * scale, mn = fmaxf(m, sc);
    const float corr = (m == -INFINITY) ? 0.f : expf(m - mn), pj = expf(sc - mn);
    l = l * corr + pj;
#pragma unroll
    for (int c = 0; c < HD; ++c) o[c] = o[c] * corr + pj * to_f32(vr[c]);
    m = mn;
  }
  sm_m[tid] = m;
  __syncthreads();
  float M = -INFINITY;
  for (int i = 0; i < 256; ++i) M = fmaxf(M, sm_m[i]);
  const float f = (m == -INFINITY) ? 0.f : expf(m - M);
  sm_l[tid] = l * f;
#pragma unroll
  for (int c = 0; c < HD; ++c) sm_o[tid][c] = o[c] * f;
  __syncthreads();
  if (tid < HD) {
    float so = 0.f, sl = 0.f;
    for (int i = 0; i < 256; ++i) { so += sm_o[i][tid]; sl += sm_l[i]; }
    float* p = part + (size_t)blockIdx.x * ATT_PSTRIDE;
    if (tid == 0) { p[0] = M; p[1] = sl; }
    p[4 + tid] = so;
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
  // batched decode: workgroup = slot; per-slot strides (0 in the batch-1 step, grid = 1)
  int logits_stride, tok_stride;
  int kid;  // probe builds: VX_KSTAMP id (-1 = not stamped)
  unsigned* epoch;  // batch-1 step: launch counter, bumped once per step whatever the decode state (tags the hand-overs of ar_granules.hpp)
  long long* zero_acc;  // sharded step: (1024,) int64 accumulator this launch zeroes for the step it opens (ar_tp.hpp), or null
};

__device__ __forceinline__ uint32_t order_key(float v) {
  const uint32_t u = __float_as_uint(v);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// Exp(1) draw of the on-device sampler: counter-based 32-bit hash of (seed, pass, index).  64-bit
// multiplies are quarter-rate VALU sequences; this mix uses two 32-bit ones.
__device__ __forceinline__ float device_exp1(unsigned long long seed, int pass, int i) {
  uint32_t x = (uint32_t)(seed ^ (seed >> 32)) + 0x9E3779B9u * (uint32_t)(pass + 1) + 0x632BE5ABu * (uint32_t)(i + 1);
  x ^= x >> 16; x *= 0x7feb352du;
  x ^= x >> 15; x *= 0x846ca68bu;
  x ^= x >> 16;
  const float u = ((float)(x >> 8) + 0.5f) * (1.0f / 16777216.0f);  // (0,1)
  return -__logf(u);
}

// k-th largest of one key per lane (needs >= k non-zero keys): bitwise select, ballot counts.
__device__ __forceinline__ uint32_t kth_largest_1reg(uint32_t x, int k) {
  uint32_t T = 0u;
  for (int b = 31; b >= 0; --b) {
    const uint32_t cand = T | (1u << b);
    const int c = __popcll(__ballot(x >= cand));
    if (c >= k) {
      T = cand;
      if (c == k) return wave_umin_dpp(x >= cand ? x : 0xffffffffu);
    }
  }
  return T;
}

// Exact k-th largest of the NV*64 keys one wave holds (NV per lane, 0 = absent): the top-k threshold of
// valle.py:1254-1260 on order-preserving keys.  Counts are ballots (scalar), so every branch is wave-uniform.
// For k <= 64 the search is first narrowed: the k-th largest of the 64 per-lane maxima is a lower bound L of the
// answer, so only keys >= L (usually k..k+3 of them) can matter; they are compacted into one register through
// LDS (same wave writes and reads: the LDS queue is in order, no barrier) and selected there.
template <int NV>
__device__ __forceinline__ uint32_t topk_threshold_wave(const uint32_t (&key)[NV], int top_k, int lane, uint32_t* cand_lds) {
  if (top_k <= 64) {
    uint32_t lm = key[0];
#pragma unroll
    for (int j = 1; j < NV; ++j) lm = max(lm, key[j]);
    const uint32_t L = kth_largest_1reg(lm, top_k);
    int cL = 0;
#pragma unroll
    for (int j = 0; j < NV; ++j) cL += __popcll(__ballot(key[j] >= L));
    if (cL == top_k) return L;
    if (cL <= 64) {
      int base = 0;
#pragma unroll
      for (int j = 0; j < NV; ++j) {
        const unsigned long long mk = __ballot(key[j] >= L);
        const int pos = base + __builtin_amdgcn_mbcnt_hi((unsigned)(mk >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mk, 0u));
        if (key[j] >= L) cand_lds[pos] = key[j];
        base += __popcll(mk);
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      const uint32_t c = (lane < cL) ? cand_lds[lane] : 0u;
      return kth_largest_1reg(c, top_k);
    }
  }
  uint32_t T = 0u;
  for (int b = 31; b >= 0; --b) {
    const uint32_t cand = T | (1u << b);
    int c = 0;
#pragma unroll
    for (int j = 0; j < NV; ++j) c += __popcll(__ballot(key[j] >= cand));
    if (c >= top_k) {
      T = cand;
      if (c == top_k) {  // the kept set is exactly {key >= cand}: its minimum is the k-th largest
        uint32_t mn = 0xffffffffu;
#pragma unroll
        for (int j = 0; j < NV; ++j) mn = (key[j] >= cand) ? min(mn, key[j]) : mn;
        return wave_umin_dpp(mn);
      }
    }
  }
  return T;
}

// ONE wave: lane l owns logits l, l+64, ..., l+64*(NV-1) in registers.  Everything the stop rule
// depends on is a wave-level DPP reduction or a ballot — no LDS, no barrier on the token's critical path.
template <int NV>
__global__ __launch_bounds__(64) void sample_embed_kernel(const SampleArgs a) {
  const int slot = blockIdx.x;
  ArState* st = a.st + slot;
  if (st->done) return;  // uniform
  const int lane = threadIdx.x;
  const int V = a.V;
  const int pass = st->pass;
  const float* lg = a.logits + (size_t)slot * a.logits_stride;  // newest row, fixed address
  int* const tokens = a.tokens + (size_t)slot * a.tok_stride;
  int* const sampled = a.sampled + (size_t)slot * a.tok_stride;
  int* const argmaxes = a.argmaxes + (size_t)slot * a.tok_stride;
  float* const xout = a.x + (size_t)slot * a.d;
  const float* nz = st->exp_noise;
  if (nz != nullptr) nz += (size_t)min((long long)pass, st->noise_rows - 1) * V;
  float v[NV], qn[NV];
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const int i = j * 64 + lane;
    v[j] = (i < V) ? lg[i] : -INFINITY;
    qn[j] = (nz != nullptr && i < V) ? nz[i] : 1.f;
  }
  const float temp = st->temperature;
  const int top_k = st->top_k;
  const unsigned long long seed = st->seed;
  const int n_gen = st->n_gen, bos = st->bos, S = st->S, max_new = st->max_new, n_forced = st->n_forced;
  const long long* forced = st->forced;
  const int row = st->row + 1;  // KV row of the new token
  const float alpha = a.alpha[0];

  // argmax of the raw logits (valle.py:1045); first index on ties
  ValIdx am{v[0], lane};
#pragma unroll
  for (int j = 1; j < NV; ++j) am = better(am, ValIdx{v[j], j * 64 + lane});
  am = wave_argmax_dpp(am);

  if (temp != 1.0f) {  // valle.py:1296-1297
#pragma unroll
    for (int j = 0; j < NV; ++j) v[j] = v[j] / temp;
  }

  // top-k: keep v >= (k-th largest), ties kept (valle.py:1254-1260).  The threshold is the k-th largest
  // order-preserving key, found exactly by bitwise select with ballot counts (all branches wave-uniform).
  // For k <= 64 the search is first narrowed: the k-th largest of the 64 per-lane maxima is a lower bound L
  // of the answer, so only keys >= L (usually k..k+3 of them) can matter; they are compacted into one
  // register through LDS and selected there (1 compare per round instead of 17).
  uint32_t T = 0u;
  if (top_k > 0 && top_k < V) {
    __shared__ uint32_t cand_lds[64];
    uint32_t key[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) key[j] = (j * 64 + lane < V) ? order_key(v[j]) : 0u;
    T = topk_threshold_wave<NV>(key, top_k, lane, cand_lds);
  }
  bool keep[NV];
#pragma unroll
  for (int j = 0; j < NV; ++j) keep[j] = (j * 64 + lane < V) && (T == 0u || order_key(v[j]) >= T);

  // softmax over the kept entries (F.softmax, valle.py:1301).  Registers in which no lane kept anything
  // (most of them under top-k) skip exp / RNG / the two IEEE divisions wave-uniformly.
  bool any[NV];
  float mx = -INFINITY;
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    any[j] = __ballot(keep[j]) != 0ull;
    mx = fmaxf(mx, keep[j] ? v[j] : -INFINITY);
  }
  mx = wave_max_dpp(mx);
  float e[NV], zs = 0.f;
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    e[j] = 0.f;
    if (any[j]) { e[j] = keep[j] ? expf(v[j] - mx) : 0.f; zs += e[j]; }
  }
  const float Z = wave_sum_dpp(zs);

  // multinomial(p, 1) == argmax(p / q), q ~ Exp(1); entries with p = 0 can only win if nothing is kept
  ValIdx sm{-1.f, 0x7fffffff};
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const int i = j * 64 + lane;
    if (any[j]) {
      const float q = (nz != nullptr) ? qn[j] : device_exp1(seed, pass, i);
      if (i < V) sm = better(sm, ValIdx{(e[j] / Z) / q, i});
    } else if (i < V) {
      sm = better(sm, ValIdx{0.f, i});
    }
  }
  sm = wave_argmax_dpp(sm);

  sm.i = min(sm.i, V - 1);  // non-finite logits leave the argmax sentinel: keep the embedding lookup inside the table
  // stop rule + append (valle.py:1044-1057); every lane evaluates the same scalars
  int tok = sm.i, reason = 0;
  bool append = false, go = false;
  if (forced != nullptr) {
    if (pass >= n_forced) reason = 4;
    else { tok = (int)forced[pass]; append = true; go = true; }
  } else if (am.i == NUM_AUDIO_TOKENS) reason = 1;
  else if (sm.i == NUM_AUDIO_TOKENS) reason = 2;
  else if (bos + n_gen > 16 * S) reason = 3;
  else if (max_new >= 0 && n_gen >= max_new) reason = 4;
  else {
    append = true; go = true;
    // after this append the next pass could only stop (valle.py:1047): do not compute it
    if (bos + n_gen + 1 > 16 * S) { reason = 3; go = false; }
    else if (max_new >= 0 && n_gen + 1 >= max_new) { reason = 4; go = false; }
  }
  if (lane == 0) {
    sampled[pass] = sm.i;
    argmaxes[pass] = am.i;
    if (append) { tokens[n_gen] = tok; st->n_gen = n_gen + 1; }
    if (reason != 0) { st->done = 1; st->stop_reason = reason; }
    if (go) { st->row = row; st->pass = pass + 1; }
  }
  if (!go) return;
  // x = E[tok] * 1.0 + alpha * pe[audio position] (valle.py:1013-1015; embedding.py:93-97)
  const int apos = row - st->kv_text;
  for (int c = lane * 4; c < a.d; c += 256) {
    const float4 ev = *reinterpret_cast<const float4*>(a.emb + (size_t)tok * a.d + c);
    const float4 pv = *reinterpret_cast<const float4*>(a.pe + (size_t)apos * a.d + c);
    float4 o;
    o.x = __fadd_rn(ev.x, __fmul_rn(alpha, pv.x)); o.y = __fadd_rn(ev.y, __fmul_rn(alpha, pv.y));
    o.z = __fadd_rn(ev.z, __fmul_rn(alpha, pv.z)); o.w = __fadd_rn(ev.w, __fmul_rn(alpha, pv.w));
    *reinterpret_cast<float4*>(xout + c) = o;
  }
}

// Four-wave variant used by the decode step: the single-wave kernel above executes ~3000 instructions
// serially; here thread t owns logits t, t+256, ... (NVT each) for the elementwise work and the four
// block-level reductions, while wave 0 alone also holds all keys (NV0 per lane) for the exact top-k select.
// Body of the four-wave sampler.  v[j] = logit j * 256 + tid (any value past V), `lg` = the whole row for wave 0's exact top-k
// select (global memory in the stand-alone kernel, LDS when the predict-layer launch samples in place, ar_tp.hpp).
template <int NVT, int NV0, typename Book>
__device__ __forceinline__ void sample4_body(const SampleArgs& a, ArState* st, const ArState& s, float (&v)[NVT], float (&w0)[NV0], int slot, Book&& book) {
  // `s` = the decode state as the kernel read it in its FIRST round trip (one copy of the whole struct next to the logits loads;
  // read field by field where it was used, the state cost five dependent round trips: done, pass + exp_noise, noise_rows, ...);
  // `st` is written only.  w0 = wave 0's copy of the whole row for the exact top-k select, requested up front as well.
  __shared__ float s_av[4], s_sv[4], s_f[4];
  __shared__ int s_ai[4], s_si[4];
  __shared__ uint32_t cand_lds[64];
  __shared__ uint32_t s_T;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int V = a.V;
  float qn[NVT];
  const int pass = s.pass;
#ifdef VX_STAMPS
  const unsigned long long vx_t0 = __builtin_amdgcn_s_memrealtime();
#endif
  int* const tokens = a.tokens + (size_t)slot * a.tok_stride;
  int* const sampled = a.sampled + (size_t)slot * a.tok_stride;
  int* const argmaxes = a.argmaxes + (size_t)slot * a.tok_stride;
  float* const xout = a.x + (size_t)slot * a.d;
  const float* nz = s.exp_noise;
  if (nz != nullptr) nz += (size_t)min((long long)pass, s.noise_rows - 1) * V;
#pragma unroll
  for (int j = 0; j < NVT; ++j) {
    const int i = j * 256 + tid;
    if (i >= V) v[j] = -INFINITY;
    qn[j] = (nz != nullptr && i < V) ? nz[i] : 1.f;
  }
  const float temp = s.temperature;
  const int top_k = s.top_k;
  const bool filt = top_k > 0 && top_k < V;  // uniform
  const unsigned long long seed = s.seed;
  const int n_gen = s.n_gen, bos = s.bos, S = s.S, max_new = s.max_new, n_forced = s.n_forced;
  const long long* forced = s.forced;
  const int row = s.row + 1;
  const float alpha = a.alpha[0];

  // argmax of the raw logits (valle.py:1045); first index on ties
  ValIdx am{v[0], tid};
#pragma unroll
  for (int j = 1; j < NVT; ++j) am = better(am, ValIdx{v[j], j * 256 + tid});
  am = wave_argmax_dpp(am);
  if (lane == 0) { s_av[wave] = am.v; s_ai[wave] = am.i; }
  __syncthreads();
  book();  // the step's bookkeeping stores: behind the first use of the logits (see sample_embed4_kernel)
  am = ValIdx{s_av[0], s_ai[0]};
#pragma unroll
  for (int w = 1; w < 4; ++w) am = better(am, ValIdx{s_av[w], s_ai[w]});

  if (temp != 1.0f) {  // valle.py:1296-1297
#pragma unroll
    for (int j = 0; j < NVT; ++j) v[j] = v[j] / temp;
  }
  uint32_t T = 0u;
  if (filt) {
    if (wave == 0) {
      uint32_t key[NV0];
#pragma unroll
      for (int j = 0; j < NV0; ++j) {
        const float x = (temp != 1.0f) ? w0[j] / temp : w0[j];
        key[j] = (j * 64 + lane < V) ? order_key(x) : 0u;
      }
      const uint32_t t0 = topk_threshold_wave<NV0>(key, top_k, lane, cand_lds);
      if (lane == 0) s_T = t0;
    }
    __syncthreads();
    T = s_T;
  }
  // softmax over the kept entries: the maximum always survives the filter and x -> x / temp is monotone, so
  // the kept maximum is the (scaled) raw maximum
  const float mx = (temp != 1.0f) ? am.v / temp : am.v;
  bool keep[NVT];
  float e[NVT], zs = 0.f;
#pragma unroll
  for (int j = 0; j < NVT; ++j) {
    keep[j] = (j * 256 + tid < V) && (T == 0u || order_key(v[j]) >= T);
    e[j] = keep[j] ? expf(v[j] - mx) : 0.f;
    zs += e[j];
  }
  zs = wave_sum_dpp(zs);
  if (lane == 0) s_f[wave] = zs;
  __syncthreads();
  const float Z = ((s_f[0] + s_f[1]) + s_f[2]) + s_f[3];

  // multinomial(p, 1) == argmax(p / q), q ~ Exp(1)
  ValIdx sm{-1.f, 0x7fffffff};
#pragma unroll
  for (int j = 0; j < NVT; ++j) {
    const int i = j * 256 + tid;
    if (i < V) {
      float r = 0.f;
      if (keep[j]) r = (e[j] / Z) / ((nz != nullptr) ? qn[j] : device_exp1(seed, pass, i));
      sm = better(sm, ValIdx{r, i});
    }
  }
  sm = wave_argmax_dpp(sm);
  if (lane == 0) { s_sv[wave] = sm.v; s_si[wave] = sm.i; }
  __syncthreads();
  sm = ValIdx{s_sv[0], s_si[0]};
#pragma unroll
  for (int w = 1; w < 4; ++w) sm = better(sm, ValIdx{s_sv[w], s_si[w]});

  // stop rule + append (valle.py:1044-1057); every thread evaluates the same scalars.  Non-finite logits leave the
  // argmax sentinel in sm.i: clamp so that the embedding lookup below can never leave the table
  sm.i = min(sm.i, V - 1);
  int tok = sm.i, reason = 0;
  bool append = false, go = false;
  if (forced != nullptr) {
    if (pass >= n_forced) reason = 4;
    else { tok = (int)forced[pass]; append = true; go = true; }
  } else if (am.i == NUM_AUDIO_TOKENS) reason = 1;
  else if (sm.i == NUM_AUDIO_TOKENS) reason = 2;
  else if (bos + n_gen > 16 * S) reason = 3;
  else if (max_new >= 0 && n_gen >= max_new) reason = 4;
  else {
    append = true; go = true;
    if (bos + n_gen + 1 > 16 * S) { reason = 3; go = false; }
    else if (max_new >= 0 && n_gen + 1 >= max_new) { reason = 4; go = false; }
  }
  {
    // alpha's load (issued long ago) is retired HERE, in front of the state stores: its first use was behind them, and the wait the
    // compiler put there (vmcnt counts loads and stores alike) also waited for the stores' acknowledgements before wave 0 could
    // request the embedding row
    float alpha_now = alpha;
    asm volatile("" : "+v"(alpha_now));
  }
  if (tid == 0) {
    sampled[pass] = sm.i;
    argmaxes[pass] = am.i;
    if (append) { tokens[n_gen] = tok; st->n_gen = n_gen + 1; }
    if (reason != 0) { st->done = 1; st->stop_reason = reason; }
    if (go) { st->row = row; st->pass = pass + 1; }
  }
  if (!go) return;
  const int apos = row - s.kv_text;
  for (int c = tid * 4; c < a.d; c += 1024) {
    const float4 ev = *reinterpret_cast<const float4*>(a.emb + (size_t)tok * a.d + c);
    const float4 pv = *reinterpret_cast<const float4*>(a.pe + (size_t)apos * a.d + c);
    float4 o;
    o.x = __fadd_rn(ev.x, __fmul_rn(alpha, pv.x)); o.y = __fadd_rn(ev.y, __fmul_rn(alpha, pv.y));
    o.z = __fadd_rn(ev.z, __fmul_rn(alpha, pv.z)); o.w = __fadd_rn(ev.w, __fmul_rn(alpha, pv.w));
    *reinterpret_cast<float4*>(xout + c) = o;
  }
#ifdef VX_STAMPS
  VX_KSTAMP_SELF(a.kid, pass + 1, vx_t0);  // slot = the pass index every later kernel of this step reads
#endif
}

// Four-wave variant used by the decode step: the single-wave kernel above executes ~3000 instructions
// serially; here thread t owns logits t, t+256, ... (NVT each) for the elementwise work and the four
// block-level reductions, while wave 0 alone also holds all keys (NV0 per lane) for the exact top-k select.
template <int NVT, int NV0>
__global__ __launch_bounds__(256) void sample_embed4_kernel(const SampleArgs a) {
  const int slot = blockIdx.x;
  ArState* st = a.st + slot;
  const int tid = threadIdx.x;
  const int V = a.V;
  // ONE round trip for everything the sampling needs: the decode state (whole struct), the newest logits row (fixed address) and
  // wave 0's second copy of the row - all requested before the first store of the kernel (a store to memory the compiler cannot
  // tell from the state would pin every later state read behind it, one dependent load at a time)
  const ArState s = *st;
  const unsigned epoch_old = a.epoch != nullptr ? *a.epoch : 0u;
  __builtin_amdgcn_sched_barrier(0);  // the state's (scalar) loads first: left to the scheduler they went out behind the logits' return
  const float* lg = a.logits + (size_t)slot * a.logits_stride;
  float v[NVT], w0[NV0];
#pragma unroll
  for (int j = 0; j < NVT; ++j) v[j] = lg[min(j * 256 + tid, V - 1)];
  if (tid < 64) {  // wave 0
#pragma unroll
    for (int j = 0; j < NV0; ++j) w0[j] = lg[min(j * 64 + tid, V - 1)];
  }
  // The step's bookkeeping for the launches BEHIND this one (nothing here reads it; on every path: the step's launches run and
  // accumulate on every replay).  Not at the head of the kernel: the wait in front of the first use of the logits counts stores
  // too (and, the store count differing per path, waits for all of them), so the sampling stood still until these stores were
  // acknowledged; behind the first barrier their acknowledgements come back under the top-k select.
  auto book = [&]() {
    if (a.epoch != nullptr && tid == 0) {  // 0 is never a tag (fresh granules are zero-filled)
      const unsigned n = epoch_old + 1u;
      *a.epoch = n ? n : 1u;
    }
    if (a.zero_acc != nullptr) {
      *reinterpret_cast<uint4*>(a.zero_acc + 4 * tid) = make_uint4(0u, 0u, 0u, 0u);
      *reinterpret_cast<uint4*>(a.zero_acc + 4 * tid + 2) = make_uint4(0u, 0u, 0u, 0u);
    }
  };
  if (s.done) { book(); return; }  // uniform
  sample4_body<NVT, NV0>(a, st, s, v, w0, slot, book);
}

}  // namespace vx
