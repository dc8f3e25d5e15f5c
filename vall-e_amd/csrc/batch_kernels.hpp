// Batched AR decode: up to 32 utterances ("slots") advance one token per step and share ONE stream of
// the 304 MB of weights (BASELINE configs[2]: padded per-slot KV cache, hipGraph-captured step).
// Per layer: ln_batch -> bgemm(QKV) -> attn_batch -> bgemm(out, partial) -> ln_batch(+partials) ->
// bgemm(FFN1) -> bgemm(FFN2, partial); the partial sums of the two N = d GEMMs are reduced, in a fixed
// order, by the LayerNorm kernel that follows them anyway.
#pragma once
#include "common.hpp"
#include "ar_kernels.hpp"
#include "mfma_kernels.hpp"

namespace vx {

constexpr int BMAX = 64;  // slots per engine (up to four 16-row MFMA halves)
typedef __bf16 bf16x8b_t __attribute__((ext_vector_type(8)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));

enum BgemmEpi { BE_QKV = 0, BE_RELU = 1, BE_PARTIAL = 2, BE_LOGITS = 3 };

struct BgemmArgs {
  const bf16* A;      // (32, K) activations, rows >= B hold finite stale values
  const bf16* W;      // (N, K)
  const float* bias;  // (N,) [BE_QKV, BE_RELU]
  int N, K, B;
  int kgroups;        // K split over workgroups (BE_PARTIAL); 1 otherwise
  const ArState* st;  // (32,) per-slot state
  // outputs
  float* q;           // BE_QKV: (32, d)
  bf16* kv;           // BE_QKV: this layer's cache base of slot 0; slot stride kv_slot_stride elements
  size_t kv_slot_stride, kv_v_offset;  // elements
  int d, hd, ctx_max;
  bf16* f;            // BE_RELU: (32, N)
  float* part;        // BE_PARTIAL: (kgroups, 32, N)
  float* logits;      // BE_LOGITS: (32, logits_stride)
  int logits_stride;
  float* trace;       // BE_LOGITS, optional (VX_FLAG_TRACE_LOGITS): (slots, trace_rows, N) - row `pass` of every live slot
  int trace_rows;
  // cache warm-up for a LATER GEMM of the step (as GemvArgs.pf of the batch-1 step): workgroup b touches bytes
  // [b pf_slice, (b+1) pf_slice) of `pf` with 8 unused 16-byte loads per lane.  Speed only.
  const void* pf;
  unsigned pf_slice, pf_total;
};

// C[b][n] = sum_k A[b][k] W[n][k] on v_mfma_f32_16x16x32_bf16: one workgroup = one 16-row n tile (x one K
// group), its 4 waves take 4 K slices of NS steps and are summed through LDS in wave order.  Lane
// (c = l&15, g = l>>4) holds act[b = c (+16)][8g..8g+8) and W[n0 + c][8g..8g+8) of each 32-wide step; the
// accumulator has n on the lane and b = 4g + v in its 4 registers.
// The operands of the kernel's FIRST loads (A, W, sizes) are explicit leading arguments: with the build's
// `-amdgpu-kernarg-preload-count` they arrive in SGPRs at wave launch (hipcc does not preload by-value structs), so the weight
// and activation loads do not wait for a kernarg fetch.  nk = (N << 16) | K.
template <int EPI, int NS, int NH, bool PF = false>  // NH 16-row halves of slots: 2 (B <= 32) or 4 (B <= 64)
__global__ __launch_bounds__(256) void bgemm_kernel(const bf16* __restrict__ A_, const bf16* __restrict__ W_, unsigned nk, int kgroups,
                                                    const BgemmArgs a) {
  __shared__ float red[4][4 * NH][64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c = lane & 15, g = lane >> 4;
  const int ntile = blockIdx.x / kgroups, kg = blockIdx.x - ntile * kgroups;
  const int n0 = ntile * 16;
  const int K = (int)(nk & 0xffffu), Nn = (int)(nk >> 16);
  const int kbeg = kg * (K / kgroups) + wave * NS * 32;
  const bf16* wp = W_ + (size_t)min(n0 + c, Nn - 1) * K + kbeg + 8 * g;
  const bf16* ap = A_ + (size_t)c * K + kbeg + 8 * g;
  // epilogue operands first (clamped, unconditional): fetched after the K loop they are one more exposed memory round trip
  float bias_v = 0.f;
  if (EPI == BE_QKV || EPI == BE_RELU) bias_v = a.bias[min(n0 + c, a.N - 1)];
  int st_done[NH], st_row[NH];  // st_row: KV row (BE_QKV) / pass index of the logits row being produced (BE_LOGITS trace)
#pragma unroll
  for (int u = 0; u < NH; ++u) {
    st_done[u] = 0; st_row[u] = 0;
    if (EPI == BE_QKV || EPI == BE_LOGITS) {
      const int i = wave * NH + u, b = min(16 * (i >> 2) + 4 * g + (i & 3), a.B - 1);
      st_done[u] = a.st[b].done;
      st_row[u] = EPI == BE_QKV ? a.st[b].row : a.st[b].pass;
    }
  }
  bf16x8b_t wf[NS], af[NH][NS];
#pragma unroll
  for (int s = 0; s < NS; ++s) wf[s] = *reinterpret_cast<const bf16x8b_t*>(wp + s * 32);
#pragma unroll
  for (int s = 0; s < NS; ++s)
#pragma unroll
    for (int h = 0; h < NH; ++h) af[h][s] = *reinterpret_cast<const bf16x8b_t*>(ap + (size_t)16 * h * K + s * 32);
  // a later GEMM's weights, requested behind this kernel's own loads (vmcnt retires in order: the waits of the MFMAs do not cover them)
  uint4 pfv[PF ? 8 : 1];
  if (PF) {
    const unsigned lim = min(a.pf_slice, a.pf_total - min(a.pf_total, blockIdx.x * a.pf_slice));
    const char* pb = reinterpret_cast<const char*>(a.pf) + min((size_t)blockIdx.x * a.pf_slice, (size_t)a.pf_total - 16);
#pragma unroll
    for (int i = 0; i < 8; ++i) pfv[i] = *reinterpret_cast<const uint4*>(pb + min((unsigned)(i * 4096 + tid * 16), max(lim, 16u) - 16u));
  }
  f32x4_t acc[NH];
#pragma unroll
  for (int h = 0; h < NH; ++h) acc[h] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int s = 0; s < NS; ++s)
#pragma unroll
    for (int h = 0; h < NH; ++h) acc[h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[h][s], wf[s], acc[h], 0, 0, 0);
#pragma unroll
  for (int h = 0; h < NH; ++h)
#pragma unroll
    for (int v = 0; v < 4; ++v) red[wave][4 * h + v][lane] = acc[h][v];
  __syncthreads();
  // wave w finishes values {NH w, .., NH w + NH-1} of the 4 NH per lane: value i -> batch row b = 16*(i>>2) + 4g + (i&3)
  const int n = n0 + c;
#pragma unroll
  for (int u = 0; u < NH; ++u) {
    const int i = wave * NH + u;
    const float x = ((red[0][i][lane] + red[1][i][lane]) + red[2][i][lane]) + red[3][i][lane];
    const int b = 16 * (i >> 2) + 4 * g + (i & 3);
    if (b >= a.B || n >= a.N) continue;
    if (EPI == BE_PARTIAL) {
      a.part[((size_t)kg * BMAX + b) * a.N + n] = x;
    } else if (EPI == BE_RELU) {
      a.f[(size_t)b * a.N + n] = (bf16)fmaxf(x + bias_v, 0.f);
    } else if (EPI == BE_LOGITS) {
      if (!st_done[u]) {
        a.logits[(size_t)b * a.logits_stride + n] = x;
        if (a.trace != nullptr && st_row[u] < a.trace_rows) a.trace[((size_t)b * a.trace_rows + st_row[u]) * a.N + n] = x;
      }
    } else {  // BE_QKV
      const float v = x + bias_v;
      const int sec = n / a.d, ii = n - sec * a.d;
      if (sec == 0) {
        a.q[(size_t)b * a.d + ii] = v;
      } else if (!st_done[u]) {
        const int h = ii / a.hd, cc = ii - h * a.hd;
        bf16* base = a.kv + (size_t)b * a.kv_slot_stride + (sec == 2 ? a.kv_v_offset : 0);
        base[((size_t)h * a.ctx_max + st_row[u]) * a.hd + cc] = (bf16)v;
      }
    }
  }
  if (PF) {  // the warm-up loads stay live (and unwaited) until here
#pragma unroll
    for (int i = 0; i < 8; ++i) asm volatile("" ::"v"(pfv[i].x), "v"(pfv[i].y), "v"(pfv[i].z), "v"(pfv[i].w));
  }
}

// x[b] += bias + sum_g part[g][b] (KG > 0; written back), then h[b] = bf16(LN(x[b]) * gamma + beta).
// One workgroup per slot, thread t owns columns [4t, 4t+4): every load of the kernel (x, bias, the KG partials,
// gamma, beta) is issued up front, so the kernel is one memory round trip plus two workgroup reductions.
template <int KG>
__global__ __launch_bounds__(256) void ln_batch_kernel(float* __restrict__ x, const float* __restrict__ part,
                                                       const float* __restrict__ pbias, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, bf16* __restrict__ h, int d) {
  __shared__ float red[2][4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.x;
  const bool live = tid * 4 < d;
  const int k = live ? tid * 4 : 0;  // idle threads (d < 1024) load column 0 and contribute zeros
  float* xr = x + (size_t)b * d;
  float4 v = *reinterpret_cast<const float4*>(xr + k);
  const float4 gm = *reinterpret_cast<const float4*>(gamma + k);
  const float4 bt = *reinterpret_cast<const float4*>(beta + k);
  if (KG > 0) {
    float4 t = *reinterpret_cast<const float4*>(pbias + k);
    float4 p[KG > 0 ? KG : 1];
#pragma unroll
    for (int gi = 0; gi < KG; ++gi) p[gi] = *reinterpret_cast<const float4*>(part + ((size_t)gi * BMAX + b) * d + k);
#pragma unroll
    for (int gi = 0; gi < KG; ++gi) { t.x += p[gi].x; t.y += p[gi].y; t.z += p[gi].z; t.w += p[gi].w; }
    v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
    if (live) *reinterpret_cast<float4*>(xr + k) = v;
  }
  if (!live) v = make_float4(0.f, 0.f, 0.f, 0.f);
  const float s = wave_sum_dpp((v.x + v.y) + (v.z + v.w));
  if (lane == 0) red[0][wave] = s;
  __syncthreads();
  const float mean = ((red[0][0] + red[0][1]) + (red[0][2] + red[0][3])) / (float)d;
  const float d0 = v.x - mean, d1 = v.y - mean, d2 = v.z - mean, d3 = v.w - mean;
  const float ss = wave_sum_dpp(live ? (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3) : 0.f);
  if (lane == 0) red[1][wave] = ss;
  __syncthreads();
  const float rstd = 1.0f / sqrtf(((red[1][0] + red[1][1]) + (red[1][2] + red[1][3])) / (float)d + LN_EPS);
  if (live) {
    union { bf16 e[4]; uint2 u; } pk;
    pk.e[0] = (bf16)(d0 * rstd * gm.x + bt.x);
    pk.e[1] = (bf16)(d1 * rstd * gm.y + bt.y);
    pk.e[2] = (bf16)(d2 * rstd * gm.z + bt.z);
    pk.e[3] = (bf16)(d3 * rstd * gm.w + bt.w);
    *reinterpret_cast<uint2*>(h + (size_t)b * d + k) = pk.u;
  }
}

// Single-query attention of every (slot, head): grid = (nhead, B), one workgroup walks all cached keys of its
// head (running max / sum across passes) and writes the normalised output as the bf16 A operand of the
// out-projection.  Same inner structure as attn_decode_kernel.
template <int HD>
__global__ __launch_bounds__(256) void attn_batch_kernel(const float* __restrict__ q, const bf16* __restrict__ kv,
                                                         size_t kv_slot_stride, size_t kv_v_offset,
                                                         const ArState* __restrict__ st, int ctx_max, int d, float scale,
                                                         bf16* __restrict__ out) {
  constexpr int VEC = 8, LPK = HD / VEC, KPW = 64 / LPK, KPB = 4 * KPW, UNR = 4;
  __shared__ float sm_red[4];
  __shared__ __attribute__((aligned(16))) float sm_o[4 * KPW][HD + 1];
  __shared__ float sm_l[4 * KPW];
  const int h = blockIdx.x, slot = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int sub = lane % LPK, grp = lane / LPK;
  const int ctx = st[slot].row + 1;
  float qv[VEC];
#pragma unroll
  for (int i = 0; i < VEC; i += 4) {
    const float4 t = *reinterpret_cast<const float4*>(q + (size_t)slot * d + h * HD + sub * VEC + i);
    qv[i] = t.x; qv[i + 1] = t.y; qv[i + 2] = t.z; qv[i + 3] = t.w;
  }
  const bf16* kb = kv + (size_t)slot * kv_slot_stride + (size_t)h * ctx_max * HD + sub * VEC;
  const bf16* vb = kb + kv_v_offset;
  float M = -INFINITY, L = 0.f, acc[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) acc[i] = 0.f;
  constexpr int STEP = UNR * KPB;
  // Two register sets: the loads of pass p+1 are in flight while pass p is scored.  They are issued
  // unconditionally on clamped rows (the pass after the last re-reads the last key and is never scored), so the
  // loop body has no branch around a load and the waits stay counted.
  uint4 kr0[UNR], vr0[UNR], kr1[UNR], vr1[UNR];
  auto issue = [&](uint4 (&kr)[UNR], uint4 (&vr)[UNR], int base) {
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int j = min(base + u * KPB + wave * KPW + grp, ctx - 1);
      kr[u] = ld16(kb + (size_t)j * HD);
      vr[u] = ld16(vb + (size_t)j * HD);
    }
    __builtin_amdgcn_sched_barrier(0);  // keep the loads here, ahead of the scoring of the other set
  };
  auto score = [&](const uint4 (&kr)[UNR], const uint4 (&vr)[UNR], int base) {
    float sc[UNR], mloc = -INFINITY;
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int j = base + u * KPB + wave * KPW + grp;
      float kf[VEC];
      unpack<bf16>(kr[u], kf);
      float dot = 0.f;
#pragma unroll
      for (int i = 0; i < VEC; ++i) dot = fmaf(kf[i], qv[i], dot);
      dot = group8_sum_dpp(dot);
      sc[u] = (j < ctx) ? dot * scale : -INFINITY;
      mloc = fmaxf(mloc, sc[u]);
    }
    // running maximum per WAVE (merged across the four waves once, at the end): a workgroup-wide maximum per pass cost two
    // barriers per 128 keys, and with them the waves' loads and arithmetic stopped overlapping each other's
    const float Mn = fmaxf(M, wave_max_dpp(mloc));
    const float corr = (M == -INFINITY) ? 0.f : expf(M - Mn);
    L *= corr;
#pragma unroll
    for (int i = 0; i < VEC; ++i) acc[i] *= corr;
    M = Mn;
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      float vf[VEC];
      unpack<bf16>(vr[u], vf);
      const float p = (sc[u] == -INFINITY) ? 0.f : expf(sc[u] - M);
      L += p;
#pragma unroll
      for (int i = 0; i < VEC; ++i) acc[i] = fmaf(p, vf[i], acc[i]);
    }
  };
  issue(kr0, vr0, 0);
  for (int base = 0;;) {
    issue(kr1, vr1, base + STEP);
    score(kr0, vr0, base);
    base += STEP;
    if (base >= ctx) break;
    issue(kr0, vr0, base + STEP);
    score(kr1, vr1, base);
    base += STEP;
    if (base >= ctx) break;
  }
  const int gi = wave * KPW + grp;
#pragma unroll
  for (int i = 0; i < VEC; ++i) sm_o[gi][sub * VEC + i] = acc[i];
  if (sub == 0) sm_l[gi] = L;
  if (lane == 0) sm_red[wave] = M;
  __syncthreads();
  if (tid < HD) {
    // every wave scored at least one live key (ctx >= 1 and the first pass covers key 0 in wave 0; a wave whose keys were all past
    // ctx has M = -inf, L = 0, acc = 0 and weight 0)
    const float Ma = fmaxf(fmaxf(sm_red[0], sm_red[1]), fmaxf(sm_red[2], sm_red[3]));
    float o = 0.f, l = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const float wgt = (sm_red[w] == -INFINITY) ? 0.f : expf(sm_red[w] - Ma);
      float ow = 0.f, lw = 0.f;
#pragma unroll
      for (int gidx = 0; gidx < KPW; ++gidx) { ow += sm_o[w * KPW + gidx][tid]; lw += sm_l[w * KPW + gidx]; }
      o = fmaf(wgt, ow, o);
      l = fmaf(wgt, lw, l);
    }
    out[(size_t)slot * d + h * HD + tid] = (bf16)(o / l);
  }
}

}  // namespace vx
