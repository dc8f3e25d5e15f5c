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

constexpr int BMAX = 32;  // slots per engine (two 16-row MFMA halves)
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
};

// C[b][n] = sum_k A[b][k] W[n][k] on v_mfma_f32_16x16x32_bf16: one workgroup = one 16-row n tile (x one K
// group), its 4 waves take 4 K slices of NS steps and are summed through LDS in wave order.  Lane
// (c = l&15, g = l>>4) holds act[b = c (+16)][8g..8g+8) and W[n0 + c][8g..8g+8) of each 32-wide step; the
// accumulator has n on the lane and b = 4g + v in its 4 registers.
template <int EPI, int NS>
__global__ __launch_bounds__(256) void bgemm_kernel(const BgemmArgs a) {
  __shared__ float red[4][8][64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c = lane & 15, g = lane >> 4;
  const int ntile = blockIdx.x / a.kgroups, kg = blockIdx.x - ntile * a.kgroups;
  const int n0 = ntile * 16;
  const int K = a.K;
  const int kbeg = kg * (K / a.kgroups) + wave * NS * 32;
  const bf16* wp = a.W + (size_t)min(n0 + c, a.N - 1) * K + kbeg + 8 * g;
  const bf16* ap0 = a.A + (size_t)c * K + kbeg + 8 * g;
  const bf16* ap1 = ap0 + (size_t)16 * K;
  bf16x8b_t wf[NS], a0[NS], a1[NS];
#pragma unroll
  for (int s = 0; s < NS; ++s) wf[s] = *reinterpret_cast<const bf16x8b_t*>(wp + s * 32);
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    a0[s] = *reinterpret_cast<const bf16x8b_t*>(ap0 + s * 32);
    a1[s] = *reinterpret_cast<const bf16x8b_t*>(ap1 + s * 32);
  }
  f32x4_t acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0[s], wf[s], acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1[s], wf[s], acc1, 0, 0, 0);
  }
#pragma unroll
  for (int v = 0; v < 4; ++v) { red[wave][v][lane] = acc0[v]; red[wave][4 + v][lane] = acc1[v]; }
  __syncthreads();
  // wave w finishes values {2w, 2w+1} of the 8 per lane: value i -> batch row b = 16*(i>>2) + 4g + (i&3)
  const int n = n0 + c;
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int i = wave * 2 + u;
    const float x = ((red[0][i][lane] + red[1][i][lane]) + red[2][i][lane]) + red[3][i][lane];
    const int b = 16 * (i >> 2) + 4 * g + (i & 3);
    if (b >= a.B || n >= a.N) continue;
    if (EPI == BE_PARTIAL) {
      a.part[((size_t)kg * BMAX + b) * a.N + n] = x;
    } else if (EPI == BE_RELU) {
      a.f[(size_t)b * a.N + n] = (bf16)fmaxf(x + a.bias[n], 0.f);
    } else if (EPI == BE_LOGITS) {
      if (!a.st[b].done) a.logits[(size_t)b * a.logits_stride + n] = x;
    } else {  // BE_QKV
      const float v = x + a.bias[n];
      const int sec = n / a.d, ii = n - sec * a.d;
      if (sec == 0) {
        a.q[(size_t)b * a.d + ii] = v;
      } else if (!a.st[b].done) {
        const int h = ii / a.hd, cc = ii - h * a.hd;
        bf16* base = a.kv + (size_t)b * a.kv_slot_stride + (sec == 2 ? a.kv_v_offset : 0);
        base[((size_t)h * a.ctx_max + a.st[b].row) * a.hd + cc] = (bf16)v;
      }
    }
  }
}

// x[b] += bias + sum_g part[g][b] (optional; written back), then h[b] = bf16(LN(x[b]) * gamma + beta).
// One wave per slot.
__global__ __launch_bounds__(256) void ln_batch_kernel(float* __restrict__ x, const float* __restrict__ part, int kgroups,
                                                       const float* __restrict__ pbias, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, bf16* __restrict__ h, int B, int d) {
  constexpr int MAXV = 4;  // d <= 1024
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= B) return;
  float* xr = x + (size_t)b * d;
  float4 v[MAXV];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int k = (i * 64 + lane) * 4;
    if (k < d) {
      v[i] = *reinterpret_cast<const float4*>(xr + k);
      if (part != nullptr) {
        const float4 pb = *reinterpret_cast<const float4*>(pbias + k);
        float4 t = pb;
        for (int gi = 0; gi < kgroups; ++gi) {
          const float4 p = *reinterpret_cast<const float4*>(part + ((size_t)gi * BMAX + b) * d + k);
          t.x += p.x; t.y += p.y; t.z += p.z; t.w += p.w;
        }
        v[i].x += t.x; v[i].y += t.y; v[i].z += t.z; v[i].w += t.w;
        *reinterpret_cast<float4*>(xr + k) = v[i];
      }
    } else {
      v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
  }
  const float mean = wave_sum_dpp(s) / (float)d;
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int k = (i * 64 + lane) * 4;
    if (k < d) {
      const float d0 = v[i].x - mean, d1 = v[i].y - mean, d2 = v[i].z - mean, d3 = v[i].w - mean;
      ss += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
    }
  }
  const float rstd = 1.0f / sqrtf(wave_sum_dpp(ss) / (float)d + LN_EPS);
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int k = (i * 64 + lane) * 4;
    if (k < d) {
      const float4 gm = *reinterpret_cast<const float4*>(gamma + k);
      const float4 bt = *reinterpret_cast<const float4*>(beta + k);
      union { bf16 e[4]; uint2 u; } pk;
      pk.e[0] = (bf16)((v[i].x - mean) * rstd * gm.x + bt.x);
      pk.e[1] = (bf16)((v[i].y - mean) * rstd * gm.y + bt.y);
      pk.e[2] = (bf16)((v[i].z - mean) * rstd * gm.z + bt.z);
      pk.e[3] = (bf16)((v[i].w - mean) * rstd * gm.w + bt.w);
      *reinterpret_cast<uint2*>(h + (size_t)b * d + k) = pk.u;
    }
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
  constexpr int VEC = 8, LPK = HD / VEC, KPW = 64 / LPK, KPB = 4 * KPW, UNR = 6;
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
  for (int base = 0; base < ctx; base += UNR * KPB) {
    uint4 kr[UNR], vr[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int j = min(base + u * KPB + wave * KPW + grp, ctx - 1);
      kr[u] = ld16(kb + (size_t)j * HD);
      vr[u] = ld16(vb + (size_t)j * HD);
    }
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
    mloc = wave_max_dpp(mloc);
    __syncthreads();
    if (lane == 0) sm_red[wave] = mloc;
    __syncthreads();
    const float mb = fmaxf(fmaxf(sm_red[0], sm_red[1]), fmaxf(sm_red[2], sm_red[3]));
    const float Mn = fmaxf(M, mb);
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
  }
  const int gi = wave * KPW + grp;
#pragma unroll
  for (int i = 0; i < VEC; ++i) sm_o[gi][sub * VEC + i] = acc[i];
  if (sub == 0) sm_l[gi] = L;
  __syncthreads();
  if (tid < HD) {
    float o = 0.f, l = 0.f;
#pragma unroll
    for (int gidx = 0; gidx < 4 * KPW; ++gidx) { o += sm_o[gidx][tid]; l += sm_l[gidx]; }
    out[(size_t)slot * d + h * HD + tid] = (bf16)(o / l);
  }
}

}  // namespace vx
