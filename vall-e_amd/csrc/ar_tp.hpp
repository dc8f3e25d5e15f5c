// Batch-1 decode step sharded over the 8 XCDs of the chip: TWO launches per layer instead of five.
//
// What a launch boundary costs on this part (1.5-1.8 us + a first load that returns 2.1-2.9 us after it was issued, ~3.8 us per
// dependent launch before it has a single byte, profiles/r03_notes.md) is what an exchange between workgroups on DIFFERENT XCDs
// costs whichever way it is done - but an exchange among the 32 workgroups of ONE XCD goes through that XCD's L2 and costs
// 0.35-0.6 us (measured in situ).  So the layer is cut like a tensor-parallel transformer over 8 devices, an XCD playing the
// device (d = 1024, 16 heads; workgroup b: XCD x = b % 8, local index i = b / 8):
//
//   tp_attn_kernel  XCD x owns heads 2x, 2x+1.   LN1 of the residual row (every workgroup, cooperatively), the 12 q / k / v rows
//                   of its quarter-head slice, [exchange inside the head: 192 values], attention over one sixteenth of the
//                   cached keys (requested at kernel start), [exchange inside the XCD: the 2 x 16 partial softmaxes], combine,
//                   and the XCD's K-slice of the out-projection: rows 32i..32i+31 over the 128 channels of its two heads,
//                   added into the residual accumulator.
//   tp_ffn_kernel   LN2 of the residual row, XCD x owns hidden units 512x..512x+511: 16 FFN1 rows per workgroup,
//                   [exchange inside the XCD: 512 values], the XCD's K-slice of FFN2: rows 32i..32i+31 over its 512 hidden
//                   units, added into the residual accumulator.
//   tp_head_kernel  final norm of the residual row and the 1025 logit rows.
//
// The cross-XCD sums travel over the launch boundary as ONE vector: every workgroup adds its partial rows into a 64-bit
// FIXED-POINT accumulator that carries the residual stream itself (TpAccArgs: integer adds commute, so the result is
// bit-reproducible whatever the arrival order).  Everything inside a launch is tagged granules (ar_granules.hpp).  All weights of
// a launch are requested at its start, so the in-launch stages pay a hop and their arithmetic, not a memory latency.  Placement
// (blockIdx % 8 == XCD) is a speed assumption only: the granules are written through (sc1) and polled with sc1 loads, valid
// across XCDs.
// Reference arithmetic: valle/modules/transformer.py:297-334 (pre-norm encoder layer), activation.py:407-427.
#pragma once
#include <type_traits>
#include "ar_granules.hpp"

namespace vx {

#ifdef VX_STAMPS
#define TP_STAMP(which, i)                                                                                            \
  do {                                                                                                                \
    unsigned long long* fq_r_ = g_fq_stamps;                                                                          \
    if (fq_r_ != nullptr && threadIdx.x == 0)                                                                         \
      fq_r_[(((size_t)(which) * 16 + a.layer) * 256 + blockIdx.x) * 16 + (i)] = __builtin_amdgcn_s_memrealtime();      \
  } while (0)
#else
#define TP_STAMP(which, i) do { } while (0)
#endif

constexpr int TP_D = 1024, TP_H = 16, TP_FF = 4096, TP_X = 8, TP_WG = 32;  // model width, heads, FFN width, XCDs, workgroups per XCD
constexpr int TP_HID = TP_FF / TP_X;   // hidden units per XCD
constexpr int TP_ACH = TP_D / TP_X;    // attention channels per XCD (two heads)

// Residual stream between the launches of a step: 1024 x int64 fixed point (value x 2^32), three buffers in rotation.  Launch n
// normalises buffer R = n % 3 (complete: the previous launch's adds and the launch boundary), adds its own output into
// A = (n + 1) % 3 - every workgroup its 32 partial rows of the sharded GEMV, XCD 0's workgroups also the carry (the value of R)
// and the GEMV's bias - and zeroes Z = (n + 2) % 3 for the launch after next.  One accumulator instead of eight partial
// vectors + bias + x: the row every workgroup reads shrinks from 12 loads per thread to 4 (2 x int64 quads, gamma, beta) - with all
// 128 waves of an XCD asking its L2 for the same lines at the same moment that is what the row's latency follows
// (12 -> 4 loads: 213.5 -> 204.6 us per token in a timing-only experiment, profiles/r03_notes.md).  Exactness: a partial p is added
// as rint(p 2^32); the sums are exact integers, the residual stream is rounded to fp32 once, where a norm reads it.
struct TpAccArgs {
  long long* add;          // A: this launch's output accumulates here
  long long* zero;         // Z: zeroed by workgroup 1
  const float* bias;       // bias of this launch's sharded GEMV (out-projection / linear2), added once (by XCD 0)
};
constexpr float TP_FIX = 4294967296.0f, TP_UNFIX = 2.3283064365386963e-10f;
__device__ __forceinline__ long long tp_fix(float v) { return __float2ll_rn(v * TP_FIX); }
// The LayerNorm affine travels as ONE (3, 1024) block {gamma, beta, (unused: the bias is added by the producer)} per norm
// site (packed at vx_finalize_weights), so that a single preloaded pointer reaches them: their loads go out with the row's,
// in front of the weight stream (vmcnt retires in order - behind the weights the norm would wait for the whole stream).

struct TpAttnArgs {
  const float* qkv_bias;   // (3d,)
  unsigned* err;
  fq_gran* gq;             // this layer's (16, FQ_QKV)
  fq_gran* gp;             // this layer's (16, FQ_G, FQ_PART)
  TpAccArgs acc;
  void* kcache;
  void* vcache;
  int ctx_max;
  float scale;
  int layer;
};

struct TpFfnArgs {
  const float* b1;         // (4d,)
  unsigned* err;
  fq_gran* gh;             // this layer's (8, TP_HID)
  TpAccArgs acc;
  int layer;
};

struct TpHeadArgs {
  float* logits;
  int N;
};

// Row prologue: the residual row (fp32 embedding or the int64 accumulator), LayerNorm over the 1024 channels by the whole workgroup
// (thread t owns channels 4t..4t+3), result in LDS.  Every load is issued before the first wait.  `red` = 8 floats of LDS.
// Every global load of a launch's prologue is an asm statement: volatile asm keeps source order, so the ROW's loads really are
// the first in the queue (left to the compiler they were scheduled behind the whole weight stream - vmcnt retires in order, and
// the norm waited for 16 MB of weights), and the counted waits below are exact by construction: tp_wait<N> = "at most N of the
// loads issued after the ones I need are still in flight".  A wait names the registers it releases ("+v"), which orders their
// first use behind it.
typedef float tp_f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void tp_ld16(vx_u32x4& v, const void* p) {
  asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(p) : "memory");
}
// the weight stream (non-temporal loads measured: no difference, profiles/r03_notes.md)
__device__ __forceinline__ void tp_ld16w(vx_u32x4& v, const void* p) {
  asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(p) : "memory");
}
__device__ __forceinline__ void tp_ld16f(tp_f4& v, const void* p) {
  asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(p) : "memory");
}
__device__ __forceinline__ void tp_ld4f(float& v, const void* p) {
  asm volatile("global_load_dword %0, %1, off" : "=v"(v) : "v"(p) : "memory");
}
template <int N> __device__ __forceinline__ void tp_wait() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
template <int N> __device__ __forceinline__ void tp_wait(vx_u32x4& a) { asm volatile("s_waitcnt vmcnt(%1)" : "+v"(a) : "n"(N) : "memory"); }
template <int N> __device__ __forceinline__ void tp_wait(float& a) { asm volatile("s_waitcnt vmcnt(%1)" : "+v"(a) : "n"(N) : "memory"); }
__device__ __forceinline__ uint4 tp_u4(const vx_u32x4& v) { return make_uint4(v.x, v.y, v.z, v.w); }

struct TpRowLoads { tp_f4 x, g, be; vx_u32x4 xf[2]; };
// FIRST: the row is the sampler's fp32 embedding vector (layer 0's attention half); otherwise four int64 of the accumulator R
template <bool FIRST>
__device__ __forceinline__ void tp_row_issue(TpRowLoads& r, const float* __restrict__ x_f32, const long long* __restrict__ racc,
                                             const float* __restrict__ gbb, int tid) {
  if (FIRST) {
    tp_ld16f(r.x, x_f32 + 4 * tid);
  } else {
    tp_ld16(r.xf[0], racc + 4 * tid);
    tp_ld16(r.xf[1], racc + 4 * tid + 2);
  }
  tp_ld16f(r.g, gbb + 4 * tid);
  tp_ld16f(r.be, gbb + TP_D + 4 * tid);
}
// AFTER = loads issued behind the row's
template <bool FIRST, int AFTER> __device__ __forceinline__ void tp_row_wait(TpRowLoads& r) {
  if (FIRST) asm volatile("s_waitcnt vmcnt(%3)" : "+v"(r.x), "+v"(r.g), "+v"(r.be) : "n"(AFTER) : "memory");
  else asm volatile("s_waitcnt vmcnt(%4)" : "+v"(r.xf[0]), "+v"(r.xf[1]), "+v"(r.g), "+v"(r.be) : "n"(AFTER) : "memory");
}
__device__ __forceinline__ float tp_unfix(unsigned lo, unsigned hi) {
  return __ll2float_rn((long long)(((unsigned long long)hi << 32) | lo)) * TP_UNFIX;
}
template <typename WT> __device__ __forceinline__ float tp_exp(float v) {  // fp32 engine: the precise exp of the plain kernels
  if constexpr (std::is_same<WT, float>::value) return expf(v);
  else return __expf(v);
}
template <bool FIRST>
__device__ __forceinline__ void tp_row_norm(TpRowLoads& r, float* xs, float* red, int tid) {
  const int lane = tid & 63, wave = tid >> 6;
  float4 v;
  if (FIRST) v = make_float4(r.x.x, r.x.y, r.x.z, r.x.w);
  else v = make_float4(tp_unfix(r.xf[0].x, r.xf[0].y), tp_unfix(r.xf[0].z, r.xf[0].w), tp_unfix(r.xf[1].x, r.xf[1].y), tp_unfix(r.xf[1].z, r.xf[1].w));
  float s1 = wave_sum_dpp((v.x + v.y) + (v.z + v.w));
  if (lane == 0) red[wave] = s1;
  __syncthreads();
  const float mean = ((red[0] + red[1]) + (red[2] + red[3])) * (1.0f / TP_D);
  const float d0 = v.x - mean, d1 = v.y - mean, d2 = v.z - mean, d3 = v.w - mean;
  float s2 = wave_sum_dpp((d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3));
  if (lane == 0) red[4 + wave] = s2;
  __syncthreads();
  const float rstd = 1.0f / sqrtf(((red[4] + red[5]) + (red[6] + red[7])) * (1.0f / TP_D) + LN_EPS);
  *reinterpret_cast<float4*>(xs + 4 * tid) =
      make_float4(d0 * rstd * r.g.x + r.be.x, d1 * rstd * r.g.y + r.be.y, d2 * rstd * r.g.z + r.be.z, d3 * rstd * r.g.w + r.be.w);
  __syncthreads();
}

// The tail of both halves: row 32i + tid/8 of the sharded GEMV's partial (in lanes tid % 8 == 0) goes into the accumulator;
// XCD 0 also adds the carry and the bias (requested in the prologue: `carry` = R's word / the fp32 embedding, `bo` = bias).
template <bool FIRST>
__device__ __forceinline__ void tp_acc_tail(const TpAccArgs& a, float partial, int xcd, int i, int tid, unsigned long long carry_ll,
                                            float carry_f, float bo) {
  if ((tid & 7) == 0) {
    long long add = tp_fix(partial);
    if (xcd == 0) add += tp_fix(bo) + (FIRST ? tp_fix(carry_f) : (long long)carry_ll);
    atomicAdd(reinterpret_cast<unsigned long long*>(a.add) + 32 * i + (tid >> 3), (unsigned long long)add);
  }
  if (blockIdx.x == 1) {
    *reinterpret_cast<uint4*>(a.zero + 4 * tid) = make_uint4(0u, 0u, 0u, 0u);
    *reinterpret_cast<uint4*>(a.zero + 4 * tid + 2) = make_uint4(0u, 0u, 0u, 0u);
  }
}

// lane's slice of the normalised row in the GEMV layout: chunk c holds channels (c * 64 + lane) * VEC .. + VEC
template <int KCH, int VEC>
__device__ __forceinline__ void tp_row_read(const float* xs, int lane, float (&xr)[KCH][VEC]) {
#pragma unroll
  for (int c = 0; c < KCH; ++c)
#pragma unroll
    for (int i = 0; i < VEC; i += 4) {
      const float4 t = *reinterpret_cast<const float4*>(xs + (c * 64 + lane) * VEC + i);
      xr[c][i] = t.x; xr[c][i + 1] = t.y; xr[c][i + 2] = t.z; xr[c][i + 3] = t.w;
    }
}
template <typename WT, int KCH>
__device__ __forceinline__ float tp_dot(const uint4 (&w)[KCH], const float (&xr)[KCH][Vec16<WT>::N]) {
  constexpr int VEC = Vec16<WT>::N;
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < KCH; ++c) {
    float wf[VEC];
    unpack<WT>(w[c], wf);
#pragma unroll
    for (int i = 0; i < VEC; ++i) s = fmaf(wf[i], xr[c][i], s);
  }
  return s;
}

// ------------------------------------------------------------------------------------------------ attention half
// Wo_ = this layer's out-projection re-laid out by tp_repack_kernel: [x][i][m][t] 16-byte words, word (m, t) = row 32i + t/8,
// channels 128x + (t%8 + 8m) * VEC .. + VEC.
template <typename WT, bool FIRST>
__global__ __launch_bounds__(256) void tp_attn_kernel(const void* __restrict__ Wqkv_, const float* __restrict__ x_f32,
                                                      const long long* __restrict__ racc, const float* __restrict__ gbb,
                                                      const ArState* __restrict__ st, const unsigned* __restrict__ epoch,
                                                      const void* __restrict__ Wo_, const TpAttnArgs a) {
  constexpr int VEC = Vec16<WT>::N;
  constexpr int KCH = TP_D / (64 * VEC);        // 2 (bf16) / 4 (fp32)
  constexpr int HD = 64;
  constexpr int LPK = HD / VEC, KPW = 64 / LPK, KPB = 4 * KPW, UNR = 4;
  constexpr int NGRP = 4 * KPW;                 // key groups of the workgroup: 32 (bf16) / 16 (fp32)
  constexpr int OCH = TP_ACH / (8 * VEC);       // out-projection words per thread: 2 (bf16) / 4 (fp32)
  __shared__ __attribute__((aligned(16))) float xs[TP_D];
  __shared__ float red[8];
  __shared__ __attribute__((aligned(16))) float s_qkv[FQ_QKV];
  __shared__ __attribute__((aligned(16))) float sm_o[NGRP][HD + 4];
  __shared__ float sm_m[4], sm_l[4];
  __shared__ float s_wo[4][HD];
  __shared__ float s_part[2 * FQ_G * FQ_PART];
  __shared__ float s_fac[2 * FQ_G];
  __shared__ __attribute__((aligned(16))) float s_att[TP_ACH];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int x = blockIdx.x & (TP_X - 1), i = blockIdx.x >> 3;
  const int hl = i >> 4, j = i & 15, h = 2 * x + hl;
  const int cih = 4 * j + wave, ch = h * HD + cih;
  const int st_row = st->row, st_done = st->done;
  const unsigned tag = *epoch;
  const WT* __restrict__ W = reinterpret_cast<const WT*>(Wqkv_);

  // ---- every load of the launch, in this order, before the first wait ----
  TpRowLoads rl;
  tp_row_issue<FIRST>(rl, x_f32, racc, gbb, tid);
  TP_STAMP(0, 0);
  vx_u32x4 w[3][KCH];
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < KCH; ++c) tp_ld16w(w[r][c], W + (size_t)(r * TP_D + ch) * TP_D + (c * 64 + lane) * VEC);
  float e_bias;
  {
    const float* bp = a.qkv_bias;  // first use of the by-value argument struct: its kernarg wait belongs HERE, behind the loads above
    asm volatile("" : "+s"(bp));
    tp_ld4f(e_bias, bp + min(lane, 2) * TP_D + ch);
  }
  // the accumulator tail's operands of row 32i + tid/8 (used by XCD 0 only; requested by all: the counted waits stay uniform)
  unsigned long long carry_ll = 0ull;
  float carry_f = 0.f, bo;
  if (FIRST) tp_ld4f(carry_f, x_f32 + 32 * i + (tid >> 3));
  else asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(carry_ll) : "v"(racc + 32 * i + (tid >> 3)) : "memory");
  tp_ld4f(bo, a.acc.bias + 32 * i + (tid >> 3));
  // second-stage operands: the out-projection slice and this split's cached keys
  const int n_old = st_row;  // the newest row travels in the granules
  const int chunk = (n_old + FQ_G - 1) / FQ_G;
  const int j0 = j * chunk, j1 = min(n_old, j0 + chunk);
  const int sub = lane % LPK, grp = lane / LPK;
  vx_u32x4 wo[OCH];
  vx_u32x4 kr[UNR], vr[UNR];
  const WT* kb;
  const WT* vb;
  auto tp_stage2 = [&]() {
    const uint4* wb = reinterpret_cast<const uint4*>(Wo_) + (size_t)(x * TP_WG + i) * OCH * 256;
#pragma unroll
    for (int m = 0; m < OCH; ++m) tp_ld16w(wo[m], wb + m * 256 + tid);
    const void* kcp = a.kcache;
    const void* vcp = a.vcache;
    int ctxm = a.ctx_max;
    asm volatile("" : "+s"(kcp), "+s"(vcp), "+s"(ctxm));
    kb = reinterpret_cast<const WT*>(kcp) + (size_t)h * ctxm * HD + sub * VEC;
    vb = reinterpret_cast<const WT*>(vcp) + (size_t)h * ctxm * HD + sub * VEC;
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      // slots past the split's end re-read the split's OWN last key (row 0 always exists): clamped to the context's last row they
      // were one more line every workgroup of the head asked for at the same moment
      const int jk = min(j0 + u * KPB + wave * KPW + grp, max(j1, 1) - 1);
      tp_ld16(kr[u], kb + (size_t)jk * HD);
      tp_ld16(vr[u], vb + (size_t)jk * HD);
    }
  };
  tp_stage2();
  auto load_pass = [&](int base) {  // later passes of a long context (> 16 * UNR * KPB rows): plain loads
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int jk = min(base + u * KPB + wave * KPW + grp, max(j1, 1) - 1);
      const uint4 k4 = ld16(kb + (size_t)jk * HD), v4 = ld16(vb + (size_t)jk * HD);
      kr[u] = vx_u32x4{k4.x, k4.y, k4.z, k4.w};
      vr[u] = vx_u32x4{v4.x, v4.y, v4.z, v4.w};
    }
  };
  constexpr int N_KV = 2 * UNR, N_WO = OCH, N_W = 3 * KCH;

  // ---- LN1, the three dot products, publish ----
  tp_row_wait<FIRST, N_W + 3 + N_WO + N_KV>(rl);
  TP_STAMP(0, 7);
  tp_row_norm<FIRST>(rl, xs, red, tid);
  TP_STAMP(0, 1);
  {
    float xr[KCH][VEC];
    tp_row_read<KCH, VEC>(xs, lane, xr);
    float acc[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      uint4 wr[KCH];
#pragma unroll
      for (int c = 0; c < KCH; ++c) { tp_wait<N_WO + N_KV + 3>(w[r][c]); wr[c] = tp_u4(w[r][c]); }
      acc[r] = wave_sum_dpp(tp_dot<WT, KCH>(wr, xr));
    }
    tp_wait<N_WO + N_KV + 2>(e_bias);
    if (lane < 3) {
      float v = (lane == 0 ? acc[0] : lane == 1 ? acc[1] : acc[2]) + e_bias;
      if (lane > 0) {  // K / V travel rounded to the cache's element type: the values later passes read back
        const WT rv = from_f32<WT>(v);
        v = to_f32(rv);
        if (!st_done) {
          WT* cache = reinterpret_cast<WT*>(lane == 1 ? a.kcache : a.vcache);
          cache[((size_t)h * a.ctx_max + st_row) * HD + cih] = rv;
        }
      }
      gran_store(a.gq + (size_t)h * FQ_QKV + lane * HD + cih, v, tag);
    }
  }
  TP_STAMP(0, 2);
  // ---- the head's q / newest k / newest v ----
  {
    float v1[1];
    gran_gather<1>(a.gq + (size_t)h * FQ_QKV, FQ_QKV, tag, v1, a.err, 1u);
    if (tid < FQ_QKV) s_qkv[tid] = v1[0];
  }
#pragma unroll
  for (int u = 0; u < UNR; ++u) { tp_wait<0>(kr[u]); tp_wait<0>(vr[u]); }  // the gather waited for everything
#pragma unroll
  for (int m = 0; m < OCH; ++m) tp_wait<0>(wo[m]);
  asm volatile("" : "+v"(carry_ll), "+v"(carry_f), "+v"(bo));
  TP_STAMP(0, 3);
  __syncthreads();
  TP_STAMP(0, 8);
  float qv[VEC];
#pragma unroll
  for (int c = 0; c < VEC; ++c) qv[c] = s_qkv[sub * VEC + c] * a.scale;

  // ---- this split's keys: every WAVE keeps its own running softmax (no workgroup reduction in the loop) ----
  float M = -INFINITY, L = 0.f, acc[VEC];
#pragma unroll
  for (int c = 0; c < VEC; ++c) acc[c] = 0.f;
  const bool owner = (j == FQ_G - 1) && wave == 0;  // the newest key: wave 0 of the last split (uniform per wave)
  float sn = -INFINITY;
  if (owner) {
    float dot = 0.f;
#pragma unroll
    for (int c = 0; c < VEC; ++c) dot = fmaf(s_qkv[HD + sub * VEC + c], qv[c], dot);
    sn = (LPK == 8) ? group8_sum_dpp(dot) : group16_sum_dpp(dot);  // the same number in every lane group
  }
  for (int base = j0; base < j1 || (owner && base == j0); base += UNR * KPB) {
    if (base != j0) load_pass(base);
    const int nr = (j1 - base + KPB - 1) / KPB;  // rounds of this pass that hold keys (uniform; <= 0: the newest key only)
    float sc[UNR], mloc = (base == j0) ? sn : -INFINITY;
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      sc[u] = -INFINITY;
      if (u < nr) {
        const int jk = base + u * KPB + wave * KPW + grp;
        float kf[VEC];
        unpack<WT>(tp_u4(kr[u]), kf);
        float dot = 0.f;
#pragma unroll
        for (int c = 0; c < VEC; ++c) dot = fmaf(kf[c], qv[c], dot);
        dot = (LPK == 8) ? group8_sum_dpp(dot) : group16_sum_dpp(dot);
        sc[u] = (jk < j1) ? dot : -INFINITY;
        mloc = fmaxf(mloc, sc[u]);
      }
    }
    mloc = wave_max_dpp(mloc);
    if (mloc != -INFINITY) {  // wave-uniform
      const float Mn = fmaxf(M, mloc);
      const float corr = (M == -INFINITY) ? 0.f : tp_exp<WT>(M - Mn);
      L *= corr;
#pragma unroll
      for (int c = 0; c < VEC; ++c) acc[c] *= corr;
      M = Mn;
#pragma unroll
      for (int u = 0; u < UNR; ++u)
        if (u < nr) {
          float vf[VEC];
          unpack<WT>(tp_u4(vr[u]), vf);
          const float p = tp_exp<WT>(sc[u] - M);  // exp(-inf) = 0 for the masked keys
          L += p;
#pragma unroll
          for (int c = 0; c < VEC; ++c) acc[c] = fmaf(p, vf[c], acc[c]);
        }
      if (owner && base == j0 && grp == 0) {
        const float p = tp_exp<WT>(sn - M);
        L += p;
#pragma unroll
        for (int c = 0; c < VEC; ++c) acc[c] = fmaf(p, s_qkv[2 * HD + sub * VEC + c], acc[c]);
      }
    }
  }
  TP_STAMP(0, 9);
  // merge: a wave's key groups share its maximum, so they add up plainly - through the wave's own LDS rows, no workgroup
  // barrier (LDS operations of one wave execute in order); lane c then owns channel c of the wave's partial
  {
#pragma unroll
    for (int c = 0; c < VEC; c += 4)
      *reinterpret_cast<float4*>(&sm_o[wave * KPW + grp][sub * VEC + c]) = make_float4(acc[c], acc[c + 1], acc[c + 2], acc[c + 3]);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    float oc = 0.f;
#pragma unroll
    for (int g = 0; g < KPW; ++g) oc += sm_o[wave * KPW + g][lane];
    const float Lw = wave_sum_dpp(sub == 0 ? L : 0.f);
    s_wo[wave][lane] = oc;
    if (lane == 0) { sm_m[wave] = M; sm_l[wave] = Lw; }
  }
  TP_STAMP(0, 10);
  __syncthreads();
  TP_STAMP(0, 11);
  if (tid < FQ_PART) {
    const float Mx = fmaxf(fmaxf(sm_m[0], sm_m[1]), fmaxf(sm_m[2], sm_m[3]));
    float v = 0.f;
    if (tid == HD) {
      v = Mx;
    } else {
#pragma unroll
      for (int wv = 0; wv < 4; ++wv)
        v = fmaf(tid < HD ? s_wo[wv][tid] : sm_l[wv], (sm_m[wv] == -INFINITY) ? 0.f : tp_exp<WT>(sm_m[wv] - Mx), v);
    }
    gran_store(a.gp + ((size_t)h * FQ_G + j) * FQ_PART + tid, v, tag);
  }
  TP_STAMP(0, 4);
  // ---- both heads' partial softmaxes (2 x 16 x 66 granules), combined by every workgroup of the XCD ----
  {
    constexpr int NG = (2 * FQ_G * FQ_PART + 255) / 256;
    float pv[NG];
    gran_gather<NG>(a.gp + (size_t)(2 * x) * FQ_G * FQ_PART, 2 * FQ_G * FQ_PART, tag, pv, a.err, 2u);
#pragma unroll
    for (int g = 0; g < NG; ++g)
      if (tid + 256 * g < 2 * FQ_G * FQ_PART) s_part[tid + 256 * g] = pv[g];
  }
  TP_STAMP(0, 5);
  __syncthreads();
  if (tid < 2 * FQ_G) {  // thread (head, split): exp(m_s - M) / L of its split; a head's 16 threads are one DPP row
    const float* hp = s_part + (tid >> 4) * FQ_G * FQ_PART;
    float Mx = hp[HD];
#pragma unroll
    for (int s = 1; s < FQ_G; ++s) Mx = fmaxf(Mx, hp[s * FQ_PART + HD]);
    const float pm = hp[(tid & 15) * FQ_PART + HD];
    const float es = (pm == -INFINITY) ? 0.f : tp_exp<WT>(pm - Mx);
    const float Ls = group16_sum_dpp(hp[(tid & 15) * FQ_PART + HD + 1] * es);
    s_fac[tid] = es * (1.0f / Ls);
  }
  __syncthreads();
  if (tid < TP_ACH) {
    const float* hp = s_part + (tid >> 6) * FQ_G * FQ_PART + (tid & 63);
    const float* fp = s_fac + (tid >> 6) * FQ_G;
    float o = 0.f;
#pragma unroll
    for (int s = 0; s < FQ_G; ++s) o = fmaf(hp[s * FQ_PART], fp[s], o);
    s_att[tid] = o;
  }
  __syncthreads();
  // ---- the XCD's K-slice of the out-projection: row 32i + tid/8 over the XCD's 128 channels, 8 lanes per row ----
  {
    float s = 0.f;
#pragma unroll
    for (int m = 0; m < OCH; ++m) {
      float wf[VEC];
      unpack<WT>(tp_u4(wo[m]), wf);
      const float* ap = s_att + ((tid & 7) + 8 * m) * VEC;
#pragma unroll
      for (int c = 0; c < VEC; ++c) s = fmaf(wf[c], ap[c], s);
    }
    s = group8_sum_dpp(s);
    tp_acc_tail<FIRST>(a.acc, s, x, i, tid, carry_ll, carry_f, bo);
  }
  TP_STAMP(0, 6);
}

// ------------------------------------------------------------------------------------------------ feed-forward half
// W2_ = this layer's linear2 re-laid out by tp_repack_kernel: [x][i][m][t], word (m, t) = row 32i + t/8, hidden units
// 512x + (t%8 + 8m) * VEC .. + VEC.
template <typename WT>
__global__ __launch_bounds__(256) void tp_ffn_kernel(const void* __restrict__ W1_, const long long* __restrict__ racc,
                                                     const float* __restrict__ gbb,
                                                     const unsigned* __restrict__ epoch, const void* __restrict__ W2_,
                                                     const TpFfnArgs a) {
  constexpr int VEC = Vec16<WT>::N;
  constexpr int KCH = TP_D / (64 * VEC);
  constexpr int FCH = TP_HID / (8 * VEC);  // linear2 words per thread: 8 (bf16) / 16 (fp32)
  __shared__ __attribute__((aligned(16))) float xs[TP_D];
  __shared__ float red[8];
  __shared__ __attribute__((aligned(16))) float hs[TP_HID];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int x = blockIdx.x & (TP_X - 1), i = blockIdx.x >> 3;
  const unsigned tag = *epoch;
  const WT* __restrict__ W1 = reinterpret_cast<const WT*>(W1_);
  const int row0 = TP_HID * x + 16 * i + 4 * wave;  // this wave's four hidden units

  TpRowLoads rl;
  tp_row_issue<false>(rl, nullptr, racc, gbb, tid);
  TP_STAMP(1, 0);
  vx_u32x4 w1[4][KCH];
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int c = 0; c < KCH; ++c) tp_ld16w(w1[r][c], W1 + (size_t)(row0 + r) * TP_D + (c * 64 + lane) * VEC);
  float e_b1;
  {
    const float* bp = a.b1;  // first use of the by-value argument struct: its kernarg wait belongs HERE
    asm volatile("" : "+s"(bp));
    tp_ld4f(e_b1, bp + row0 + min(lane, 3));
  }
  unsigned long long carry_ll;
  float bo;
  asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(carry_ll) : "v"(racc + 32 * i + (tid >> 3)) : "memory");
  tp_ld4f(bo, a.acc.bias + 32 * i + (tid >> 3));
  vx_u32x4 w2[FCH];
  auto tp_stage2 = [&]() {
    const uint4* wb = reinterpret_cast<const uint4*>(W2_) + (size_t)(x * TP_WG + i) * FCH * 256;
#pragma unroll
    for (int m = 0; m < FCH; ++m) tp_ld16w(w2[m], wb + m * 256 + tid);
  };
  tp_stage2();
  tp_row_wait<false, 4 * KCH + 3 + FCH>(rl);
  TP_STAMP(1, 7);
  tp_row_norm<false>(rl, xs, red, tid);
  TP_STAMP(1, 1);
  {
    float xr[KCH][VEC];
    tp_row_read<KCH, VEC>(xs, lane, xr);
    float acc[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      uint4 wr[KCH];
#pragma unroll
      for (int c = 0; c < KCH; ++c) { tp_wait<3 + FCH>(w1[r][c]); wr[c] = tp_u4(w1[r][c]); }
      acc[r] = wave_sum_dpp(tp_dot<WT, KCH>(wr, xr));
    }
    tp_wait<2 + FCH>(e_b1);
    if (lane < 4) {
      const float v = (lane == 0 ? acc[0] : lane == 1 ? acc[1] : lane == 2 ? acc[2] : acc[3]) + e_b1;
      gran_store(a.gh + (size_t)x * TP_HID + 16 * i + 4 * wave + lane, fmaxf(v, 0.f), tag);
    }
  }
  TP_STAMP(1, 2);
  {
    float hv[2];
    gran_gather<2>(a.gh + (size_t)x * TP_HID, TP_HID, tag, hv, a.err, 3u);
    hs[tid] = hv[0]; hs[tid + 256] = hv[1];
  }
#pragma unroll
  for (int m = 0; m < FCH; ++m) tp_wait<0>(w2[m]);  // the gather waited for everything
  asm volatile("" : "+v"(carry_ll), "+v"(bo));
  TP_STAMP(1, 3);
  __syncthreads();
  {
    float s = 0.f;
#pragma unroll
    for (int m = 0; m < FCH; ++m) {
      float wf[VEC];
      unpack<WT>(tp_u4(w2[m]), wf);
      const float* hp = hs + ((tid & 7) + 8 * m) * VEC;
#pragma unroll
      for (int c = 0; c < VEC; ++c) s = fmaf(wf[c], hp[c], s);
    }
    s = group8_sum_dpp(s);
    tp_acc_tail<false>(a.acc, s, x, i, tid, carry_ll, 0.f, bo);
  }
  TP_STAMP(1, 4);
}

// ------------------------------------------------------------------------------------------------ head
// logits = ar_predict_layer(norm(x + bias + partials)) (valle.py:1039): 4 rows per wave, grid = ceil(N / 16).
template <typename WT>
__global__ __launch_bounds__(256) void tp_head_kernel(const void* __restrict__ W_, const long long* __restrict__ racc,
                                                      const float* __restrict__ gbb,
                                                      const ArState* __restrict__ st, const TpHeadArgs a) {
  constexpr int VEC = Vec16<WT>::N;
  constexpr int KCH = TP_D / (64 * VEC);
  __shared__ __attribute__((aligned(16))) float xs[TP_D];
  __shared__ float red[8];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const WT* __restrict__ W = reinterpret_cast<const WT*>(W_);
  const int row0 = (blockIdx.x * 4 + wave) * 4;
  TpRowLoads rl;
  tp_row_issue<false>(rl, nullptr, racc, gbb, tid);
  vx_u32x4 w[4][KCH];
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int c = 0; c < KCH; ++c) tp_ld16w(w[r][c], W + (size_t)min(row0 + r, a.N - 1) * TP_D + (c * 64 + lane) * VEC);
  const int st_pass = st->pass, st_trace = st->trace_logits, st_done = st->done;
  tp_row_wait<false, 4 * KCH>(rl);
  tp_row_norm<false>(rl, xs, red, tid);
  float xr[KCH][VEC];
  tp_row_read<KCH, VEC>(xs, lane, xr);
  float acc[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    uint4 wr[KCH];
#pragma unroll
    for (int c = 0; c < KCH; ++c) { tp_wait<0>(w[r][c]); wr[c] = tp_u4(w[r][c]); }
    acc[r] = wave_sum_dpp(tp_dot<WT, KCH>(wr, xr));
  }
  const int row = row0 + lane;
  if (lane < 4 && row < a.N && !st_done) {  // a finished decode keeps replaying the step: leave its last logits row intact
    const float v = lane == 0 ? acc[0] : lane == 1 ? acc[1] : lane == 2 ? acc[2] : acc[3];
    a.logits[row] = v;
    if (st_trace) a.logits[LOGITS_CUR + (size_t)st_pass * a.N + row] = v;
  }
}

// ---- one-time re-layout of a (1024, K) matrix into the sharded word order the two kernels above stream:
// dst word [(x * 32 + i) * NW + m][t] = src[row 32i + t/8][cols KX * x + (t%8 + 8m) * VEC .. + VEC], KX = K / 8, NW = KX / (8 VEC)
template <typename WT>
__global__ void tp_repack_kernel(const WT* __restrict__ src, uint4* __restrict__ dst, int K) {
  constexpr int VEC = Vec16<WT>::N;
  const int KX = K / TP_X, NW = KX / (8 * VEC);
  const size_t word = (size_t)blockIdx.x * 256 + threadIdx.x;  // one 16-byte word per thread
  const size_t total = (size_t)TP_D * K / VEC;
  if (word >= total) return;
  const int t = (int)(word & 255);
  const size_t blk = word >> 8;
  const int m = (int)(blk % NW);
  const int wg = (int)(blk / NW);
  const int x = wg / TP_WG, i = wg % TP_WG;
  const int row = 32 * i + (t >> 3), col = KX * x + ((t & 7) + 8 * m) * VEC;
  dst[word] = *reinterpret_cast<const uint4*>(src + (size_t)row * K + col);
}

}  // namespace vx
