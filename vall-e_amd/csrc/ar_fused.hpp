// Fused first half of a decode-step layer: QKV projection + single-query attention over the KV cache + combine of the
// key splits, ONE launch instead of three (valle/modules/transformer.py:297-301 + activation.py:407-427 with a KV cache).
//
// Why: every dependent launch of the step costs 3.2-4.5 us whatever it computes (profiles/r02_notes.md), and the two edges
// QKV -> attention -> out-projection carry almost no data: a head's attention needs that head's 64 q values and its newest
// K / V row, nothing else.  So the 16 H workgroups of the launch are grouped by HEAD: workgroup (h, j) projects rows
// 4j..4j+3 of head h's q, k and v (the same 12 weight rows per workgroup as the plain GEMV), the 16 workgroups of a head
// exchange those 192 values among themselves, and workgroup (h, j) then attends over the j-th sixteenth of the cached keys -
// whose K / V rows it requested at kernel START, next to its weight rows, so the cache traffic is off the critical path.
// Workgroup (h, 0) finally merges the head's 16 partial softmaxes and writes the 64 normalised values the out-projection
// reads (4 KB instead of the 35 KB of partials per workgroup the PRO_ATTN prologue fetched).
//
// Hand-over = tagged granules (cdna_hip_programming.md, decode rows): every value travels as one 8-byte {value, tag} word
// written with an sc1 (write-through) store and polled with sc1 loads; tag = the step counter the sampling kernel bumps once
// per step, so no flag, fence or ordering between stores is needed and a stale word can never match.  Workgroup ids put a
// head's workgroups on one XCD (blockIdx % 8 == h % 8 when nhead % 8 == 0), where such a hop is 0.5 us (tests/probes/
// xcd_hop.hip); nothing depends on the placement for correctness.  Every spin is bounded (FQ_SPIN_MAX polls, then an error
// word is set and the wave goes on), and no workgroup waits before it has published its own values, so the grid drains
// whenever all 16 H workgroups are resident (checked on the host with the occupancy API before the path is chosen).
#pragma once
#include "ar_kernels.hpp"

namespace vx {

constexpr int FQ_G = 16;             // workgroups per head = key splits
constexpr int FQ_QKV = 192;          // granules a head exchanges after the projection: q[64], k_new[64], v_new[64]
constexpr int FQ_PART = 66;          // granules of one split's partial softmax: o[64], m, l
constexpr int FQ_SPIN_MAX = 1 << 18; // polls before a wave gives up (~0.3 s)

typedef unsigned long long fq_gran;

#ifdef VX_STAMPS
// stamps build: thread 0 of every workgroup records s_memrealtime (100 MHz) at its phase boundaries into
// [layer][workgroup][8] of the buffer vx_debug_fqstamps arms (the last step's values survive)
}  // namespace vx
extern __device__ unsigned long long* g_fq_stamps;
namespace vx {
#define FQ_STAMP(i)                                                                                                   \
  do {                                                                                                                \
    unsigned long long* fq_r_ = g_fq_stamps;                                                                          \
    if (fq_r_ != nullptr && threadIdx.x == 0) fq_r_[((size_t)a.layer * 256 + blockIdx.x) * 16 + (i)] = __builtin_amdgcn_s_memrealtime(); \
  } while (0)
#else
#define FQ_STAMP(i) do { } while (0)
#endif

struct FusedArgs {
  const float* bias;       // (3d,)
  const ArState* st;
  const unsigned* epoch;   // step counter (sampling kernel): the tag of this step's granules
  unsigned* err;           // != 0: a bounded spin ran out (vx_ar_decode reports it)
  fq_gran* gq;             // this layer's (nhead, FQ_QKV) granules
  fq_gran* gp;             // this layer's (nhead, FQ_G, FQ_PART) granules
  float* out;              // (d,) normalised attention output = input of the out-projection
  void* kcache;            // this layer's K: (nhead, ctx_max, 64) WT
  void* vcache;
  float* xnorm_out;        // PRO_LN: if set, workgroup 0 / wave 0 also stores LN(x) here (post-norm residual base)
  int d, nhead, ctx_max;
  float scale;
  const void* pf;          // weight warm-up of a later GEMV (GemvArgs.pf)
  unsigned pf_slice, pf_total;
  int layer;
};

__device__ __forceinline__ void gran_store(fq_gran* p, float v, unsigned tag) {
  const fq_gran g = ((fq_gran)tag << 32) | (fq_gran)__float_as_uint(v);
  asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(p), "v"(g) : "memory");
}
__device__ __forceinline__ void gran_load(fq_gran& v, const fq_gran* p) {
  asm volatile("global_load_dwordx2 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
}

// Thread t gathers granules t, t + 256, ... (NG per thread, indices clamped to count - 1) until every one carries `tag`.
// Wave-uniform exit; bounded.
template <int NG>
__device__ __forceinline__ void gran_gather(const fq_gran* base, int count, unsigned tag, float (&val)[NG], unsigned* err, unsigned code) {
  fq_gran g[NG];
  for (int spin = 0;; ++spin) {
#pragma unroll
    for (int i = 0; i < NG; ++i) gran_load(g[i], base + min((int)threadIdx.x + 256 * i, count - 1));
    if (NG == 1) asm volatile("s_waitcnt vmcnt(0)" : "+v"(g[0])::"memory");
    else {
#pragma unroll
      for (int i = 0; i < NG; ++i) asm volatile("s_waitcnt vmcnt(0)" : "+v"(g[i])::"memory");
    }
    bool ok = true;
#pragma unroll
    for (int i = 0; i < NG; ++i) ok = ok && ((unsigned)(g[i] >> 32) == tag);
    if (__ballot(!ok) == 0ull) break;
    if (spin >= FQ_SPIN_MAX) {
      if ((threadIdx.x & 63) == 0) atomicExch(err, code);
      break;
    }
  }
#pragma unroll
  for (int i = 0; i < NG; ++i) val[i] = __uint_as_float((unsigned)g[i]);
}

// Cheap pre-poll in front of a large gather: lane l of every wave watches sentinel granule base[min(l, count-1) * stride] until
// all `count` (<= 64... or 128 with two per lane) carry the tag.  Bounded; the gather behind it still verifies every granule.
__device__ __forceinline__ void gran_wait(const fq_gran* base, int stride, int count, unsigned tag) {
  const int lane = threadIdx.x & 63;
  const fq_gran* p0 = base + (size_t)min(lane, count - 1) * stride;
  const fq_gran* p1 = base + (size_t)min(lane + 64, count - 1) * stride;
  for (int spin = 0; spin < FQ_SPIN_MAX; ++spin) {
    fq_gran g0, g1;
    gran_load(g0, p0);
    gran_load(g1, p1);
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(g0), "+v"(g1)::"memory");
    if (__ballot((unsigned)(g0 >> 32) != tag || (unsigned)(g1 >> 32) != tag) == 0ull) break;
  }
}

// grid = 16 * nhead workgroups of 256; workgroup b: head h = b % nhead, split j = b / nhead.
template <typename WT, int KCH, int PRO>
__global__ __launch_bounds__(256) void qkv_attn_kernel(const void* __restrict__ W_, const float* __restrict__ xin,
                                                       const float* __restrict__ gamma_, const float* __restrict__ beta_,
                                                       unsigned K_, const FusedArgs a) {
  constexpr int VEC = Vec16<WT>::N;
  constexpr int V4 = VEC / 4;
  constexpr int HD = 64;
  constexpr int LPK = HD / VEC;   // lanes per key: 8 (bf16) / 16 (fp32)
  constexpr int KPW = 64 / LPK;   // keys per wave-load
  constexpr int KPB = 4 * KPW;    // keys per workgroup round
  constexpr int UNR = 4;          // rounds held in registers: UNR * KPB old keys per pass (one pass up to ctx 16 * UNR * KPB)
  __shared__ __attribute__((aligned(16))) float s_qkv[FQ_QKV];
  __shared__ float sm_red[4];
  __shared__ __attribute__((aligned(16))) float sm_o[4 * KPW][HD + 1];
  __shared__ float sm_l[4 * KPW];
  __shared__ float s_part[FQ_G * FQ_PART];
  const int K = (int)K_, d = a.d;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = blockIdx.x % a.nhead, j = blockIdx.x / a.nhead;
  const int cih = 4 * j + wave;   // channel inside the head this wave projects (q, k and v row)
  const int ch = h * HD + cih;
  const WT* __restrict__ W = reinterpret_cast<const WT*>(W_);
  FQ_STAMP(0);

  // ---- (A) small loads first, unconditional on clamped addresses (see gemv_kernel) ----
  float4 x4[KCH][V4], g4[KCH][V4], b4[KCH][V4];
  bool kok[KCH];
#pragma unroll
  for (int c = 0; c < KCH; ++c) kok[c] = (c * 64 + lane) * VEC < K;
#pragma unroll
  for (int c = 0; c < KCH; ++c) {
    const int k = min((c * 64 + lane) * VEC, K - VEC);
#pragma unroll
    for (int i = 0; i < V4; ++i) {
      x4[c][i] = *reinterpret_cast<const float4*>(xin + k + 4 * i);
      if (PRO == PRO_LN) {
        g4[c][i] = *reinterpret_cast<const float4*>(gamma_ + k + 4 * i);
        b4[c][i] = *reinterpret_cast<const float4*>(beta_ + k + 4 * i);
      }
    }
  }
  // ---- (B) this wave's three weight rows ----
  uint4 w[3][KCH];
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < KCH; ++c) {
      const int k = min((c * 64 + lane) * VEC, K - VEC);
      w[r][c] = ld16(W + (size_t)(r * d + ch) * K + k);
    }
  __builtin_amdgcn_sched_barrier(0);
  const float e_bias = a.bias[min(lane, 2) * d + ch];
  uint4 pfv[8];
  const bool warm = a.pf != nullptr;  // uniform
  if (warm) {
    const unsigned lim = min(a.pf_slice, a.pf_total - min(a.pf_total, blockIdx.x * a.pf_slice));
    const char* pb = reinterpret_cast<const char*>(a.pf) + min((size_t)blockIdx.x * a.pf_slice, (size_t)a.pf_total - 16);
#pragma unroll
    for (int i = 0; i < 8; ++i)
      pfv[i] = *reinterpret_cast<const uint4*>(pb + min((unsigned)(i * 4096 + tid * 16), max(lim, 16u) - 16u));
  }
  __builtin_amdgcn_sched_barrier(0);
  // ---- (B') this split's cached keys, requested before anything waits: keys [j0, j1) of the n_old rows earlier passes wrote
  const int st_row = a.st->row, st_done = a.st->done;
  const unsigned tag = *a.epoch;
  const int n_old = st_row;  // the newest row (index st_row) is produced by this launch and travels in the granules
  const int chunk = (n_old + FQ_G - 1) / FQ_G;
  const int j0 = j * chunk, j1 = min(n_old, j0 + chunk);
  const int sub = lane % LPK, grp = lane / LPK;
  const WT* kb = reinterpret_cast<const WT*>(a.kcache) + (size_t)h * a.ctx_max * HD + sub * VEC;
  const WT* vb = reinterpret_cast<const WT*>(a.vcache) + (size_t)h * a.ctx_max * HD + sub * VEC;
  uint4 kr[UNR], vr[UNR];
  auto load_pass = [&](int base) {
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int jk = min(base + u * KPB + wave * KPW + grp, max(n_old, 1) - 1);  // clamped: row 0 always exists
      kr[u] = ld16(kb + (size_t)jk * HD);
      vr[u] = ld16(vb + (size_t)jk * HD);
    }
  };
  load_pass(j0);
  __builtin_amdgcn_sched_barrier(0);

  // ---- (C) LayerNorm of the whole row in this wave's registers (modules/transformer.py:57-74) ----
  float xr[KCH][VEC];
#pragma unroll
  for (int c = 0; c < KCH; ++c)
#pragma unroll
    for (int i = 0; i < V4; ++i) {
      xr[c][4 * i] = kok[c] ? x4[c][i].x : 0.f; xr[c][4 * i + 1] = kok[c] ? x4[c][i].y : 0.f;
      xr[c][4 * i + 2] = kok[c] ? x4[c][i].z : 0.f; xr[c][4 * i + 3] = kok[c] ? x4[c][i].w : 0.f;
    }
  if (PRO == PRO_LN) {
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < KCH; ++c)
#pragma unroll
      for (int i = 0; i < VEC; ++i) s += xr[c][i];
    const float mean = wave_sum_dpp(s) / (float)K;
    float ss = 0.f;
#pragma unroll
    for (int c = 0; c < KCH; ++c)
#pragma unroll
      for (int i = 0; i < VEC; ++i) {
        const float dv = xr[c][i] - mean;
        ss += kok[c] ? dv * dv : 0.f;
      }
    const float rstd = 1.0f / sqrtf(wave_sum_dpp(ss) / (float)K + LN_EPS);
#pragma unroll
    for (int c = 0; c < KCH; ++c)
#pragma unroll
      for (int i = 0; i < V4; ++i) {
        const float o0 = (xr[c][4 * i] - mean) * rstd * g4[c][i].x + b4[c][i].x;
        const float o1 = (xr[c][4 * i + 1] - mean) * rstd * g4[c][i].y + b4[c][i].y;
        const float o2 = (xr[c][4 * i + 2] - mean) * rstd * g4[c][i].z + b4[c][i].z;
        const float o3 = (xr[c][4 * i + 3] - mean) * rstd * g4[c][i].w + b4[c][i].w;
        xr[c][4 * i] = kok[c] ? o0 : 0.f; xr[c][4 * i + 1] = kok[c] ? o1 : 0.f;
        xr[c][4 * i + 2] = kok[c] ? o2 : 0.f; xr[c][4 * i + 3] = kok[c] ? o3 : 0.f;
      }
    if (a.xnorm_out != nullptr && blockIdx.x == 0 && wave == 0) {
#pragma unroll
      for (int c = 0; c < KCH; ++c)
        if (kok[c]) {
#pragma unroll
          for (int i = 0; i < V4; ++i)
            *reinterpret_cast<float4*>(a.xnorm_out + (c * 64 + lane) * VEC + 4 * i) =
                make_float4(xr[c][4 * i], xr[c][4 * i + 1], xr[c][4 * i + 2], xr[c][4 * i + 3]);
        }
    }
  }

  // ---- (D) the three dot products; lane r publishes row r (0 q, 1 k, 2 v) ----
  {
    float acc[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      float s = 0.f;
#pragma unroll
      for (int c = 0; c < KCH; ++c) {
        float wf[VEC];
        unpack<WT>(w[r][c], wf);
#pragma unroll
        for (int i = 0; i < VEC; ++i) s = fmaf(wf[i], xr[c][i], s);
      }
      acc[r] = wave_sum_dpp(s);
    }
    if (lane < 3) {
      float v = (lane == 0 ? acc[0] : lane == 1 ? acc[1] : acc[2]) + e_bias;
      if (lane > 0) {  // K / V travel rounded to the cache's element type: the same values later passes read back
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
  FQ_STAMP(1);

  // ---- (E) the head's q / newest k / newest v, from its 16 workgroups ----
  {
    float v1[1];
    gran_gather<1>(a.gq + (size_t)h * FQ_QKV, FQ_QKV, tag, v1, a.err, 1u);
    if (tid < FQ_QKV) s_qkv[tid] = v1[0];
  }
  FQ_STAMP(2);
  __syncthreads();
  float qv[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) qv[i] = s_qkv[sub * VEC + i];

  // ---- (F) scores, softmax and P.V over this split's keys (attn_decode_kernel's two-pass-over-registers scheme) ----
  float M = -INFINITY, L = 0.f, acc[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) acc[i] = 0.f;
  for (int base = j0; base < j1; base += UNR * KPB) {
    if (base != j0) load_pass(base);
    float sc[UNR], mloc = -INFINITY;
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int jk = base + u * KPB + wave * KPW + grp;
      float kf[VEC];
      unpack<WT>(kr[u], kf);
      float dot = 0.f;
#pragma unroll
      for (int i = 0; i < VEC; ++i) dot = fmaf(kf[i], qv[i], dot);
      dot = (LPK == 8) ? group8_sum_dpp(dot) : group16_sum_dpp(dot);
      sc[u] = (jk < j1) ? dot * a.scale : -INFINITY;
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
      unpack<WT>(vr[u], vf);
      const float p = (sc[u] == -INFINITY) ? 0.f : expf(sc[u] - M);
      L += p;
#pragma unroll
      for (int i = 0; i < VEC; ++i) acc[i] = fmaf(p, vf[i], acc[i]);
    }
  }
  if (j == FQ_G - 1) {  // the newest key belongs to the last split; its score is the same number in every lane group
    float dot = 0.f;
#pragma unroll
    for (int i = 0; i < VEC; ++i) dot = fmaf(s_qkv[HD + sub * VEC + i], qv[i], dot);
    dot = (LPK == 8) ? group8_sum_dpp(dot) : group16_sum_dpp(dot);
    const float sn = dot * a.scale;
    const float Mn = fmaxf(M, sn);
    const float corr = (M == -INFINITY) ? 0.f : expf(M - Mn);
    L *= corr;
#pragma unroll
    for (int i = 0; i < VEC; ++i) acc[i] *= corr;
    M = Mn;
    if (wave == 0 && grp == 0) {
      const float p = expf(sn - M);
      L += p;
#pragma unroll
      for (int i = 0; i < VEC; ++i) acc[i] = fmaf(p, s_qkv[2 * HD + sub * VEC + i], acc[i]);
    }
  }
  // sum the 4 * KPW key groups through LDS; thread c sums channel c
  const int gi = wave * KPW + grp;
#pragma unroll
  for (int i = 0; i < VEC; ++i) sm_o[gi][sub * VEC + i] = acc[i];
  if (sub == 0) sm_l[gi] = L;
  __syncthreads();
  if (tid < FQ_PART) {
    float v;
    if (tid < HD) {
      v = 0.f;
#pragma unroll
      for (int gidx = 0; gidx < 4 * KPW; ++gidx) v += sm_o[gidx][tid];
    } else if (tid == HD) {
      v = M;
    } else {
      v = 0.f;
#pragma unroll
      for (int gidx = 0; gidx < 4 * KPW; ++gidx) v += sm_l[gidx];
    }
    gran_store(a.gp + ((size_t)h * FQ_G + j) * FQ_PART + tid, v, tag);
  }
  FQ_STAMP(3);
  if (warm) {  // the warm-up loads stay live (and unwaited by the compiler) until here
#pragma unroll
    for (int i = 0; i < 8; ++i) asm volatile("" ::"v"(pfv[i].x), "v"(pfv[i].y), "v"(pfv[i].z), "v"(pfv[i].w));
  }
  if (j != 0) return;

  // ---- (G) workgroup (h, 0): flash-decoding combine of the head's 16 partials (the PRO_ATTN prologue's arithmetic) ----
  {
    constexpr int NG = (FQ_G * FQ_PART + 255) / 256;
    float pv[NG];
    gran_gather<NG>(a.gp + (size_t)h * FQ_G * FQ_PART, FQ_G * FQ_PART, tag, pv, a.err, 2u);
#pragma unroll
    for (int i = 0; i < NG; ++i)
      if (tid + 256 * i < FQ_G * FQ_PART) s_part[tid + 256 * i] = pv[i];
  }
  FQ_STAMP(4);
  __syncthreads();
  if (tid < HD) {
    float Mx = s_part[HD];
#pragma unroll
    for (int s = 1; s < FQ_G; ++s) Mx = fmaxf(Mx, s_part[s * FQ_PART + HD]);
    float Ls = 0.f, o = 0.f;
#pragma unroll
    for (int s = 0; s < FQ_G; ++s) {
      const float pm = s_part[s * FQ_PART + HD];
      const float f = (pm == -INFINITY) ? 0.f : expf(pm - Mx);
      Ls += s_part[s * FQ_PART + HD + 1] * f;
      o += s_part[s * FQ_PART + tid] * f;
    }
    a.out[h * HD + tid] = o * (1.0f / Ls);
  }
  FQ_STAMP(5);
}

}  // namespace vx
