// Probe (not product): a persistent decode-step skeleton.  ONE launch runs T tokens x L layers x 6 dependent stages with the
// real step's weight bytes (26 MB bf16 per layer, streamed from a 317 MB buffer); stages hand their output vectors over as
// 8-byte {value, tag} granules written with sc1 stores into EIGHT replicas (one per XCD) and gathered by a dedicated fifth
// wave of every workgroup from its own XCD's replica (tests/probes/xcd_hop.hip: a gather of 8 KB costs 1.3 us when 32
// workgroups share a replica, 2.7 us when all 256 poll one copy).
//
//   hipcc --offload-arch=gfx950 -O3 -I vall-e_amd/csrc -o tests/probes/persist_chain tests/probes/persist_chain.hip
//
// Stages per layer (G = active workgroups, rows split evenly over them, 4 consumer waves each):
//   A  QKV projection      N 3072, K 1024, G 256   (LayerNorm in the gathering wave)
//   B  attention stand-in  N 2048, K  256, G 128   (8 KB of "K/V" per workgroup instead of weights)
//   C  combine stand-in    N 1024, K  128, G  16   (no weights)
//   D  out-projection      N 1024, K 1024, G 256
//   E  FFN1                N 4096, K 1024, G 256   (LayerNorm; output published as bf16 pairs)
//   F  FFN2                N 1024, K 4096, G 256   (input = 2048 bf16-pair granules)
// The arithmetic is a stand-in (no residual stream, no softmax); what is measured is the cost of a dependent stage.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>
#include "common.hpp"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

using namespace vx;
typedef unsigned long long u64;
constexpr int NWG = 256, NREP = 8;
constexpr int SPIN_MAX = 1 << 17;

// granule vectors (per replica): offsets in granules
constexpr int V_XF = 0, V_QKV = 1024, V_PART = V_QKV + 3072, V_A = V_PART + 2048, V_XD = V_A + 1024, V_H = V_XD + 1024, V_END = V_H + 2048;
// weight offsets (elements) inside one layer
constexpr size_t W_A = 0, W_B = W_A + (size_t)3072 * 1024, W_D = W_B + (size_t)2048 * 256, W_E = W_D + (size_t)1024 * 1024,
                 W_F = W_E + (size_t)4096 * 1024, W_LAYER = W_F + (size_t)1024 * 4096;

struct Ctl {
  unsigned count[8];
  unsigned abort_;
  unsigned pad[7];
  u64 t_tok[64];      // workgroup 0: time at the start of each token (and one past the end)
  u64 t_stage[16];    // workgroup 0, last token, first layer: time at the start of each stage
  u64 t_g[6][NWG];    // last token, layer 1: gather of stage i done (every workgroup's gathering wave)
  u64 t_b[6][NWG];    // ... consumer wave 0 past the barrier of stage i
  u64 t_w[6][NWG];    // ... its weights landed (dot products done)
  u64 t_p[6][NWG];    // ... its publish stores issued
};

__device__ __forceinline__ unsigned xcc_id() { unsigned v; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v)); return v & 7; }
__device__ __forceinline__ void st8_sc1(u64* p, u64 v) { asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory"); }
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void ld16_sc1(u32x4& v, const void* p) { asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory"); }

// ---- gathering wave: NL 16-byte loads per lane = 2 NL granules; lane l, load j holds granules (j*64 + l)*2 + {0,1}
template <int NL> __device__ __forceinline__ bool gather(const u64* src, unsigned tag, u32x4 (&g)[NL], int lane, int nvalid, Ctl* c) {
  for (int s = 0;; ++s) {
#pragma unroll
    for (int j = 0; j < NL; ++j) ld16_sc1(g[j], src + (size_t)min((j * 64 + lane) * 2, nvalid - 2));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int j = 0; j < NL; ++j) asm volatile("" : "+v"(g[j]));  // uses of g stay behind the wait (volatile asms keep their order)
    bool ok = true;
#pragma unroll
    for (int j = 0; j < NL; ++j) ok = ok && g[j].y == tag && g[j].w == tag;
    if (__ballot(!ok) == 0ull) return true;
    if (s > SPIN_MAX) { atomicExch(&c->abort_, 5u); return false; }
    if ((s & 63) == 63 && __hip_atomic_load(&c->abort_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return false;
  }
}

// publish RPW wave-uniform results as fp32 granules into all replicas: lane = r*8 + rep
template <int RPW> __device__ __forceinline__ void publish_f32(u64* gb, int vec_off, int row0, const float (&acc)[RPW], unsigned tag, int lane) {
  if (lane < RPW * NREP) {
    const int r = lane >> 3, rep = lane & 7;
    // two-level select (a chain of `r == i` selects is turned into an indexed read of a stack array)
    const float a1 = RPW > 1 ? acc[RPW > 1 ? 1 : 0] : acc[0], a2 = RPW > 2 ? acc[RPW > 2 ? 2 : 0] : acc[0], a3 = RPW > 3 ? acc[RPW > 3 ? 3 : 0] : a2;
    const float v01 = (r & 1) ? a1 : acc[0], v23 = (r & 1) ? a3 : a2;
    const float v = (r & 2) ? v23 : v01;
    st8_sc1(gb + (size_t)rep * V_END + vec_off + row0 + r, ((u64)tag << 32) | (u64)__float_as_uint(v));
  }
}
// bf16 pairs: rows (row0 + 2i, row0 + 2i + 1) -> granule (row0/2 + i)
template <int RPW> __device__ __forceinline__ void publish_bf16x2(u64* gb, int vec_off, int row0, const float (&acc)[RPW], unsigned tag, int lane) {
  static_assert(RPW % 2 == 0, "pairs");
  if (lane < (RPW / 2) * NREP) {
    const int r = lane >> 3, rep = lane & 7;
    float lo = acc[0], hi = acc[1];
#pragma unroll
    for (int i = 1; i < RPW / 2; ++i) { lo = (r == i) ? acc[2 * i] : lo; hi = (r == i) ? acc[2 * i + 1] : hi; }
    const unsigned pk = (__float_as_uint(lo) >> 16) | (__float_as_uint(hi) & 0xffff0000u);  // truncation is fine for a probe
    st8_sc1(gb + (size_t)rep * V_END + vec_off + (row0 >> 1) + r, ((u64)tag << 32) | (u64)pk);
  }
}

template <int KCH, int RPW> __device__ __forceinline__ void issue_w(uint4 (&w)[RPW * KCH], const bf16* W, int row0, int K, int lane) {
#pragma unroll
  for (int r = 0; r < RPW; ++r)
#pragma unroll
    for (int c = 0; c < KCH; ++c) w[r * KCH + c] = ld16(W + (size_t)(row0 + r) * K + min((c * 64 + lane) * 8, K - 8));
}
// dot of RPW rows with the x vector in LDS (fp32); lanes past K/8 contribute zero
template <int KCH, int RPW> __device__ __forceinline__ void dot_rows(const uint4 (&w)[RPW * KCH], const float* xs, int K, int lane, float (&acc)[RPW]) {
  float s[RPW];
#pragma unroll
  for (int r = 0; r < RPW; ++r) s[r] = 0.f;
#pragma unroll
  for (int c = 0; c < KCH; ++c) {
    const int k = (c * 64 + lane) * 8;
    const bool okk = k < K;
    const float4 a = *reinterpret_cast<const float4*>(xs + min(k, K - 8)), b = *reinterpret_cast<const float4*>(xs + min(k, K - 8) + 4);
    float xr[8];
    xr[0] = okk ? a.x : 0.f; xr[1] = okk ? a.y : 0.f; xr[2] = okk ? a.z : 0.f; xr[3] = okk ? a.w : 0.f;
    xr[4] = okk ? b.x : 0.f; xr[5] = okk ? b.y : 0.f; xr[6] = okk ? b.z : 0.f; xr[7] = okk ? b.w : 0.f;
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
      float wf[8];
      unpack<bf16>(w[r * KCH + c], wf);
#pragma unroll
      for (int j = 0; j < 8; ++j) s[r] = fmaf(wf[j], xr[j], s[r]);
    }
  }
#pragma unroll
  for (int r = 0; r < RPW; ++r) acc[r] = wave_sum_dpp(s[r]);
}

// gathering wave: unpack NL loads of fp32 granules into LDS (optionally LayerNorm'ed)
template <int NL, bool LN> __device__ __forceinline__ void stage_in_f32(const u32x4 (&g)[NL], float* xs, int lane, int K) {
  float v[NL][2];
#pragma unroll
  for (int j = 0; j < NL; ++j) { v[j][0] = __uint_as_float(g[j].x); v[j][1] = __uint_as_float(g[j].z); }
  if (LN) {
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < NL; ++j) s += v[j][0] + v[j][1];
    const float mean = wave_sum_dpp(s) / (float)K;
    float ss = 0.f;
#pragma unroll
    for (int j = 0; j < NL; ++j) { const float a = v[j][0] - mean, b = v[j][1] - mean; ss += a * a + b * b; }
    const float rstd = 1.0f / sqrtf(wave_sum_dpp(ss) / (float)K + 1e-5f);
#pragma unroll
    for (int j = 0; j < NL; ++j) { v[j][0] = (v[j][0] - mean) * rstd; v[j][1] = (v[j][1] - mean) * rstd; }
  }
#pragma unroll
  for (int j = 0; j < NL; ++j) {
    const int k = (j * 64 + lane) * 2;
    if (k < K) *reinterpret_cast<float2*>(xs + k) = make_float2(v[j][0], v[j][1]);
  }
}
template <int NL> __device__ __forceinline__ void stage_in_bf16x2(const u32x4 (&g)[NL], float* xs, int lane) {
#pragma unroll
  for (int j = 0; j < NL; ++j) {
    const int k = (j * 64 + lane) * 4;  // 2 granules = 4 values
    *reinterpret_cast<float4*>(xs + k) = make_float4(__uint_as_float(g[j].x << 16), __uint_as_float(g[j].x & 0xffff0000u),
                                                     __uint_as_float(g[j].z << 16), __uint_as_float(g[j].z & 0xffff0000u));
  }
}

template <bool late>
__global__ __launch_bounds__(320) void chain_kernel(Ctl* c, u64* gb, const bf16* __restrict__ Wt, int L, int T, unsigned base_tag) {
  __shared__ __attribute__((aligned(16))) float xs[2][4096];
  __shared__ unsigned s_x, s_i, s_abort;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) {
    s_x = xcc_id();
    s_i = atomicAdd(&c->count[s_x], 1u);
    s_abort = 0;
    if (s_i >= 32) atomicExch(&c->abort_, 1u);
  }
  __syncthreads();
  const int b = (int)(s_x * 32 + s_i);  // role: workgroup b of the step (XCD-major)
  const u64* rep = gb + (size_t)s_x * V_END;
  if (tid == 0) {  // everyone has a role before anyone spins on anyone
    for (int s = 0;; ++s) {
      unsigned tot = 0;
      for (int x = 0; x < 8; ++x) tot += __hip_atomic_load(&c->count[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (tot == NWG) break;
      if (s > SPIN_MAX) { atomicExch(&c->abort_, 2u); break; }
      __builtin_amdgcn_s_sleep(4);
    }
  }
  __syncthreads();
  if (__hip_atomic_load(&c->abort_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;
  const bool gw = wave == 4;      // the gathering wave
  const bool actB = b < 128, actC = b < 16;
  unsigned tag = base_tag;
  int buf = 0;
  auto bar = [&]() -> bool { __syncthreads(); return s_abort != 0; };
#define STAMP_STAGE(i) if (b == 0 && lane == 0 && t == T - 1 && l == 0) c->t_stage[i] = __builtin_amdgcn_s_memrealtime();
#define STAMP_WG(arr, i) if (lane == 0 && t == T - 1 && l == 1) c->arr[i][b] = __builtin_amdgcn_s_memrealtime();
#define STAMP_W0(arr, i) if (wave == 0 && lane == 0 && t == T - 1 && l == 1) c->arr[i][b] = __builtin_amdgcn_s_memrealtime();

  // The two roles run their own loops (same sequence of workgroup barriers): the gathering wave holds granules, the consumer
  // waves hold weights; in one shared loop both sets were live everywhere and the kernel spilled.
  if (gw) {
    for (int t = 0; t < T; ++t) {
      if (b == 0 && lane == 0) c->t_tok[t] = __builtin_amdgcn_s_memrealtime();
      for (int l = 0; l < L; ++l) {
        STAMP_STAGE(0)
        { u32x4 g[8]; if (!gather<8>(rep + V_XF, tag, g, lane, 1024, c)) s_abort = 1; STAMP_WG(t_g, 0) stage_in_f32<8, true>(g, xs[buf], lane, 1024); }
        if (bar()) return;
        ++tag; buf ^= 1;
        STAMP_STAGE(1)
        if (actB) {
          { u32x4 g[2]; if (!gather<2>(rep + V_QKV + (b & 7) * 256, tag, g, lane, 256, c)) s_abort = 1; STAMP_WG(t_g, 1) stage_in_f32<2, false>(g, xs[buf], lane, 256); }
          if (bar()) return;
          buf ^= 1;
        }
        ++tag;
        STAMP_STAGE(2)
        if (actC) {
          { u32x4 g[1]; if (!gather<1>(rep + V_PART + b * 128, tag, g, lane, 128, c)) s_abort = 1; STAMP_WG(t_g, 2) stage_in_f32<1, false>(g, xs[buf], lane, 128); }
          if (bar()) return;
          buf ^= 1;
        }
        ++tag;
        STAMP_STAGE(3)
        { u32x4 g[8]; if (!gather<8>(rep + V_A, tag, g, lane, 1024, c)) s_abort = 1; STAMP_WG(t_g, 3) stage_in_f32<8, false>(g, xs[buf], lane, 1024); }
        if (bar()) return;
        ++tag; buf ^= 1;
        STAMP_STAGE(4)
        { u32x4 g[8]; if (!gather<8>(rep + V_XD, tag, g, lane, 1024, c)) s_abort = 1; STAMP_WG(t_g, 4) stage_in_f32<8, true>(g, xs[buf], lane, 1024); }
        if (bar()) return;
        ++tag; buf ^= 1;
        STAMP_STAGE(5)
        { u32x4 g[16]; if (!gather<16>(rep + V_H, tag, g, lane, 2048, c)) s_abort = 1; STAMP_WG(t_g, 5) stage_in_bf16x2<16>(g, xs[buf], lane); }
        if (bar()) return;
        ++tag; buf ^= 1;
        STAMP_STAGE(6)
      }
    }
    if (b == 0 && lane == 0) c->t_tok[T] = __builtin_amdgcn_s_memrealtime();
    return;
  }

  // ---- consumer waves.  Weight register sets: A and E use w0, B and F use w1, D uses w3 (consecutive active stages of a
  // workgroup never share a set, so the next stage's weights load while the current stage waits for its input).
  uint4 w0[8], w1[8], w3[2];
  { float a0[1] = {0.01f * (float)((b * 4 + wave) % 37)}; publish_f32<1>(gb, V_XF, b * 4 + wave, a0, tag, lane); }  // token 0's input
  issue_w<2, 3>(reinterpret_cast<uint4(&)[6]>(w0), Wt + W_A, b * 12 + wave * 3, 1024, lane);
  for (int t = 0; t < T; ++t) {
    for (int l = 0; l < L; ++l) {
      const bf16* Wl = Wt + (size_t)l * W_LAYER;
      const bf16* Wn = Wt + (size_t)((l + 1) % L) * W_LAYER;  // next layer (wraps into the next token)
      // The next active stage's weights are requested EARLY (before this stage's barrier: two stages' worth can be in flight in
      // front of the gathering wave's polls) or LATE (after this stage's publish: one stage's worth, landing during the hop).
      // A: QKV
      if (!late) { if (actB) issue_w<1, 4>(reinterpret_cast<uint4(&)[4]>(w1), Wl + W_B, b * 16 + wave * 4, 256, lane); else issue_w<2, 1>(w3, Wl + W_D, b * 4 + wave, 1024, lane); }
      if (bar()) return;
      STAMP_W0(t_b, 0)
      ++tag;
      { float acc[3]; dot_rows<2, 3>(reinterpret_cast<uint4(&)[6]>(w0), xs[buf], 1024, lane, acc); STAMP_W0(t_w, 0) publish_f32<3>(gb, V_QKV, b * 12 + wave * 3, acc, tag, lane); STAMP_W0(t_p, 0) }
      if (late) { if (actB) issue_w<1, 4>(reinterpret_cast<uint4(&)[4]>(w1), Wl + W_B, b * 16 + wave * 4, 256, lane); else issue_w<2, 1>(w3, Wl + W_D, b * 4 + wave, 1024, lane); }
      buf ^= 1;
      // B: attention stand-in
      if (actB) {
        if (!late && !actC) issue_w<2, 1>(w3, Wl + W_D, b * 4 + wave, 1024, lane);
        if (bar()) return;
      }
      ++tag;
      if (actB) {
        STAMP_W0(t_b, 1) float acc[4]; dot_rows<1, 4>(reinterpret_cast<uint4(&)[4]>(w1), xs[buf], 256, lane, acc); STAMP_W0(t_w, 1) publish_f32<4>(gb, V_PART, b * 16 + wave * 4, acc, tag, lane); STAMP_W0(t_p, 1)
        if (late && !actC) issue_w<2, 1>(w3, Wl + W_D, b * 4 + wave, 1024, lane);
        buf ^= 1;
      }
      // C: combine stand-in
      if (actC) {
        if (!late) issue_w<2, 1>(w3, Wl + W_D, b * 4 + wave, 1024, lane);
        if (bar()) return;
      }
      ++tag;
      if (actC) {  // 64 outputs per workgroup, 16 per wave: lanes 0..15 own one each, 8 replica stores per lane
        if (lane < 16) {
          const int row = b * 64 + wave * 16 + lane;
          const float v = 0.5f * (xs[buf][(wave * 16 + lane) & 127] + xs[buf][(wave * 16 + lane + 64) & 127]);
#pragma unroll
          for (int r = 0; r < NREP; ++r) st8_sc1(gb + (size_t)r * V_END + V_A + row, ((u64)tag << 32) | (u64)__float_as_uint(v));
        }
        if (late) issue_w<2, 1>(w3, Wl + W_D, b * 4 + wave, 1024, lane);
        buf ^= 1;
      }
      // D: out-projection
      if (!late) issue_w<2, 4>(w0, Wl + W_E, b * 16 + wave * 4, 1024, lane);
      if (bar()) return;
      ++tag;
      { STAMP_W0(t_b, 3) float acc[1]; dot_rows<2, 1>(w3, xs[buf], 1024, lane, acc); STAMP_W0(t_w, 3) publish_f32<1>(gb, V_XD, b * 4 + wave, acc, tag, lane); STAMP_W0(t_p, 3) }
      if (late) issue_w<2, 4>(w0, Wl + W_E, b * 16 + wave * 4, 1024, lane);
      buf ^= 1;
      // E: FFN1
      if (!late) issue_w<8, 1>(w1, Wl + W_F, b * 4 + wave, 4096, lane);
      if (bar()) return;
      ++tag;
      {
        STAMP_W0(t_b, 4) float acc[4]; dot_rows<2, 4>(w0, xs[buf], 1024, lane, acc); STAMP_W0(t_w, 4)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[r] = fmaxf(acc[r], 0.f) + 0.01f;
        publish_bf16x2<4>(gb, V_H, b * 16 + wave * 4, acc, tag, lane); STAMP_W0(t_p, 4)
      }
      if (late) issue_w<8, 1>(w1, Wl + W_F, b * 4 + wave, 4096, lane);
      buf ^= 1;
      // F: FFN2
      if (!late) issue_w<2, 3>(reinterpret_cast<uint4(&)[6]>(w0), Wn + W_A, b * 12 + wave * 3, 1024, lane);
      if (bar()) return;
      ++tag;
      {
        STAMP_W0(t_b, 5) float acc[1]; dot_rows<8, 1>(w1, xs[buf], 4096, lane, acc); STAMP_W0(t_w, 5)
        acc[0] = acc[0] * 0.05f + 0.01f * (float)((b * 4 + wave) % 37);
        publish_f32<1>(gb, V_XF, b * 4 + wave, acc, tag, lane); STAMP_W0(t_p, 5)
      }
      if (late) issue_w<2, 3>(reinterpret_cast<uint4(&)[6]>(w0), Wn + W_A, b * 12 + wave * 3, 1024, lane);
      buf ^= 1;
    }
  }
}

int main(int argc, char** argv) {
  const int L = argc > 1 ? atoi(argv[1]) : 12, T = argc > 2 ? atoi(argv[2]) : 24;
  const bool late = argc > 3 && atoi(argv[3]) != 0;
  printf("weights requested %s\n", late ? "LATE (after the publish)" : "EARLY (before the barrier)");
  CK(hipSetDevice(0));
  Ctl* dc; u64* gb; bf16* W;
  const size_t wel = (size_t)L * W_LAYER;
  CK(hipMalloc(&dc, sizeof(Ctl)));
  CK(hipMalloc(&gb, (size_t)NREP * V_END * sizeof(u64)));
  CK(hipMalloc(&W, wel * 2));
  {  // small pseudo-random bf16 weights
    std::vector<unsigned short> h(wel);
    unsigned s = 12345u;
    for (size_t i = 0; i < wel; ++i) { s = s * 1664525u + 1013904223u; const float f = ((float)(s >> 8) / 16777216.0f - 0.5f) * 0.06f; unsigned u; memcpy(&u, &f, 4); h[i] = (unsigned short)(u >> 16); }
    CK(hipMemcpy(W, h.data(), wel * 2, hipMemcpyHostToDevice));
  }
  Ctl* hcp = (Ctl*)malloc(sizeof(Ctl));
  Ctl& hc = *hcp;
  unsigned base = 1;
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipMemset(dc, 0, sizeof(Ctl)));
    if (rep == 0) CK(hipMemset(gb, 0, (size_t)NREP * V_END * sizeof(u64)));
    CK(hipDeviceSynchronize());
    if (late) chain_kernel<true><<<NWG, 320>>>(dc, gb, W, L, T, base); else chain_kernel<false><<<NWG, 320>>>(dc, gb, W, L, T, base);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(&hc, dc, sizeof(Ctl), hipMemcpyDeviceToHost));
    base += (unsigned)(T * L * 6 + 8);
    if (hc.abort_) { printf("ABORT code %u (counts %u %u %u %u %u %u %u %u)\n", hc.abort_, hc.count[0], hc.count[1], hc.count[2], hc.count[3], hc.count[4], hc.count[5], hc.count[6], hc.count[7]); return 1; }
    const double tot = (double)(hc.t_tok[T] - hc.t_tok[1]) * 0.01 / (T - 1);
    printf("run %d: L=%d  %.1f us per token (first token %.1f us) = %.2f us per layer, %.3f us per stage; weights %.1f MB/token -> %.2f TB/s\n", rep, L, tot,
           (double)(hc.t_tok[1] - hc.t_tok[0]) * 0.01, tot / L, tot / (L * 6), (double)wel * 2 / 1e6, (double)wel * 2 / tot / 1e6);
    printf("   last token, layer 0, stage spans (A QKV, B attn, C combine, D out, E FFN1, F FFN2):");
    for (int i = 0; i < 6; ++i) printf(" %.2f", (double)(hc.t_stage[i + 1] - hc.t_stage[i]) * 0.01);
    printf(" us\n");
    if (rep == 2) {
      const char* nm[6] = {"A QKV", "B attn", "C comb", "D out", "E FFN1", "F FFN2"};
      const int G[6] = {256, 128, 16, 256, 256, 256};
      u64 t0 = ~0ull;
      for (int i = 0; i < 6; ++i) for (int w = 0; w < G[i]; ++w) if (hc.t_g[i][w] && hc.t_g[i][w] < t0) t0 = hc.t_g[i][w];
      printf("   last token, layer 1, all workgroups, us since the first stamp (min / median / max over the active workgroups):\n");
      for (int i = 0; i < 6; ++i) {
        auto stat = [&](u64 (*arr)[NWG], const char* what) {
          std::vector<double> v;
          for (int w = 0; w < G[i]; ++w) if (arr[i][w]) v.push_back((double)(arr[i][w] - t0) * 0.01);
          if (v.empty()) { printf("      %-7s %-12s -\n", nm[i], what); return; }
          std::sort(v.begin(), v.end());
          printf("      %-7s %-12s %6.2f %6.2f %6.2f\n", nm[i], what, v.front(), v[v.size() / 2], v.back());
        };
        stat(hc.t_g, "gathered"); stat(hc.t_b, "past barrier"); stat(hc.t_w, "dot done"); stat(hc.t_p, "published");
      }
    }
  }
  return 0;
}
