// Tagged-granule hand-over between workgroups INSIDE a launch (cdna_hip_programming.md, decode rows): every value travels as one
// 8-byte {value, tag} word written with an sc1 (write-through) store and gathered with sweeps of sc1 loads; tag = a step counter
// the sampling launch bumps once per step, so no flag, fence or ordering between stores is needed and a stale word can never
// match.  Valid across XCDs; among the workgroups of ONE XCD a hop costs 0.35-0.6 us in situ (profiles/r03_notes.md).  Every spin
// is bounded (FQ_SPIN_MAX sweeps, then an error word is set and the wave goes on).  Users: ar_tp.hpp.
// (Round 3 first built a fused QKV + attention + combine launch on this transport, 16 workgroups per head; it measured the hop
// but was slower than the launches it replaced - 277 vs 236 us per token, profiles/r03_fused_qkv_attn_stamps.json - and is gone.)
#pragma once
#include "ar_kernels.hpp"

namespace vx {

constexpr int FQ_G = 16;             // workgroups per head = key splits
constexpr int FQ_QKV = 192;          // granules a head exchanges after the projection: q[64], k_new[64], v_new[64]
constexpr int FQ_PART = 66;          // granules of one split's partial softmax: o[64], m, l
constexpr int FQ_SPIN_MAX = 1 << 18; // polls before a wave gives up (~0.3 s)

typedef unsigned long long fq_gran;

#ifdef VX_STAMPS
// stamps build: buffer of vx_debug_fqstamps (ar_tp.hpp TP_STAMP)
}  // namespace vx
extern __device__ unsigned long long* g_fq_stamps;
namespace vx {
#endif

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

}  // namespace vx
