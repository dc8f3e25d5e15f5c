// Probe for a persistent (single-launch) AR decode step: a chain of dependent GEMV stages inside ONE kernel,
// separated by a device-wide barrier instead of kernel boundaries.  Each workgroup owns a fixed slice of every
// stage's weights, issues the loads for the NEXT stage before it waits at the barrier (the weights do not depend
// on the previous stage), then reads the 1024-float activation vector the other workgroups just wrote.
//
// This answers one question with a measurement: what does one dependent stage cost when the kernel boundary is
// replaced by a barrier + prefetch?  (A launch boundary costs 1.65 us in a hipGraph, and the AR step's 62 kernels
// average 4.1 us each)
//
// Every spin is bounded and checks a shared error flag, so the grid always drains even if workgroups were not
// co-resident.
#pragma once
#include "common.hpp"

namespace vx {

struct ChainArgs {
  const bf16* W;      // [stages][nwg][rows][1024]
  float* xbuf;        // [2][nwg*rows]  (ping-pong; a stage reads the first 1024 of its input half)
  unsigned* ctr;      // monotonically increasing arrival counter
  unsigned* err;      // set when a spin ran out
  int stages, rows, mode;
};

enum { CHAIN_BARRIER_ONLY = 0, CHAIN_FENCE = 1, CHAIN_BYPASS = 2, CHAIN_GROUP8 = 3, CHAIN_XCD = 4 };  // XCD: as GROUP8, but a group = the workgroups with equal blockIdx % 8 (one XCD under round-robin placement)  // GROUP8: BYPASS data path, barrier among 8 neighbours only (timing probe: results unchecked)

template <int ROWS, int MODE>
__global__ __launch_bounds__(256) void chain_kernel(ChainArgs a) {
  constexpr int NI = ROWS / 2;  // 16-byte chunks per thread: chunk c = t + 256 i -> row c/128, cols 8 (c%128)
  const int t = threadIdx.x, wg = blockIdx.x, nwg = gridDim.x;
  const int wave = t >> 6, lane = t & 63;
  __shared__ float red[4][NI];
  const size_t slice = (size_t)ROWS * 1024;
  const int nout = nwg * ROWS;
  constexpr bool GROUP = MODE == CHAIN_GROUP8 || MODE == CHAIN_XCD;
  unsigned* ctr = MODE == CHAIN_XCD ? a.ctr + 64 * (1 + (wg & 7)) : GROUP ? a.ctr + 64 * (1 + wg / 8) : a.ctr;  // one counter per 256-byte line
  const unsigned arrivals = MODE == CHAIN_XCD ? (unsigned)(nwg / 8) : GROUP ? 8u : (unsigned)nwg;
  for (int s = 0; s < a.stages; ++s) {
    uint4 w[NI];
    if (MODE != CHAIN_BARRIER_ONLY) {
      const bf16* wp = a.W + ((size_t)s * nwg + wg) * slice;
#pragma unroll
      for (int i = 0; i < NI; ++i) w[i] = ld16(wp + (size_t)(t + 256 * i) * 8);
      __builtin_amdgcn_sched_barrier(0);
    }
    // ---- wait until every workgroup has finished stage s-1
    if (s > 0) {
      if (t == 0) {
        const unsigned target = (unsigned)s * arrivals;
        int spins = 0;
        bool ok = false;
        while (spins < (1 << 20)) {
          unsigned v = (MODE == CHAIN_FENCE) ? __hip_atomic_load(ctr, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT)
                                             : __hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (v >= target) { ok = true; break; }
          if ((++spins & 255) == 0 && __hip_atomic_load(a.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
          __builtin_amdgcn_s_sleep(1);
        }
        if (!ok) __hip_atomic_store(a.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      __syncthreads();
    }
    if (MODE != CHAIN_BARRIER_ONLY) {
      const float* xin = a.xbuf + (size_t)(s & 1) * nout + 8 * (t & 127);
      float x[8];
      if (MODE == CHAIN_FENCE) {
        float4 x0 = *(const float4*)xin, x1 = *(const float4*)(xin + 4);
        x[0] = x0.x; x[1] = x0.y; x[2] = x0.z; x[3] = x0.w; x[4] = x1.x; x[5] = x1.y; x[6] = x1.z; x[7] = x1.w;
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = __hip_atomic_load(xin + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      float acc[NI];
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        float f[8];
        unpack<bf16>(w[i], f);
        float v = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) v += f[j] * x[j];
        acc[i] = wave_sum_dpp(v);
      }
      if (lane == 0) {
#pragma unroll
        for (int i = 0; i < NI; ++i) red[wave][i] = acc[i];
      }
      __syncthreads();
      if (t < ROWS) {
        const int i = t >> 1, h = t & 1;  // row t = 2 i + h; h selects threads 128..255 (waves 2,3)
        const float y = red[2 * h][i] + red[2 * h + 1][i];
        float* yo = a.xbuf + (size_t)((s + 1) & 1) * nout + (size_t)wg * ROWS + t;
        if (MODE == CHAIN_FENCE) *yo = y;
        else __hip_atomic_store(yo, y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    // ---- arrive
    if (MODE == CHAIN_BYPASS || GROUP) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (t == 0) {
      if (MODE == CHAIN_FENCE) __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      else __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

}  // namespace vx

// ---- probe: L2 -> CU fill rate.  Every workgroup streams a small, L2-resident window (all workgroups share one
// `region_bytes` region, so after the first touch everything is served by the XCD's L2) with U independent 16-byte
// loads in flight per lane.  Answers: how many bytes per clock can one CU pull from L2, and does it scale with
// the number of resident waves or with the loads in flight per wave?  (The GEMM tiles are sized against this.)
namespace vx {
template <int U>
__global__ __launch_bounds__(256) void l2_fill_kernel(const uint4* __restrict__ buf, size_t region_vec, int iters,
                                                      unsigned* __restrict__ sink) {
  const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  uint4 acc = make_uint4(0u, 0u, 0u, 0u);
  size_t p = tid % region_vec;
  const size_t stride = ((size_t)gridDim.x * blockDim.x) % region_vec;  // < region_vec: one conditional subtraction keeps p in range
  for (int it = 0; it < iters; ++it) {
    uint4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      v[u] = buf[p];
      p += stride;
      if (p >= region_vec) p -= region_vec;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) { acc.x ^= v[u].x; acc.y ^= v[u].y; acc.z ^= v[u].z; acc.w ^= v[u].w; }
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = acc.x;  // keeps the loads alive
}
}  // namespace vx
