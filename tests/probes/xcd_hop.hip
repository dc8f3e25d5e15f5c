// Probe (not product): what does a dependent hand-over between workgroups cost when it stays inside ONE XCD's L2,
// against the device-scope (sc1, memory-side) hand-over every cross-XCD edge needs?
//
//   hipcc --offload-arch=gfx950 -O3 -o tests/probes/xcd_hop tests/probes/xcd_hop.hip && tests/probes/xcd_hop
//
// 256 workgroups (one per CU) read their XCC id from the hardware register and take a local index inside their XCD from
// an atomic counter; nothing depends on the dispatcher's placement.  Every spin is bounded and watches a global abort word.
//   chain     one token walks around the 32 workgroups of an XCD (1-to-1 hop latency), or around all 256
//   gather    every workgroup of a group publishes its share of 1024 tagged 8-byte granules {value, tag}, then gathers all
//             of them (one quarter per wave, staged in LDS); round r+1's values depend on round r's
// Modes: L2 = plain stores + sc0 loads (miss the CU's L1, may hit the XCD's L2); SC1 = sc1 stores (write-through) + sc1 loads.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef unsigned long long u64;
constexpr int NWG = 256;
constexpr int SPIN_MAX = 1 << 19;

enum { MODE_L2 = 0 /* plain store, sc0 load */, MODE_SC1 = 1 /* sc1 store, sc1 load */, MODE_P_SC1 = 2 /* plain store, sc1 load */,
       MODE_P_INV = 3 /* plain store, buffer_inv sc0 + plain load */, MODE_P_ATOM = 4 /* plain store, L2 atomic-or poll */, MODE_SC1_ATOM = 5 };

__device__ __forceinline__ unsigned xcc_id() {
  unsigned v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return v & 0xf;
}
__device__ __forceinline__ unsigned hw_id() {
  unsigned v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(v));
  return v;
}

template <int MODE> __device__ __forceinline__ void st8(u64* p, u64 v) {
  if (MODE == MODE_SC1 || MODE == MODE_SC1_ATOM) asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
  else asm volatile("global_store_dwordx2 %0, %1, off" ::"v"(p), "v"(v) : "memory");
}
template <int MODE> __device__ __forceinline__ void ld8_issue(u64& v, const u64* p) {
  if (MODE == MODE_L2) asm volatile("global_load_dwordx2 %0, %1, off sc0" : "=v"(v) : "v"(p) : "memory");
  else if (MODE == MODE_P_INV) asm volatile("buffer_inv sc0\n\tglobal_load_dwordx2 %0, %1, off" : "=v"(v) : "v"(p) : "memory");
  else if (MODE == MODE_P_ATOM || MODE == MODE_SC1_ATOM) { u64 z = 0; asm volatile("global_atomic_or_x2 %0, %1, %2, off sc0" : "=v"(v) : "v"(p), "v"(z) : "memory"); }
  else asm volatile("global_load_dwordx2 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
}
__device__ __forceinline__ void wait1(u64& a) { asm volatile("s_waitcnt vmcnt(0)" : "+v"(a)::"memory"); }
__device__ __forceinline__ void wait4(u64& a, u64& b, u64& c, u64& d) {
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d)::"memory");
}

struct Ctl {
  unsigned count[8];     // arrivals per XCD
  unsigned abort_;       // any bounded spin that ran out sets it
  unsigned pad[7];
  unsigned xcc_of[NWG];  // per workgroup: xcc id, local index, HW_ID
  unsigned idx_of[NWG];
  unsigned hw_of[NWG];
  u64 t[NWG][2];         // start / end stamps per workgroup (s_memrealtime, 100 MHz)
};

// role assignment: (xcc, local index).  Returns false when an XCD received more than `per` workgroups.
__device__ __forceinline__ bool take_role(Ctl* c, unsigned& xcc, unsigned& idx, unsigned per) {
  __shared__ unsigned s_x, s_i;
  if (threadIdx.x == 0) {
    s_x = xcc_id();
    s_i = atomicAdd(&c->count[s_x & 7], 1u);
    c->xcc_of[blockIdx.x] = s_x; c->idx_of[blockIdx.x] = s_i; c->hw_of[blockIdx.x] = hw_id();
    if (s_i >= per) atomicExch(&c->abort_, 1u);
  }
  __syncthreads();
  xcc = s_x & 7; idx = s_i;
  return idx < per;
}

// ---- chain: a token walks around the ring.  local = ring inside each XCD (8 independent rings of 32), else one ring of 256
// in blockIdx order (neighbours sit on different XCDs under round-robin placement).
template <int MODE>
__global__ __launch_bounds__(320) void chain_kernel(Ctl* c, u64* slots, int rounds, int local, const uint4* bg, size_t bg_n, float* sink) {
  __shared__ volatile int s_done;
  unsigned xcc, idx;
  if (threadIdx.x == 0) s_done = 0;
  if (!take_role(c, xcc, idx, 32)) return;
  if (threadIdx.x >= 64) {  // background: every other wave streams 16-byte loads from a large buffer until wave 0 is done
    uint4 acc = make_uint4(0, 0, 0, 0);
    size_t pos = ((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 1237 * 64 + (threadIdx.x & 63);
    for (int it = 0; it < (1 << 20) && !s_done; ++it) {
      uint4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) { v[u] = bg[pos % bg_n]; pos += 64 * 4099; }
#pragma unroll
      for (int u = 0; u < 8; ++u) { acc.x ^= v[u].x; acc.y += v[u].y; acc.z ^= v[u].z; acc.w += v[u].w; }
    }
    if (acc.x == 0x12345 && acc.y == 77) sink[threadIdx.x] = 1.f;
    return;
  }
  if (threadIdx.x != 0) return;
  const int ring = local ? 32 : NWG;
  const int me = local ? (int)idx : (int)blockIdx.x;
  u64* base = slots + (local ? (size_t)xcc * 512 : 0);  // this ring's slot array
  u64* mine = base + (size_t)me * 16;              // one slot per 128 bytes
  u64* next = base + (size_t)((me + 1) % ring) * 16;
  // wait until every workgroup has a role (so the ring is complete): all counts summed == NWG
  for (int s = 0;; ++s) {
    unsigned tot = 0;
    for (int x = 0; x < 8; ++x) tot += __hip_atomic_load(&c->count[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (tot == NWG) break;
    if (s > SPIN_MAX || __hip_atomic_load(&c->abort_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { atomicExch(&c->abort_, 2u); s_done = 1; return; }
    __builtin_amdgcn_s_sleep(4);
  }
  if (__hip_atomic_load(&c->abort_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { s_done = 1; return; }
  const u64 t0 = __builtin_amdgcn_s_memrealtime();
  for (int r = 0; r < rounds; ++r) {
    const u64 want = (u64)r * ring + me + 1;   // token value when it reaches me in round r
    if (!(r == 0 && me == 0)) {
      u64 v;
      int s = 0;
      for (;;) {
        ld8_issue<MODE>(v, mine);
        wait1(v);
        if (v == want) break;
        if (++s > SPIN_MAX) { atomicExch(&c->abort_, 3u); s_done = 1; return; }
        if ((s & 1023) == 0 && __hip_atomic_load(&c->abort_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { s_done = 1; return; }
      }
    }
    st8<MODE>(next, want + 1);
  }
  // the value handed to `next` is want+1 = r*ring + me + 2 = what (me+1) expects in round r; for me = ring-1 the next is 0 in round
  // r+1 expecting (r+1)*ring + 1 = r*ring + ring + 1 = want + 1 as well
  c->t[blockIdx.x][0] = t0;
  c->t[blockIdx.x][1] = __builtin_amdgcn_s_memrealtime();
  s_done = 1;
}

// ---- gather: groups of G workgroups (G = 32: one XCD; G = 256: the chip).  Each publishes 1024/G granules per round into
// buf[group][round & 1][1024]; every workgroup gathers all 1024 (thread t: granules t, t+256, t+512, t+768).
template <int MODE>
__global__ __launch_bounds__(256) void gather_kernel(Ctl* c, u64* buf, int rounds, int local, float* sink) {
  __shared__ float xs[1024];
  unsigned xcc, idx;
  if (!take_role(c, xcc, idx, 32)) return;
  const int tid = threadIdx.x;
  const int G = local ? 32 : NWG;
  const int me = local ? (int)idx : (int)(xcc * 32 + idx);
  u64* gb = buf + (size_t)(local ? xcc : 0) * 2 * 1024;
  const int share = 1024 / G;
  if (tid == 0) {
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
  float val = 1.0f + 0.001f * me;
  const u64 t0 = __builtin_amdgcn_s_memrealtime();
  for (int r = 0; r < rounds; ++r) {
    u64* rb = gb + (size_t)(r & 1) * 1024;
    if (tid < share) {
      const u64 gr = ((u64)(unsigned)(r + 1) << 32) | (u64)__float_as_uint(val + tid);
      st8<MODE>(rb + me * share + tid, gr);
    }
    u64 g0, g1, g2, g3;
    int s = 0;
    for (;;) {
      ld8_issue<MODE>(g0, rb + tid);
      ld8_issue<MODE>(g1, rb + tid + 256);
      ld8_issue<MODE>(g2, rb + tid + 512);
      ld8_issue<MODE>(g3, rb + tid + 768);
      wait4(g0, g1, g2, g3);
      const bool ok = (unsigned)(g0 >> 32) == (unsigned)(r + 1) && (unsigned)(g1 >> 32) == (unsigned)(r + 1) &&
                      (unsigned)(g2 >> 32) == (unsigned)(r + 1) && (unsigned)(g3 >> 32) == (unsigned)(r + 1);
      if (__ballot(!ok) == 0ull) break;  // wave-uniform exit
      if (++s > SPIN_MAX) { if ((tid & 63) == 0) atomicExch(&c->abort_, 4u); break; }
      if ((s & 255) == 0 && __hip_atomic_load(&c->abort_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
    }
    xs[tid] = __uint_as_float((unsigned)g0); xs[tid + 256] = __uint_as_float((unsigned)g1);
    xs[tid + 512] = __uint_as_float((unsigned)g2); xs[tid + 768] = __uint_as_float((unsigned)g3);
    __syncthreads();
    if (__hip_atomic_load(&c->abort_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;
    // next round's values depend on what was gathered (a few LDS reads; stands in for the dot products)
    val = 0.25f * (xs[(tid * 7) & 1023] + xs[(tid * 13 + 5) & 1023] + xs[(me * 3) & 1023] + xs[1023 - ((tid + me) & 1023)]);
    val = val - floorf(val) + 1.0f;
    __syncthreads();
  }
  if (tid == 0) {
    c->t[blockIdx.x][0] = t0;
    c->t[blockIdx.x][1] = __builtin_amdgcn_s_memrealtime();
    sink[blockIdx.x] = val;
  }
}

static void report(const char* name, Ctl* hc, int rounds, int hops_per_round) {
  if (hc->abort_) { printf("%-44s ABORT code %u (counts", name, hc->abort_); for (int x = 0; x < 8; ++x) printf(" %u", hc->count[x]); printf(")\n"); return; }
  double worst = 0, best = 1e30;
  for (int b = 0; b < NWG; ++b) {
    if (hc->t[b][1] == 0) continue;
    const double us = (double)(hc->t[b][1] - hc->t[b][0]) * 0.01;
    worst = us > worst ? us : worst; best = us < best ? us : best;
  }
  printf("%-44s %8.1f us total  -> %.3f us per %s (fastest workgroup %.1f us)\n", name, worst, worst / ((double)rounds * hops_per_round),
         hops_per_round > 1 ? "hop" : "round", best);
}

int main(int argc, char** argv) {
  const int rounds = argc > 1 ? atoi(argv[1]) : 200;
  CK(hipSetDevice(0));
  Ctl* dc; u64* buf; float* sink; uint4* bg;
  const size_t bg_n = (size_t)1 << 26;  // 1 GiB of 16-byte elements
  CK(hipMalloc(&dc, sizeof(Ctl)));
  CK(hipMalloc(&buf, 8 * 2 * 1024 * sizeof(u64) + NWG * 16 * sizeof(u64)));
  CK(hipMalloc(&sink, 1024 * sizeof(float)));
  CK(hipMalloc(&bg, bg_n * 16));
  CK(hipMemset(bg, 1, bg_n * 16));
  Ctl* hc = (Ctl*)malloc(sizeof(Ctl));
  auto reset = [&]() {
    CK(hipMemset(dc, 0, sizeof(Ctl)));
    CK(hipMemset(buf, 0, 8 * 2 * 1024 * sizeof(u64) + NWG * 16 * sizeof(u64)));
    CK(hipDeviceSynchronize());
  };
  auto fetch = [&]() { CK(hipDeviceSynchronize()); CK(hipMemcpy(hc, dc, sizeof(Ctl), hipMemcpyDeviceToHost)); };

  reset();
  chain_kernel<MODE_SC1><<<NWG, 64>>>(dc, buf, 1, 1, bg, bg_n, sink);
  fetch();
  printf("workgroups per XCD:");
  for (int x = 0; x < 8; ++x) printf(" %u", hc->count[x]);
  int rr = 0;
  for (int b = 0; b < NWG; ++b) rr += (hc->xcc_of[b] & 7) == (unsigned)(b & 7);
  printf("   blockIdx %% 8 == xcc for %d of %d workgroups; abort %u\n", rr, NWG, hc->abort_);

#define CHAIN(MODE, threads, local, name) do { reset(); chain_kernel<MODE><<<NWG, threads>>>(dc, buf, (local) ? rounds : rounds / 4 + 1, local, bg, bg_n, sink); fetch(); \
    report(name, hc, (local) ? rounds : rounds / 4 + 1, (local) ? 32 : 256); } while (0)
  for (int load = 0; load < 2; ++load) {
    const int th = load ? 320 : 64;
    printf("---- 1-to-1 chain, %s\n", load ? "4 waves per CU streaming a 1 GiB buffer meanwhile" : "idle chip");
    CHAIN(MODE_SC1, th, 1, "inside one XCD, sc1 store + sc1 load");
    CHAIN(MODE_SC1, th, 0, "all 256 (cross-XCD), sc1 store + sc1 load");
    CHAIN(MODE_P_SC1, th, 1, "inside one XCD, plain store + sc1 load");
    CHAIN(MODE_P_INV, th, 1, "inside one XCD, plain store + buffer_inv sc0 + plain load");
    CHAIN(MODE_P_ATOM, th, 1, "inside one XCD, plain store + L2 atomic poll");
    CHAIN(MODE_SC1_ATOM, th, 0, "all 256 (cross-XCD), sc1 store + atomic poll (no sc1)");
  }
  if (argc > 2) {
    printf("---- gather of 8 KB (idle chip)\n");
    reset(); gather_kernel<MODE_SC1><<<NWG, 256>>>(dc, buf, rounds, 1, sink); fetch();
    report("gather 8 KB, 32 WGs of one XCD, sc1 + sc1", hc, rounds, 1);
    reset(); gather_kernel<MODE_P_INV><<<NWG, 256>>>(dc, buf, rounds, 1, sink); fetch();
    report("gather 8 KB, 32 WGs of one XCD, plain + inv sc0", hc, rounds, 1);
    reset(); gather_kernel<MODE_P_ATOM><<<NWG, 256>>>(dc, buf, rounds, 1, sink); fetch();
    report("gather 8 KB, 32 WGs of one XCD, plain + L2 atomic", hc, rounds, 1);
    reset(); gather_kernel<MODE_SC1><<<NWG, 256>>>(dc, buf, rounds, 0, sink); fetch();
    report("gather 8 KB, all 256 WGs, sc1 + sc1", hc, rounds, 1);
  }
  return 0;
}
