// Host side of the measurement probes (vx_debug_*, declared in probes.h).  Compiled only into the probe builds
// (`build.py --probes` -> libvallex_probes.so, `--stamps` -> libvallex_stamps.so): the product library carries none of it.
// Included at the end of engine.hip, inside its translation unit (uses its HIPC / fail helpers).
#pragma once
#include "persist_probe.hpp"
#include "probes.h"

// ------------------------------------------------------------------------------ measurement aid
__global__ void noop_kernel(float* p) {
  if (p != nullptr && threadIdx.x == 0 && blockIdx.x == 0x7fffffff) p[0] = 0.f;
}

// Launch floor of this box: time of a dependent chain of n trivial kernels (grid x block), replayed
// `iters` times as a hipGraph and launched eagerly.  out[0] = us per kernel (graph), out[1] = eager.
extern "C" int vx_debug_launch_floor(int32_t n_kernels, int32_t grid, int32_t block, int32_t iters, double* out) {
  if (!out || n_kernels <= 0 || iters <= 0) return fail(VX_ERR_ARG, "bad argument");
  hipStream_t s;
  HIPC(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  hipEvent_t e0, e1;
  HIPC(hipEventCreate(&e0));
  HIPC(hipEventCreate(&e1));
  hipGraph_t g;
  hipGraphExec_t ge;
  HIPC(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
  for (int i = 0; i < n_kernels; ++i) noop_kernel<<<grid, block, 0, s>>>(nullptr);
  HIPC(hipStreamEndCapture(s, &g));
  HIPC(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  for (int i = 0; i < 3; ++i) HIPC(hipGraphLaunch(ge, s));
  HIPC(hipStreamSynchronize(s));
  float ms = 0.f;
  HIPC(hipEventRecord(e0, s));
  for (int i = 0; i < iters; ++i) HIPC(hipGraphLaunch(ge, s));
  HIPC(hipEventRecord(e1, s));
  HIPC(hipStreamSynchronize(s));
  HIPC(hipEventElapsedTime(&ms, e0, e1));
  out[0] = (double)ms * 1e3 / ((double)iters * n_kernels);
  HIPC(hipEventRecord(e0, s));
  for (int i = 0; i < iters; ++i)
    for (int k = 0; k < n_kernels; ++k) noop_kernel<<<grid, block, 0, s>>>(nullptr);
  HIPC(hipEventRecord(e1, s));
  HIPC(hipStreamSynchronize(s));
  HIPC(hipEventElapsedTime(&ms, e0, e1));
  out[1] = (double)ms * 1e3 / ((double)iters * n_kernels);
  (void)hipGraphExecDestroy(ge); (void)hipGraphDestroy(g);
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipStreamDestroy(s);
  return VX_OK;
}


// Persistent-step probe (persist_probe.hpp): a chain of `stages` dependent 1024-wide GEMV stages in one launch.
// out[0] = us per launch, out[1] = us per stage, out[2] = max |y - host| over the final vector, out[3] = spin-timeout flag.
template <int ROWS>
static hipError_t launch_chain(int mode, int nwg, const ChainArgs& a, hipStream_t s) {
  if (mode == CHAIN_BARRIER_ONLY) chain_kernel<ROWS, CHAIN_BARRIER_ONLY><<<nwg, 256, 0, s>>>(a);
  else if (mode == CHAIN_FENCE) chain_kernel<ROWS, CHAIN_FENCE><<<nwg, 256, 0, s>>>(a);
  else if (mode == CHAIN_GROUP8) chain_kernel<ROWS, CHAIN_GROUP8><<<nwg, 256, 0, s>>>(a);
  else if (mode == CHAIN_XCD) chain_kernel<ROWS, CHAIN_XCD><<<nwg, 256, 0, s>>>(a);
  else chain_kernel<ROWS, CHAIN_BYPASS><<<nwg, 256, 0, s>>>(a);
  return hipGetLastError();
}

extern "C" int vx_debug_stage_chain(int32_t nwg, int32_t stages, int32_t rows, int32_t mode, int32_t iters, double* out) {
  if (!out || nwg <= 0 || nwg > 1024 || stages <= 0 || stages > 256 || iters <= 0 || mode < 0 || mode > 4 || (mode == 4 && nwg % 8))
    return fail(VX_ERR_ARG, "bad argument");
  if (rows != 4 && rows != 12 && rows != 16) return fail(VX_ERR_ARG, "rows must be 4, 12 or 16");
  if (nwg * rows < 1024) return fail(VX_ERR_ARG, "nwg*rows must cover the 1024-wide vector");
  hipDeviceProp_t prop;
  HIPC(hipGetDeviceProperties(&prop, 0));
  if (nwg > 2 * prop.multiProcessorCount) return fail(VX_ERR_ARG, "nwg exceeds what is certainly co-resident");
  const size_t slice = (size_t)rows * 1024, nW = (size_t)stages * nwg * slice;
  const int nout = nwg * rows;
  std::vector<uint16_t> hW(nW);
  uint32_t st = 12345u;
  for (size_t i = 0; i < nW; ++i) {  // uniform in +-sqrt(3)/32 -> unit gain per stage
    st = st * 1664525u + 1013904223u;
    float v = (((st >> 8) & 0xFFFF) / 65535.0f * 2.f - 1.f) * 0.0541f;
    uint32_t u; memcpy(&u, &v, 4);
    hW[i] = (uint16_t)((u + 0x7FFF + ((u >> 16) & 1)) >> 16);
  }
  std::vector<float> hx(2 * (size_t)nout, 0.f);
  for (int i = 0; i < 1024; ++i) hx[i] = sinf(0.37f * i);
  bf16* dW = nullptr; float* dx = nullptr; unsigned* dc = nullptr;
  HIPC(hipMalloc(&dW, nW * 2));
  HIPC(hipMalloc(&dx, hx.size() * 4));
  const size_t ctr_bytes = 256 * (2 + (size_t)nwg / 8);
  HIPC(hipMalloc(&dc, ctr_bytes));
  HIPC(hipMemcpy(dW, hW.data(), nW * 2, hipMemcpyHostToDevice));
  hipStream_t s;
  HIPC(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  hipEvent_t e0, e1;
  HIPC(hipEventCreate(&e0));
  HIPC(hipEventCreate(&e1));
  ChainArgs a{dW, dx, dc, dc + 1, stages, rows, mode};  // group counters live at dc + 64 * (1 + group)
  auto once = [&]() -> hipError_t {
    hipError_t r = hipMemsetAsync(dc, 0, ctr_bytes, s);
    if (r != hipSuccess) return r;
    return rows == 4 ? launch_chain<4>(mode, nwg, a, s) : rows == 12 ? launch_chain<12>(mode, nwg, a, s) : launch_chain<16>(mode, nwg, a, s);
  };
  HIPC(hipMemcpy(dx, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
  HIPC(once());
  HIPC(hipStreamSynchronize(s));
  std::vector<float> got(hx.size());
  unsigned flags[2] = {0, 0};
  HIPC(hipMemcpy(got.data(), dx, hx.size() * 4, hipMemcpyDeviceToHost));
  HIPC(hipMemcpy(flags, dc, 8, hipMemcpyDeviceToHost));
  double maxerr = 0.0;
  if (mode == 1 || mode == 2) {  // host chain: only the first 1024 outputs feed the next stage
    std::vector<float> x(hx.begin(), hx.begin() + 1024), y(nout);
    auto w = [&](size_t i) { uint32_t u = (uint32_t)hW[i] << 16; float f; memcpy(&f, &u, 4); return f; };
    for (int sidx = 0; sidx < stages; ++sidx) {
      const int need = sidx + 1 == stages ? nout : 1024;
      for (int o = 0; o < need; ++o) {
        const size_t base = ((size_t)sidx * nwg + o / rows) * slice + (size_t)(o % rows) * 1024;
        double acc = 0.0;
        for (int k = 0; k < 1024; ++k) acc += (double)w(base + k) * x[k];
        y[o] = (float)acc;
      }
      for (int k = 0; k < 1024; ++k) x[k] = y[k];
      if (sidx + 1 == stages)
        for (int o = 0; o < nout; ++o) maxerr = fmax(maxerr, fabs((double)got[(size_t)(stages & 1) * nout + o] - y[o]));
    }
  }
  float ms = 0.f;
  if (!flags[1]) {
    for (int i = 0; i < 2; ++i) HIPC(once());
    HIPC(hipEventRecord(e0, s));
    for (int i = 0; i < iters; ++i) HIPC(once());
    HIPC(hipEventRecord(e1, s));
    HIPC(hipStreamSynchronize(s));
    HIPC(hipEventElapsedTime(&ms, e0, e1));
    HIPC(hipMemcpy(flags, dc, 8, hipMemcpyDeviceToHost));
  }
  out[0] = (double)ms * 1e3 / iters;
  out[1] = out[0] / stages;
  out[2] = maxerr;
  out[3] = (double)flags[1];
  (void)hipFree(dW); (void)hipFree(dx); (void)hipFree(dc);
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipStreamDestroy(s);
  return VX_OK;
}


// L2 -> CU fill-rate probe (persist_probe.hpp): grid workgroups of `threads` lanes, `unroll` 16-byte loads in flight
// per lane, all over one shared region of `region_bytes` (keep it below the 4 MB of an XCD's L2).
// out[0] = GB/s over the whole chip, out[1] = bytes per clock per CU at the clock in out[2] (GHz, from wall_clock).
extern "C" int vx_debug_l2_fill(int32_t grid, int32_t threads, int32_t unroll, int64_t region_bytes, int32_t iters, double* out) {
  if (!out || grid <= 0 || grid > 4096 || threads <= 0 || threads > 256 || threads % 64 || iters <= 0 || region_bytes < 65536)
    return fail(VX_ERR_ARG, "bad argument");
  if (unroll != 1 && unroll != 2 && unroll != 4 && unroll != 8 && unroll != 16) return fail(VX_ERR_ARG, "unroll must be 1/2/4/8/16");
  const size_t nvec = (size_t)region_bytes / 16;
  uint4* buf = nullptr; unsigned* sink = nullptr;
  HIPC(hipMalloc((void**)&buf, nvec * 16));
  HIPC(hipMalloc((void**)&sink, 16));
  HIPC(hipMemset(buf, 1, nvec * 16));
  hipStream_t s;
  HIPC(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  hipEvent_t e0, e1;
  HIPC(hipEventCreate(&e0));
  HIPC(hipEventCreate(&e1));
  auto launch = [&]() {
    switch (unroll) {
      case 1: l2_fill_kernel<1><<<grid, threads, 0, s>>>(buf, nvec, iters, sink); break;
      case 2: l2_fill_kernel<2><<<grid, threads, 0, s>>>(buf, nvec, iters, sink); break;
      case 4: l2_fill_kernel<4><<<grid, threads, 0, s>>>(buf, nvec, iters, sink); break;
      case 8: l2_fill_kernel<8><<<grid, threads, 0, s>>>(buf, nvec, iters, sink); break;
      default: l2_fill_kernel<16><<<grid, threads, 0, s>>>(buf, nvec, iters, sink); break;
    }
  };
  launch();
  HIPC(hipStreamSynchronize(s));
  HIPC(hipEventRecord(e0, s));
  launch();
  HIPC(hipEventRecord(e1, s));
  HIPC(hipStreamSynchronize(s));
  HIPC(hipGetLastError());
  float ms = 0.f;
  HIPC(hipEventElapsedTime(&ms, e0, e1));
  const double bytes = (double)grid * threads * 16.0 * unroll * iters;
  hipDeviceProp_t prop;
  HIPC(hipGetDeviceProperties(&prop, 0));
  const double ghz = prop.clockRate * 1e-6;
  const int cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  const int busy = grid < cus ? grid : cus;
  out[0] = bytes / (ms * 1e-3) / 1e9;
  out[1] = bytes / (ms * 1e-3) / (ghz * 1e9) / busy;
  out[2] = ghz;
  (void)hipFree(buf); (void)hipFree(sink);
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipStreamDestroy(s);
  return VX_OK;
}


// Phase stamps of the last stamped kernel launch (probe builds, common.hpp VX_STAMP): out[i] = 10 ns ticks.
extern "C" int vx_debug_read_stamps(unsigned long long* out, int32_t n) {
#ifdef VX_STAMPS
  if (!out || n < 1 || n > 32) return fail(VX_ERR_ARG, "bad argument");
  HIPC(hipDeviceSynchronize());
  HIPC(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_vx_stamps), (size_t)n * 8));
  return VX_OK;
#else
  (void)out; (void)n;
  return fail(VX_ERR_UNSUPPORTED, "library built without -DVX_STAMPS (python vall-e_amd/csrc/build.py --stamps)");
#endif
}


// In-graph stamps of the AR decode step (common.hpp VX_KSTAMP_WG): dst == NULL (re)arms the device ring ([16 passes][64 kernels]
// x 8 bytes, zeroed); otherwise copies it to host memory.
extern "C" int vx_debug_kstamps(void* dst, int64_t nbytes, int32_t* dims) {
#ifdef VX_STAMPS
  static unsigned long long* ring = nullptr;
  const size_t bytes = (size_t)16 * 64 * 8;
  if (dims) { dims[0] = 16; dims[1] = 64; dims[2] = 0; dims[3] = 0; }
  HIPC(hipDeviceSynchronize());
  if (dst == nullptr) {
    if (!ring) HIPC(hipMalloc((void**)&ring, bytes));
    HIPC(hipMemset(ring, 0, bytes));
    HIPC(hipMemcpyToSymbol(HIP_SYMBOL(g_vx_kstamps), &ring, sizeof ring));
    HIPC(hipDeviceSynchronize());
    return VX_OK;
  }
  if (!ring) return fail(VX_ERR_STATE, "stamps not armed");
  HIPC(hipMemcpy(dst, ring, (size_t)nbytes < bytes ? (size_t)nbytes : bytes, hipMemcpyDeviceToHost));
  return VX_OK;
#else
  (void)dst; (void)nbytes; (void)dims;
  return fail(VX_ERR_UNSUPPORTED, "library built without -DVX_STAMPS (python vall-e_amd/csrc/build.py --stamps)");
#endif
}

// Stamps build only: phase times of the sharded decode step's launches (ar_tp.hpp TP_STAMP), [layer 16][workgroup 256][8] uint64 of
// s_memrealtime (100 MHz).  dst == NULL arms (allocates + zeroes), otherwise copies out.
extern "C" int vx_debug_fqstamps(void* dst, int64_t nbytes) {
#ifdef VX_STAMPS
  static unsigned long long* buf = nullptr;
  const size_t bytes = (size_t)2 * 16 * 256 * 16 * 8;  // [launch kind][layer][workgroup][16]
  HIPC(hipDeviceSynchronize());
  if (dst == nullptr) {
    if (!buf) HIPC(hipMalloc((void**)&buf, bytes));
    HIPC(hipMemset(buf, 0, bytes));
    HIPC(hipMemcpyToSymbol(HIP_SYMBOL(g_fq_stamps), &buf, sizeof buf));
    HIPC(hipDeviceSynchronize());
    return VX_OK;
  }
  if (!buf) return fail(VX_ERR_STATE, "stamps not armed");
  HIPC(hipMemcpy(dst, buf, (size_t)nbytes < bytes ? (size_t)nbytes : bytes, hipMemcpyDeviceToHost));
  return VX_OK;
#else
  (void)dst; (void)nbytes;
  return fail(VX_ERR_UNSUPPORTED, "library built without -DVX_STAMPS");
#endif
}
