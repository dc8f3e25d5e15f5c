// Shared device helpers for the gfx950 kernels: storage types, 16-byte vector loads,
// wave64 / workgroup reductions.  Wave width is hard-coded to 64 (CDNA4).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// In-kernel phase stamps (probe builds only: `build.py --stamps` compiles libvallex_stamps.so with -DVX_STAMPS).  Lane 0 of
// workgroup (0,0,0) records wall_clock64() (10 ns ticks) at VX_STAMP(i); read back with vx_debug_read_stamps.
#ifdef VX_STAMPS
extern __device__ unsigned long long g_vx_stamps[32];
#define VX_STAMP(i)                                                                                        \
  do {                                                                                                     \
    if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0) g_vx_stamps[i] = wall_clock64(); \
  } while (0)
// Per-kernel stamps of the AR decode step INSIDE the hipGraph replay (probe builds).  The stamped kernels of the step are launched
// with ONE EXTRA workgroup (the last) that does no work: its thread 0 records s_memrealtime (100 MHz) into ring slot
// [pass & 15][kernel id] of the buffer vx_debug_kstamps owns and the workgroup returns (VX_KSTAMP_WG), so no wave of the real
// computation executes anything the product does not.  The extra workgroup is dispatched last, i.e. its stamp trails the
// kernel's start by the dispatch time of the grid (~0.5 us, about the same for every kernel): only DIFFERENCES between consecutive
// kernels' stamps are used (tests/probes/ar_step_stamps.py).  The one-workgroup sampling kernel stamps from its own thread 0.
extern __device__ unsigned long long* g_vx_kstamps;
#define VX_KSTAMP_WG(kid, st_ptr)                                                                                  \
  do {                                                                                                             \
    if ((kid) >= 0 && (kid) < 64 && blockIdx.x == gridDim.x - 1) {                                                 \
      unsigned long long* vx_r_ = g_vx_kstamps;                                                                    \
      if (vx_r_ != nullptr && threadIdx.x == 0)                                                                    \
        vx_r_[(size_t)(((st_ptr)->pass) & 15) * 64 + (kid)] = __builtin_amdgcn_s_memrealtime();                   \
      return;                                                                                                      \
    }                                                                                                              \
  } while (0)
#define VX_KSTAMP_SELF(kid, pass, t0)                                                                              \
  do {                                                                                                             \
    unsigned long long* vx_r_ = g_vx_kstamps;                                                                      \
    if (vx_r_ != nullptr && (kid) >= 0 && (kid) < 64 && threadIdx.x == 0) vx_r_[(size_t)((pass) & 15) * 64 + (kid)] = (t0); \
  } while (0)
constexpr int VX_KSTAMP_EXTRA = 1;  // workgroups added to a stamped launch
#else
#define VX_STAMP(i) do { } while (0)
#define VX_KSTAMP_WG(kid, st_ptr) do { } while (0)
#define VX_KSTAMP_SELF(kid, pass, t0) do { } while (0)
constexpr int VX_KSTAMP_EXTRA = 0;
#endif

namespace vx {

// Host: per-DEVICE caches for launch helpers (an engine may live on any device of the process: a process-wide `static bool
// done` would set a kernel's max-dynamic-LDS attribute on the first device only and reuse its CU count everywhere).
inline int vx_cur_device() {
  int d = 0;
  (void)hipGetDevice(&d);
  return (d < 0 || d >= 16) ? 0 : d;
}
inline int vx_cu_count() {
  static int cu[16] = {};
  const int d = vx_cur_device();
  if (cu[d] <= 0) {
    int n = 256;
    (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, d);
    cu[d] = n > 0 ? n : 256;
  }
  return cu[d];
}

constexpr int WAVE = 64;
constexpr int NUM_AUDIO_TOKENS = 1024;  // valle/models/macros.py:5
constexpr int AR_VOCAB = 1025;          // ar_predict_layer rows (valle.py:153-155)
constexpr float LN_EPS = 1e-5f;

typedef __bf16 bf16;

// ---- storage <-> float ---------------------------------------------------------------------
__device__ __forceinline__ float to_f32(float v) { return v; }
__device__ __forceinline__ float to_f32(bf16 v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16 from_f32<bf16>(float v) { return (bf16)v; }  // RNE, v_cvt_pk_bf16_f32

template <typename T> struct Vec16;  // elements per 16-byte access
template <> struct Vec16<float> { static constexpr int N = 4; };
template <> struct Vec16<bf16> { static constexpr int N = 8; };

// One 16-byte global load, unpacked to floats.
__device__ __forceinline__ void unpack16(const uint4& r, float (&o)[4], float*) {
  o[0] = __uint_as_float(r.x); o[1] = __uint_as_float(r.y); o[2] = __uint_as_float(r.z); o[3] = __uint_as_float(r.w);
}
__device__ __forceinline__ void unpack16(const uint4& r, float (&o)[8], bf16*) {
  o[0] = __uint_as_float(r.x << 16); o[1] = __uint_as_float(r.x & 0xffff0000u);
  o[2] = __uint_as_float(r.y << 16); o[3] = __uint_as_float(r.y & 0xffff0000u);
  o[4] = __uint_as_float(r.z << 16); o[5] = __uint_as_float(r.z & 0xffff0000u);
  o[6] = __uint_as_float(r.w << 16); o[7] = __uint_as_float(r.w & 0xffff0000u);
}
template <typename T> __device__ __forceinline__ void unpack(const uint4& r, float (&o)[Vec16<T>::N]) {
  unpack16(r, o, (T*)nullptr);
}
template <typename T> __device__ __forceinline__ uint4 ld16(const T* p) { return *reinterpret_cast<const uint4*>(p); }
typedef unsigned vx_u32x4 __attribute__((ext_vector_type(4)));
template <typename T> __device__ __forceinline__ uint4 ld16nt(const T* p) {  // global_load_dwordx4 ... nt
  const vx_u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const vx_u32x4*>(p));
  return make_uint4(v.x, v.y, v.z, v.w);
}

// ---- DPP reductions (VALU cross-lane, no LDS round trip; ds_bpermute-based __shfl costs ~10x) ----
// dpp_ctrl: 0xB1 quad_perm[1,0,3,2], 0x4E quad_perm[2,3,0,1], 0x124/0x128 row_ror:4/8,
// 0x141 row_half_mirror, 0x140 row_mirror, 0x142 row_bcast:15 (rows 1,3), 0x143 row_bcast:31 (rows 2,3).
template <int CTRL, int ROW_MASK = 0xF> __device__ __forceinline__ float dpp_f(float old, float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, v),
                                                                CTRL, ROW_MASK, 0xF, false));
}
template <int CTRL, int ROW_MASK = 0xF> __device__ __forceinline__ int dpp_i(int old, int v) {
  return __builtin_amdgcn_update_dpp(old, v, CTRL, ROW_MASK, 0xF, false);
}
__device__ __forceinline__ float lane63(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
// every lane returns the sum over the 64 lanes (fixed tree order -> deterministic)
__device__ __forceinline__ float wave_sum_dpp(float v) {
  v += dpp_f<0xB1>(0.f, v);
  v += dpp_f<0x4E>(0.f, v);
  v += dpp_f<0x124>(0.f, v);
  v += dpp_f<0x128>(0.f, v);
  v += dpp_f<0x142, 0xA>(0.f, v);
  v += dpp_f<0x143, 0xC>(0.f, v);
  return lane63(v);
}
__device__ __forceinline__ float wave_max_dpp(float v) {
  v = fmaxf(v, dpp_f<0xB1>(v, v));
  v = fmaxf(v, dpp_f<0x4E>(v, v));
  v = fmaxf(v, dpp_f<0x124>(v, v));
  v = fmaxf(v, dpp_f<0x128>(v, v));
  v = fmaxf(v, dpp_f<0x142, 0xA>(v, v));
  v = fmaxf(v, dpp_f<0x143, 0xC>(v, v));
  return lane63(v);
}
__device__ __forceinline__ uint32_t wave_umin_dpp(uint32_t v) {
  v = min(v, (uint32_t)dpp_i<0xB1>((int)v, (int)v));
  v = min(v, (uint32_t)dpp_i<0x4E>((int)v, (int)v));
  v = min(v, (uint32_t)dpp_i<0x124>((int)v, (int)v));
  v = min(v, (uint32_t)dpp_i<0x128>((int)v, (int)v));
  v = min(v, (uint32_t)dpp_i<0x142, 0xA>((int)v, (int)v));
  v = min(v, (uint32_t)dpp_i<0x143, 0xC>((int)v, (int)v));
  return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
// sum over aligned groups of 8 / 16 lanes; every lane of the group gets the group sum
__device__ __forceinline__ float group8_sum_dpp(float v) {
  v += dpp_f<0xB1>(0.f, v);
  v += dpp_f<0x4E>(0.f, v);
  v += dpp_f<0x141>(0.f, v);
  return v;
}
__device__ __forceinline__ float group16_sum_dpp(float v) {
  v = group8_sum_dpp(v);
  v += dpp_f<0x140>(0.f, v);
  return v;
}

// ---- reductions ------------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, WAVE);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, WAVE));
  return v;
}

// Sum over a workgroup of NW waves; every thread gets the result.  `red` = NW floats of LDS.
// Two barriers; may be called repeatedly with the same scratch.
template <int NW> __device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  const int w = threadIdx.x >> 6;
  __syncthreads();  // scratch free
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NW; ++i) s += red[i];
  return s;
}
template <int NW> __device__ __forceinline__ float block_max(float v, float* red) {
  v = wave_max(v);
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  float s = red[0];
#pragma unroll
  for (int i = 1; i < NW; ++i) s = fmaxf(s, red[i]);
  return s;
}

// argmax with "first index wins" on ties (torch.argmax on CPU returns the first maximum).
struct ValIdx { float v; int i; };
__device__ __forceinline__ ValIdx better(ValIdx a, ValIdx b) {
  return (b.v > a.v || (b.v == a.v && b.i < a.i)) ? b : a;
}
template <int CTRL, int ROW_MASK = 0xF> __device__ __forceinline__ ValIdx argmax_step_dpp(ValIdx a) {
  ValIdx b;
  b.v = dpp_f<CTRL, ROW_MASK>(a.v, a.v);  // masked rows read themselves: better(a, a) == a
  b.i = dpp_i<CTRL, ROW_MASK>(a.i, a.i);
  return better(a, b);
}
__device__ __forceinline__ ValIdx wave_argmax_dpp(ValIdx a) {
  a = argmax_step_dpp<0xB1>(a);
  a = argmax_step_dpp<0x4E>(a);
  a = argmax_step_dpp<0x124>(a);
  a = argmax_step_dpp<0x128>(a);
  a = argmax_step_dpp<0x142, 0xA>(a);
  a = argmax_step_dpp<0x143, 0xC>(a);
  ValIdx r;
  r.v = lane63(a.v);
  r.i = __builtin_amdgcn_readlane(a.i, 63);
  return r;
}
__device__ __forceinline__ ValIdx wave_argmax(ValIdx a) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    ValIdx b; b.v = __shfl_xor(a.v, o, WAVE); b.i = __shfl_xor(a.i, o, WAVE);
    a = better(a, b);
  }
  return a;
}
template <int NW> __device__ __forceinline__ ValIdx block_argmax(ValIdx a, float* redv, int* redi) {
  a = wave_argmax(a);
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) { redv[w] = a.v; redi[w] = a.i; }
  __syncthreads();
  ValIdx r; r.v = redv[0]; r.i = redi[0];
#pragma unroll
  for (int i = 1; i < NW; ++i) { ValIdx b; b.v = redv[i]; b.i = redi[i]; r = better(r, b); }
  return r;
}

// Device-resident decode state: read by every kernel of the captured AR step so that one
// hipGraph serves every pass (no per-launch kernel arguments change).
struct ArState {
  int32_t S;          // text rows
  int32_t bos;        // 1 if a BOS row precedes the prompt (valle.py:1006-1007)
  int32_t P;          // prompt frames
  int32_t row;        // KV row (0-based, text rows first) of the token currently being processed
  int32_t pass;       // forward passes completed so far minus 1 == index of the newest logits row
  int32_t n_gen;      // tokens appended so far
  int32_t done;       // stop flag
  int32_t stop_reason;
  // decode parameters (vx_decode_params)
  int32_t top_k;
  float temperature;
  int32_t max_new;
  int32_t n_forced;
  const float* exp_noise;
  long long noise_rows;
  unsigned long long seed;
  const long long* forced;
  int32_t trace_logits;
  int32_t kv_text;    // rows in front of the audio sub-sequence in the KV cache: S for VALL-E (text rows are cached), 0 for VALL-F
                      // (the text is cross-attention memory); audio position of KV row r = r - kv_text (valle.py:1013-1016)
};

}  // namespace vx
