// Shared device helpers for the gfx950 kernels: storage types, 16-byte vector loads,
// wave64 / workgroup reductions.  Wave width is hard-coded to 64 (CDNA4).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace vx {

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
  int32_t pad_;
};

}  // namespace vx
