// MXFP8 row GEMMs for the NAR stages (BASELINE configs[4]: "fp8 MFMA QKV/FFN path for NAR stages"):
//   C[M,N] = A[M,K] . W[N,K]^T with both operands in OCP e4m3 (gfx950's fp8: e4m3fn, not MI300's fnuz) and one E8M0
//   scale per 32 consecutive k of every row (OCP MX block scaling), multiplied by v_mfma_scale_f32_32x32x64_f8f6f4:
//   the matrix core applies the two block scales itself, at twice the bf16 rate per clock, fp32 accumulate.
// Why block scales and not per-row / per-channel ones: the A operand of FFN2 is FFN1's output, whose row maximum spans
// sixteen 256-wide tiles of sixteen workgroups; a 32-wide block lives inside one accumulator fragment, so FFN1's epilogue
// can quantise its own output (and the LayerNorm kernel its own row) with no extra pass over the activations.
//
// Operand storage:  bytes [rows][K] e4m3, K-contiguous;  scales [K/32][ld_s] E8M0 bytes (k-block major, ld_s >= rows rounded
// up to 256 and zero-padded; inside a k-block's run row r sits at mx_spos(r)), so that the scales one 256-row tile needs for one
// 64-deep K stage are two runs of 256 bytes.
// Quantisation (mx_block_scale / mx_pack4, mirrored by tests/mx_ref.py): with E the biased exponent of the block's largest
// magnitude, scale byte = max(E, 8) - 8 (i.e. 2^(floor(log2 amax) - 8): e4m3's largest binade), values multiplied by the
// exact inverse power of two, clamped to +-448 and rounded to nearest even by v_cvt_pk_fp8_f32.
#pragma once
#include "common.hpp"
#include "mfma_kernels.hpp"

namespace vx {

typedef int i32x8_t __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float group8_max_dpp(float v) {  // max over aligned groups of 8 lanes, in every lane of the group
  v = fmaxf(v, dpp_f<0xB1>(v, v));
  v = fmaxf(v, dpp_f<0x4E>(v, v));
  v = fmaxf(v, dpp_f<0x141>(v, v));
  return v;
}
// Position of a row's scale inside a k-block's run of the scale array: rows r, r + 32, r + 64, r + 96 of every 128-row group sit
// in one dword, so a lane of the GEMM reads the scales of its four A fragments (two W fragments) with one LDS read and picks
// the byte with the MFMA's op_sel.
__host__ __device__ __forceinline__ int mx_spos(int row) { return (row & ~127) | ((row & 31) << 2) | ((row >> 5) & 3); }

__device__ __forceinline__ uint32_t mx_block_scale(float amax, float& inv) {
  const uint32_t E = (__float_as_uint(amax) >> 23) & 0xffu;
  const uint32_t byte = max(E, 8u) - 8u;
  inv = __uint_as_float((254u - byte) << 23);  // 2^(127 - byte), exact
  return byte;
}
__device__ __forceinline__ uint32_t mx_pack4(float a, float b, float c, float d, float inv) {
  a = fminf(fmaxf(a * inv, -448.f), 448.f); b = fminf(fmaxf(b * inv, -448.f), 448.f);
  c = fminf(fmaxf(c * inv, -448.f), 448.f); d = fminf(fmaxf(d * inv, -448.f), 448.f);
  int p = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
  p = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, p, true);
  return (uint32_t)p;
}

// rows of fp32 -> MXFP8 (weights at load time, operands of the op-level test).  One wave per row, lane l owns k = 4 (64 i + l) .. +3.
__global__ __launch_bounds__(256) void mx_quant_rows_kernel(const float* __restrict__ x, uint8_t* __restrict__ q,
                                                            uint8_t* __restrict__ sc, int rows, int K, int ld_s) {
  const int lane = threadIdx.x & 63, r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  for (int k0 = 0; k0 < K; k0 += 256) {
    const int k = k0 + lane * 4;
    const bool ok = k < K;  // K % 32 == 0: a block is all in or all out
    const float4 v = ok ? *reinterpret_cast<const float4*>(x + (size_t)r * K + k) : make_float4(0.f, 0.f, 0.f, 0.f);
    const float am = group8_max_dpp(fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
    float inv;
    const uint32_t sb = mx_block_scale(am, inv);
    if (ok) {
      *reinterpret_cast<uint32_t*>(q + (size_t)r * K + k) = mx_pack4(v.x, v.y, v.z, v.w, inv);
      if ((lane & 7) == 0) sc[(size_t)(k >> 5) * ld_s + mx_spos(r)] = (uint8_t)sb;
    }
  }
}

// (Adaptive)LayerNorm with an MXFP8 result: layernorm_rows_kernel's arithmetic (rows_kernels.hpp; same fold prologue for the
// split-K slabs of the preceding GEMM), the normalised row quantised in registers: the 32-wide blocks are aligned groups of 8 lanes.
template <int MAXV>
__global__ __launch_bounds__(256) void layernorm_rows_mx_kernel(const float* x, const float* __restrict__ gamma,
                                                                const float* __restrict__ beta, const float* __restrict__ ada_w,
                                                                const float* __restrict__ ada_b, uint8_t* __restrict__ q,
                                                                uint8_t* __restrict__ sc, int rows, int d, int ld_s) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  const float* xr = x + (size_t)r * d;
  const bool ada = ada_w != nullptr;
  float4 v[MAXV], g[MAXV], b[MAXV], w[MAXV], c[MAXV];
  int kk[MAXV];
  bool ok[MAXV];
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int k = (i * 64 + lane) * 4;
    ok[i] = k < d;
    kk[i] = ok[i] ? k : 0;
    v[i] = *reinterpret_cast<const float4*>(xr + kk[i]);
    g[i] = *reinterpret_cast<const float4*>(gamma + kk[i]);
    b[i] = *reinterpret_cast<const float4*>(beta + kk[i]);
    if (ada) {
      w[i] = *reinterpret_cast<const float4*>(ada_w + kk[i]);
      c[i] = *reinterpret_cast<const float4*>(ada_b + kk[i]);
    }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) s += ok[i] ? (v[i].x + v[i].y) + (v[i].z + v[i].w) : 0.f;
  const float mean = wave_sum(s) / (float)d;
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const float d0 = v[i].x - mean, d1 = v[i].y - mean, d2 = v[i].z - mean, d3 = v[i].w - mean;
    ss += ok[i] ? (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3) : 0.f;
  }
  const float rstd = 1.0f / sqrtf(wave_sum(ss) / (float)d + LN_EPS);
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    float o[4] = {(v[i].x - mean) * rstd * g[i].x + b[i].x, (v[i].y - mean) * rstd * g[i].y + b[i].y,
                  (v[i].z - mean) * rstd * g[i].z + b[i].z, (v[i].w - mean) * rstd * g[i].w + b[i].w};
    if (ada) {
      o[0] = __fadd_rn(__fmul_rn(w[i].x, o[0]), c[i].x); o[1] = __fadd_rn(__fmul_rn(w[i].y, o[1]), c[i].y);
      o[2] = __fadd_rn(__fmul_rn(w[i].z, o[2]), c[i].z); o[3] = __fadd_rn(__fmul_rn(w[i].w, o[3]), c[i].w);
    }
    if (!ok[i]) { o[0] = o[1] = o[2] = o[3] = 0.f; }
    const float am = group8_max_dpp(fmaxf(fmaxf(fabsf(o[0]), fabsf(o[1])), fmaxf(fabsf(o[2]), fabsf(o[3]))));
    float inv;
    const uint32_t sb = mx_block_scale(am, inv);
    if (ok[i]) {
      *reinterpret_cast<uint32_t*>(q + (size_t)r * d + kk[i]) = mx_pack4(o[0], o[1], o[2], o[3], inv);
      if ((lane & 7) == 0) sc[(size_t)(kk[i] >> 5) * ld_s + mx_spos(r)] = (uint8_t)sb;
    }
  }
}

// ---- 256x256 tile MXFP8 GEMM: mfma256_kernel's structure (persistent workgroups, XCD-aware tile walk, global -> LDS by
// LDS-DMA into a 4-deep ring of 64-byte-row stages, counted vmcnt + raw barrier, fragments double-buffered in registers) with
// 64 k per stage instead of 32 and ONE v_mfma_scale_f32_32x32x64_f8f6f4 per (A fragment, W fragment) per stage:
// 8 waves as 2 (M) x 4 (N), each 128 x 64 = 4 x 2 fragments of 32 x 32, 8 MFMAs of 64 cycles per stage = the matrix time of the
// bf16 kernel's 32 MFMAs of 16 cycles at twice the K.
// Lane maps (tests/probes/mx_probe.hip, exact data with unequal block scales): lane (r = l & 31, h = l >> 5) holds, of row r,
// k = 16 h .. 16 h + 15 in operand bytes 0-15 and k = 32 + 16 h .. 32 + 16 h + 15 in bytes 16-31; the scale byte lane (r, h)
// supplies applies to k-block h of the row (k = 32 h .. 32 h + 31), i.e. to bytes 16 h .. 16 h + 15 of BOTH lanes of the row -
// not to the lane's own 32 bytes (with equal scales per row the two readings cannot be told apart).
// The MFMA is issued as (W fragment, A fragment), so the accumulator has m on the lane and n = (v&3) + 8 (v>>2) + 4 h in its
// registers: 4 consecutive n per register quad.
// Per stage the ring slot also carries the stage's 2 x 256 scale bytes of each operand (every wave brings 128 of the 1024 bytes
// with one 2-byte-per-lane LDS-DMA).
enum MxOut { MX_OUT_F32 = 0, MX_OUT_BF16 = 1, MX_OUT_MX = 2 };

template <int EPI, int OUT>
__global__ __launch_bounds__(512) void mx256_kernel(const uint8_t* __restrict__ A, const uint8_t* __restrict__ SA, int lda_s,
                                                    const uint8_t* __restrict__ W, const uint8_t* __restrict__ SW, int ldw_s,
                                                    const float* __restrict__ bias, void* __restrict__ Cv,
                                                    uint8_t* __restrict__ SC, int ldc_s, int M, int N, int K,
                                                    bf16* __restrict__ vt, int vt_n0, int vt_ld, int ntn, int ntiles) {
  constexpr int STAGE = 32768 + 1024;  // A 16 KB | W 16 KB | A scales 2 x 256 | W scales 2 x 256
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  // `wave` in an SGPR: the branches on it below enclose scalar instructions (s_waitcnt / s_barrier / the LDS-DMA's M0 setup),
  // which ignore EXEC - under a branch the compiler takes for divergent a wave would run BOTH sides' waits and barriers
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int r = lane & 31, h = lane >> 5;
  const int G = gridDim.x;
  const int q8 = ntiles >> 3, r8 = ntiles & 7;
  auto tile_of = [&](int v) {
    const int xcd = v & 7, loc = v >> 3;
    return (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + loc;
  };
  const int cnt = (ntiles - (int)blockIdx.x + G - 1) / G;

  // one stage = 2048 16-byte slots (A: 1024, W: 1024), 4 per thread; slot q -> row q>>2, position q&3 holds source chunk
  // (q&3) ^ ((-(row>>2))&3)  (rows are 64 bytes: the swizzle of mfma256_kernel, conflict-free for these fragment reads too)
  const int q0 = tid, q1 = tid + 512;
  const int rowa0 = q0 >> 2, rowa1 = q1 >> 2;
  const int ca0 = ((q0 & 3) ^ ((0 - (rowa0 >> 2)) & 3)) * 16, ca1 = ((q1 & 3) ^ ((0 - (rowa1 >> 2)) & 3)) * 16;
  const int d0 = (q0 - lane) * 16, d1 = (q1 - lane) * 16;
  const int nk = K / 64;

  const uint8_t *srcA0, *srcA1, *srcW0, *srcW1, *srcS;
  int l_ord = 0, l_k = 0, l_slot = 0;
  size_t s_step = 0;  // bytes between two stages in the scale array this wave streams
  auto set_load_tile = [&](int ord) {
    const int tile = tile_of((int)blockIdx.x + min(ord, cnt - 1) * G);
    const int mt = tile / ntn, nt = tile - mt * ntn;
    srcA0 = A + (size_t)min(mt * 256 + rowa0, M - 1) * K + ca0;
    srcA1 = A + (size_t)min(mt * 256 + rowa1, M - 1) * K + ca1;
    srcW0 = W + (size_t)min(nt * 256 + rowa0, N - 1) * K + ca0;
    srcW1 = W + (size_t)min(nt * 256 + rowa1, N - 1) * K + ca1;
    // scales: the stage's block is [A kb0 | A kb1 | W kb0 | W kb1] x 256 bytes = one 16-byte LDS-DMA load of one wave; every
    // wave issues it (same bytes to the same place), so that all waves run the same five loads per stage: one counted wait, no
    // branch in the loop, and only 16-byte LDS-DMA in the kernel (a narrower one next to the ds_reads made hipcc drain vmcnt(0))
    {
      const int part = lane >> 4, off = (lane & 15) * 16;
      if (part < 2) { srcS = SA + (size_t)part * lda_s + mt * 256 + off; s_step = (size_t)2 * lda_s; }
      else { srcS = SW + (size_t)(part - 2) * ldw_s + nt * 256 + off; s_step = (size_t)2 * ldw_s; }
    }
  };
  auto stage = [&]() {
    unsigned char* base = lds + (l_slot & 3) * STAGE;
    const int k0 = l_k * 64;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(srcA0 + k0),
                                     (__attribute__((address_space(3))) void*)(base + d0), 16, 0, 0);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(srcA1 + k0),
                                     (__attribute__((address_space(3))) void*)(base + d1), 16, 0, 0);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(srcW0 + k0),
                                     (__attribute__((address_space(3))) void*)(base + 16384 + d0), 16, 0, 0);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(srcW1 + k0),
                                     (__attribute__((address_space(3))) void*)(base + 16384 + d1), 16, 0, 0);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(srcS + (size_t)l_k * s_step),
                                     (__attribute__((address_space(3))) void*)(base + 32768), 16, 0, 0);
    ++l_slot;
    if (++l_k == nk) { l_k = 0; set_load_tile(++l_ord); }
  };

  // Registers: 128 accumulators leave room for ONE and a half stages of fragments, not two: the W fragments (and the scale
  // dwords) of the current and of the next stage live in two named sets, the A fragments in two halves (rows i = 0,1 / 2,3) that
  // are refilled one half-stage (4 MFMAs = 256 matrix cycles) ahead of their use.
  f32x16_t acc[4][2];
  u32x4 w0[2][2], w1[2][2], aX[2][2], aY[2][2];
  unsigned sw0 = 0, sw1 = 0, sa0 = 0, sa1 = 0;
  auto frag = [](const u32x4 (&p)[2]) {
    i32x8_t v;
    v[0] = (int)p[0].x; v[1] = (int)p[0].y; v[2] = (int)p[0].z; v[3] = (int)p[0].w;
    v[4] = (int)p[1].x; v[5] = (int)p[1].y; v[6] = (int)p[1].z; v[7] = (int)p[1].w;
    return v;
  };
  auto read_w = [&](int slot, u32x4 (&w)[2][2], unsigned& sw, unsigned& sa) {
    const unsigned char* ba = lds + (slot & 3) * STAGE;
    const unsigned char* bw = ba + 16384;
    const unsigned char* bs = ba + 32768 + h * 256;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int row = wn * 64 + j * 32 + r, sz = (0 - (row >> 2)) & 3;
      w[j][0] = *reinterpret_cast<const u32x4*>(bw + row * 64 + ((h ^ sz) << 4));        // k = 16 h .. +15   (block 0)
      w[j][1] = *reinterpret_cast<const u32x4*>(bw + row * 64 + (((2 + h) ^ sz) << 4));  // k = 32 + 16 h .. (block 1)
    }
    // the scale dwords of this lane: bytes i = 0..3 are rows r + 32 i of the wave's 128-row group (mx_spos)
    sa = *reinterpret_cast<const unsigned*>(bs + wm * 128 + r * 4);
    sw = *reinterpret_cast<const unsigned*>(bs + 512 + (wn >> 1) * 128 + r * 4) >> ((wn & 1) * 16);  // bytes 0,1 = fragments j = 0,1
  };
  auto read_a = [&](int slot, int half, u32x4 (&a)[2][2]) {
    const unsigned char* ba = lds + (slot & 3) * STAGE;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int row = wm * 128 + (2 * half + i) * 32 + r, sz = (0 - (row >> 2)) & 3;
      a[i][0] = *reinterpret_cast<const u32x4*>(ba + row * 64 + ((h ^ sz) << 4));
      a[i][1] = *reinterpret_cast<const u32x4*>(ba + row * 64 + (((2 + h) ^ sz) << 4));
    }
  };
  // 4 MFMAs: A fragments 2 half .. 2 half + 1 against both W fragments (op_sel = byte of the scale dword)
#define MX_MM(HALF, W, SW, A, SA)                                                                                                   \
  do {                                                                                                                             \
    acc[2 * HALF][0] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(frag(W[0]), frag(A[0]), acc[2 * HALF][0], 0, 0, 0, (int)SW, 2 * HALF, (int)SA);         \
    acc[2 * HALF][1] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(frag(W[1]), frag(A[0]), acc[2 * HALF][1], 0, 0, 1, (int)SW, 2 * HALF, (int)SA);         \
    acc[2 * HALF + 1][0] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(frag(W[0]), frag(A[1]), acc[2 * HALF + 1][0], 0, 0, 0, (int)SW, 2 * HALF + 1, (int)SA); \
    acc[2 * HALF + 1][1] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(frag(W[1]), frag(A[1]), acc[2 * HALF + 1][1], 0, 0, 1, (int)SW, 2 * HALF + 1, (int)SA); \
  } while (0)
  // counted wait: the youngest stage (5 loads per wave) may stay in flight across the barrier
  auto wait_barrier = [&]() { asm volatile("s_waitcnt vmcnt(5)\n\ts_barrier" ::: "memory"); };

  // Stage s of the stream lives in ring slot s & 3.  On stage s: [issue stage s+3] [read A rows 2,3 of s] [MFMAs rows 0,1]
  // [read W + scales + A rows 0,1 of stage s+1] [MFMAs rows 2,3] [vmcnt: stage s+2 landed] [barrier].  Slot (s+3)&3 = (s-1)&3
  // was last read (A rows 2,3) during stage s-1, whose MFMAs consumed those reads before that stage's barrier.
  set_load_tile(0);
  stage();
  stage();
  stage();
  wait_barrier();  // stages 0 and 1 landed everywhere
  read_w(0, w0, sw0, sa0);
  read_a(0, 0, aX);
  int c_slot = 0;
  for (int ord = 0; ord < cnt; ++ord) {
    const int tile = tile_of((int)blockIdx.x + ord * G);
    const int mt = tile / ntn, nt = tile - mt * ntn;
    const int m0 = mt * 256, n0 = nt * 256;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[i][j][v] = 0.f;
    for (int t = 0; t < nk; t += 2) {  // nk is even (K % 128 == 0)
      stage();
      read_a(c_slot, 1, aY);
      MX_MM(0, w0, sw0, aX, sa0);
      read_w(c_slot + 1, w1, sw1, sa1);
      read_a(c_slot + 1, 0, aX);
      MX_MM(1, w0, sw0, aY, sa0);
      wait_barrier();
      stage();
      read_a(c_slot + 1, 1, aY);
      MX_MM(0, w1, sw1, aX, sa1);
      read_w(c_slot + 2, w0, sw0, sa0);
      read_a(c_slot + 2, 0, aX);
      MX_MM(1, w1, sw1, aY, sa1);
      wait_barrier();
      c_slot += 2;
    }
#undef MX_MM

    // epilogue: acc[i][j][v] = C[m = m0 + wm*128 + i*32 + r][n = n0 + wn*64 + j*32 + (v&3) + 8*(v>>2) + 4*h]
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int nf = n0 + wn * 64 + j * 32;  // N % 256 == 0: always in range
      float4 bq[4];
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) bq[qd] = EPI != GE_PLAIN ? *reinterpret_cast<const float4*>(bias + nf + 8 * qd + 4 * h) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int m = m0 + wm * 128 + i * 32 + r;
        float x[16];
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
          x[4 * qd] = acc[i][j][4 * qd] + bq[qd].x; x[4 * qd + 1] = acc[i][j][4 * qd + 1] + bq[qd].y;
          x[4 * qd + 2] = acc[i][j][4 * qd + 2] + bq[qd].z; x[4 * qd + 3] = acc[i][j][4 * qd + 3] + bq[qd].w;
        }
        if (EPI == GE_RELU) {
#pragma unroll
          for (int v = 0; v < 16; ++v) x[v] = fmaxf(x[v], 0.f);
        }
        if (OUT == MX_OUT_MX) {  // the row's 32 columns of this fragment are one MX block: 16 here, 16 in the other half-wave
          float am = 0.f;
#pragma unroll
          for (int v = 0; v < 16; ++v) am = fmaxf(am, fabsf(x[v]));
          am = fmaxf(am, xor32_f(am));
          float inv;
          const uint32_t sb = mx_block_scale(am, inv);
          if (m < M) {
            uint8_t* cp = reinterpret_cast<uint8_t*>(Cv) + (size_t)m * N + nf + 4 * h;
#pragma unroll
            for (int qd = 0; qd < 4; ++qd)
              *reinterpret_cast<uint32_t*>(cp + 8 * qd) = mx_pack4(x[4 * qd], x[4 * qd + 1], x[4 * qd + 2], x[4 * qd + 3], inv);
            if (h == 0) SC[(size_t)(nf >> 5) * ldc_s + mx_spos(m)] = (uint8_t)sb;
          }
        } else if (m < M) {
          if (OUT == MX_OUT_F32) {
            float* cp = reinterpret_cast<float*>(Cv) + (size_t)m * N + nf + 4 * h;
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) {
              float4* p4 = reinterpret_cast<float4*>(cp + 8 * qd);
              if (EPI == GE_RESID) {
                const float4 o = *p4;
                *p4 = make_float4(o.x + x[4 * qd], o.y + x[4 * qd + 1], o.z + x[4 * qd + 2], o.w + x[4 * qd + 3]);
              } else {
                *p4 = make_float4(x[4 * qd], x[4 * qd + 1], x[4 * qd + 2], x[4 * qd + 3]);
              }
            }
          } else {
            bf16* cp = reinterpret_cast<bf16*>(Cv) + (size_t)m * N + nf + 4 * h;
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) {
              union { bf16 e[4]; uint2 u; } pk;
#pragma unroll
              for (int v = 0; v < 4; ++v) pk.e[v] = (bf16)x[4 * qd + v];
              *reinterpret_cast<uint2*>(cp + 8 * qd) = pk.u;
              if (vt != nullptr && nf >= vt_n0) {
#pragma unroll
                for (int v = 0; v < 4; ++v) vt[(size_t)(nf + 8 * qd + 4 * h + v - vt_n0) * vt_ld + m] = pk.e[v];
              }
            }
          }
        }
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the stages issued past the last tile
}

// ---- the MXFP8 256x256 GEMM on the 8-phase ping-pong schedule of mfma256p_kernel (mfma_kernels.hpp): e4m3 rows of 128 k are 128
// bytes, exactly a bf16 row of 64 k, so the byte geometry is the same - a K-tile (128 k) is four 16 KB half-tiles (B-h0, A-h0,
// B-h1, A-h1) in one of two 64 KB buffers, one half-tile staged per phase with four in flight, the two row groups one barrier
// apart - and so is the matrix time: a phase is one quadrant of the wave's 128 x 64 output over the K-tile's two 64-k steps =
// 4 v_mfma_scale_f32_32x32x64_f8f6f4 of 64 cycles (bf16: 16 MFMAs of 16).  Fragment reads per phase 8 / 4 / 8 / 4 + the next
// K-tile's four scale dwords.  The K-tile's 2 KB of E8M0 scales ([A | W][4 k-blocks][256 rows]) travel with its first half-tile:
// wave w brings k-block w & 3 of operand w >> 2 with one 4-byte-per-lane LDS-DMA, so every wave issues nine loads per K-tile and
// the counted wait is vmcnt(9).  Epilogues as in mx256_kernel; full tiles store unguarded (32 store instructions per lane, 40
// with the scale bytes of the quantised form) so that the four waits behind an epilogue can count them in.
// Three things this kernel needs that the bf16 one got away without (each seen in the ISA as 400-900 bytes of scratch per lane):
// the stream's step to the next item is branch-free (a branch inside the phases splits the block); every phase ends with an empty
// asm that uses its accumulators (the scaled MFMAs are pure and next used a K-tile later: the machine sinker moved them out of
// their phases, all 32 to the bottom of the iteration); the fragment read addresses are rebuilt per phase from laundered lane
// constants (held across the loop, eight XOR-ed variants per buffer were spilled).
template <int EPI, int OUT>
__global__ __launch_bounds__(512) void mx256p_kernel(const uint8_t* __restrict__ A, const uint8_t* __restrict__ SA, int lda_s,
                                                     const uint8_t* __restrict__ W, const uint8_t* __restrict__ SW, int ldw_s,
                                                     const float* __restrict__ bias, void* __restrict__ Cv,
                                                     uint8_t* __restrict__ SC, int ldc_s, int M, int N, int K,
                                                     bf16* __restrict__ vt, int vt_n0, int vt_ld, int ntn, int ntiles, P8Tail tl) {
  constexpr int BUF = 65536 + 2048;  // four half-tiles | A scales 4 x 256 | W scales 4 x 256
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];  // [2][BUF], then 8 x 256 B of bias values
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int r = lane & 31, h = lane >> 5;
  const int G = gridDim.x;
  const int q8 = ntiles >> 3, r8 = ntiles & 7;
  auto tile_of = [&](int v) {
    const int xcd = v & 7, loc = v >> 3;
    return (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + loc;
  };
  const int nk = K >> 7;  // K-tiles of 128 k; K % 256 == 0: an even number
  const int S = tl.split;
  const int R = S > 1 ? ntiles / G : 0;
  const bool has_tail = S > 1 && (int)blockIdx.x < (ntiles - R * G) * S;
  const int cnt = S > 1 ? R + (has_tail ? 1 : 0) : (ntiles - (int)blockIdx.x + G - 1) / G;
  const int tail_a = S > 1 ? (int)blockIdx.x / S : 0, tail_s = S > 1 ? (int)blockIdx.x - tail_a * S : 0;
  auto item = [&](int ord, int& tile, int& kt0, int& kte) {
    if (S > 1 && ord >= R) {
      tile = tile_of(R * G + tail_a);
      kt0 = tail_s * (nk / S);
      kte = kt0 + nk / S;
    } else {
      tile = tile_of((int)blockIdx.x + ord * G);
      kt0 = 0;
      kte = nk;
    }
  };

  // ---- load side (see mfma256p_kernel): two 16-byte slots per thread and half-tile, local row (tid >> 3) + 64 i, chunk swizzle
  // position = chunk ^ ((row >> 1) & 7)
  const int lrow = tid >> 3;
  const unsigned csrc = (unsigned)(((tid & 7) ^ ((lrow >> 1) & 7)) * 16);
  const unsigned ldst = (unsigned)(wave * 1024);
  const unsigned tb = (unsigned)((lrow >> 5) * 64 + (lrow & 31));
  const unsigned Kb = (unsigned)K;  // bytes per row
  // The stream's item (first row, first column, K-tile, last K-tile + 1) and the one after it (n_*), all scalars.  The step to the
  // next item is a handful of scalar selects: a branch inside the unrolled phases splits the block, and the machine sinker then
  // moves the MFMAs out of their phase into the block of their next use (seen in the ISA: empty setprio pairs, 600 bytes of
  // scratch).  The item after next is looked up once per output tile, outside the phases (set_next).
  int l_kt = 0, l_kend = 0, l_m0 = 0, l_n0 = 0, n_kt = 0, n_kend = 0, n_m0 = 0, n_n0 = 0;
  auto lookup = [&](int ord, int& m0_, int& n0_, int& kt_, int& kend_) {
    int tile;
    item(min(ord, cnt - 1), tile, kt_, kend_);  // past the end: the last item again (loaded, never used)
    const int mt = tile / ntn;
    m0_ = mt * 256;
    n0_ = (tile - mt * ntn) * 256;
  };
  // kind 0: B-h0 (+ the K-tile's scales), 1: A-h0, 2: B-h1, 3: A-h1; kind 0 opens the next K-tile
  auto stage = [&](auto kindc, auto bufc) {
    constexpr int kind = decltype(kindc)::value, buf = decltype(bufc)::value;
    constexpr bool isA = kind == 1 || kind == 3;
    constexpr int sub = kind < 2 ? 0 : 1;
    if (kind == 0) {
      const bool adv = l_kt + 1 == l_kend;
      l_kt = adv ? n_kt : l_kt + 1;
      l_m0 = adv ? n_m0 : l_m0;
      l_n0 = adv ? n_n0 : l_n0;
      l_kend = adv ? n_kend : l_kend;
    }
    unsigned char* base = lds + buf * BUF + kind * 16384 + ldst;
    const unsigned kb = (unsigned)l_kt * 128u + csrc;
    unsigned o0, o1;
    if (isA) {
      const int row = l_m0 + lrow + 64 * sub;
      o0 = (unsigned)min(row, M - 1) * Kb + kb;
      o1 = (unsigned)min(row + 128, M - 1) * Kb + kb;
    } else {
      o0 = ((unsigned)(l_n0 + 32 * sub) + tb) * Kb + kb;
      o1 = o0 + 128u * Kb;
    }
    const uint8_t* src = isA ? A : W;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (size_t)o0),
                                     (__attribute__((address_space(3))) void*)(base), 16, 0, 0);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (size_t)o1),
                                     (__attribute__((address_space(3))) void*)(base + 8192), 16, 0, 0);
    if (kind == 0) {
      // scales: wave w -> operand w >> 2, k-block w & 3 of this K-tile: 256 bytes = the tile's 256 rows (mx_spos order), 4 per lane
      const uint8_t* sp = wave < 4 ? SA + (size_t)(4 * l_kt + wave) * lda_s + l_m0 : SW + (size_t)(4 * l_kt + wave - 4) * ldw_s + l_n0;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(sp + (unsigned)(lane * 4)),
                                       (__attribute__((address_space(3))) void*)(lds + buf * BUF + 65536 + wave * 256), 4, 0, 0);
    }
  };

  // ---- read side: lane (r = l & 31, h = l >> 5) reads, of row r of a 32-row fragment and 64-k step kk, chunk 4 kk + h (k-block
  // 2 kk: operand bytes 0-15) and chunk 4 kk + 2 + h (k-block 2 kk + 1: bytes 16-31), and supplies the scale of k-block 2 kk + h
  const unsigned pos0 = (unsigned)((h ^ ((r >> 1) & 7)) << 4);
  const unsigned aoff = (unsigned)((wr * 64 + r) * 128), boff = (unsigned)((wc * 32 + r) * 128);
  const unsigned soff = (unsigned)(h * 256 + r * 4);
  f32x16_t acc[4][2];
  i32x8_t fa[2][2], fb0[2], fb1[2];  // [fragment][kk]: the operand tuple is formed where it is read, once, not at each MFMA
  auto ld8 = [](const unsigned char* p, unsigned o0, unsigned o1) {
    const u32x4 lo = *reinterpret_cast<const u32x4*>(p + o0), hi = *reinterpret_cast<const u32x4*>(p + o1);
    i32x8_t v;
    v[0] = (int)lo.x; v[1] = (int)lo.y; v[2] = (int)lo.z; v[3] = (int)lo.w;
    v[4] = (int)hi.x; v[5] = (int)hi.y; v[6] = (int)hi.z; v[7] = (int)hi.w;
    return v;
  };
  unsigned saX[2], swX[2], saY[2], swY[2];  // scale dwords per 64-k step of the even / odd K-tile
  auto read_a = [&](auto bufc, auto subc) {
    constexpr int buf = decltype(bufc)::value, sub = decltype(subc)::value;
    unsigned ao = aoff, ps = pos0;  // rebuilt per phase: held across the loop, the read addresses of all buffers / blocks were spilled
    asm volatile("" : "+v"(ao), "+v"(ps));
    const unsigned pos0 = ps;
    const unsigned char* b = lds + buf * BUF + (sub == 0 ? 1 : 3) * 16384 + ao;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) fa[i][kk] = ld8(b + i * 4096, pos0 ^ ((4 * kk) << 4), pos0 ^ ((4 * kk + 2) << 4));
  };
  auto read_b = [&](auto bufc, auto subc, i32x8_t (&fb)[2]) {
    constexpr int buf = decltype(bufc)::value, sub = decltype(subc)::value;
    unsigned bo = boff, ps = pos0;
    asm volatile("" : "+v"(bo), "+v"(ps));
    const unsigned pos0 = ps;
    const unsigned char* b = lds + buf * BUF + (sub == 0 ? 0 : 2) * 16384 + bo;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) fb[kk] = ld8(b, pos0 ^ ((4 * kk) << 4), pos0 ^ ((4 * kk + 2) << 4));
  };
  auto read_s = [&](auto bufc, unsigned (&sa)[2], unsigned (&sw)[2]) {
    constexpr int buf = decltype(bufc)::value;
    unsigned so = soff;
    asm volatile("" : "+v"(so));
    const unsigned char* bs = lds + buf * BUF + 65536 + so;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      sa[kk] = *reinterpret_cast<const unsigned*>(bs + kk * 512 + wr * 128);
      sw[kk] = *reinterpret_cast<const unsigned*>(bs + 1024 + kk * 512 + (wc >> 1) * 128) >> ((wc & 1) * 16);  // bytes 0, 1 = columns j = 0, 1
    }
  };
  // quadrant (A-sub a, column fragment j) over the K-tile's two 64-k steps: 4 MFMAs; op_sel = byte of the scale dword
  auto mm = [&](auto ac, auto jc, const i32x8_t (&fb)[2], const unsigned (&sa)[2], const unsigned (&sw)[2]) {
    constexpr int a = decltype(ac)::value, j = decltype(jc)::value;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      acc[2 * a][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fb[kk], fa[0][kk], acc[2 * a][j], 0, 0, j, (int)sw[kk], 2 * a, (int)sa[kk]);
      acc[2 * a + 1][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fb[kk], fa[1][kk], acc[2 * a + 1][j], 0, 0, j, (int)sw[kk], 2 * a + 1, (int)sa[kk]);
    }
    // a use of the results inside the phase: without it the MFMAs (pure, next used a K-tile later) are sunk out of their phase
    asm volatile("" : "+v"(acc[2 * a][j]), "+v"(acc[2 * a + 1][j]));
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>;
  using I3 = std::integral_constant<int, 3>;
  constexpr int NST = OUT == MX_OUT_MX ? 40 : 32;  // store instructions per lane of a full tile's epilogue
  auto phase = [&](auto xc, auto laxc) {
    constexpr int x = decltype(xc)::value;
    constexpr bool lax = decltype(laxc)::value;
    constexpr int p = x & 3;
    constexpr bool odd = ((x >> 2) & 1) != 0;  // K-tile parity: buffer and scale register set
    using CB = std::integral_constant<int, (x >> 2) & 1>;
    using NB = std::integral_constant<int, ((x >> 2) & 1) ^ 1>;
    using SK = std::integral_constant<int, (x + 2) & 3>;
    using SB = std::integral_constant<int, ((x + 6) >> 2) & 1>;
    if (p == 0) read_a(CB{}, I0{});
    else if (p == 1) read_b(CB{}, I1{}, fb1);
    else if (p == 2) read_a(CB{}, I1{});
    else {
      read_b(NB{}, I0{}, fb0);
      if (odd) read_s(NB{}, saX, swX); else read_s(NB{}, saY, swY);
    }
    stage(SK{}, SB{});
    if (lax) {
      if (NST == 40) asm volatile("s_waitcnt vmcnt(50)\n\ts_barrier" ::: "memory");  // 9 operand loads + 40 stores + the bias load
      else asm volatile("s_waitcnt vmcnt(42)\n\ts_barrier" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(9)\n\ts_barrier" ::: "memory");
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
    if (!odd) {
      if (p == 0) mm(I0{}, I0{}, fb0, saX, swX);
      else if (p == 1) mm(I0{}, I1{}, fb1, saX, swX);
      else if (p == 2) mm(I1{}, I0{}, fb0, saX, swX);
      else mm(I1{}, I1{}, fb1, saX, swX);
    } else {
      if (p == 0) mm(I0{}, I0{}, fb0, saY, swY);
      else if (p == 1) mm(I0{}, I1{}, fb1, saY, swY);
      else if (p == 2) mm(I1{}, I0{}, fb0, saY, swY);
      else mm(I1{}, I1{}, fb1, saY, swY);
    }
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  };
  using P0 = std::integral_constant<int, 0>;
  using P1 = std::integral_constant<int, 1>;
  using P2 = std::integral_constant<int, 2>;
  using P3 = std::integral_constant<int, 3>;
  using P4 = std::integral_constant<int, 4>;
  using P5 = std::integral_constant<int, 5>;
  using P6 = std::integral_constant<int, 6>;
  using P7 = std::integral_constant<int, 7>;

  // prologue: events 0..5; events 0 (B-h0 + scales of K-tile 0) and 1 landed everywhere behind the barrier
  lookup(0, l_m0, l_n0, l_kt, l_kend);
  lookup(1, n_m0, n_n0, n_kt, n_kend);
  --l_kt;
  stage(I0{}, I0{});
  stage(I1{}, I0{});
  stage(I2{}, I0{});
  stage(I3{}, I0{});
  stage(I0{}, I1{});
  stage(I1{}, I1{});
  asm volatile("s_waitcnt vmcnt(9)\n\ts_barrier" ::: "memory");
  read_b(I0{}, I0{}, fb0);
  read_s(I0{}, saX, swX);
  if (wr == 1) asm volatile("s_barrier" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);

  int lax = 0;
  unsigned char* blds = lds + 2 * BUF + wave * 256;
  for (int ord = 0; ord < cnt; ++ord) {
    int tile, kt0, kte;
    item(ord, tile, kt0, kte);
    const bool is_tail = S > 1 && ord >= R;
    // the stream is inside item `ord` now (it runs at most a K-tile and a half ahead and an item has at least two): its next is ord + 1
    lookup(ord + 1, n_m0, n_n0, n_kt, n_kend);
    const int mt = tile / ntn, nt = tile - mt * ntn;
    const int m0 = mt * 256, n0 = nt * 256;
    if (EPI != GE_PLAIN) {
      unsigned l4 = (unsigned)lane * 4u;
      asm volatile("" : "+v"(l4));
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(reinterpret_cast<const unsigned char*>(bias + n0 + wc * 64) + l4),
                                       (__attribute__((address_space(3))) void*)(blds), 4, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[i][j][v] = 0.f;
    if (lax) {
      phase(P0{}, std::true_type{});
      phase(P1{}, std::true_type{});
      phase(P2{}, std::true_type{});
      phase(P3{}, std::true_type{});
    } else {
      phase(P0{}, std::false_type{});
      phase(P1{}, std::false_type{});
      phase(P2{}, std::false_type{});
      phase(P3{}, std::false_type{});
    }
    phase(P4{}, std::false_type{});
    phase(P5{}, std::false_type{});
    phase(P6{}, std::false_type{});
    phase(P7{}, std::false_type{});
    for (int t = kt0 + 2; t < kte; t += 2) {
      phase(P0{}, std::false_type{});
      phase(P1{}, std::false_type{});
      phase(P2{}, std::false_type{});
      phase(P3{}, std::false_type{});
      phase(P4{}, std::false_type{});
      phase(P5{}, std::false_type{});
      phase(P6{}, std::false_type{});
      phase(P7{}, std::false_type{});
    }

    // epilogue: acc[i][j][v] = C[m = m0 + wr*128 + i*32 + r][n = n0 + wc*64 + j*32 + (v&3) + 8*(v>>2) + 4*h]
    const bool has_vt = OUT == MX_OUT_BF16 && vt != nullptr && n0 + 256 > vt_n0;
    const bool full = m0 + 256 <= M;
    unsigned hq = (unsigned)(h * 16);
    asm volatile("" : "+v"(hq));
    auto emit = [&](auto guardc) {
      constexpr bool guard = decltype(guardc)::value;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int nf = n0 + wc * 64 + j * 32;
        f32x4v_t bq[4];
#pragma unroll
        for (int qd = 0; qd < 4; ++qd)
          bq[qd] = EPI != GE_PLAIN ? *reinterpret_cast<const f32x4v_t*>(blds + hq + j * 128 + qd * 32) : f32x4v_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int m = m0 + wr * 128 + i * 32 + r;
          float x[16];
#pragma unroll
          for (int qd = 0; qd < 4; ++qd)
#pragma unroll
            for (int e = 0; e < 4; ++e) x[4 * qd + e] = acc[i][j][4 * qd + e] + bq[qd][e];
          if (EPI == GE_RELU) {
#pragma unroll
            for (int v = 0; v < 16; ++v) x[v] = fmaxf(x[v], 0.f);
          }
          if (OUT == MX_OUT_MX) {  // the row's 32 columns of this fragment are one MX block: 16 here, 16 in the other half-wave
            float am = 0.f;
#pragma unroll
            for (int v = 0; v < 16; ++v) am = fmaxf(am, fabsf(x[v]));
            am = fmaxf(am, xor32_f(am));
            float inv;
            const uint32_t sb = mx_block_scale(am, inv);
            if (!guard || m < M) {
              uint8_t* cp = reinterpret_cast<uint8_t*>(Cv) + (size_t)m * N + nf + 4 * h;
#pragma unroll
              for (int qd = 0; qd < 4; ++qd)
                *reinterpret_cast<uint32_t*>(cp + 8 * qd) = mx_pack4(x[4 * qd], x[4 * qd + 1], x[4 * qd + 2], x[4 * qd + 3], inv);
              if (h == 0) SC[(size_t)(nf >> 5) * ldc_s + mx_spos(m)] = (uint8_t)sb;
            }
          } else if (!guard || m < M) {
            if (OUT == MX_OUT_F32) {
              float* cp = reinterpret_cast<float*>(Cv) + (size_t)m * N + nf + 4 * h;
              f32x4v_t old[4];
              if (EPI == GE_RESID) {  // the four old values together behind one wait (see mfma256p_kernel)
#pragma unroll
                for (int qd = 0; qd < 4; ++qd) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(old[qd]) : "v"(cp + 8 * qd) : "memory");
                asm volatile("s_waitcnt vmcnt(0)" : "+v"(old[0]), "+v"(old[1]), "+v"(old[2]), "+v"(old[3]) : : "memory");
              }
#pragma unroll
              for (int qd = 0; qd < 4; ++qd) {
                float4* p4 = reinterpret_cast<float4*>(cp + 8 * qd);
                if (EPI == GE_RESID) *p4 = make_float4(old[qd][0] + x[4 * qd], old[qd][1] + x[4 * qd + 1], old[qd][2] + x[4 * qd + 2], old[qd][3] + x[4 * qd + 3]);
                else *p4 = make_float4(x[4 * qd], x[4 * qd + 1], x[4 * qd + 2], x[4 * qd + 3]);
              }
            } else {
              bf16* cp = reinterpret_cast<bf16*>(Cv) + (size_t)m * N + nf + 4 * h;
#pragma unroll
              for (int qd = 0; qd < 4; ++qd) {
                union { bf16 e[4]; uint2 u; } pk;
#pragma unroll
                for (int v = 0; v < 4; ++v) pk.e[v] = (bf16)x[4 * qd + v];
                *reinterpret_cast<uint2*>(cp + 8 * qd) = pk.u;
                if (vt != nullptr && nf >= vt_n0) {
#pragma unroll
                  for (int v = 0; v < 4; ++v) vt[(size_t)(nf + 8 * qd + 4 * h + v - vt_n0) * vt_ld + m] = pk.e[v];
                }
              }
            }
          }
        }
      }
    };
    if (is_tail) {
      // tail item (fp32 forms only): partial tile to the workspace, [item][256][256] fp32; p8_tail_reduce_kernel finishes it
      unsigned lo = (unsigned)((r * 256 + 4 * h) * 4);
      asm volatile("" : "+v"(lo));
      unsigned char* slab = reinterpret_cast<unsigned char*>(tl.ws + (size_t)(tail_a * S + tail_s) * 65536 + wr * 32768 + wc * 64) + lo;
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int qd = 0; qd < 4; ++qd)
            *reinterpret_cast<f32x4v_t*>(slab + i * 32768 + j * 128 + qd * 32) =
                f32x4v_t{acc[i][j][4 * qd], acc[i][j][4 * qd + 1], acc[i][j][4 * qd + 2], acc[i][j][4 * qd + 3]};
      lax = 0;
    } else if (full) {
      emit(std::false_type{});
      lax = __builtin_amdgcn_readfirstlane((has_vt || EPI == GE_RESID) ? 0 : 1);
    } else {
      emit(std::true_type{});
      lax = 0;
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (wr == 0) asm volatile("s_barrier" ::: "memory");
}

// C = A . W^T on MXFP8 operands.  M any (rows clamped, scale arrays padded to 256 rows), N % 256 == 0, K % 128 == 0.
static inline int mx_gemm_dispatch(const uint8_t* A, const uint8_t* SA, int lda_s, const uint8_t* W, const uint8_t* SW, int ldw_s,
                                   const float* bias, void* C, uint8_t* SC, int ldc_s, int M, int N, int K, int epi, int out,
                                   hipStream_t s, bf16* vt = nullptr, int vt_n0 = 0, int vt_ld = 0) {
  if (N % 256 != 0 || K % 128 != 0 || M < 1) return 1;
  const int ntn = N / 256, ntm = (M + 255) / 256;
  const int ncu = vx_cu_count();
  const int grid = ntn * ntm < ncu ? ntn * ntm : ncu;
  // the 8-phase kernel needs whole pairs of 128-k tiles and 32-bit byte offsets; any other K or size runs the 32-k ring (mx256_kernel)
  const bool p8 = K % 256 == 0 && (size_t)M * K < 0xFFFF0000ull && (size_t)N * K < 0xFFFF0000ull;
  // tail split (VX_GEMM_TAIL=1, fp32 forms only: the second launch's epilogue is the bf16 GEMM's); a K-tile here is 128 k
  const P8Tail tl = (p8 && out == MX_OUT_F32) ? p8_tail_plan(ntn * ntm, grid, K / 2, s) : P8Tail{nullptr, 0};
#define MX(E, O)                                                                                                         \
  do {                                                                                                                  \
    static bool attr_dev[16] = {}; bool& attr_done = attr_dev[vx_cur_device()];                                                                                      \
    if (!attr_done) {                                                                                                   \
      (void)hipFuncSetAttribute((const void*)mx256_kernel<E, O>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * (32768 + 1024));  \
      attr_done = true;                                                                                                 \
    }                                                                                                                   \
    if (p8) {                                                                                                           \
      static bool attr_pdev[16] = {}; bool& attr_p = attr_pdev[vx_cur_device()];                                                                                       \
      if (!attr_p) {                                                                                                    \
        (void)hipFuncSetAttribute((const void*)mx256p_kernel<E, O>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * (65536 + 2048) + 2048); \
        attr_p = true;                                                                                                  \
      }                                                                                                                 \
      mx256p_kernel<E, O><<<grid, 512, 2 * (65536 + 2048) + 2048, s>>>(A, SA, lda_s, W, SW, ldw_s, bias, C, SC, ldc_s, M, N, K, vt, vt_n0, vt_ld, ntn, ntn * ntm, tl); \
      if (tl.split > 1) {                                                                                               \
        const int nt_all = ntn * ntm, rem_t = nt_all % grid;                                                            \
        p8_tail_reduce_kernel<E, true><<<dim3(rem_t, 16), 256, 0, s>>>(tl.ws, tl.split, nt_all - rem_t, ntn, nt_all, bias, C, M, N, nullptr, 0, 0); \
      }                                                                                                                 \
    } else                                                                                                              \
      mx256_kernel<E, O><<<grid, 512, 4 * (32768 + 1024), s>>>(A, SA, lda_s, W, SW, ldw_s, bias, C, SC, ldc_s, M, N, K, vt, vt_n0, vt_ld, ntn, ntn * ntm);  \
  } while (0)
  if (out == MX_OUT_MX && epi == GE_RELU) MX(GE_RELU, MX_OUT_MX);
  else if (out == MX_OUT_BF16 && epi == GE_BIAS) MX(GE_BIAS, MX_OUT_BF16);
  else if (out == MX_OUT_F32 && epi == GE_RESID) MX(GE_RESID, MX_OUT_F32);
  else if (out == MX_OUT_F32 && epi == GE_BIAS) MX(GE_BIAS, MX_OUT_F32);
  else if (out == MX_OUT_F32 && epi == GE_RELU) MX(GE_RELU, MX_OUT_F32);
  else if (out == MX_OUT_F32 && epi == GE_PLAIN) MX(GE_PLAIN, MX_OUT_F32);
  else return 2;
#undef MX
  return 0;
}

}  // namespace vx
